"""The C-ABI shared library loads and exports every symbol include/*.h declares.
No compute calls here (no GPU in the CPU suite)."""
import ctypes as C
import os
import re

import pytest

import lucille_amd as la
from lucille_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:lh|ri|lucille)_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_loads():
    path = la.build_library()
    assert os.path.exists(path)
    C.CDLL(path)


@pytest.mark.parametrize("header", sorted(h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")))
def test_exports_every_declared_symbol(header):
    L = C.CDLL(la.build_library())
    names = _declared(header)
    assert names, "no declarations parsed from " + header
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/%s but not exported: %s" % (header, missing)


def test_binding_lists_the_whole_header():
    assert sorted(binding.ABI_SYMBOLS) == _declared("lucille_hip.h")


def test_no_gpu_fails_loudly_not_silently():
    """there is no CPU fallback in the product: without a device, creation fails with a message"""
    if la.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(la.LucilleHipError, match="no HIP device"):
        la.HipAccel(0)


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under lucille_amd/ may reference it"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lucille_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"(import|from)\s+oracle|pyoracle|liblucille_oracle|lucille_oracle\.h|liblucille_ref|liblh_model", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    # bench.py: only the cpu_baseline leg may touch oracle/ (the checker timed as the CPU baseline, never the thing measured)
    for f in ["bench.py"] + [os.path.join("benchlegs", x) for x in sorted(os.listdir(os.path.join(ROOT, "benchlegs"))) if x.endswith(".py")]:
        txt = open(os.path.join(ROOT, f)).read()
        uses = re.search(r"(import|from)\s+oracle|pyoracle|CDLL\([^)]*oracle", txt) is not None
        assert uses == (f == os.path.join("benchlegs", "cpu.py")), f
