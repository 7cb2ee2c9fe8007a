"""Fresh-seed fuzz inside what the driver runs: tools/fuzz_parity.py and tools/fuzz_ao.py for a
fixed time budget, every output array poisoned before the launch (LH_POISON_OUTPUTS=1: an answer slot nobody writes is a
mismatch), both builders, against the oracle.  The seed is derived from the CODE under test -- `git rev-parse HEAD` where a
work tree is there, else (the GPU box gets the snapshot without .git) a digest of the product's sources -- so every commit
that changes a kernel walks through scenes no earlier commit has seen; the seed is printed, a failure is replayed with
`python tools/fuzz_parity.py <rounds> <seed>`.  Round 5's two defects (zero-area noise determinants; any-hit slots left
unwritten) were found by exactly these tools and by none of the fixtures."""
import glob
import hashlib
import os
import re
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = [pytest.mark.gpu]


def code_seed():
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=20)
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "lucille_amd", "include"], capture_output=True, text=True, timeout=20)
        if head.returncode == 0 and dirty.returncode == 0 and not dirty.stdout.strip():
            return int(head.stdout.strip()[:8], 16), "git HEAD " + head.stdout.strip()[:12]
    except (OSError, subprocess.SubprocessError):
        pass
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "lucille_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "lucille_amd", "csrc", "*.c"))
                    + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(f, "rb").read())
    return int(h.hexdigest()[:8], 16), "sha256 of lucille_amd/csrc + include = " + h.hexdigest()[:12]


def run_tool(args, budget_s, timeout):
    env = dict(os.environ, LH_POISON_OUTPUTS="1", FUZZ_BUDGET_S=str(budget_s))
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    tail = (r.stdout[-2500:] + "\n" + r.stderr[-1500:])
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, tail
    return r.stdout


def test_fresh_seed_parity_fuzz_both_builders():
    seed, why = code_seed()
    print("fuzz_parity seed %d (%s)" % (seed, why))
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_parity.py"), "400", str(seed)], budget_s=40, timeout=600)
    m = re.search(r"(\d+) rays over (\d+) scenes x 2 builders: closest-hit records and any-hit flags equal to the oracle", out)
    assert m, out[-2000:]
    assert int(m.group(2)) >= 9 and int(m.group(1)) >= 300000, "the budget did not even cover one scene of every kind: %s" % m.group(0)


def test_fresh_seed_ao_pipeline_fuzz_both_builders():
    seed, why = code_seed()
    print("fuzz_ao seed %d (%s)" % (seed, why))
    out = run_tool([os.path.join(ROOT, "tools", "fuzz_ao.py"), str(seed % 1000003), "200"], budget_s=20, timeout=600)
    m = re.search(r"(\d+) AO rays over (\d+) frames x 2 builders: fused == materialised bit for bit, occlusion equal to the oracle", out)
    assert m, out[-2000:]
    assert int(m.group(2)) >= 3, m.group(0)
