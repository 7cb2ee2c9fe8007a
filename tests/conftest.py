import os
import sys

import pytest

# every ray dump of the suite starts from output arrays filled with 0x77 (lh_query.hip): an answer slot that no kernel writes is a wrong
# record in the comparison, not whatever a recycled buffer held (round 5: lost any-hit rays had hidden behind earlier results)
os.environ.setdefault("LH_POISON_OUTPUTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _any_hit_answers_are_zero_or_one():
    """with LH_POISON_OUTPUTS an any-hit slot that nobody wrote holds 0x77 -- which `.astype(bool)` in a comparison would read as
    "occluded".  Every any-hit dump made through the bindings is checked here for values other than 0 and 1."""
    if not _has_gpu():
        yield
        return
    import numpy as np
    import lucille_amd as la
    host0, dev0 = la.HipAccel.intersect_host, la.HipAccel.intersect_device

    def host(self, org, dr, mode=la.MODE_CLOSEST):
        r = host0(self, org, dr, mode=mode)
        if mode == la.MODE_ANY:
            assert int(np.asarray(r).max(initial=0)) <= 1, "an any-hit answer slot was left unwritten"
        return r

    def dev(self, org, dr, out=None, mode=la.MODE_CLOSEST, **kw):
        r = dev0(self, org, dr, out=out, mode=mode, **kw)
        if mode == la.MODE_ANY and not kw.get("counters") and r[0].numel():
            assert int(r[0].max().item()) <= 1, "an any-hit answer slot was left unwritten"
        return r

    la.HipAccel.intersect_host, la.HipAccel.intersect_device = host, dev
    yield
    la.HipAccel.intersect_host, la.HipAccel.intersect_device = host0, dev0
