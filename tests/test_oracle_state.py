"""The whole hit epilogue -- ri_intersection_state_build with colours, tangents / binormals, shared and
unshared texture coordinates, two-sided meshes (intersection_state.c:99-248) -- oracle restatement against the
compiled reference's records (tests/golden/state_attr.npz, made by tests/golden/make_golden.py --state) and,
where the reference can be built, against the live reference.  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.golden.make_golden import apply_state_scene, state_scene
from tests.helpers import load_golden


def test_oracle_state_records_equal_the_reference_goldens():
    g = load_golden("state_attr")
    meshes, org, dr = state_scene(int(g["seed"]))
    o = po.Oracle(); apply_state_scene(o, meshes, False); o.build()
    prim, st = o.state_batch(org, dr)
    assert np.array_equal(prim, g["prim"])
    assert np.array_equal(st, g["state"])                    # all 24 doubles of every ray, bit for bit
    hit = prim != po.MISS
    assert hit.sum() > 3000 and st[hit, 23].sum() > 100      # two-sided back faces are exercised
    assert (st[hit, 15:18] != 1.0).any() and (st[hit, 18:20] != 0.0).any()


@pytest.mark.skipif(not po.ref_available(), reason="the compiled reference is not available here")
def test_oracle_state_records_equal_the_live_reference():
    meshes, org, dr = state_scene(77)
    ref = po.RefLib(); apply_state_scene(ref, meshes, True); ref.build()
    o = po.Oracle(); apply_state_scene(o, meshes, False); o.build()
    rp, rs = ref.state_batch(org, dr); op, os_ = o.state_batch(org, dr)
    assert np.array_equal(rp, op) and np.array_equal(rs, os_)
