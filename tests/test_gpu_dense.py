"""Round 4's launch policy: every launch of 65 536 rays or more walks at most 34 CHECKED LDS stack rows (four workgroups per CU;
rows4 in lh_kernels.hip) -- a ray that would overrun them is finished by the cooperative walk -- and the fused AO stage has a
visit budget of its own ("ao_budget").  Neither may change a bit: records and frames with the policy (the default) must equal
those of the unchecked walk of rounds 1-3 ("stack_cap" 64) and those of a stack capped far below what the tree asks for, on trees
that ask for more than 34 rows (a deep chain of nested boxes; the device builder's tree over an exponentially spaced line)."""
import numpy as np
import pytest

import lucille_amd as la
from oracle import pyoracle as po
from tests.helpers import chain_scene, load_golden, random_rays

pytestmark = pytest.mark.gpu


def _records(acc, d_org, d_dir):
    import torch
    out = acc.intersect_device(d_org, d_dir)
    occ = acc.intersect_device(d_org, d_dir, mode=la.MODE_ANY)[0]
    torch.cuda.synchronize()
    return [x.clone() for x in out] + [occ.clone()]


@pytest.mark.parametrize("build", ["host", "device"])
def test_big_dumps_walk_checked_rows_and_keep_every_bit(build):
    import torch
    rng = np.random.default_rng(41)
    if build == "host":
        P, idx = chain_scene(60)                                   # nested geometry: a tree far deeper than 34 rows cover
        org, dr = random_rays(rng, 200000, lo=-1.0, hi=2.0)
    else:
        # an exponentially spaced line of small triangles: the Morton order degenerates, the radix tree is a chain
        n = 3000
        c = np.stack([2.0 ** (-np.arange(n) * 0.02), np.zeros(n), np.zeros(n)], 1)
        P = (c[:, None, :] + rng.uniform(-1e-4, 1e-4, (n, 3, 3))).reshape(-1, 3)
        idx = np.arange(3 * n, dtype=np.uint32)
        org = rng.uniform(-0.2, 1.2, (200000, 3)); org[:, 1:] *= 0.05
        dr = rng.normal(size=(200000, 3)); dr[:, 0] *= 4.0
    acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(build=build)
    d_org = torch.from_numpy(np.ascontiguousarray(org)).cuda(); d_dir = torch.from_numpy(np.ascontiguousarray(dr)).cuda()
    base = _records(acc, d_org, d_dir)                             # the default: 34 checked rows for 200 000 rays
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    exp = o.intersect(org[:40000], dr[:40000], nthreads=16)
    assert np.array_equal(base[0][:40000].cpu().numpy().view(np.uint32), exp[0])
    for k in (1, 2, 3):
        assert np.array_equal(base[k][:40000].cpu().numpy(), exp[k])
    assert np.array_equal(base[4][:40000].cpu().numpy().astype(bool), exp[0] != po.MISS)
    for cap in (64, 20, 8):                                        # rounds 1-3's unchecked rows; caps far below the tree's depth
        acc.set_param("stack_cap", cap)
        got = _records(acc, d_org, d_dir)
        assert all(torch.equal(a, b) for a, b in zip(got, base)), (build, cap, info["max_depth"])
    # a small batch keeps unchecked rows and no queue: same records
    acc.set_param("stack_cap", 0)
    small = _records(acc, d_org[:3000], d_dir[:3000])
    assert all(torch.equal(a, b[:3000]) for a, b in zip(small, base))
    acc.close()


def test_ao_frames_do_not_depend_on_rows_or_budgets():
    """the fused AO stage at the default (34 checked rows, its own budget of 384 iterations) == unchecked rows == tiny caps ==
    any pair of budgets; "ray_budget" still sets both budgets (the tests of rounds 2-3 rely on it)"""
    import torch
    from lucille_amd import render, scenes
    g = load_golden("ao_c1")
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 4); acc.add_mesh(P, I)
    acc.commit(build="device")
    c = g["camera"]; cam = la.Camera.make(320, 240, c[16], c[:16], int(c[19]))
    ref_img, ref_stats = render.render_ao_frame(acc, cam, 2, 16, tile=320, seed=9)
    assert 0.05 < float(ref_img.mean()) < 0.95
    for cap, rb, ab in ((64, 128, 384), (16, 128, 384), (0, 128, 8), (0, 8, 4096), (0, 4096, 4096), (12, 3, 3)):
        acc.set_param("stack_cap", cap); acc.set_param("ray_budget", rb); acc.set_param("ao_budget", ab)
        img, stats = render.render_ao_frame(acc, cam, 2, 16, tile=320, seed=9)
        torch.cuda.synchronize()
        assert stats == ref_stats and torch.equal(img, ref_img), (cap, rb, ab)
    acc.close()
