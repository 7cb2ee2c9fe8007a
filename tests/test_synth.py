"""The product's synthetic workload generator (lucille_amd/csrc/lh_synth.c: SURVEY.md Appendix C) against
the checker's copy of the same specification, the jump-ahead that lets a rank start at its slice of a
ray dump, and the tessellation BASELINE config 5 is built with.  CPU only."""
import numpy as np

from lucille_amd import scenes, shard
from oracle import pyoracle as po


def test_soup_equals_the_oracle_generator_and_the_survey_stream():
    P, idx, org, dr = scenes.soup(2000, 5000, 0.005)
    P2, idx2, org2, dr2 = po.soup(2000, 5000, 0.005, po.SOUP_SEED)
    assert np.array_equal(P, P2) and np.array_equal(idx, idx2)
    assert np.array_equal(org, org2) and np.array_equal(dr, dr2)
    assert scenes.SOUP_SEED == 88172645463325252 == po.SOUP_SEED
    assert np.allclose(np.linalg.norm(dr, axis=1), 1.0)
    assert np.array_equal(idx, np.arange(6000, dtype=np.uint32))


def test_jump_ahead_reaches_every_slice_of_a_dump():
    _, _, st = scenes.soup_triangles(100, 0.01)
    assert scenes.skip(scenes.SOUP_SEED, 12 * 100) == st            # a triangle is 12 draws
    org, dr, end = scenes.soup_rays(10007, st)
    assert scenes.skip(st, 5 * 10007) == end                        # a ray is 5 draws
    for world in (2, 3, 8):
        parts_o, parts_d = [], []
        for r in range(world):
            b, e = shard.ray_slice(10007, r, world)
            o, d, _ = scenes.soup_rays(e - b, scenes.skip(st, 5 * b))
            parts_o.append(o.copy()); parts_d.append(d.copy())
        assert np.array_equal(np.concatenate(parts_o), org) and np.array_equal(np.concatenate(parts_d), dr)
    assert scenes.skip(st, 0) == st
    assert scenes.skip(scenes.skip(st, 2 ** 40 + 12345), 7) == scenes.skip(st, 2 ** 40 + 12352)


def test_tessellation_is_midpoint_subdivision_without_shared_vertices():
    rng = np.random.default_rng(5)
    P = rng.uniform(-1, 1, (30, 3)); idx = rng.integers(0, 30, 36).astype(np.uint32)
    for lv in (0, 1, 3):
        Q, J = scenes.tessellate(P, idx, lv)
        tri = P[idx.astype(np.int64).reshape(-1, 3)]
        for _ in range(lv):
            a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
            ab, bc, ca = 0.5 * (a + b), 0.5 * (b + c), 0.5 * (c + a)
            tri = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1),
                            np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
        assert np.array_equal(Q, tri.reshape(-1, 3)) and np.array_equal(J, np.arange(Q.shape[0], dtype=np.uint32))
    # positions may be lucille's double[4]
    Q4, _ = scenes.tessellate(np.concatenate([P, np.ones((30, 1))], 1), idx, 2)
    assert np.array_equal(Q4, scenes.tessellate(P, idx, 2)[0])
