"""Generate the committed golden fixtures by running the COMPILED REFERENCE
(oracle/_ref, built from /root/reference by oracle/Makefile).  Runs only where
/root/reference exists.  Fixtures are data (inputs by seed, expected outputs);
no reference source is stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def soup_fixture(name, ntri, nrays, half_extent):
    """S-soup (SURVEY.md Appendix C generator): inputs are reproducible from
    (seed, ntri, nrays, half_extent); outputs are the reference's hit records,
    per-batch traversal counters (RI_BVH_TRACE_STATISTICS build) and tree shape."""
    P, idx, org, dr = po.soup(ntri, nrays, half_extent)
    ref = po.RefLib(stat=True)
    ref.add_mesh(P, idx)
    ref.build()
    prim, t, u, v, cnt = ref.intersect(org, dr, counters=True)
    tree = ref.tree_stats()
    bmin, bmax = ref.bbox()
    # exact-t ties between two different triangles would make the winner
    # tree-dependent; record that none occur
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    bf = o.brute_force(org[:2000], dr[:2000], nthreads=os.cpu_count())
    assert np.array_equal(bf[0], prim[:2000]) and np.array_equal(bf[1], t[:2000])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), seed=np.uint64(po.SOUP_SEED), ntri=ntri, nrays=nrays,
        half_extent=half_extent, prim=prim, t=t, u=u, v=v,
        counters=np.array([cnt[k] for k in ("ninner", "nleaf", "ntested", "nhit", "nrays")], np.uint64),
        tree=np.array([tree[k] for k in ("ninner", "nleaf", "max_depth", "max_leaf_tris", "ntriangles")], np.uint64),
        bmin=bmin, bmax=bmax)
    print(name, "hits", int((prim != po.MISS).sum()), "of", nrays, "sum t", float(t[prim != po.MISS].sum()), cnt, tree)


if __name__ == "__main__":
    if not po.ref_available(stat=True):
        po.build_ref()
    soup_fixture("soup_20k", 20000, 20000, 0.005)
    soup_fixture("soup_3k_fat", 3000, 10000, 0.05)
    # check values of the full S-soup-1M are pinned in
    # tests/test_oracle_vs_ref.py against the live reference, not stored here.
