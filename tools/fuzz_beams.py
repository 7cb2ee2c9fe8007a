"""Randomised sweep of the beam queries against the oracle: beam visibility (ri_bvh_intersect_beam_visibility) over random soups and beam
spreads, beam raster (ri_bvh_intersect_beam) over tests/test_beam_raster.py's case generator with other seeds, both builders.
python tools/fuzz_beams.py [seed] [rounds]"""
import os, sys
os.environ.setdefault("LH_POISON_OUTPUTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lucille_amd as la
from oracle import pyoracle as po
from tests.helpers import random_beams
from tests.test_beam_raster import raster_case, oracle_planes, hip_planes

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed); nb = 0; nr = 0
for r in range(rounds):
    ntri = int(rng.choice([1, 5, 40, 700, 30000, 250000])); he = float(10.0 ** rng.uniform(-2.6, -0.3))
    P, idx, _, _ = po.soup(ntri, 1, he, int(rng.integers(1, 1 << 30)))
    o = po.Oracle(); o.add_mesh(P, idx); o.build()
    for build in ("host", "device"):
        acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=build); acc.wait_exact()
        for s in (0.0003, 0.01, 0.1, 0.5):
            org, d = random_beams(np.random.default_rng(seed * 1000 + r), 4000, s)
            a, b = acc.beam_visibility(org, d), o.beam_visibility(org, d)
            if not np.array_equal(a, b):
                print("MISMATCH visibility round %d build %s spread %g: %d beams" % (r, build, s, int((a != b).sum()))); sys.exit(1)
            nb += org.shape[0]
        acc.close()
    kw = dict(seed=int(rng.integers(1000, 1 << 30)), ntri=int(rng.choice([1, 30, 400, 3000])), width=int(rng.choice([16, 48, 64, 130])), height=int(rng.choice([16, 40, 64])),
              nbeams=6, tri_size=float(rng.choice([0.05, 0.15, 0.6])), fov=float(rng.choice([30.0, 45.0, 80.0])), inside=bool(rng.integers(0, 2)))
    try:
        c = raster_case(**kw)
    except TypeError:
        kw.pop("tri_size", None); kw.pop("fov", None); kw.pop("inside", None); c = raster_case(**kw)
    rc, t_exp, fl_exp = oracle_planes(c)
    for build in ("host", "device"):
        t, st, fl = hip_planes(c, build=build)
        if not (np.array_equal(np.where(st == 1, 0, st), rc) and np.array_equal(t, t_exp) and np.array_equal(fl, fl_exp)):
            print("MISMATCH raster round %d build %s case %s" % (r, build, kw)); sys.exit(1)
    nr += kw["nbeams"]
print("%d visibility beams and %d raster planes over %d rounds x 2 builders: equal to the oracle" % (nb, nr, rounds))
