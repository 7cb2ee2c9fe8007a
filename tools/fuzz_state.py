"""Randomised sweep of the hit epilogue (lh_accel_state_build_*: P, Ng, Ns, tangent, binormal, colour, st, I, inside) against the
oracle's ri_intersection_state_build, bit for bit, on the test fixtures' scene generator with other seeds.
python tools/fuzz_state.py [first seed] [count]"""
import os, sys
os.environ.setdefault("LH_POISON_OUTPUTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lucille_amd as la
from oracle import pyoracle as po
from tests.golden.make_golden import apply_state_scene, state_scene

class Prod:
    def __init__(self): self.acc = la.HipAccel(0)
    def add_mesh(self, P, idx): self.acc.add_mesh(P, idx)
    def set_normals(self, k, N, two_side): self.acc.set_normals(k, N, two_side)
    def set_attribute(self, k, kind, data): self.acc.set_attribute(k, kind, data)

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rays = 0
for seed in range(s0, s0 + cnt):
    meshes, org, dr = state_scene(seed)
    for unshared in (False, True):
        try:
            p = Prod(); apply_state_scene(p, meshes, unshared); p.acc.commit()
            o = po.Oracle(); apply_state_scene(o, meshes, unshared); o.build()
        except TypeError:
            break
        prim, t, u, v = p.acc.intersect_host(org, dr)
        op, ost = o.state_batch(org, dr)
        st = p.acc.state_build(org, dr, prim, t, u, v)
        if not (np.array_equal(prim, op) and np.array_equal(st, ost)):
            bad = np.nonzero((st != ost).any(1))[0]
            print("MISMATCH seed %d unshared %s: %d records, first %s" % (seed, unshared, bad.size, bad[:5])); sys.exit(1)
        rays += org.shape[0]; p.acc.close()
print("%d rays over %d scenes: hit records and epilogue records equal to the oracle" % (rays, cnt))
