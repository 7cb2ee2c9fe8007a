"""one closest-hit launch of a kernel variant over S-soup-1M (for counter passes): python tools/variant_once.py <variant> [nrays] [param=value ...]
LH_SOUP_NTRI / LH_SOUP_HALF / LH_SOUP_BUILD in the environment pick another soup (10000000 / 0.002 / device: the bench's HBM leg)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
v = int(sys.argv[1]); nr = int(sys.argv[2]) if len(sys.argv) > 2 else 20000000
P, idx, st = scenes.soup_triangles(int(os.environ.get("LH_SOUP_NTRI", 1000000)), float(os.environ.get("LH_SOUP_HALF", 0.005)))
ho, hd, _ = scenes.soup_rays(nr, st)
o = torch.from_numpy(ho).cuda(); d = torch.from_numpy(hd).cuda()
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit(build=os.environ.get("LH_SOUP_BUILD", "host"))
for kv in sys.argv[3:]:
    k, val = kv.split("="); acc.set_param(k, int(val))
out = acc.intersect_device(o[:1000000], d[:1000000], variant=v); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); acc.intersect_device(o, d, variant=v); e1.record(); torch.cuda.synchronize()
print("variant %d %s: %.0f Mrays/s" % (v, " ".join(sys.argv[3:]), nr / e0.elapsed_time(e1) / 1e3))
if os.environ.get("LH_VARIANT_COUNT"):
    ns = min(nr, 4000000); _, c = acc.intersect_device(o[:ns], d[:ns], variant=v, counters=True)
    print("per ray: %.3f node records, %.3f triangle records, %.4f fp64 tests" % (c["nodes"] / ns, c["tris"] / ns, c["exact"] / ns))
