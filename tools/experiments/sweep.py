"""Within-process A/B sweeps of the trace kernel on S-soup (interleaved rounds).
python tools/sweep.py <what> [nrays]     what in: grid, minact, bvh, sort """
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po

what = sys.argv[1]; nt = int(sys.argv[2]) if len(sys.argv) > 2 else 20000000
ntri = int(os.environ.get("NTRI", "1000000"))
P, idx, org, dr = po.soup(ntri, nt)
dev = torch.device("cuda:0")
d_org = torch.from_numpy(org).to(dev); d_dir = torch.from_numpy(dr).to(dev)

def timeit(acc, mode, variant, reps=3, o=d_org, d=d_dir):
    outs = acc.intersect_device(o, d, mode=mode, variant=variant); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(o, d, out=outs, mode=mode, variant=variant); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return o.shape[0] / min(ts) / 1e3

def mk(env=None):
    for k, v in (env or {}).items(): os.environ[k] = str(v)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(build="host")
    for k in (env or {}): os.environ.pop(k, None)
    return acc, info

if what == "grid":
    acc, info = mk()
    for g in (256, 512, 768, 1024, 1280, 1536, 2048, 2560):
        acc.set_grid(g)
        print("grid", g, "closest %.1f any %.1f Mrays/s" % (timeit(acc, 0, 4), timeit(acc, 1, 4)), flush=True)
elif what == "minact":
    for m in (8, 16, 24, 32, 40, 48, 56, 62):
        acc, info = mk({"LH_MIN_ACTIVE": m})
        print("min_active", m, "closest %.1f any %.1f Mrays/s" % (timeit(acc, 0, 4), timeit(acc, 1, 4)), flush=True)
        acc.close()
elif what == "bvh":
    for ci, ct in ((1, 1), (1, 0.5), (1, 0.25), (1, 2), (2, 1), (0.5, 1)):
        acc, info = mk({"LH_BVH_CI": ci, "LH_BVH_CT": ct})
        _, cnt = acc.intersect_device(d_org[:2000000].contiguous(), d_dir[:2000000].contiguous(), counters=True, variant=0)
        print("ci", ci, "ct", ct, "nodes", info["nnodes"], "depth", info["max_depth"], "nodes/ray %.2f tris/ray %.2f" % (cnt["nodes"] / 2e6, cnt["tris"] / 2e6),
              "closest %.1f any %.1f Mrays/s" % (timeit(acc, 0, 4), timeit(acc, 1, 4)), flush=True)
        acc.close()
elif what == "sort":
    acc, info = mk()
    base = timeit(acc, 0, 4)
    # coherence potential: sort rays by a Morton key of the origin (+ direction octant) with torch (experiment only)
    def morton(o, bits):
        q = (o.clamp(0, 0.999999) * (1 << bits)).long()
        key = torch.zeros(o.shape[0], dtype=torch.long, device=o.device)
        for b in range(bits):
            for k in range(3):
                key |= ((q[:, k] >> b) & 1) << (3 * b + k)
        return key
    for bits in (3, 4, 5, 6, 8):
        key = morton(d_org, bits) * 8 + ((d_dir[:, 0] < 0).long() | ((d_dir[:, 1] < 0).long() << 1) | ((d_dir[:, 2] < 0).long() << 2))
        perm = torch.argsort(key)
        so = d_org[perm].contiguous(); sd = d_dir[perm].contiguous()
        print("sorted bits", bits, "closest %.1f (unsorted %.1f) any %.1f Mrays/s" % (timeit(acc, 0, 4, o=so, d=sd), base, timeit(acc, 1, 4, o=so, d=sd)), flush=True)
        key2 = ((d_dir[:, 0] < 0).long() | ((d_dir[:, 1] < 0).long() << 1) | ((d_dir[:, 2] < 0).long() << 2)) * (1 << (3 * bits)) + morton(d_org, bits)
        perm = torch.argsort(key2); so = d_org[perm].contiguous(); sd = d_dir[perm].contiguous()
        print("  octant-major bits", bits, "closest %.1f any %.1f" % (timeit(acc, 0, 4, o=so, d=sd), timeit(acc, 1, 4, o=so, d=sd)), flush=True)
