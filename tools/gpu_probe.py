"""Scratch GPU probe: parity vs oracle + per-variant timing on S-soup scenes.
Usage: python tools/gpu_probe.py [ntri] [nrays_time] """
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lucille_amd as la
from oracle import pyoracle as po

ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0), "CUs", torch.cuda.get_device_properties(0).multi_processor_count, flush=True)

P, idx, org, dr = po.soup(ntri, nt)
acc = la.HipAccel(0); acc.add_mesh(P, idx); info = acc.commit(); print("accel", info, flush=True)

# ---- parity on the first 1M rays (oracle 8 threads)
npar = min(1000000, nt)
o = po.Oracle(); o.add_mesh(P, idx); o.build()
t0 = time.time(); ref = o.intersect(org[:npar], dr[:npar], nthreads=os.cpu_count()); print("oracle s", time.time() - t0, "cores", os.cpu_count(), flush=True)
d_org = torch.from_numpy(org).to(dev); d_dir = torch.from_numpy(dr).to(dev)
for variant in (0, 1, 2):
    out = acc.intersect_device(d_org[:npar].contiguous(), d_dir[:npar].contiguous(), variant=variant)
    torch.cuda.synchronize()
    prim = out[0].cpu().numpy().view(np.uint32); t = out[1].cpu().numpy(); u = out[2].cpu().numpy(); v = out[3].cpu().numpy()
    print("variant", variant, "closest parity prim/t/u/v:", np.array_equal(prim, ref[0]), np.array_equal(t, ref[1]), np.array_equal(u, ref[2]), np.array_equal(v, ref[3]),
          "nbad", int((prim != ref[0]).sum()), flush=True)
    occ = acc.intersect_device(d_org[:npar].contiguous(), d_dir[:npar].contiguous(), mode=la.MODE_ANY, variant=variant)[0]
    torch.cuda.synchronize()
    print("variant", variant, "anyhit parity:", np.array_equal(occ.cpu().numpy().astype(bool), ref[0] != po.MISS), flush=True)

# ---- counters
(_, cnt) = acc.intersect_device(d_org[:npar].contiguous(), d_dir[:npar].contiguous(), counters=True, variant=0)
print("counters/ray", {k: v / npar for k, v in cnt.items()}, flush=True)
bray = 32 + 16 + 64 * cnt["nodes"] / npar + 40 * cnt["tris"] / npar

# ---- timing
res = {}
outs = None
for mode in (la.MODE_CLOSEST, la.MODE_ANY):
    for variant in (0, 1, 2):
        outs = acc.intersect_device(d_org, d_dir, mode=mode, variant=variant)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); acc.intersect_device(d_org, d_dir, out=outs, mode=mode, variant=variant); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        mr = nt / ms / 1e3
        res[(mode, variant)] = mr
        print("mode", mode, "variant", variant, "ms", [round(x, 2) for x in ts], "Mrays/s %.1f" % mr, "alg GB/s %.0f" % (mr * 1e6 * bray / 1e9), flush=True)
print("B_ray", bray)
