import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
P, idx, org, dr = po.soup(1000000, nt)
d_org = torch.from_numpy(org).cuda(); d_dir = torch.from_numpy(dr).cuda()
def timeit(acc, mode, variant, reps=3):
    outs = acc.intersect_device(d_org, d_dir, mode=mode, variant=variant); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); acc.intersect_device(d_org, d_dir, out=outs, mode=mode, variant=variant); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return nt / min(ts) / 1e3
o = po.Oracle(); o.add_mesh(P, idx); o.build(); exp = o.intersect(org[:500000], dr[:500000], nthreads=64)
for env in ({}, {"LH_BVH_CT": 0.5}):
    for m in (24, 40, 48, 56, 60):
        os.environ["LH_MIN_ACTIVE"] = str(m)
        for k, v in env.items(): os.environ[k] = str(v)
        acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
        for k in env: os.environ.pop(k)
        out = acc.intersect_device(d_org[:500000].contiguous(), d_dir[:500000].contiguous(), variant=3); torch.cuda.synchronize()
        ok = all(np.array_equal(out[k].cpu().numpy().view(np.uint32) if k == 0 else out[k].cpu().numpy(), exp[k]) for k in range(4))
        for g in (1024, 1280):
            acc.set_grid(g)
            print(env, "min_active", m, "grid", g, "parity", ok, "v2 closest %.1f any %.1f | v3 closest %.1f any %.1f" % (timeit(acc, 0, 2), timeit(acc, 1, 2), timeit(acc, 0, 3), timeit(acc, 1, 3)), flush=True)
        acc.close()
