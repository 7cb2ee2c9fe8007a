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
for fmt in ("f32", "q16"):
    for tb, m in ((4, 16), (4, 8), (2, 12), (8, 12), (1, 12)):
        os.environ["LH_MIN_ACTIVE"] = str(m); os.environ["LH_TRI_BATCH"] = str(tb); os.environ["LH_NODE_FORMAT"] = fmt
        acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
        oks = []
        for var in (2, 4):
            out = acc.intersect_device(d_org[:500000].contiguous(), d_dir[:500000].contiguous(), variant=var); torch.cuda.synchronize()
            oks.append(all(np.array_equal(out[k].cpu().numpy().view(np.uint32) if k == 0 else out[k].cpu().numpy(), exp[k]) for k in range(4)))
            occ = acc.intersect_device(d_org[:500000].contiguous(), d_dir[:500000].contiguous(), mode=1, variant=var)[0]; torch.cuda.synchronize()
            oks.append(np.array_equal(occ.cpu().numpy().astype(bool), exp[0] != po.MISS))
        _, cnt = acc.intersect_device(d_org[:2000000].contiguous(), d_dir[:2000000].contiguous(), variant=4, counters=True)
        print(fmt, "tri_batch", tb, "min_active", m, "parity", oks, "nodes/ray %.1f tris/ray %.2f" % (cnt["nodes"] / 2e6, cnt["tris"] / 2e6),
              "v2 closest %.1f any %.1f | v4 closest %.1f any %.1f" % (timeit(acc, 0, 2), timeit(acc, 1, 2), timeit(acc, 0, 4), timeit(acc, 1, 4)), flush=True)
        acc.close()
