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
o = po.Oracle(); o.add_mesh(P, idx); o.build(); exp = o.intersect(org[:300000], dr[:300000], nthreads=64)
configs = [eval(c) for c in sys.argv[2:]] or [{}]
for env in configs:
    for k, v in env.items(): os.environ[k] = str(v)
    acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
    for k in env: os.environ.pop(k)
    out = acc.intersect_device(d_org[:300000].contiguous(), d_dir[:300000].contiguous()); torch.cuda.synchronize()
    ok = all(np.array_equal(out[k].cpu().numpy().view(np.uint32) if k == 0 else out[k].cpu().numpy(), exp[k]) for k in range(4))
    print(env, "parity", ok, "closest %.1f any %.1f" % (timeit(acc, 0, -1), timeit(acc, 1, -1)), flush=True)
    acc.close()
