"""A/B of differently built liblucille_hip.so files (gpurun_variants/*.so), one subprocess per
library, interleaved rounds.  python tools/ab_libs.py [nrays] [rounds]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nr = sys.argv[1] if len(sys.argv) > 1 else "50000000"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
libs = sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "*.so")))
code = ("import sys,os; sys.path.insert(0,%r); import numpy as np, torch; import lucille_amd as la; from oracle import pyoracle as po;"
        "P,idx,org,dr=po.soup(1000000,%s); acc=la.HipAccel(0); acc.add_mesh(P,idx); acc.commit(build='host');"
        "o=torch.from_numpy(org).cuda(); d=torch.from_numpy(dr).cuda();\n"
        "def t(mode):\n"
        "    out=acc.intersect_device(o,d,mode=mode); torch.cuda.synchronize(); ts=[]\n"
        "    for _ in range(4):\n"
        "        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record(); acc.intersect_device(o,d,out=out,mode=mode); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))\n"
        "    return o.shape[0]/min(ts)/1e3\n"
        "print('%%.1f %%.1f' %% (t(0), t(1)))") % (ROOT, nr)
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, LH_LIBRARY=l)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        res[l].append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-300:])
        print(os.path.basename(l), "round", r, res[l][-1], flush=True)
