import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lucille_amd as la
from oracle import pyoracle as po
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
P, idx, org, dr = po.soup(1000, 1000, 0.05)
acc = la.HipAccel(0); acc.add_mesh(P, idx); acc.commit()
if mode in ("a", "b"):
    o = torch.from_numpy(org).cuda(); d = torch.from_numpy(dr).cuda()
    out = acc.intersect_device(o, d); torch.cuda.synchronize()
if mode == "h":
    acc.intersect_host(org, dr)
if mode != "b":
    acc.close()
print("end of script", mode, flush=True)
