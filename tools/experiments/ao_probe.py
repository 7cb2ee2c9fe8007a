"""AO leg alone (for rocprofv3 --kernel-trace --stats): python tools/ao_probe.py [size] [samples] [tess]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lucille_amd as la
import bench
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tess = int(sys.argv[3]) if len(sys.argv) > 3 else 6
print(bench.ao_frame_leg(la, acc_device=0, rank=0, world=1, size=size, nsamples=ns, steps=3, dev=torch.device("cuda:0"), tess=tess))
