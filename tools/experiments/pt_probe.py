"""PT leg alone (for rocprofv3 --kernel-trace --stats): python tools/pt_probe.py [size] [spp]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import lucille_amd as la
import bench
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
r = bench.pt_frame_leg(la, 0, 0, 1, size, spp, torch.device("cuda:0"))
print(r)
