"""bench.py's AO leg alone: python tools/ao_leg_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lucille_amd as la
import bench
r = bench.ao_frame_leg(la, acc_device=0, rank=0, world=1, size=4096, nsamples=64, steps=2, dev=torch.device("cuda:0"), tess=8)
print(r["frame_ms"], r["device_build"])
