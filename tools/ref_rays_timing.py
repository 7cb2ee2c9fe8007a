"""BASELINE config 1 (the AO example scene, 256 x 256, 16 AO samples) through the REFERENCE's own renderer and render threads,
every ray through ri_raytrace -> accel->intersect (RI_HIP_RENDER=rays): round 5's host walk (lh_hostwalk.c), the coalesced one-ray device path (LH_HOST_WALK=0) against one launch per call
(LH_COMBINE=0), against rounds 1-3's path (LH_COMBINE=0 LH_SMALL_BATCH=0) and against the reference's CPU BVH.  GPU box.
  python tools/ref_rays_timing.py [threads] [size] [gather]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_rib
from tests.helpers import load_golden
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
gather = int(sys.argv[3]) if len(sys.argv) > 3 else 16
g = load_golden("ao_c1")
c2w = np.asarray(g["camera"][:16]).reshape(4, 4)
w2c = np.linalg.inv(c2w @ np.linalg.inv(np.diag([1.0, 1.0, -1.0, 1.0])))
scene = {"ngeoms": int(g["ngeoms"]), "w2c": w2c, "fov": 45.0}
for k in range(int(g["ngeoms"])):
    scene["pos%d" % k] = g["pos%d" % k]; scene["idx%d" % k] = g["idx%d" % k]
tmp = tempfile.mkdtemp(); sp = os.path.join(tmp, "scene.npz"); np.savez(sp, **scene)
kw = dict(width=size, height=size, gather_nsamples=gather, pixel_samples=1, lib="liblucille_ref_hip.so", record=False, nthreads=threads)
nrays = None
def run(tag, method, env, **over):
    k = dict(kw); k.update(over)
    t0 = time.time()
    o = ref_rib.render_scene_subprocess(sp, os.path.join(tmp, "".join(ch if ch.isalnum() else "_" for ch in tag) + ".npz"), accel_method=method, env=dict(env, RI_HIP_RENDER="rays"), **k)
    dt = time.time() - t0
    print("%-58s %7.2f s wall (process start, Ri ingest, build, frame; image mean %.4f)" % (tag, dt, float(o["image"].mean())), flush=True)
    return dt, o
DEV = {"LH_HOST_WALK": "0"}          # the device paths of rounds 1-4; round 5's default answers one ray on the calling thread (lh_hostwalk.c)
t_hw, o_hw = run("hip accel, %d threads, host walk (round 5 default)" % threads, 2, {})
t_hw1, o_hw1 = run("hip accel, 1 thread, host walk (round 5 default)", 2, {}, nthreads=1)
t_cpu, _ = run("reference CPU BVH, %d threads" % threads, 1, {})
t_cpu1, _ = run("reference CPU BVH, 1 thread", 1, {}, nthreads=1)
t_new, o_new = run("hip accel, %d threads, device, coalesced (round 4 default)" % threads, 2, DEV)
t_one, o_one = run("hip accel, 1 thread, device, coalesced (batches of one)", 2, DEV, nthreads=1)
small = dict(width=size // 4, height=size // 4)           # 1/16 of the frame: these paths are slow
t_nc, _ = run("hip accel, %d threads, device, LH_COMBINE=0, 1/16 frame" % threads, 2, dict(DEV, LH_COMBINE="0"), **small)
t_r3, _ = run("hip accel, %d threads, device, rounds 1-3 path, 1/16 frame" % threads, 2, dict(DEV, LH_COMBINE="0", LH_SMALL_BATCH="0"), **small)
t_ns, _ = run("hip accel, %d threads, device, coalesced, 1/16 frame" % threads, 2, DEV, **small)
print("coalesced vs rounds 1-3 (1/16 frame, same process overheads): %.1fx; vs one launch per call: %.1fx" % (t_r3 / t_ns, t_nc / t_ns))
# one render thread: the reference's frame is a function of the scene (with more, buckets go to threads as they come and every thread
# has its own MT19937 stream: SURVEY 8c)
print("frames with one render thread: host walk == device path: %s" % bool(np.array_equal(o_hw1["image"], o_one["image"])))
