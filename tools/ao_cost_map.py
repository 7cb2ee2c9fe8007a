"""round 5: a per-pixel cost map of the config-5 AO frame (sequential-walk node visits: deterministic), to see WHERE the
expensive 1024-ray ranges of the fused AO launch are and what predicts them.  Every `step`-th line of the 4096^2 frame, every
pixel of it: node visits of the camera ray (closest hit) and of its 64 AO rays (any-hit) -> gpurun_out/ao_cost_map.npz
  python tools/ao_cost_map.py [step] [size] [tess]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import scenes
step = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
tess = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ns = 64
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I)
acc.commit()
acc.set_param("ao_fused", 0)          # the AO rays materialised: scratch buffers 8 (origins) and 9 (directions)
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
rows = list(range(0, size, step))
cam_cost = np.zeros((len(rows), size), np.uint16); ao_cost = np.zeros((len(rows), size), np.float32); ao_tri = np.zeros((len(rows), size), np.float32)
hit = np.zeros((len(rows), size), np.bool_); ao_max = np.zeros((len(rows), size), np.uint16)
hip = __import__("ctypes").CDLL("libamdhip64.so")
import ctypes as C
def dev_view(which, dtype, width):
    p = C.c_void_p(); n = C.c_size_t()
    assert acc.L.lh_render_scratch(acc.h, which, C.byref(p), C.byref(n)) == 0
    return p.value, n.value
t0 = time.time()
for ri, y in enumerate(rows):
    o, d = acc.primary_rays(cam, 0, y, size, 1, 1)
    dg = acc.intersect_diag_device(o, d, la.MODE_CLOSEST)
    cam_cost[ri] = dg[:, 0].cpu().numpy().astype(np.uint16)
    img, st = acc.render_ao_tile(cam, 0, y, size, 1, 1, ns, seed=1)
    torch.cuda.synchronize()
    slot = torch.from_numpy(acc.scratch(6, np.uint32, 1).astype(np.int64))          # slot of every sample (0xFFFFFFFF: miss)
    nh = st["primary_hits"]
    if nh == 0:
        continue
    po, n_ao = dev_view(8, np.float64, 3); pd, _ = dev_view(9, np.float64, 3)
    assert n_ao == nh * ns, (n_ao, nh)
    ao_o = torch.empty((n_ao, 3), dtype=torch.float64, device="cuda"); ao_d = torch.empty_like(ao_o)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(ao_o.data_ptr()), C.c_void_p(po), n_ao * 24, 3) == 0
    assert hip.hipMemcpy(C.c_void_p(ao_d.data_ptr()), C.c_void_p(pd), n_ao * 24, 3) == 0
    dg = acc.intersect_diag_device(ao_o, ao_d, la.MODE_ANY).view(nh, ns, 4)
    per_slot = dg[:, :, 0].float().mean(1).cpu().numpy(); per_slot_t = dg[:, :, 2].float().mean(1).cpu().numpy()
    per_max = dg[:, :, 0].max(1).values.cpu().numpy()
    m = (slot != 0xFFFFFFFF).numpy()
    hit[ri] = m
    ao_cost[ri, m] = per_slot[slot[m].numpy()]; ao_tri[ri, m] = per_slot_t[slot[m].numpy()]; ao_max[ri, m] = np.minimum(per_max[slot[m].numpy()], 65535)
    if ri % 64 == 0:
        print("line %d: hits %d, camera nodes %.1f, AO nodes / ray %.2f (max of a pixel's mean %.1f), %.1f s" % (y, nh, cam_cost[ri].mean(), per_slot.mean(), per_slot.max(), time.time() - t0), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "ao_cost_map.npz"), rows=np.array(rows), cam_cost=cam_cost, ao_cost=ao_cost.astype(np.float16), ao_tri=ao_tri.astype(np.float16), ao_max=ao_max, hit=hit)
print("done: %d lines in %.1f s; AO node visits per ray %.2f over %d hit pixels" % (len(rows), time.time() - t0, ao_cost[hit].mean(), hit.sum()))
