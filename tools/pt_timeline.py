"""round 4: the kernel timeline of one path-traced config-4 frame (2048^2 x 256 spp as one pass) from a rocprofv3 --kernel-trace csv:
rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/pt_timeline.py run [world] ; python tools/pt_timeline.py show DIR/.../kernel_trace.csv"""
import os, sys, csv, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import numpy as np, torch
    import lucille_amd as la
    from lucille_amd import render
    g = np.load(os.path.join(ROOT, "tests", "golden", "ao_ps.npz"))
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        acc.add_mesh(g["pos%d" % k], g["idx%d" % k])
        if ("nrm%d" % k) in g.files: acc.set_normals(k, g["nrm%d" % k], int(g["two_side%d" % k]))
    acc.commit()
    c = g["camera"]; cam = la.Camera.make(2048, 2048, c[16], c[:16], int(c[19]))
    import time
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        world = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # > 1: rank 0's share of the frame (its interleaved 4-line bands as one pass)
        if world == 1:
            img, st = render.render_pt_frame_sharded(acc, cam, 256, 0, 1, tile=2048, spp_chunk=256, kd=0.8, env=(1.0, 1.0, 1.0), max_vertices=8, seed=7)
        else:
            acc.set_environment((1.0, 1.0, 1.0), None)
            out = torch.zeros((2048 // 4 // world, 4, 2048, 3), dtype=torch.float32, device="cuda")
            acc.render_pt_bands(cam, 0, 4, 4 * world, 2048 // 4 // world, 0, 256, 256, max_vertices=8, override=la.Material.make(kd=(0.8,) * 3), seed=7, out=out)
        torch.cuda.synchronize(); print("frame ms %.2f" % ((time.perf_counter() - t0) * 1e3), flush=True)
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    k0 = [k for k, r in enumerate(rows) if "k_pt_begin" in r["Kernel_Name"]][-1]
    t0 = int(rows[k0]["Start_Timestamp"]); prev_end = t0
    for r in rows[k0:]:
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"]); s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-46s start %8.3f end %8.3f dur %7.3f" % ((m.group(1) if m else r["Kernel_Name"])[:46], (s_ - t0) / 1e6, (e_ - t0) / 1e6, (e_ - s_) / 1e6))
