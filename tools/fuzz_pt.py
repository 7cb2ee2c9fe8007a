"""Randomised sweep of the path tracer against oracle/lucille_oracle_pt.c (the one-path-at-a-time restatement of lh_pt.h): the two
example scenes, random per-mesh materials (diffuse / mirror / glass mixes), constant and light-probe environments, both weightings,
random tiles, sample ranges, vertex limits and seeds: rays traced, paths and longest path equal, the frame within 1e-6.
python tools/fuzz_pt.py [seed] [rounds]"""
import os, sys
os.environ.setdefault("LH_POISON_OUTPUTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lucille_amd as la
from tests.test_gpu_ao import load_case

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
cases = {k: load_case(k) for k in ("ao_ps", "ao_c1")}
def mat10(m): return [m.kd[0], m.kd[1], m.kd[2], m.ks[0], m.ks[1], m.ks[2], m.kt[0], m.kt[1], m.kt[2], m.ior]
rays = 0
for r in range(rounds):
    c = cases["ao_ps" if r % 2 == 0 else "ao_c1"]
    acc, cam, o, ocam, g = c["acc"], c["cam"], c["oracle"], c["ocam"], c["g"]
    W, H = cam.width, cam.height; nm = int(g["ngeoms"])
    mats = []
    for k in range(nm):
        wgt = rng.dirichlet([1.0, 0.5, 0.5]) * rng.uniform(0.3, 1.0)
        if rng.random() < 0.4: wgt = np.array([rng.uniform(0.2, 1.0), 0.0, 0.0])
        col = lambda s: tuple(float(x) for x in np.clip(s * rng.uniform(0.6, 1.0, 3), 0.0, 1.0))
        mats.append(la.Material.make(kd=col(wgt[0]), ks=col(wgt[1]), kt=col(wgt[2]), ior=float(rng.choice([1.0, 1.33, 1.5, 2.4]))))
    envmap = rng.uniform(0.05, 3.0, (int(rng.integers(2, 40)), int(rng.integers(2, 60)), 4)).astype(np.float32) if rng.random() < 0.5 else None
    env = tuple(float(x) for x in rng.uniform(0.2, 2.0, 3))
    w = int(rng.integers(1, W + 1)); h = int(rng.integers(1, H + 1)); x0 = int(rng.integers(0, W - w + 1)); y0 = int(rng.integers(0, H - h + 1))
    tot = int(rng.integers(1, 33)); s0 = int(rng.integers(0, tot)); cnt = int(rng.integers(1, tot - s0 + 1))
    mv = int(rng.choice([2, 3, 5, 8, 12, 40])); flags = int(rng.choice([0, la.PT_REFERENCE_WEIGHTS])); sd = int(rng.integers(0, 1 << 31))
    try:
        acc.set_environment(env, envmap)
        for k, m in enumerate(mats): acc.set_material(k, m)
        img, st = acc.render_pt_tile2(cam, x0, y0, w, h, s0, cnt, tot, max_vertices=mv, flags=flags, seed=sd)
        exp, est, per = o.render_pt(ocam, x0, y0, w, h, s0, cnt, tot, max_vertices=mv, materials=[mat10(m) for m in mats], env_rgb=env, env_map=envmap, ref_weights=flags, seed=sd)
    finally:
        acc.set_environment((1.0, 1.0, 1.0), None); acc.set_material(la.ALL_MESHES, la.Material.make())
    got = img.cpu().numpy()
    ok = np.abs(got - exp) <= 1e-6 * np.maximum(1.0, np.abs(exp))
    if st != est or not ok.all():
        print("MISMATCH round %d: stats %s vs %s, pixels off %d, max %g; tile %s spp %s mv %d flags %d seed %d probe %s" % (r, st, est, int((~ok).sum()), float(np.abs(got - exp).max()), (x0, y0, w, h), (s0, cnt, tot), mv, flags, sd, envmap is not None)); sys.exit(1)
    rays += st["rays"]
print("%d rays over %d passes: ray counts, path counts and longest paths equal to the oracle's, frames within 1e-6" % (rays, rounds))
