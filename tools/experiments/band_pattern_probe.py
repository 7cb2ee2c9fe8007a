"""Round 6: which way of dealing 16-line bands to 8 ranks balances the config-5 frame best?  Every candidate rule (a function band -> rank that
looks at nothing but the band's number) is measured the way tools/predict8.py solo-ao does: every rank's batch alone on the GPU, best of 4.
python tools/experiments/band_pattern_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
size, tess, ns, W = 4096, 8, 64, 8
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); del P, I
acc.commit()
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
brow, y0s = render.bands_for(size, W); nb = len(y0s)
slab = torch.zeros((nb // W + 1, brow, size, 3), dtype=torch.float32, device="cuda")
BITREV = [0, 4, 2, 6, 1, 5, 3, 7]
def four(gp, r):
    s = (r + W // 2) % W
    return (r, W - 1 - r, s, W - 1 - s)[gp % 4]
RULES = {
    "four patterns (the product's)": four,
    "plain interleave": lambda gp, r: r,
    "serpentine": lambda gp, r: r if gp % 2 == 0 else W - 1 - r,
    "cyclic latin square: (r + g) mod 8": lambda gp, r: (r + gp) % W,
    "latin square, every second cycle reversed": lambda gp, r: ((r + gp) % W) if (gp // W) % 2 == 0 else (W - 1 - (r + gp) % W),
    "latin square in bit-reversed steps: (r + bitrev(g mod 8)) mod 8": lambda gp, r: (r + BITREV[gp % W]) % W,
    "bit-reversed latin, reversed in odd groups": lambda gp, r: ((r + BITREV[gp % W]) % W) if gp % 2 == 0 else (W - 1 - (r + BITREV[gp % W]) % W),
    "multiplicative: (3 r + 5 g) mod 8": lambda gp, r: (3 * r + 5 * gp) % W,
}
for name, rule in RULES.items():
    per = []; hits = []
    for r in range(W):
        mine = [y0s[gp * W + rule(gp, r)] for gp in range(nb // W)]
        assert len(set(mine)) == len(mine)
        out = slab[:len(mine)]
        acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize(); ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); _, st = acc.render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        per.append(min(ts)); hits.append(st["primary_hits"] // 1000)
    # every band exactly once over the ranks
    allb = sorted(gp * W + rule(gp, r) for r in range(W) for gp in range(nb // W)); assert allb == list(range(nb))
    print("%-66s busiest %.2f  mean %.2f  least %.2f  | %s | hits %s" % (name, max(per), sum(per) / W, min(per), " ".join("%.2f" % x for x in per), " ".join(str(h) for h in hits)), flush=True)
