"""Round 6: would two half-batches of a rank's share in flight pay?  Two accelerators (two replicas of the config-5 scene on one device, each with its
own scratch and streams), rank 0's bands of 8 cut in two, the halves rendered by two host threads -- started together, or the second one later (the
second half's camera stage would then fall into the first half's AO launch as it ends).  Against the share as ONE batch.
python tools/experiments/two_halves_probe.py [rank]

(The first version of this probe -- profiles/r06_two_halves_probe.txt, 9.0-9.6 ms against 7.32 -- launched both halves on torch's current stream, which is
the SAME default stream in both threads: the halves ran one after the other by stream order.  This version gives each half a stream of its own, created at
the lowest priority: a pool of hardware queues nobody else is in -- streams of one priority share four in-order queues, profiles/r06_hostpath.txt.)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
import lucille_amd as la
from lucille_amd import render, scenes, shard
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
lo_, hi_ = C.c_int(0), C.c_int(0); hip.hipDeviceGetStreamPriorityRange(C.byref(lo_), C.byref(hi_))
PRIO = {"low": lo_.value, "normal": 0, "high": hi_.value}[os.environ.get("HALVES_PRIORITY", "low")]
streams = []
for _ in range(2):
    h = C.c_void_p(); assert hip.hipStreamCreateWithPriority(C.byref(h), 1, PRIO) == 0; streams.append(h.value)
size, tess, ns, world = 4096, 8, 64, 8
g = np.load(os.path.join(ROOT, "tests", "golden", "ao_c1.npz"))
accs = []
for _ in range(2):
    acc = la.HipAccel(0)
    for k in range(int(g["ngeoms"])):
        P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], tess); acc.add_mesh(P, I); del P, I
    acc.commit(); accs.append(acc)
c = g["camera"]; cam = la.Camera.make(size, size, c[16], c[:16], int(c[19]))
brow, y0s = render.bands_for(size, world)
mine = [y0s[b] for b in shard.bands_of_rank(len(y0s), rank, world)]
half = len(mine) // 2
out = torch.zeros((len(mine), brow, size, 3), dtype=torch.float32, device="cuda")
ref = torch.zeros_like(out)
def one(): accs[0].render_ao_bands(cam, mine, brow, 1, ns, seed=1, out=ref)
one(); torch.cuda.synchronize(); ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); one(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("rank %d of 8, one batch of %d bands: %.2f ms" % (rank, len(mine), min(ts)), flush=True)
for k, (lo, hi) in enumerate(((0, half), (half, len(mine)))):          # warm both accelerators on their halves
    accs[k].render_ao_bands(cam, mine[lo:hi], brow, 1, ns, seed=1, out=out[lo:hi], stream=streams[k])
torch.cuda.synchronize()
for k, (lo, hi) in enumerate(((0, half), (half, len(mine)))):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); accs[k].render_ao_bands(cam, mine[lo:hi], brow, 1, ns, seed=1, out=out[lo:hi], stream=streams[k]); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("half %d alone on its stream: %.2f ms" % (k, min(ts)), flush=True)
for delay_ms in (0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5):
    best = 1e9
    for _ in range(5):
        bar = threading.Barrier(3)
        def work(k, lo, hi, d):
            bar.wait()
            if d > 0:
                t_ = time.perf_counter()
                while (time.perf_counter() - t_) * 1e3 < d: pass
            accs[k].render_ao_bands(cam, mine[lo:hi], brow, 1, ns, seed=1, out=out[lo:hi], stream=streams[k])
        th = [threading.Thread(target=work, args=(0, 0, half, 0.0)), threading.Thread(target=work, args=(1, half, len(mine), delay_ms))]
        for t in th: t.start()
        torch.cuda.synchronize(); bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print("two halves, the second started %.1f ms later: %.2f ms   equal to one batch: %s" % (delay_ms, best, bool(torch.equal(out, ref))), flush=True)
