"""What a frame's two barriers cost between real processes: bench.py brackets a frame with
barrier / render / synchronize / barrier and takes the max over ranks, so a sharded frame pays, beyond its busiest rank's batch
and the gather, (a) the spread with which the ranks LEAVE the first barrier (the last one out starts late) and (b) the latency
of the second barrier.  Measured between `world` processes on one host (gloo, as bench.py's control plane; CLOCK_MONOTONIC is
shared), no GPU involved:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/skew_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
from lucille_amd import shard
if torch.cuda.is_available():
    os.environ.setdefault("LH_DEVICE_OVERRIDE", "0")          # one-GPU box: every rank on device 0 (the shared-memory transport; the host barrier is the same for RCCL)
rank, world, _ = shard.init_process_group()
N = 400


def measure(name, fn):
    enter = np.zeros(N); leave = np.zeros(N)
    for it in range(N + 20):
        time.sleep(0.002 * ((rank * 7 + it) % 5) / 5.0)       # ranks arrive at different times, as after a frame
        t0 = time.monotonic_ns()
        fn()
        t1 = time.monotonic_ns()
        if it >= 20:
            enter[it - 20] = t0; leave[it - 20] = t1
    allv = [None] * world
    dist.all_gather_object(allv, (enter, leave))
    if rank == 0:
        E = np.stack([a for a, _ in allv]); L = np.stack([b for _, b in allv])
        spread = (L.max(0) - L.min(0)) / 1e3                   # us: last rank out minus first rank out
        latency = (L.max(0) - E.max(0)) / 1e3                  # us: last rank out minus last rank in
        p = lambda a, q: float(np.percentile(a, q))
        print("world %d, %s: exit spread p50 %.0f us, p95 %.0f us; latency (last in -> last out) p50 %.0f us, p95 %.0f us; "
              "skew term of a bracketed frame = spread + latency: p50 %.0f us, p95 %.0f us" %
              (world, name, p(spread, 50), p(spread, 95), p(latency, 50), p(latency, 95), p(spread + latency, 50), p(spread + latency, 95)), flush=True)


measure("gloo barrier (torch.distributed)", dist.barrier)
if shard.dist() is not None:
    measure("lh_dist_host_barrier (shared memory)", shard.barrier)
    shard.dist().close()
dist.destroy_process_group()
