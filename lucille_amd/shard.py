"""Sharding of the hot path over the GPUs of one node.

The reference decomposes a frame into independent 32x32 buckets pulled from a
queue by worker threads (lucille src/render/render.c:582-710,1043-1105) and its
(compiled-out) MPI design is "every rank renders, rank 0 owns the display"
(render.c:468-514, src/base/parallel.c:101-119).  Here: one process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI), the BVH replicated in
every GPU's HBM (each rank builds the same deterministic tree; no broadcast
needed), units sharded with NO per-ray communication:

  * ray dumps   -> contiguous slices            (ray_slice)
  * image tiles -> tile_id % world == rank      (tiles_of_rank)

and ONE exchange step: the gather of finished tiles to rank 0 (gather_tiles).
"""
import os


def ray_slice(n, rank, world):
    """contiguous [begin, end) of an n-ray dump owned by `rank`"""
    return n * rank // world, n * (rank + 1) // world


def chunk_capacity(n, world, nchunks):
    """rays per chunk when every rank cuts its slice of an n-ray dump into nchunks equal-capacity chunks
    (the largest slice decides; gathers need the same size on every rank)"""
    largest = -(-n // world)
    return max(1, -(-largest // nchunks))


def tile_grid(width, height, tile):
    """row-major list of (x0, y0, w, h) tiles covering the image (ragged edges kept)"""
    out = []
    for y0 in range(0, height, tile):
        for x0 in range(0, width, tile):
            out.append((x0, y0, min(tile, width - x0), min(tile, height - y0)))
    return out


def tiles_of_rank(ntiles, rank, world):
    """interleaved assignment (load balance): tile_id % world == rank"""
    return list(range(rank, ntiles, world))


def init_process_group(backend=None):
    """env:// rendezvous as launched by torch.distributed.run; returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # N ranks on one host: each builds its BVH replica with its share of the cores, not all of them
        # (lh_accel_commit reads LH_BUILD_THREADS when build_threads <= 0)
        os.environ.setdefault("LH_BUILD_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LH_DEVICE_OVERRIDE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_bytes(buf, dst_list, async_op=False):
    """the exchange step: every rank's `buf` (same size everywhere) lands in rank 0's dst_list[r]
    (None on the other ranks).  RCCL: seven point-to-point transfers into rank 0 (xGMI is a full mesh:
    they run in parallel, one link each), enqueued behind the work already on the current stream;
    with async_op the caller keeps tracing the next chunk while this one is on the links.
    gloo (CPU tests, 2 ranks on one GPU): staged through the host, synchronous.
    Returns a handle for wait()."""
    import torch.distributed as dist
    if dist.get_world_size() == 1:
        return None
    if dist.get_backend() == "nccl":
        return dist.gather(buf, dst_list, dst=0, async_op=async_op)
    hb = buf.cpu() if buf.is_cuda else buf
    dist.gather(hb, dst_list, dst=0)
    return None


def wait(work):
    if work is not None:
        work.wait()


def gather_slabs(slab, rank, world):
    """every rank's slab tensor (same shape everywhere) -> list of `world` tensors on rank 0, None elsewhere.
    The display owner is rank 0 (the reference's compiled-out MPI design: "everyone renders, rank 0 owns the display",
    render.c:468-514): a GATHER -- seven point-to-point xGMI transfers landing on rank 0 in parallel -- not an
    all-gather whose ring would carry every slab past every GPU."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [slab]
    if dist.get_backend() != "nccl" and slab.is_cuda:          # gloo (tests): stage through the host
        hp = slab.cpu(); ho = [torch.empty_like(hp) for _ in range(world)] if rank == 0 else None
        dist.gather(hp, ho, dst=0)
        return [t.to(slab.device) for t in ho] if rank == 0 else None
    out = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
    dist.gather(slab, out, dst=0)
    return out
