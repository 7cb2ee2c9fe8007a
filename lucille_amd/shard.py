"""Sharding of the hot path over the GPUs of one node.

The reference decomposes a frame into independent 32x32 buckets pulled from a
queue by worker threads (lucille src/render/render.c:582-710,1043-1105) and its
(compiled-out) MPI design is "every rank renders, rank 0 owns the display"
(render.c:468-514, src/base/parallel.c:101-119).  Here: one process per GPU
(launched by torch.distributed.run; data through lh_dist_* in the C ABI = RCCL over
xGMI), the BVH replicated in every GPU's HBM (ONE host build on rank 0, then a
broadcast of the flattened arrays), units sharded with NO per-ray communication:

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


def _band_pos(g, rank, world):
    """position of `rank`'s band inside group g (a permutation of range(world) for every g): rank, reversed, shifted by half the
    world, shifted and reversed -- then again"""
    s = (rank + world // 2) % world
    return (rank, world - 1 - rank, s, world - 1 - s)[g % 4]


def bands_of_rank(nbands, rank, world):
    """full-width bands of an AO frame, dealt out in groups of `world`: band g * world + pos is the g-th band of the rank whose
    position in group g is pos (_band_pos: rank order, reversed, shifted by half the world, shifted and reversed, ...).
    Plain interleaving (b % world) hands rank r the band r lines-worth below rank 0's in EVERY group: where the cost of a line
    changes steadily down the image (sky, objects, floor) the last rank carries all of that slope (config 5, 64-line bands: rank 7
    7 % above rank 0).  Rank order + reversed (a serpentine) cancels a linear slope exactly; the two shifted groups take most of
    the curvature as well (a rank's squared positions over four groups: 66 .. 74 of 8 ranks, against 50 .. 98 for the plain
    serpentine: the ranks' camera-ray hits on config 5 at 16-line bands spread 830 .. 897 thousand with the serpentine alone).
    So bands can be tall -- and tall bands are coherent (lh_dist.hip k_place_bands has the same rule)."""
    out = []
    for g in range((nbands + world - 1) // world):
        b = g * world + _band_pos(g, rank, world)
        if b < nbands:
            out.append(b)
    return out


def band_owner(b, world):
    """(rank, index among that rank's bands) of band b under bands_of_rank's rule"""
    g, pos = divmod(b, world)
    for r in range(world):
        if _band_pos(g, r, world) == pos:
            return r, g
    raise AssertionError("not a permutation")


_DIST = None      # this process's lh_dist_t (binding.HipDist) when world > 1
_RCCL_STATUS = "not tried (world 1)"


def rccl_status():
    """how this rank's communicator came up: 'ok', 'shm: ranks share a device', 'shm: asked for', or the error that forced the fallback"""
    return _RCCL_STATUS


def dist():
    """the C-ABI communicator of this rank (RCCL over xGMI; shared memory when ranks share a device), or None at world 1"""
    return _DIST


def init_process_group(backend=None, device=None):
    """env:// rendezvous as launched by torch.distributed.run; returns (rank, world, local_rank).

    torch.distributed is the LAUNCHER only: a gloo group carries the control plane (the 128-byte RCCL id, barriers, the
    max-over-ranks of a timing).  Everything that moves device data -- the scene broadcast, the gather of hit records and
    tile slabs -- goes through lh_dist_* in the C ABI (ncclBroadcast / grouped ncclSend + ncclRecv), the same entry points
    `lsh_hip --rank R --world N` uses without Python.  `backend` is accepted for compatibility and ignored."""
    global _DIST, _RCCL_STATUS
    import torch
    import torch.distributed as tdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not tdist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if world > 1 and _DIST is None and torch.cuda.is_available():
        from . import binding
        dev = int(os.environ.get("LH_DEVICE_OVERRIDE", local if device is None else device))
        devs = [None] * world
        tdist.all_gather_object(devs, dev)
        shared = len(set(devs)) < world                       # one node: equal ordinals = one GPU (tests on a one-GPU box)
        transport = binding.DIST_SHM if shared or os.environ.get("LH_DIST_TRANSPORT") == "shm" else binding.DIST_RCCL
        if shared and os.environ.get("LH_DIST_TRANSPORT") == "rccl" and os.environ.get("LH_RCCL_LIBRARY"):
            transport = binding.DIST_RCCL      # a stand-in library that accepts ranks on one device (tests/mock_rccl): the RCCL branch with peers on a one-GPU box
        _RCCL_STATUS = "ok" if transport == binding.DIST_RCCL else ("shm: ranks share a device" if shared else "shm: asked for (LH_DIST_TRANSPORT)")
        torch.cuda.set_device(dev)
        if transport == binding.DIST_RCCL:
            # RCCL first; if ANY rank cannot bring its communicator up (no librccl, a fabric problem) every rank falls back to the
            # shared-memory transport of lh_dist_* (host staging: slower, same results) instead of dying -- said loudly
            ok, err = 1.0, ""
            ids = [None]
            try:
                ids = [binding.HipDist.unique_id() if rank == 0 else None]
            except Exception as e:                                   # noqa: BLE001 -- whatever dlopen / ncclGetUniqueId raised
                ok, err = 0.0, repr(e)
            flag = torch.tensor([ok], dtype=torch.float64); tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
            if float(flag.item()) > 0.5:
                tdist.broadcast_object_list(ids, src=0)
                try:
                    _DIST = binding.HipDist(rank, world, dev, unique_id=ids[0], transport=binding.DIST_RCCL)
                except Exception as e:                               # noqa: BLE001
                    ok, err = 0.0, repr(e)
                flag = torch.tensor([ok], dtype=torch.float64); tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
            if float(flag.item()) < 0.5:
                if _DIST is not None:
                    _DIST.close(); _DIST = None
                import sys
                print("[lucille_amd] rank %d: RCCL transport unavailable%s -- falling back to the shared-memory transport of lh_dist_*"
                      % (rank, (": " + err) if err else " on another rank"), file=sys.stderr, flush=True)
                transport = binding.DIST_SHM
                _RCCL_STATUS = "shm: RCCL unavailable" + ((": " + err) if err else " on another rank")
        if _DIST is None:
            ids = [os.urandom(128) if rank == 0 else None]
            tdist.broadcast_object_list(ids, src=0)
            _DIST = binding.HipDist(rank, world, dev, unique_id=ids[0], transport=transport)
    return rank, world, local


def commit_shared(acc, add_meshes, rank, world, **commit_kw):
    """ONE host build: rank 0 stages the meshes (add_meshes(acc)) and commits, the flattened scene is broadcast into every
    rank's HBM (lh_dist_broadcast_scene).  -> (info dict, seconds spent in rank 0's commit, seconds in the broadcast)"""
    import time
    t0 = time.perf_counter()
    err = None
    if rank == 0 or world == 1:
        try:
            add_meshes(acc)
            acc.commit(**commit_kw)
        except Exception as e:                  # noqa: BLE001 -- rank 0 still enters the broadcast: its first word tells the peers to give up
            err = e
            if world == 1:
                raise
    t1 = time.perf_counter()
    if world > 1:
        try:
            _DIST.broadcast_scene(acc)
        except Exception:
            if err is not None:
                raise err
            raise
    return acc.info(), t1 - t0, time.perf_counter() - t1


def barrier():
    """all ranks meet.  With a communicator: in shared memory (lh_dist_host_barrier: the ranks of one node, microseconds -- a gloo
    barrier between eight processes costs ~1.3 ms and releases them up to ~0.4 ms apart, tools/skew_probe.py: a tenth of a rank's
    share of a sharded frame); LH_SHARD_BARRIER=gloo or no communicator (the CPU tests): torch.distributed's."""
    if _DIST is not None and os.environ.get("LH_SHARD_BARRIER") != "gloo":
        _DIST.host_barrier()
        return
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        tdist.barrier()


def all_reduce_max(x):
    """max over the ranks of a Python float (control plane)"""
    import torch, torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64); tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(x):
    import torch, torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64); tdist.all_reduce(t)
    return float(t.item())


def all_gather_object(obj):
    """every rank's small Python object, in rank order, on every rank (control plane)"""
    import torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized()):
        return [obj]
    out = [None] * tdist.get_world_size()
    tdist.all_gather_object(out, obj)
    return out


def all_reduce_min(x):
    import torch, torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64); tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
    return float(t.item())


def gather_bytes(buf, dst, stream=None):
    """the exchange step: every rank's `buf` (same size everywhere, CUDA) lands in rank 0's dst[r] (dst: a [world, nbytes]
    CUDA tensor on rank 0, None elsewhere).  RCCL: N - 1 point-to-point transfers into rank 0 in one group (xGMI is a full
    mesh: they run in parallel, one link each), enqueued on `stream` (a torch.cuda.Stream; default: the current one) behind
    the work already there -- the caller keeps tracing the next chunk on its own stream."""
    import torch
    if not buf.is_cuda:
        # the CPU tests of the host logic (world-size-2 gloo, no GPU): the same exchange through torch.distributed
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            tdist.gather(buf, [dst[r] for r in range(tdist.get_world_size())] if dst is not None else None, dst=0)
        return
    if _DIST is None:
        return
    s = stream if stream is not None else torch.cuda.current_stream(buf.device)
    L = _DIST.L
    import ctypes as C
    L.lh_dist_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rc = L.lh_dist_gather(_DIST.h, buf.data_ptr(), buf.numel() * buf.element_size(), dst.data_ptr() if dst is not None else None, s.cuda_stream)
    if rc != 0:
        from . import binding
        raise binding.LucilleHipError("lh_dist_gather: " + L.lh_last_error().decode())


def gather_slabs(slab, rank, world):
    """every rank's slab tensor (same shape everywhere) -> a [world, ...] tensor on rank 0 (indexable like a list), None
    elsewhere.  The display owner is rank 0 (the reference's compiled-out MPI design: "everyone renders, rank 0 owns the
    display", render.c:468-514): a GATHER -- seven point-to-point xGMI transfers landing on rank 0 in parallel -- not an
    all-gather whose ring would carry every slab past every GPU."""
    if world == 1:
        return [slab]
    if not slab.is_cuda:                  # CPU tests of the assembly logic: gloo
        import torch, torch.distributed as tdist
        out = torch.empty((world,) + tuple(slab.shape), dtype=slab.dtype) if rank == 0 else None
        tdist.gather(slab.contiguous(), [out[r] for r in range(world)] if rank == 0 else None, dst=0)
        return out
    return _DIST.gather(slab)
