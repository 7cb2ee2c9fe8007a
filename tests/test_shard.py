"""N>1 path on CPU: world_size-2 gloo processes exercise the host logic of the sharding -- slices, chunks, the
placement of gathered tile slabs on the display owner -- that lucille_amd.render / bench.py run on top of the
C-ABI exchange (lh_dist_*: RCCL on a GPU node; tests/test_gpu_dist.py, test_gpu_shard.py cover that side)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lucille_amd import render, shard


def test_ray_slices_partition_exactly():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            sl = [shard.ray_slice(n, r, world) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in sl) - min(e - b for b, e in sl) <= 1


def test_tiles_cover_image_once():
    for (W, H, T) in ((256, 256, 64), (100, 70, 32), (33, 257, 16), (1024, 1024, 256)):
        tiles = shard.tile_grid(W, H, T)
        cover = np.zeros((H, W), int)
        for x0, y0, w, h in tiles:
            cover[y0:y0 + h, x0:x0 + w] += 1
        assert (cover == 1).all()
        for world in (1, 2, 8):
            ids = sorted(sum((shard.tiles_of_rank(len(tiles), r, world) for r in range(world)), []))
            assert ids == list(range(len(tiles)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = shard.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    tiles = shard.tile_grid(W, H, T)
    mine = shard.tiles_of_rank(len(tiles), rank, world)
    # "render": pixel value encodes its frame position (bottom-up y like the renderer's tiles)
    slab = torch.zeros((len(mine), T * T * 3))
    for k, tid in enumerate(mine):
        x0, y0, w_, h_ = tiles[tid]
        t = torch.zeros((T, T, 3))
        for ly in range(h_):
            for lx in range(w_):
                py = y0 + (h_ - 1 - ly)           # tile row 0 = top row of the tile in image orientation
                t[ly, lx] = torch.tensor([x0 + lx, py, rank], dtype=torch.float32)
        slab[k] = t.view(-1)
    img = render.assemble(slab, W, H, T, rank, world)
    # max-over-ranks timing reduction as bench.py does
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert tt.item() == world
    if rank == 0:
        q.put(img.numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H,T", [(64, 48, 16), (50, 35, 16)])
def test_tile_gather_world2_gloo(W, H, T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    assert np.array_equal(img[..., 0], xs)
    assert np.array_equal(img[..., 1], H - 1 - ys)          # image row r shows frame line H-1-r (bucket_write's flip)
    tiles = shard.tile_grid(W, H, T)
    owner = np.zeros((H, W))
    for tid, (x0, y0, w_, h_) in enumerate(tiles):
        owner[H - (y0 + h_):H - y0, x0:x0 + w_] = tid % 2
    assert np.array_equal(img[..., 2], owner)


def _dump_worker(rank, world, port, n, nchunks, q):
    """the strong-scaling ray-dump exchange of bench.py on gloo: every rank produces 'hit records' for its
    slice of one dump (here: a function of the absolute ray id), chunk by chunk, and rank 0 gathers them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    shard.init_process_group(backend="gloo")
    b0, b1 = shard.ray_slice(n, rank, world)
    per = shard.chunk_capacity(n, world, nchunks)
    got = [torch.zeros((world, per * 4), dtype=torch.uint8) for _ in range(nchunks)] if rank == 0 else None
    for c in range(nchunks):
        lo, hi = b0 + c * per, min(b1, b0 + (c + 1) * per)
        rec = torch.full((per,), -1, dtype=torch.int32)
        if hi > lo:
            rec[:hi - lo] = torch.arange(lo, hi, dtype=torch.int32) * 3 + 1
        shard.gather_bytes(rec.view(torch.uint8), got[c] if rank == 0 else None)
    if rank == 0:
        out = np.full(n, -7, np.int64)
        for r in range(world):
            rb0, rb1 = shard.ray_slice(n, r, world)
            for c in range(nchunks):
                lo, hi = rb0 + c * per, min(rb1, rb0 + (c + 1) * per)
                if hi > lo:
                    out[lo:hi] = got[c][r].view(torch.int32).numpy()[:hi - lo]
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,nchunks", [(1000, 4), (1003, 3), (5, 4)])
def test_ray_dump_gather_world2_gloo(n, nchunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dump_worker, args=(r, 2, port, n, nchunks, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(out, np.arange(n) * 3 + 1)


def test_serpentine_bands_cover_once_and_cancel_a_slope():
    """shard.bands_of_rank: every band exactly once, band_owner is its inverse, and a cost that changes linearly down the image
    is shared out evenly (plain interleaving leaves the last rank the whole slope)"""
    for nbands, world in ((128, 8), (64, 8), (37, 8), (5, 8), (1024, 3), (16, 1), (7, 2)):
        got = [shard.bands_of_rank(nbands, r, world) for r in range(world)]
        assert sorted(sum(got, [])) == list(range(nbands))
        for r in range(world):
            assert got[r] == sorted(got[r]) and len(got[r]) <= (nbands + world - 1) // world
            for k, b in enumerate(got[r]):
                assert shard.band_owner(b, world) == (r, b // world) and (len(got[r]) <= k or got[r][k] == b)
    cost = np.arange(128, dtype=np.float64) + 50.0                  # a steady slope down the image
    serp = [cost[shard.bands_of_rank(128, r, 8)].sum() for r in range(8)]
    plain = [cost[shard.tiles_of_rank(128, r, 8)].sum() for r in range(8)]
    assert max(serp) == min(serp) and max(plain) - min(plain) == 7 * 16
    # ... and most of a curvature: the shifted groups (rank order, reversed, shifted by half the world, shifted and reversed)
    quad = (np.arange(128, dtype=np.float64) - 40.0) ** 2
    q4 = [quad[shard.bands_of_rank(128, r, 8)].sum() for r in range(8)]
    q2 = [quad[[g * 8 + (r if g % 2 == 0 else 7 - r) for g in range(16)]].sum() for r in range(8)]      # the plain serpentine
    assert (max(q4) - min(q4)) * 2 < max(q2) - min(q2)


def _band_worker(rank, world, port, W, H, rows, q):
    """the AO frame's exchange step on gloo: every rank fills the slabs of ITS serpentine bands with a function of the frame
    position, one gather, rank 0 places them (render.assemble_shards: the regular fast path and the clipped-last-band path)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    shard.init_process_group(backend="gloo")
    _, y0s = render.bands_for(H, world, rows)
    mine = shard.bands_of_rank(len(y0s), rank, world)
    per = (len(y0s) + world - 1) // world
    slab = torch.zeros((per, rows * W * 3))
    for k, b in enumerate(mine):
        y0 = y0s[b]; h = min(rows, H - y0)
        t = torch.zeros((rows, W, 3))
        for ly in range(h):                          # a clipped band keeps its lines at the BOTTOM of its slab
            line = y0 + (h - 1 - ly)
            t[rows - h + ly, :, 0] = torch.arange(W, dtype=torch.float32); t[rows - h + ly, :, 1] = float(line); t[rows - h + ly, :, 2] = float(rank)
        slab[k] = t.view(-1)
    shards = [(0, y0, W, min(rows, H - y0)) for y0 in y0s]
    img = render.assemble_shards(slab, shards, W, H, rank, world, stride_rows=rows, serpentine=True)
    # a grey frame travels as ONE float per pixel (channel 1 here: the frame line) and comes back as three
    mono = slab.view(per, rows * W, 3)[:, :, 1].contiguous()
    img1 = render.assemble_shards(mono, shards, W, H, rank, world, stride_rows=rows, serpentine=True, channels=1)
    # ... and a one-sample AO frame as ONE BYTE per pixel, the numerator of (N - occluded) / N (here: line % 17 over 16)
    cnt = (mono.to(torch.int64) % 17).to(torch.uint8)
    img8 = render.assemble_shards(cnt, shards, W, H, rank, world, stride_rows=rows, serpentine=True, channels=1, count_of=16)
    if rank == 0:
        assert img1.shape == (H, W, 3) and bool((img1 == img[..., 1:2]).all())
        want = ((img[..., 1:2].to(torch.int64) % 17).to(torch.float64) / 16.0).to(torch.float32)
        assert img8.dtype == torch.float32 and img8.shape == (H, W, 3) and bool((img8 == want).all())
        q.put(img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H,rows,world", [(24, 64, 4, 2), (24, 70, 8, 2), (16, 96, 4, 3)])
def test_serpentine_band_gather_gloo(W, H, rows, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, W, H, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    assert np.array_equal(img[..., 0], xs)
    assert np.array_equal(img[..., 1], H - 1 - ys)          # image row r shows frame line H - 1 - r (bucket_write's flip)
    owner = np.array([shard.band_owner((H - 1 - r) // rows, world)[0] for r in range(H)], np.float32)
    assert np.array_equal(img[..., 2], np.repeat(owner[:, None], W, 1))
