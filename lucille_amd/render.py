"""Frame-level host logic of the AO render: the reference's ri_render_frame
(lucille src/render/render.c:317-369) re-expressed as tile batches on the device and
sharded in image space over the GPUs of a node.

  render_ao_frame(acc, cam, ...)   one process: all tiles of the frame (or this rank's)
  render_ao_frame_sharded(...)     torch.distributed: tiles `tile_id % world == rank`,
                                   replicated BVH, one gather of tile slabs to rank 0

All pixel arithmetic happens in liblucille_hip.so (lh_render_ao_tile); this module only
decides which tile goes where.
"""
from . import shard


def render_ao_frame(acc, cam, pixel_samples, gather_nsamples, tile=256, seed=1, tile_ids=None):
    """Returns (image [H,W,3] float32 CUDA tensor, stats).  Tiles are tile x tile pixels,
    row-major ids (shard.tile_grid); tile_ids=None renders them all."""
    import torch
    W, H = cam.width, cam.height
    tiles = shard.tile_grid(W, H, tile)
    ids = range(len(tiles)) if tile_ids is None else tile_ids
    dev = torch.device("cuda", acc.device)
    img = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    tot = {"primary_rays": 0, "primary_hits": 0, "ao_rays": 0, "ao_occluded": 0}
    for tid in ids:
        x0, y0, w, h = tiles[tid]
        rgb, st = acc.render_ao_tile(cam, x0, y0, w, h, pixel_samples, gather_nsamples, seed=seed)
        img[H - (y0 + h):H - y0, x0:x0 + w] = rgb       # bucket_write's y flip (render.c:962-964)
        for k in tot:
            tot[k] += st[k]
    return img, tot


def render_ao_frame_sharded(acc, cam, pixel_samples, gather_nsamples, rank, world, tile=256, seed=1):
    """Each rank renders its interleaved tiles; one gather of equal-sized tile slabs
    assembles the frame on rank 0 (None elsewhere).  Returns (image|None, local stats)."""
    import torch
    W, H = cam.width, cam.height
    tiles = shard.tile_grid(W, H, tile)
    mine = shard.tiles_of_rank(len(tiles), rank, world)
    dev = torch.device("cuda", acc.device)
    slab = torch.zeros((len(mine), tile * tile * 3), dtype=torch.float32, device=dev)
    tot = {"primary_rays": 0, "primary_hits": 0, "ao_rays": 0, "ao_occluded": 0}
    for k, tid in enumerate(mine):
        x0, y0, w, h = tiles[tid]
        rgb, st = acc.render_ao_tile(cam, x0, y0, w, h, pixel_samples, gather_nsamples, seed=seed)
        t = torch.zeros((tile, tile, 3), dtype=torch.float32, device=dev)
        t[:h, :w] = rgb
        slab[k] = t.view(-1)
        for kk in tot:
            tot[kk] += st[kk]
    img = assemble(slab, W, H, tile, rank, world)
    return img, tot


def assemble(slab, W, H, tile, rank, world):
    """the exchange step + placement with the reference's y flip"""
    import torch
    import torch.distributed as dist
    tiles = shard.tile_grid(W, H, tile)
    per_rank = (len(tiles) + world - 1) // world
    pad = torch.zeros((per_rank, tile * tile * 3), dtype=slab.dtype, device=slab.device)
    pad[:slab.shape[0]] = slab
    if world > 1:
        # the display owner is rank 0 (the reference's compiled-out MPI design: "everyone renders, rank 0
        # owns the display", render.c:468-514): a GATHER -- seven point-to-point xGMI transfers landing on
        # rank 0 in parallel -- not an all-gather whose ring would carry every slab past every GPU
        if dist.get_backend() != "nccl" and pad.is_cuda:      # gloo (tests): stage through the host
            hp = pad.cpu(); ho = [torch.empty_like(hp) for _ in range(world)] if rank == 0 else None
            dist.gather(hp, ho, dst=0)
            out = [t.to(pad.device) for t in ho] if rank == 0 else None
        else:
            out = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
            dist.gather(pad, out, dst=0)
    else:
        out = [pad]
    if rank != 0:
        return None
    img = torch.zeros((H, W, 3), dtype=slab.dtype, device=slab.device)
    for r in range(world):
        for k, tid in enumerate(shard.tiles_of_rank(len(tiles), r, world)):
            x0, y0, w, h = tiles[tid]
            t = out[r][k].view(tile, tile, 3)
            img[H - (y0 + h):H - y0, x0:x0 + w] = t[:h, :w]
    return img


def render_pt_frame_sharded(acc, cam, spp, rank, world, tile=256, spp_chunk=16, **kw):
    """Path-traced frame (BASELINE config 4): tiles `tile_id % world == rank`, samples in passes
    of `spp_chunk` per tile (bounded device memory), one gather of tile slabs.
    Returns (image on rank 0 | None, local stats)."""
    import torch
    W, H = cam.width, cam.height
    tiles = shard.tile_grid(W, H, tile)
    mine = shard.tiles_of_rank(len(tiles), rank, world)
    dev = torch.device("cuda", acc.device)
    slab = torch.zeros((len(mine), tile * tile * 3), dtype=torch.float32, device=dev)
    tot = {"paths": 0, "rays": 0}
    for k, tid in enumerate(mine):
        x0, y0, w, h = tiles[tid]
        out = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        for s0 in range(0, spp, spp_chunk):
            _, st = acc.render_pt_tile(cam, x0, y0, w, h, s0, min(spp_chunk, spp - s0), spp, out=out, **kw)
            tot["paths"] += st["paths"]; tot["rays"] += st["rays"]
        t = torch.zeros((tile, tile, 3), dtype=torch.float32, device=dev)
        t[:h, :w] = out
        slab[k] = t.view(-1)
    return assemble(slab, W, H, tile, rank, world), tot
