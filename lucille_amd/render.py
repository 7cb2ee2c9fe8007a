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


def bands_for(width, height, world, per_rank=8, min_rows=16):
    """image-space shards for `world` ranks: full-width bands of equal height, `per_rank` of them for every rank,
    interleaved (band_id % world == rank) so that sky / floor / silhouettes are spread over the ranks.  A band is one
    device batch (one lh_render_ao_tile call: ~10 kernel launches + 2 small read-backs), so per_rank bounds the host
    overhead per frame while the interleave bounds the imbalance.  -> list of (x0, y0, w, h)"""
    n = max(1, world * per_rank)
    rows = max(min_rows, -(-height // n))
    return [(0, y0, width, min(rows, height - y0)) for y0 in range(0, height, rows)]


def render_ao_frame_sharded(acc, cam, pixel_samples, gather_nsamples, rank, world, tile=None, seed=1, per_rank=8):
    """Each rank renders its interleaved shards straight into its slab; ONE gather of the equal-sized slabs assembles
    the frame on rank 0 (None elsewhere).  tile=None: full-width bands (bands_for); an int: square tiles as before.
    Returns (image|None, local stats)."""
    import torch
    W, H = cam.width, cam.height
    shards = bands_for(W, H, world, per_rank) if tile is None else shard.tile_grid(W, H, tile)
    mine = shard.tiles_of_rank(len(shards), rank, world)
    dev = torch.device("cuda", acc.device)
    cap = max(w * h for (_, _, w, h) in shards) * 3
    per = (len(shards) + world - 1) // world
    slab = torch.zeros((per, cap), dtype=torch.float32, device=dev)      # one allocation per frame; padding stays zero
    tot = {"primary_rays": 0, "primary_hits": 0, "ao_rays": 0, "ao_occluded": 0}
    for k, tid in enumerate(mine):
        x0, y0, w, h = shards[tid]
        out = slab[k, :w * h * 3].view(h, w, 3)                          # the tile renders in place
        _, st = acc.render_ao_tile(cam, x0, y0, w, h, pixel_samples, gather_nsamples, seed=seed, out=out)
        for kk in tot:
            tot[kk] += st[kk]
    return assemble_shards(slab, shards, W, H, rank, world), tot


def assemble_shards(slab, shards, W, H, rank, world):
    """the exchange step (one gather of [per_rank, cap] slabs to rank 0) + placement with the reference's y flip
    (bucket_write, render.c:962-964)"""
    import torch
    out = shard.gather_slabs(slab, rank, world)
    if rank != 0:
        return None
    img = torch.zeros((H, W, 3), dtype=slab.dtype, device=slab.device)
    for r in range(world):
        for k, tid in enumerate(shard.tiles_of_rank(len(shards), r, world)):
            x0, y0, w, h = shards[tid]
            img[H - (y0 + h):H - y0, x0:x0 + w] = out[r][k, :w * h * 3].view(h, w, 3)
    return img


def assemble(slab, W, H, tile, rank, world):
    """square-tile slabs [n, tile*tile*3] (rows of `tile` pixels, ragged tiles top-left aligned) -> frame on rank 0"""
    import torch
    tiles = shard.tile_grid(W, H, tile)
    per_rank = (len(tiles) + world - 1) // world
    pad = torch.zeros((per_rank, tile * tile * 3), dtype=slab.dtype, device=slab.device)
    pad[:slab.shape[0]] = slab
    out = shard.gather_slabs(pad, rank, world)
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
