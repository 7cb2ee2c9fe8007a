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


DEFAULT_BAND_ROWS = 16


def bands_for(height, world, rows=None):
    """image-space shards for `world` ranks: full-width bands of `rows` lines (default DEFAULT_BAND_ROWS; LH_BAND_ROWS overrides).
    AO frames deal them out in serpentine order (shard.bands_of_rank), path-traced frames interleaved (band_id % world: their
    bands are one pass with a fixed stride).  A rank renders ALL of its bands as ONE device batch (lh_render_ao_bands), so a band
    costs nothing by itself; what its height trades is coherence (tall: a wave's neighbours in the batch are neighbours in the
    image) against balance (fine: sky / floor / silhouettes spread evenly) -- profiles/r05_shard_cost_table.md.
    -> (band_rows, [first line of every band])"""
    import os
    if rows is None:
        rows = int(os.environ.get("LH_BAND_ROWS", str(DEFAULT_BAND_ROWS)))
    if world <= 1:
        rows = height                      # one rank: the frame is one band
    rows = max(1, min(rows, height))
    return rows, list(range(0, height, rows))


def render_ao_frame_sharded(acc, cam, pixel_samples, gather_nsamples, rank, world, tile=None, seed=1, band_rows=None, timing=None):
    """Each rank renders its interleaved shards; ONE gather of equal-sized slabs assembles the frame on rank 0 (None
    elsewhere).  tile=None (default): full-width bands, all of a rank's bands in one device batch; an int: square tiles,
    one batch each (the round-1 path, kept for comparison).  Returns (image|None, local stats).
    timing: a dict that receives this rank's bands / batch_ms / gather_ms (diagnostic frames only: it synchronises the device
    between the two phases)."""
    import os
    import time
    import torch
    W, H = cam.width, cam.height
    dev = torch.device("cuda", acc.device)
    if tile is None:
        rows, y0s = bands_for(H, world, band_rows)
        mine = shard.bands_of_rank(len(y0s), rank, world)
        per = (len(y0s) + world - 1) // world
        slab = torch.zeros((per, rows * W * 3), dtype=torch.float32, device=dev)
        if timing is not None:
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
        _, tot = acc.render_ao_bands(cam, [y0s[b] for b in mine], rows, pixel_samples, gather_nsamples, seed=seed,
                                     out=slab[:len(mine)].view(len(mine), rows, W, 3))
        if timing is not None:
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
        shards = [(0, y0, W, min(rows, H - y0)) for y0 in y0s]
        # an AO frame is grey (Lo = (N - occluded) / N in every channel, ambientocclusion.c:383-401): ONE float per pixel travels,
        # rank 0 writes it three times (lh_dist.hip k_take_channel0 / k_place_bands do the same for a C caller)
        mono = slab.view(per, rows * W, 3)[:, :, 0]
        # ... and with one sample per pixel that float is (N - occluded) / N for an integer numerator (lh_render.hip k_ao_resolve):
        # for N <= 255 ONE BYTE per pixel travels, the numerator -- read back exactly from the float, which is within 1e-7 N of
        # it -- and rank 0 evaluates the same fp64 quotient on it: the same bits (lh_dist.hip k_take_count8 does the same for a C
        # caller; LH_DIST_AO_BYTES=4 keeps the float)
        nsamp = ao_ray_count(gather_nsamples)
        if pixel_samples == 1 and 1 <= nsamp <= 255 and world > 1 and os.environ.get("LH_DIST_AO_BYTES", "1") != "4":
            mono = (mono * float(nsamp) + 0.5).to(torch.uint8)
            img = assemble_shards(mono, shards, W, H, rank, world, stride_rows=rows, serpentine=True, channels=1, count_of=nsamp)
        else:
            img = assemble_shards(mono.contiguous(), shards, W, H, rank, world, stride_rows=rows, serpentine=True, channels=1)
        if timing is not None:
            torch.cuda.synchronize(dev)
            timing.update(bands=len(mine), band_rows=rows, batch_ms=round((t1 - t0) * 1e3, 3), gather_ms=round((time.perf_counter() - t1) * 1e3, 3))
        return img, tot
    shards = shard.tile_grid(W, H, tile)
    mine = shard.tiles_of_rank(len(shards), rank, world)
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


def ao_ray_count(gather_nsamples):
    """AO rays per hit point: nphi = ntheta = (int)sqrt(nsamples) (ambientocclusion.c:378-380; lh_tile.hip ao_region)"""
    import math
    nphi = int(math.sqrt(float(gather_nsamples)))
    return nphi * nphi


def assemble_shards(slab, shards, W, H, rank, world, stride_rows=None, serpentine=False, channels=3, count_of=None):
    """the exchange step (one gather of [per_rank, cap] slabs to rank 0) + placement with the reference's y flip
    (bucket_write, render.c:962-964).  stride_rows: the slab of a shard holds that many rows (bands of a batch: a clipped
    last band keeps its lines at the BOTTOM of its slab, the clipped lines being below the frame); None: h rows.
    serpentine: the shards were dealt out by shard.bands_of_rank (AO bands), else by shard.tiles_of_rank.
    channels: floats per pixel in the slabs (1: a grey frame, expanded to RGB here).  count_of: the slabs hold uint8 numerators
    over that denominator (a one-sample AO frame) instead of floats."""
    import torch
    out = shard.gather_slabs(slab, rank, world)
    if rank != 0:
        return None
    odt = slab.dtype
    if count_of is not None:
        out = [(o.to(torch.float64) / float(count_of)).to(torch.float32) for o in out]
        odt = torch.float32
    if channels == 1:
        out = [o.view(o.shape[0], -1, 1).expand(-1, -1, 3).reshape(o.shape[0], -1) for o in out]
    of_rank = shard.bands_of_rank if serpentine else shard.tiles_of_rank
    if stride_rows is not None and H % stride_rows == 0:
        # regular bands: one indexed copy per rank, then one flip -- band 0 is the BOTTOM of the image, every band is
        # already top-line-first inside (thousands of bands per frame: no Python loop over them)
        nb = H // stride_rows
        bands = torch.empty((nb, stride_rows, W, 3), dtype=odt, device=slab.device)
        for r in range(world):
            ids = of_rank(nb, r, world)
            if ids:
                bands[torch.tensor(ids, device=slab.device)] = out[r][:len(ids)].view(len(ids), stride_rows, W, 3)
        return bands.flip(0).reshape(H, W, 3)
    img = torch.zeros((H, W, 3), dtype=odt, device=slab.device)
    for r in range(world):
        for k, tid in enumerate(of_rank(len(shards), r, world)):
            x0, y0, w, h = shards[tid]
            if stride_rows is None:
                img[H - (y0 + h):H - y0, x0:x0 + w] = out[r][k, :w * h * 3].view(h, w, 3)
            else:                        # image orientation inside the band: its first frame line is the LAST slab row
                img[H - (y0 + h):H - y0, x0:x0 + w] = out[r][k].view(stride_rows, w, 3)[stride_rows - h:]
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


def render_pt_frame_sharded(acc, cam, spp, rank, world, tile=256, spp_chunk=16, band_rows=None, **kw):
    """Path-traced frame (BASELINE config 4), samples in passes of `spp_chunk` (bounded device memory), one gather of slabs.
    world > 1, band_rows given, or a material / flags asked for: a rank's interleaved full-width bands (bands_for) are ONE pass
    per sample chunk (lh_render_pt_bands: a pass costs a fixed ~3 ms of kernel ramps whatever its size, and fine bands spread sky
    / floor / sphere evenly over the ranks; a frame height the band rows do not divide, or fewer bands than ranks, falls back to
    one-line bands -- never to another code path); else (one rank, kd / env only) square tiles, one pass each.
    kw: kd | material, env, max_vertices, flags, seed -- the SAME meaning on both paths: env is this frame's constant
    environment (default: the accelerator's own, white if never set; an explicit black stays black) and is restored afterwards.
    Returns (image on rank 0 | None, local stats)."""
    import torch
    from . import binding
    W, H = cam.width, cam.height
    dev = torch.device("cuda", acc.device)
    tot = {"paths": 0, "rays": 0}
    rows = None
    if world > 1 or band_rows is not None or kw.get("material") is not None or kw.get("flags"):
        rows, y0s = bands_for(H, max(world, 2) if band_rows is not None else world, band_rows)
        if H % rows != 0 or len(y0s) < world:
            rows, y0s = 1, list(range(H))                 # one-line bands always tile the frame
    if rows is not None:
        mine = shard.tiles_of_rank(len(y0s), rank, world)
        per = (len(y0s) + world - 1) // world
        slab = torch.zeros((per, rows * W * 3), dtype=torch.float32, device=dev)
        prev_env = getattr(acc, "_env", None)
        if "env" in kw:
            acc.set_environment(kw["env"], None)
        mat = kw.get("material")
        if mat is None and "kd" in kw:
            mat = binding.Material.make(kd=(kw["kd"],) * 3)
        try:
            if mine:
                out = slab[:len(mine)].view(len(mine), rows, W, 3)
                for s0 in range(0, spp, spp_chunk):
                    _, st = acc.render_pt_bands(cam, y0s[mine[0]], rows, rows * world, len(mine), s0, min(spp_chunk, spp - s0), spp,
                                                max_vertices=kw.get("max_vertices", 8), flags=kw.get("flags", 0), override=mat,
                                                seed=kw.get("seed", 1), out=out)
                    tot["paths"] += st["paths"]; tot["rays"] += st["rays"]
                torch.cuda.synchronize(dev)               # the passes read the environment: finished before it is put back
        finally:
            if "env" in kw:
                if prev_env is None:
                    acc.reset_environment()
                else:
                    acc.set_environment(*prev_env)
        shards = [(0, y0, W, rows) for y0 in y0s]
        return assemble_shards(slab, shards, W, H, rank, world, stride_rows=rows), tot
    # one rank, one diffuse reflectance, a constant environment: square tiles through lh_render_pt_tile (env passed per call)
    tkw = {k: v for k, v in kw.items() if k in ("kd", "env", "max_vertices", "seed")}
    if "env" not in tkw and getattr(acc, "_env", None) is not None and acc._env[1] is None:
        tkw["env"] = acc._env[0]                         # the accelerator's own constant environment, as the band path uses it
    tiles = shard.tile_grid(W, H, tile)
    mine = shard.tiles_of_rank(len(tiles), rank, world)
    slab = torch.zeros((len(mine), tile * tile * 3), dtype=torch.float32, device=dev)
    for k, tid in enumerate(mine):
        x0, y0, w, h = tiles[tid]
        out = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        for s0 in range(0, spp, spp_chunk):
            _, st = acc.render_pt_tile(cam, x0, y0, w, h, s0, min(spp_chunk, spp - s0), spp, out=out, **tkw)
            tot["paths"] += st["paths"]; tot["rays"] += st["rays"]
        t = torch.zeros((tile, tile, 3), dtype=torch.float32, device=dev)
        t[:h, :w] = out
        slab[k] = t.view(-1)
    return assemble(slab, W, H, tile, rank, world), tot
