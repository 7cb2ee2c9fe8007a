"""bench.py: in-run validation of the headline ray dump (timed launch == counted launch, the reference's check values, gathered slices)."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403 -- the constants and helpers every leg shares
from .common import ROOT, gather_ceiling, pmc_source, host_cores


def validate_dump(torch, la, args, mode, outs_of, cb, cnt_out, ns, gathered, world, per, n_total, bufs, nchunks, wire_bytes=28):
    """the timed launches' own outputs: (1) bit-equal to the counted launch on the sample, (2) hits and
    sum(t) of the first 1 M / 2 M rays against the reference's check values for the canonical dump,
    (3) N > 1: the gathered records on rank 0 == the ranks' slices (own slice checked bit for bit,
    every slice by hit-rate bounds)"""
    v = {"ok": True}
    (o, m) = outs_of(0)
    k = min(ns, m)
    if mode == la.MODE_CLOSEST:
        same = all(torch.equal(a[:k], b[:k]) for a, b in zip(o, cnt_out))
        v["timed_equals_counted_launch"] = bool(same); v["ok"] &= bool(same)
        canonical = (args.tris == 1_000_000 and abs(args.half_extent - 0.005) < 1e-12 and args.variant in (-1, 4))
        for nn, (hits, sumt) in SOUP1M_CHECK.items():
            if canonical and m >= nn:
                hit = o[0][:nn] != -1
                h = int(hit.sum().item()); s = float(o[1][:nn][hit].sum().item())
                good = (h == hits) and abs(s - sumt) < 5e-3
                v["first_%dM" % (nn // 1_000_000)] = {"hits": h, "sum_t": round(s, 4), "reference_hits": hits, "reference_sum_t": sumt, "ok": good}
                v["ok"] &= good
        total_hits = int(sum(int((outs_of(c)[0][0][:outs_of(c)[1]] != -1).sum().item()) for c in range(len(cb))))
        v["hits_this_rank"] = total_hits
    else:
        same = torch.equal(o[0][:k], cnt_out[0][:k])
        v["timed_equals_counted_launch"] = bool(same); v["ok"] &= bool(same)
    if world > 1:
        ok = True
        for c in range(nchunks):
            if wire_bytes == 16:        # {prim u32, t, u, v f32} per ray: rank 0's own slab against its fp64 records rounded to nearest
                rec = gathered[c][0].view(torch.int32).view(per, 4)
                (o_, m_) = outs_of(c) if len(cb) == nchunks else (None, 0)
                if o_ is not None and m_ > 0:
                    ok &= bool(torch.equal(rec[:m_, 0], o_[0][:m_]))
                    for k_ in (1, 2, 3):
                        ok &= bool(torch.equal(rec[:m_, k_].view(torch.float32), o_[k_][:m_].to(torch.float32)))
            else:
                ok &= bool(torch.equal(gathered[c][0], bufs[c]))
            if mode == la.MODE_CLOSEST:
                for r in range(world):
                    p = gathered[c][r].view(torch.int32).view(per, 4)[:, 0] if wire_bytes == 16 else gathered[c][r][24 * per:28 * per].clone().view(torch.int32)
                    frac = float((p != -1).float().mean().item())
                    ok &= (0.5 < frac < 0.99) or n_total < 100_000
        v["gathered_records_ok"] = ok; v["ok"] &= ok
    return v


def gathered_vs_world1(torch, la, scenes, shard, acc, args, mode, gathered, wire_bytes, per, nchunks, world, n_total, st_after_tris, dev, limit=16_000_000):
    """N > 1, rank 0: EVERY gathered slice against this rank's own trace of that slice's rays -- what a world-1 run returns for them
    (the rays of rank r's slice are regenerated here by jumping the stream ahead).  Dumps larger than `limit` rays are not repeated
    (the 100 M-ray headline: the driver's clock; the N = 8 tests run a few million)."""
    if mode != la.MODE_CLOSEST or n_total > limit:
        return {"gathered_equals_world1_records": None}
    ok = True; checked = 0
    for r in range(world):
        b0, b1 = shard.ray_slice(n_total, r, world); m = b1 - b0
        if m <= 0:
            continue
        ho, hd, _ = scenes.soup_rays(m, scenes.skip(st_after_tris, 5 * b0))
        ref = acc.intersect_device(torch.from_numpy(ho).to(dev), torch.from_numpy(hd).to(dev), mode=mode, variant=args.variant); torch.cuda.synchronize(dev)
        for c in range(nchunks):
            lo, hi = c * per, min(m, (c + 1) * per)
            if hi <= lo:
                continue
            if wire_bytes == 16:
                rec = gathered[c][r].view(torch.int32).view(per, 4)[:hi - lo]
                ok &= bool(torch.equal(rec[:, 0], ref[0][lo:hi]))
                for k in (1, 2, 3):
                    ok &= bool(torch.equal(rec[:, k].view(torch.float32), ref[k][lo:hi].to(torch.float32)))
            else:
                g = gathered[c][r].clone()          # a row of the [world, per * 28] slab starts at r * per * 28 bytes: not 8-byte aligned in general
                ok &= bool(torch.equal(g[24 * per:28 * per].view(torch.int32)[:hi - lo], ref[0][lo:hi]))
                for k in (1, 2, 3):
                    ok &= bool(torch.equal(g[8 * (k - 1) * per:8 * k * per].view(torch.float64)[:hi - lo], ref[k][lo:hi]))
            checked += hi - lo
    return {"gathered_equals_world1_records": bool(ok), "gathered_records_checked": int(checked)}
