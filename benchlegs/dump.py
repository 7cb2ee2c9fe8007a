"""bench.py: in-run validation of the headline ray dump (timed launch == counted launch, the reference's check values, gathered slices)."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403 -- the constants and helpers every leg shares
from .common import ROOT, gather_ceiling, pmc_source, host_cores


def validate_dump(torch, la, args, mode, outs_of, cb, cnt_out, ns, gathered, world, per, n_total, bufs, nchunks):
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
            ok &= bool(torch.equal(gathered[c][0], bufs[c]))
            if mode == la.MODE_CLOSEST:
                for r in range(world):
                    p = gathered[c][r][24 * per:28 * per].view(torch.int32)
                    frac = float((p != -1).float().mean().item())
                    ok &= (0.5 < frac < 0.99)
        v["gathered_records_ok"] = ok; v["ok"] &= ok
    return v
