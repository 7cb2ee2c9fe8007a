"""bench.py: the HBM-bound leg -- S-soup-10M, 50 M incoherent rays, the 8-wide 128-byte nodes."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403 -- the constants and helpers every leg shares
from .common import ROOT, gather_ceiling, pmc_source, host_cores


def hbm_leg(la, scenes, torch, dev, local, args, hip, sptr, node_fmt):
    """the HBM roofline: the same closest-hit kernel on S-soup-10M (10 M triangles, half-extent 0.002: the
    SURVEY's config-5 stress soup).  Hot set = 4-wide nodes + tri32 ~ 0.8 GB >> the 256 MiB Infinity Cache."""
    P, idx, st = scenes.soup_triangles(args.hbm_tris, 0.002)
    n = args.hbm_rays
    d_org, d_dir, _ = upload_rays(scenes, torch, dev, st, n)
    acc = la.HipAccel(local); acc.add_mesh(P, idx)
    t0 = time.perf_counter(); info = acc.commit(build=args.leg_build); commit1 = time.perf_counter() - t0
    dev_built = info["nnodes"] == info["nnodes_traversal"]          # a device-built scene has no 2-wide nodes of its own
    other_b = "host" if dev_built else "device"
    del P, idx
    out = acc.intersect_device(d_org, d_dir); torch.cuda.synchronize(dev)
    ns = min(n, 4_000_000)
    cnt_out, cnt = acc.intersect_device(d_org[:ns], d_dir[:ns], counters=True)
    n_nodes = cnt["nodes"] / ns; n_tris = cnt["tris"] / ns
    node_bytes = acc.dump_node_bytes()              # 128: the 8-wide nodes (hot set beyond the Infinity Cache), else the 4-wide node's 64
    if node_bytes == 128:
        node_fmt = "q16x8"
    # SURVEY 8d prices EVERY node visit at 64 B (B_node), whatever record the walk really fetches: that is `bytes_per_ray`,
    # `achieved` and `frac` below.  The 8-wide walk fetches one 128-byte record per visit (and makes fewer visits); its record
    # bytes are reported as a plain number (`record_bytes_per_ray`), not as a bandwidth -- part of them is served by caches.
    b_ray = B_IN + B_OUT + B_NODE_SURVEY * n_nodes + B_TRI * n_tris
    # the same sample through the 4-wide walk: hit records do not depend on the tree
    cross = None
    if node_bytes == 128:
        acc.set_param("wide8", 0)
        alt = acc.intersect_device(d_org[:ns], d_dir[:ns]); torch.cuda.synchronize(dev)
        cross = all(torch.equal(a, b) for a, b in zip(alt, cnt_out))
        acc.set_param("wide8", -1)
        del alt
    steps = 3

    def timed(a, o):
        ev = EventPairs(hip, steps)
        a.intersect_device(d_org, d_dir, out=o); torch.cuda.synchronize(dev)
        for _ in range(steps):
            ev.begin(sptr); a.intersect_device(d_org, d_dir, out=o); ev.end(sptr)
        torch.cuda.synchronize(dev)
        return float(np.mean(ev.ms()))
    ms = timed(acc, out)
    ok = all(torch.equal(a[:ns], b) for a, b in zip(out, cnt_out))
    hit = float((out[0] != -1).float().mean().item())
    achieved = b_ray * n / (ms * 1e-3) / 1e9
    traffic = traffic_source = traffic_ms = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest_hbm.json")
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("rays_per_launch") == n and j.get("triangles") == args.hbm_tris and j.get("kernel_tag") == node_fmt:
                traffic = j.get("hbm_bytes_per_launch"); traffic_ms = j.get("kernel_avg_ms_rocprof")
                traffic_source = pmc_source("profiles/pmc_latest_hbm.json", j)
        except Exception:
            traffic = traffic_source = None
    info = acc.info()
    hot = info["nnodes_traversal"] * 64 + info["ntriangles"] * 48
    hot8 = (info["nnodes_traversal"] * 128 * 3 // 7 if node_bytes == 128 else info["nnodes_traversal"] * 64) + info["ntriangles"] * 48       # an 8-wide tree has ~3/7 of the 4-wide tree's nodes
    acc.close()
    # the ceiling of THIS leg's access pattern, measured now: dependent random records of the size the walk fetches, at the
    # scene's footprint (HBM-resident), at the walk's occupancy (three workgroups per CU for the 8-wide walk, four for the 4-wide)
    gc = None if args.no_ceiling else gather_ceiling(min(4096.0, hot8 / 1e6), 5 if node_bytes == 128 else 0, lds=53000 if node_bytes == 128 else 40000)
    # the twin on the OTHER builder's tree: same rays, same records
    twin = None
    try:
        if args.no_other_builder:
            raise RuntimeError("skipped (--no-other-builder)")
        P2, idx2, _ = scenes.soup_triangles(args.hbm_tris, 0.002)
        acc2 = la.HipAccel(local); acc2.add_mesh(P2, idx2)
        t0 = time.perf_counter(); info2 = acc2.commit(build=other_b); commit2 = time.perf_counter() - t0
        del P2, idx2
        out2 = acc2.intersect_device(d_org, d_dir); torch.cuda.synchronize(dev)
        _, cnt2 = acc2.intersect_device(d_org[:ns], d_dir[:ns], counters=True)
        ms2 = timed(acc2, out2)
        same2 = all(bool(torch.equal(a, b)) for a, b in zip(out2, out))
        nn2 = cnt2["nodes"] / ns; nt2 = cnt2["tris"] / ns
        br2 = B_IN + B_OUT + B_NODE_SURVEY * nn2 + B_TRI * nt2
        twin = {"builder": other_b, "commit_s": round(commit2, 3), "kernel_ms": round(ms2, 3),
                "value": round(n / (ms2 * 1e-3) / 1e6, 1), "value_unit": "Mrays/s", "nodes_per_ray": round(nn2, 3), "tris_per_ray": round(nt2, 3),
                "bytes_per_ray": round(br2, 1), "achieved": round(br2 * n / (ms2 * 1e-3) / 1e9, 1),
                "frac": round(br2 * n / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "records_bit_equal": same2,
                "nodes": info2["nnodes_traversal"], "depth": info2["max_depth"]}
        ok = ok and same2
        acc2.close(); del out2
    except Exception as e:                                  # noqa: BLE001 -- the twin is context, the leg stands without it
        twin = {"error": repr(e)}
    return {"workload": "S-soup-10M ray dump: %d random triangles (half-extent 0.002), %d incoherent rays, closest-hit" % (args.hbm_tris, n),
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "traffic_frac_of_peak": None if traffic is None else round(traffic / ((traffic_ms or ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "traffic_frac_note": "bytes AND time of the same profiled run (%s ms per launch under the counters; this run: %.3f ms)" % (traffic_ms, ms),
            "gather_ceiling": gc, "records_per_s": round((n_nodes + n_tris) * n / (ms * 1e-3), 0),
            "frac_of_gather_ceiling": None if not gc or "records_per_s" not in gc else round((n_nodes + n_tris) * n / (ms * 1e-3) / gc["records_per_s"], 4),
            "traffic_over_algorithmic": None if traffic is None else round(traffic / (b_ray * n), 3),
            "formula": "bytes_per_ray = %d (ray in) + %d (hit record out) + %d x nodes_per_ray + %d x tris_per_ray -- SURVEY 8d's constants "
                       "(B_in, B_out, B_node, B_tri), whatever the walk really moves (this kernel reads 48 B of fp64 ray and writes a 28-B record per ray, "
                       "and an 8-wide visit fetches a 128-B record); achieved = bytes_per_ray x rays / kernel_ms; frac = achieved / peak; "
                       "traffic = 2 x FETCH_SIZE + WRITE_SIZE of the committed counter pass; traffic_over_algorithmic = traffic / (bytes_per_ray x rays)"
                       % (B_IN, B_OUT, B_NODE_SURVEY, B_TRI),
            "record_bytes_per_ray": round(B_IN + B_OUT + node_bytes * n_nodes + B_TRI * n_tris, 1),
            "record_bytes_note": "what the walk's own records add up to per ray (%d-B node records): includes bytes served by L2 / the Infinity Cache -- a count, not a bandwidth" % node_bytes,
            "residency": "hot set %.0f MB as 4-wide nodes + tri32 >> 256 MiB Infinity Cache: HBM" % (hot / 1e6),
            "builder": ("device" if dev_built else "host") + (": lh_accel_commit's own choice at this size" if args.leg_build == "auto" else ", asked for")
                       + "; `other_builder` is the same dump on the other builder's tree",
            "commit_s": round(commit1, 3), "other_builder": twin,
            "kernel": "k_trace_persist_lane<walk=spec8, q16x8 nodes: 128-byte 8-wide records, one cache line each>" if node_bytes == 128
                      else "k_trace_persist_lane<walk=spec,%s nodes>" % node_fmt,
            "node_bytes": node_bytes,
            "value": round(n / (ms * 1e-3) / 1e6, 1), "value_unit": "Mrays/s", "kernel_ms": round(ms, 3),
            "bytes_per_ray": round(b_ray, 1), "nodes_per_ray": round(n_nodes, 3), "tris_per_ray": round(n_tris, 3),
            "hit_rate": round(hit, 4), "device_bytes": info["device_bytes"],
            "build_s": round(info["build_seconds"], 3), "ref_tree_build_s": round(info["ref_build_seconds"], 3),
            "validation": {"timed_equals_counted_launch": bool(ok), "equals_4wide_walk_on_sample": cross,
                           "ok": bool(ok) and cross is not False and 0.5 < hit < 0.999}}
