"""gpurun_out/r06_predict8_{ao,dump}.jsonl (tools/r06_predict8.sh) -> the markdown cost tables of profiles/.   python tools/predict8_table.py ao|dump <jsonl>"""
import json, sys
what, path = sys.argv[1], sys.argv[2]
rows = sorted((json.loads(l) for l in open(path) if l.startswith("{")), key=lambda r: r["world"])
lat, bw = rows[0]["latency_us_per_call"], rows[0]["GBps_per_link"]
METHOD = ("Method (`tools/predict8.py`, `tools/r06_predict8.sh`; no number added by hand): **a rank's share** = its batch alone on the GPU, every rank's one after the "
          "other in ONE process (best of 4, after one untimed pass); **the exchange** = the real call sequence of `world` real processes on one device through "
          "`lh_dist_*`'s RCCL branch and `tests/mock_rccl`, whose link model charges %.0f us per call that reaches the library (a grouped gather = one call) + "
          "the LARGEST transfer of the call / %.0f GB/s (xGMI is point to point: the N - 1 receives of a gather cross N - 1 links at once); **the barriers** = the two "
          "barriers that bracket a timed frame, between those `world` processes arriving at different times (exit spread + last-in to last-out, p50; "
          "`tools/skew_probe.py`'s measure).  predicted = busiest share + exchange + barriers; speed-up = the whole job as ONE batch on one GPU / predicted.  "
          "Nothing here ran on more than one GPU.\n" % (lat, bw))
if what == "ao":
    r0 = rows[0]
    print("# Shard cost table, round 6: BASELINE config 5 AO frame (%d triangles, %dx%d, %d AO samples), one MI355X\n" % (r0["triangles"], r0["size"], r0["size"], r0["samples"]))
    print(METHOD)
    print("| ranks | bands x rows | one batch, whole frame (ms) | sum over ranks (ms) | busiest rank (ms) | least busy (ms) | exchange, modelled (ms; calls, MB into rank 0) | barriers (ms) | predicted frame (ms) | predicted speed-up | without the barriers | sharded frame == one batch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d x %d | %.2f | %.2f | %.2f | %.2f | %.3f (%d, %.1f) | %.3f | %.2f | **%.2fx** | %.2fx | %s |" % (
            r["world"], r["bands"], r["band_rows"], r["one_batch_ms"], r["sum_ms"], r["busiest_ms"], min(r["batch_ms"]), r["exchange_model_ms"], r["exchange_calls_rank0"],
            r["exchange_bytes_rank0"] / 1e6, r["barriers_ms"], r["predicted_ms"], r["speedup"], r["speedup_without_barriers"], r["frame_equals_one_batch"]))
    for r in rows:
        print("\nPer rank at %d ranks -- batch ms (camera-ray hits, thousands): " % r["world"] + "  ".join("%.2f (%d)" % (b, h) for b, h in zip(r["batch_ms"], r["hits_k"])))
else:
    r0 = rows[0]
    print("# Dump cost table, round 6: the headline dump (S-soup-1M, %d incoherent rays, closest hit) cut into contiguous slices, one MI355X\n" % r0["rays"])
    print(METHOD)
    print("A rank traces its slice in `chunks` launches; chunk c's records travel to rank 0 while chunk c + 1 is traced (rank 0 has chunk c when EVERY peer has traced it: "
          "the slowest rank's chunk times gate the pipeline).  Wire records: 16 bytes (`prim` u32 + `t u v` fp32 rounded from the fp64 records, which stay with "
          "the rank that traced them: `lh_dist_pack_records16`, packed behind every chunk's launch and timed with it) or the 28-byte fp64 records themselves.  "
          "`records stay` = no record exchange (a digest travels): the busiest slice + the barriers.\n")
    print("| ranks | chunks | whole dump, one launch (ms) | busiest slice, 16-B wire (ms) | gather per chunk 16 B / 28 B (ms, modelled) | predicted 16 B (ms) | **speed-up 16 B** | predicted 28 B (ms) | speed-up 28 B | records stay: speed-up |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        for nch, e in sorted(r["chunks"].items(), key=lambda kv: int(kv[0])):
            a, b, c = e["wire_16"], e["wire_28"], e["records_stay_with_rank"]
            print("| %d | %s | %.2f | %.2f | %.3f / %.3f | %.2f | **%.2fx** | %.2f | %.2fx | %.2fx |" % (
                r["world"], nch, r["one_launch_ms"], a["busiest_slice_ms"], a["gather_model_ms_per_chunk"], b["gather_model_ms_per_chunk"], a["predicted_ms"], a["speedup"],
                b["predicted_ms"], b["speedup"], c["speedup"]))
    print("\nBarriers (ms) at %s ranks: %s" % (" / ".join(str(r["world"]) for r in rows), " / ".join("%.3f" % r["barriers_ms"] for r in rows)))
