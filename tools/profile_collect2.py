"""Copy the summaries of gpurun_out/<tag>/<leg>/ (tools/profile_round2.sh) into profiles/<round>_<leg>_* and
refresh profiles/pmc_latest.json (headline) / pmc_latest_hbm.json (S-soup-10M).
  python tools/profile_collect2.py <tag> <round-name>"""
import csv, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
CORR = ("gfx950: every L2->fabric read request of this access pattern is a 128-B request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ on "
        "the random-gather calibration kernel tools/ubench/gather), while FETCH_SIZE tallies 64 B per request (MI355X_MICROARCH.md "
        "HBM section) => read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE as reported (uncalibrated, ~1% of the total)")


def main_kernel(rows):
    """the trace kernel with the largest total time"""
    cand = [r for r in rows if "k_trace" in r["Name"]]
    return max(cand, key=lambda r: float(r["TotalDurationNs"])) if cand else None


for leg in ("main", "hbm", "ao", "pt"):
    src = os.path.join(ROOT, "gpurun_out", tag, leg)
    ks = os.path.join(src, "kernel_stats.csv")
    if not os.path.exists(ks):
        continue
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (rnd, leg)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in rows[:14]:
            w.writerow(r)
    bl = os.path.join(src, "bench_line.json")
    if os.path.exists(bl) and os.path.getsize(bl):
        shutil.copy(bl, os.path.join(ROOT, "profiles", "%s_%s_bench_line.json" % (rnd, leg)))
    mk = main_kernel(rows)
    if mk is None:
        continue
    avg_ms = float(mk["AverageNs"]) / 1e6
    kt = os.path.join(src, "kernel_trace_rayquery.csv")
    if os.path.exists(kt):
        # per-dispatch durations: the timed launches are the longest ones of this kernel (the smoke pass of
        # --only runs and the counted sample are shorter and would pollute the --stats average)
        d = sorted((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(kt)) if r["Kernel_Name"] == mk["Name"])
        k = 5 if leg == "main" else 3
        if d:
            avg_ms = sum(d[-k:]) / len(d[-k:]) / 1e6
            with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_dispatches.csv" % (rnd, leg)), "w") as f:
                f.write("kernel,duration_ns\n")
                for x in d:
                    f.write("\"%s\",%d\n" % (mk["Name"], x))
    print(leg, "kernel", mk["Name"][:80], "calls", mk["Calls"], "stats avg ms", float(mk["AverageNs"]) / 1e6, "timed-launch avg ms", avg_ms)
    vals = {}
    keep = []
    for P, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(src, P + ".csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and r["Kernel_Name"] == mk["Name"]:
                vals.setdefault(counter, []).append(float(r["Counter_Value"])); keep.append(r)
    if keep:
        with open(os.path.join(ROOT, "profiles", "%s_%s_pmc_fetch_write.csv" % (rnd, leg)), "w") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys())); w.writeheader()
            for r in keep:
                w.writerow(r)
    if leg in ("main", "hbm") and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # the timed launches are the largest ones (the counted launch and the smoke pass are smaller)
        k = 5 if leg == "main" else 3
        F = sorted(vals["FETCH_SIZE"])[-k:]; W = sorted(vals["WRITE_SIZE"])[-k:]
        F = sum(F) / len(F); W = sum(W) / len(W)
        line = json.load(open(bl)) if os.path.exists(bl) and os.path.getsize(bl) else {}
        wide = leg == "hbm" and (line.get("roofline_hbm") or {}).get("node_bytes") == 128
        try:
            commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = ""
        j = {"round": rnd, "commit": commit, "source": "profiles/%s_%s_pmc_fetch_write.csv (tools/profile_round2.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes of the bench command)" % (rnd, leg),
             "kernel": mk["Name"], "kernel_tag": "q16x8" if wide else "q16x4", "mode": "closest",
             "rays_per_launch": 100000000 if leg == "main" else 50000000,
             "triangles": 1000000 if leg == "main" else 10000000,
             "FETCH_SIZE_KiB": F, "WRITE_SIZE_KiB": W, "kernel_avg_ms_rocprof": avg_ms,
             "correction": CORR, "hbm_bytes_per_launch": 2 * F * 1024 + W * 1024}
        name = "pmc_latest.json" if leg == "main" else "pmc_latest_hbm.json"
        json.dump(j, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
        print(json.dumps(j)[:400])
