"""Copy the summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/ and refresh
profiles/pmc_latest.json.  python tools/profile_collect.py <tag> <round-name>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
ks = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(ROOT, "profiles", rnd + "_bench_kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader()
    for r in rows[:8]:
        w.writerow(r)
main = [r for r in rows if "k_trace_persist_lane<false, false, 3, true>" in r["Name"]][0]
print("kernel", main["Name"][:70], "calls", main["Calls"], "avg ms", float(main["AverageNs"]) / 1e6)


def pmc(sub, counter):
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))[0]
    vals = []; keep = []
    for r in csv.DictReader(open(f)):
        if "k_trace_persist_lane<false, false, 3, true>" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"])); keep.append(r)
    return vals, keep


fv, fk = pmc("fetch", "FETCH_SIZE"); wv, wk = pmc("write", "WRITE_SIZE")
with open(os.path.join(ROOT, "profiles", rnd + "_bench_pmc_fetch_write.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=list(fk[0].keys())); w.writeheader()
    for r in fk + wk:
        w.writerow(r)
# the timed launches are the 100 M-ray ones: the largest values
fetch = sorted(fv)[-5:]; write = sorted(wv)[-5:]
F = sum(fetch) / len(fetch); W = sum(write) / len(write)
j = {"round": rnd, "kernel": "k_trace_persist_lane<false,false,3,true>", "kernel_tag": "q16x4", "mode": "closest",
     "rays_per_launch": 100000000, "FETCH_SIZE_KiB": F, "WRITE_SIZE_KiB": W,
     "kernel_avg_ms_rocprof": float(main["AverageNs"]) / 1e6,
     "correction": "gfx950: every L2->fabric read request of this access pattern is a 128-B request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ on "
                   "the random-gather calibration kernel tools/ubench/gather), while FETCH_SIZE tallies 64 B per request (MI355X_MICROARCH.md "
                   "HBM section) => read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE as reported (uncalibrated, ~1% of the total)",
     "hbm_bytes_per_launch": 2 * F * 1024 + W * 1024}
json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(j)[:300])
