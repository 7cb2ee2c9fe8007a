"""round 4: the kernel timeline of the last AO batch in a rocprofv3 --kernel-trace csv (start / end relative to the batch's first kernel).
python tools/timeline_of_batch.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_primary_rays starts the last batch
starts = [k for k, r in enumerate(rows) if "k_primary_rays" in r["Kernel_Name"]]
k0 = starts[-1]; t0 = int(rows[k0]["Start_Timestamp"])
for r in rows[k0:]:
    n = r["Kernel_Name"]; m = re.search(r"(k_\w+(?:<[^>]*>)?)", n)
    print("%-60s start %9.3f  end %9.3f  dur %8.3f ms  grid %s" % ((m.group(1) if m else n)[:60], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "")))
