#!/bin/bash
# usage (on the GPU box): tools/trace_timeline.sh <out.csv> <command...>   kernel trace (start, end, name) of the command's LAST second
R=$GRAFT_REPO_ROOT; OUTF=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- "$@" > /tmp/tl.log 2>&1 < /dev/null
grep "tile ms\|frames_ms" /tmp/tl.log
K=$(ls /tmp/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$K" ] && python3 - "$K" "$R/$OUTF" <<'P'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,dur_us,gap_us,queue,kernel\n")
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t_end - 40_000_000: continue
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
        f.write("%.1f,%.1f,%.1f,%s,%s\n" % ((s - (t_end - 40_000_000)) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3, r.get("Queue_Id", ""), name))
        prev = e
P
