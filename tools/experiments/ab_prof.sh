#!/bin/bash
# usage (on the GPU box): tools/ab_prof.sh <kernel-name regex> <command...>
# Runs <command> under rocprofv3 --kernel-trace --stats once per library in gpurun_variants/*.so (LH_LIBRARY) and prints the
# matching rows of the kernel statistics: an A/B of single kernels, not of the whole frame.
R=$GRAFT_REPO_ROOT; PAT=$1; shift
cd /tmp; export TMPDIR=/tmp
for LIB in $R/gpurun_variants/*.so; do
  OUT=/tmp/abprof_$(basename $LIB .so); rm -rf $OUT
  LH_LIBRARY=$LIB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1 < /dev/null
  K=$(ls $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $(basename $LIB)"
  [ -n "$K" ] && python3 - "$K" "$PAT" <<'P'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("  %-56s calls %4s avg %8.3f ms max %8.3f ms total %9.2f ms" % (re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:56], r["Calls"],
              float(r["AverageNs"]) / 1e6, int(r["MaxNs"]) / 1e6, int(r["TotalDurationNs"]) / 1e6))
P
  grep -h "frame_ms" $OUT.log | tail -1 | cut -c1-200
done
