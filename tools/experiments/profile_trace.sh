#!/bin/bash
# usage (on the GPU box): tools/profile_trace.sh <tag> <leg>      leg: main hbm ao pt config2
# rocprofv3 --kernel-trace --stats of one bench leg only (no counter passes): gpurun_out/<tag>/<leg>/kernel_stats.csv
R=$GRAFT_REPO_ROOT; TAG=$1; LEG=$2
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG/$LEG; mkdir -p $OUT
case $LEG in
  main) CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-ao --no-pt --no-hbm --no-config2" ;;
  *)    CMD="python $R/bench.py --only $LEG" ;;
esac
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 < /dev/null
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
K=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$K" ] && cp $K $OUT/kernel_stats.csv
rm -rf $OUT/trace
[ -f $OUT/kernel_stats.csv ] && head -16 $OUT/kernel_stats.csv | cut -c1-220
