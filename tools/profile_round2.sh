#!/bin/bash
# usage (on the GPU box): tools/profile_round2.sh <tag> [legs...]     legs: main hbm ao pt (default: all)
# For each leg: rocprofv3 --kernel-trace --stats (no counters), then separate --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes (no tracing flags), each under `timeout`.  Outputs under gpurun_out/<tag>/<leg>/;
# tools/profile_collect2.py copies the summaries into profiles/.
R=$GRAFT_REPO_ROOT; TAG=$1; shift; LEGS=${@:-main hbm ao pt}
cd /tmp; export TMPDIR=/tmp
for LEG in $LEGS; do
  OUT=$R/gpurun_out/$TAG/$LEG; mkdir -p $OUT
  case $LEG in
    main) CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-ao --no-pt --no-hbm --no-config2 --no-other-builder" ;;   # lh_accel_commit's own choice of builder, without the other-builder comparison launches
    *)    CMD="python $R/bench.py --only $LEG --no-other-builder" ;;
  esac
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
  grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
  # the raw counter CSVs are large: keep the rows of the ray-query kernels only
  for P in fetch write; do
    F=$(ls $OUT/$P/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$F" ] && { head -1 $F > $OUT/$P.csv; grep -E "k_trace|k_ref_retrace|k_ao|k_pt|k_primary|k_resolve" $F >> $OUT/$P.csv; rm -rf $OUT/$P; }
  done
  K=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  T=$(ls $OUT/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && { head -1 $T > $OUT/kernel_trace_rayquery.csv; grep -E "k_trace|k_resolve" $T >> $OUT/kernel_trace_rayquery.csv; }
  [ -n "$K" ] && { cp $K $OUT/kernel_stats.csv; rm -rf $OUT/trace; }
done
