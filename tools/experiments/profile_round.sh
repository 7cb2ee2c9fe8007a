#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench command (no counters)
#   2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (no tracing flags), each under `timeout`
# Outputs under gpurun_out/<tag>/ ; tools/experiments/profile_collect.py copies the summaries into profiles/.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-ao --no-pt"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
grep -h '"metric"' $OUT/trace.log | tail -1 | cut -c1-400
