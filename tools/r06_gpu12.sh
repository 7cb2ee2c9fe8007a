cd /tmp; export TMPDIR=/tmp
LH_DANGER_BOXES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cliff -o cliff -- python $GRAFT_REPO_ROOT/tools/r06_degenerate_cliff.py 4000000 > $GRAFT_REPO_ROOT/gpurun_out/prof_cliff.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_cliff -name "*kernel_stats.csv" | head -1); echo $f; head -12 $f | cut -c1-220
grep "Mrays" gpurun_out/prof_cliff.log
