cd $GRAFT_REPO_ROOT
P8_WHAT=dump bash tools/r06_predict8.sh > gpurun_out/r06_predict8_dump.log 2>&1
grep "solo-dump world 8" gpurun_out/r06_predict8_dump.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest2.log
tail -6 gpurun_out/r06_gputest2.log
