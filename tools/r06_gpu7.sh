cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_shard.py -q -k "n2" > gpurun_out/r06_gputest_shard2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_shard2.log
tail -4 gpurun_out/r06_gputest_shard2.log
P8_WHAT=dump bash tools/r06_predict8.sh
