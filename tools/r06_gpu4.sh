cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_dist_mock.py tests/test_gpu_dist.py -x -q > gpurun_out/r06_gputest_shard.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_shard.log
tail -12 gpurun_out/r06_gputest_shard.log
free -g | head -2; df -h /dev/shm | tail -1
bash tools/r06_predict8.sh
