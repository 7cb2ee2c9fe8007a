cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_dist_mock.py tests/test_gpu_dist.py -q -k "not test_bench_n2_code_path and not test_sharded" > gpurun_out/r06_gputest_shard.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_shard.log
grep -n "mock_rccl rank" gpurun_out/r06_gputest_shard.log | head -20
tail -12 gpurun_out/r06_gputest_shard.log
timeout 1500 python tools/r06_ab_share.py 2 > gpurun_out/r06_ab_share.txt 2>&1
cat gpurun_out/r06_ab_share.txt
