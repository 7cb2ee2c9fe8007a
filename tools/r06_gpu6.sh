cd $GRAFT_REPO_ROOT
env LH_DIST_TRANSPORT=rccl LH_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/mock_rccl/libmock_rccl.so MOCK_RCCL_TIMEOUT=120 MOCK_RCCL_LOG=$GRAFT_REPO_ROOT/gpurun_out/n2mock_log \
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 2 --warmup 1 \
  --device-override 0 --rays 2000001 --tris 100000 --half-extent 0.01 --record-bytes 28 --no-cpu --no-hbm --no-pt --ao-size 192 --ao-tess 2 --ao-samples 16 \
  > gpurun_out/n2mock.out 2> gpurun_out/n2mock.err
echo "n2mock rc $?"; grep -n "mock_rccl rank\|Error\|error" gpurun_out/n2mock.err | head -20; tail -5 gpurun_out/n2mock_log.rank1; tail -5 gpurun_out/n2mock_log.rank0
bash tools/r06_predict8.sh
