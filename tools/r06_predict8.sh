# tools/r06_predict8.sh (GPU box): tools/predict8.py for 2, 4 and 8 ranks -- every rank's share timed alone by ONE process (solo-ao / solo-dump),
# then the real exchange of `world` processes on one device through tests/mock_rccl's link model (30 us per RCCL call, 45 GB/s per xGMI link and
# direction) and the barriers' skew between them -> gpurun_out/r06_predict8_{ao,dump}.jsonl
cd $GRAFT_REPO_ROOT
for what in ${P8_WHAT:-ao dump}; do
  : > gpurun_out/r06_predict8_$what.jsonl
  for w in ${P8_WORLDS:-8 4 2}; do
    timeout 900 python tools/predict8.py solo-$what $w 2> gpurun_out/r06_solo_${what}_$w.err | tail -5
    env LH_DEVICE_OVERRIDE=0 LH_DIST_TRANSPORT=rccl LH_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/mock_rccl/libmock_rccl.so MOCK_RCCL_LATENCY_US=30 MOCK_RCCL_GBPS=45 MOCK_RCCL_TIMEOUT=900 \
      timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + w)) tools/predict8.py $what \
      2> gpurun_out/r06_predict8_${what}_$w.err | grep '^{' >> gpurun_out/r06_predict8_$what.jsonl
    tail -2 gpurun_out/r06_predict8_${what}_$w.err
  done
  cat gpurun_out/r06_predict8_$what.jsonl
done
