cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_degenerate.py tests/test_gpu_parity.py tests/test_gpu_single_ray.py -x -q > gpurun_out/r06_gputest_deg.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_deg.log
tail -15 gpurun_out/r06_gputest_deg.log
echo "== round 5's rule (LH_DANGER_BOXES=0)" > gpurun_out/r06_degenerate_cliff.txt
LH_DANGER_BOXES=0 timeout 600 python tools/r06_degenerate_cliff.py 20000000 2>&1 | grep -v amdgpu >> gpurun_out/r06_degenerate_cliff.txt
echo "== round 6 (the boxes of the leaves that hold a zero-area triangle)" >> gpurun_out/r06_degenerate_cliff.txt
timeout 600 python tools/r06_degenerate_cliff.py 20000000 2>&1 | grep -v amdgpu >> gpurun_out/r06_degenerate_cliff.txt
cat gpurun_out/r06_degenerate_cliff.txt
FUZZ_BUDGET_S=120 timeout 400 python tools/fuzz_parity.py 2000 611 2>&1 | tail -2
