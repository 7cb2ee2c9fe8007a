cd $GRAFT_REPO_ROOT
timeout -k 5 900 python tools/experiments/ab_frames.py 2 2>&1 | grep -v amdgpu | tee gpurun_out/r06_ab_variants2.txt
timeout -k 5 600 python -m pytest tests/test_gpu_degenerate.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
timeout -k 5 600 python tools/r06_degenerate_cliff.py 20000000 2>&1 | grep -v amdgpu | tee gpurun_out/r06_degenerate_cliff3.txt
FUZZ_BUDGET_S=100 timeout -k 5 400 python tools/fuzz_parity.py 2000 615 2>&1 | tail -1
