cd $GRAFT_REPO_ROOT
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest3.log
tail -5 gpurun_out/r06_gputest3.log
echo "== round 5's rule (LH_DANGER_BOXES=0)" > gpurun_out/r06_degenerate_cliff2.txt
LH_DANGER_BOXES=0 timeout -k 5 600 python tools/r06_degenerate_cliff.py 20000000 2>&1 | grep -v amdgpu >> gpurun_out/r06_degenerate_cliff2.txt
echo "== round 6 (the boxes of the leaves that hold a zero-area triangle)" >> gpurun_out/r06_degenerate_cliff2.txt
timeout -k 5 600 python tools/r06_degenerate_cliff.py 20000000 2>&1 | grep -v amdgpu >> gpurun_out/r06_degenerate_cliff2.txt
cat gpurun_out/r06_degenerate_cliff2.txt
FUZZ_BUDGET_S=100 timeout -k 5 400 python tools/fuzz_parity.py 2000 613 2>&1 | tail -1
FUZZ_BUDGET_S=60 timeout -k 5 300 python tools/fuzz_ao.py 63 400 2>&1 | tail -1
FUZZ_BUDGET_S=60 timeout -k 5 300 python tools/fuzz_parity.py 40 614 big 2>&1 | tail -1
