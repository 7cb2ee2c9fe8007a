# tools/r06_fuzz.sh (GPU box): round 6's randomised sweeps, fresh seeds, every output poisoned (LH_POISON_OUTPUTS=1 is the tools' default)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( timeout ${T:-400} "$@" 2>&1 | grep -v amdgpu | tail -2 ) >> $OUT; }
FUZZ_BUDGET_S=150 run python tools/fuzz_parity.py 2000 601
FUZZ_BUDGET_S=150 run python tools/fuzz_parity.py 2000 602
FUZZ_BUDGET_S=120 run python tools/fuzz_parity.py 40 603 big
FUZZ_BUDGET_S=100 run python tools/fuzz_ao.py 61 400
FUZZ_BUDGET_S=100 run python tools/fuzz_ao.py 62 400
T=200 run python tools/fuzz_beams.py 61
T=200 run python tools/fuzz_state.py 3000
T=300 run python tools/fuzz_pt.py 61
T=200 run python tools/fuzz_hostpath.py 61
cat $OUT
