cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz_final6.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( timeout -k 5 ${T:-500} "$@" 2>&1 | grep -v amdgpu | tail -1 ) >> $OUT; }
FUZZ_BUDGET_S=200 run python tools/fuzz_parity.py 4000 681
FUZZ_BUDGET_S=200 run python tools/fuzz_parity.py 4000 682
T=700 FUZZ_BUDGET_S=300 run python tools/fuzz_parity.py 3 683 big
cat $OUT
