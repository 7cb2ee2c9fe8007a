# tools/r06_fuzz.sh (GPU box): round 6's randomised sweeps at the round's final kernels, fresh seeds, every output poisoned (LH_POISON_OUTPUTS=1 is the
# tools' default); fuzz_parity's kind 9 = a few collinear triangles among ordinary ones (the leaf-box rule of DESIGN 4.5)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz_final4.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( timeout -k 5 ${T:-500} "$@" 2>&1 | grep -v amdgpu | tail -1 ) >> $OUT; }
FUZZ_BUDGET_S=300 run python tools/fuzz_parity.py 4000 661
FUZZ_BUDGET_S=300 run python tools/fuzz_parity.py 4000 662
T=1500 FUZZ_BUDGET_S=900 run python tools/fuzz_parity.py 3 663 big
FUZZ_BUDGET_S=150 run python tools/fuzz_ao.py 77 600
FUZZ_BUDGET_S=150 run python tools/fuzz_ao.py 78 600
T=300 run python tools/fuzz_beams.py 73 60
T=300 run python tools/fuzz_state.py 7000 150
T=400 run python tools/fuzz_pt.py 73 40
T=300 run python tools/fuzz_hostpath.py 75 30
cat $OUT
