# tools/r06_fuzz5.sh (GPU box): after lh_bvh.h lh_zero_area_weight (near-collinear triangles bound deg_dcap) -- the two rounds that failed, then kind 9
# (collinear triangles among ordinary ones) and kind 4 (slivers) for minutes each, then the whole mix again, fresh seeds
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz_final5.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( timeout -k 5 ${T:-500} "$@" 2>&1 | grep -v amdgpu | tail -1 ) >> $OUT; }
T=900 run python tools/fuzz_parity.py 4000 662 x 99
T=900 run python tools/fuzz_parity.py 4000 661 x 359
run env FUZZ_KIND=9 FUZZ_BUDGET_S=240 python tools/fuzz_parity.py 4000 671
run env FUZZ_KIND=9 FUZZ_BUDGET_S=240 python tools/fuzz_parity.py 4000 672
run env FUZZ_KIND=4 FUZZ_BUDGET_S=120 python tools/fuzz_parity.py 4000 673
run env FUZZ_KIND=3 FUZZ_BUDGET_S=120 python tools/fuzz_parity.py 4000 674
FUZZ_BUDGET_S=240 run python tools/fuzz_parity.py 4000 675
FUZZ_BUDGET_S=100 run python tools/fuzz_ao.py 81 600
cat $OUT
