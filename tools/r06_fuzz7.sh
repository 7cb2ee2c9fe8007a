# tools/r06_fuzz7.sh (GPU box): single-kind passes of fuzz_parity.py at the fixed library (both builders): ties (5), fans (7), one huge triangle (8), strip meshes (1), sheets (2)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz_final7.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( timeout -k 5 ${T:-400} "$@" 2>&1 | grep -v amdgpu | tail -1 ) >> $OUT; }
for ks in "5 731" "7 732" "8 733" "1 734" "2 735"; do set -- $ks; run env FUZZ_KIND=$1 FUZZ_BUDGET_S=80 python tools/fuzz_parity.py 4000 $2; done
cat $OUT
