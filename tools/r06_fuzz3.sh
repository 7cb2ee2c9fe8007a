# tools/r06_fuzz3.sh (GPU box): after the host batch path was rebuilt (ring of staging blocks, three streams, copy pool) -- tools/fuzz_hostpath.py at the
# default ring and at small rings that wrap many times per batch; one more fresh-seed pass of the ray and AO sweeps at the last library
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_fuzz_final3.txt; : > $OUT
run() { echo "== $*" >> $OUT; ( timeout -k 5 ${T:-500} "$@" 2>&1 | grep -v amdgpu | tail -1 ) >> $OUT; }
T=300 run python tools/fuzz_hostpath.py 71 40
T=300 run env LH_PIPE_CHUNK=65536 LH_PIPE_DEPTH=2 LH_COPY_THREADS=2 python tools/fuzz_hostpath.py 72 30
T=300 run env LH_PIPE_CHUNK=262144 LH_PIPE_DEPTH=5 LH_COPY_THREADS=12 python tools/fuzz_hostpath.py 73 30
T=300 run env LH_PIPE_CHUNK=131072 LH_PIPE_DEPTH=8 LH_COPY_THREADS=0 python tools/fuzz_hostpath.py 74 30
FUZZ_BUDGET_S=150 run python tools/fuzz_parity.py 4000 651
FUZZ_BUDGET_S=100 run python tools/fuzz_ao.py 71 600
cat $OUT
