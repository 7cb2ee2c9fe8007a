# tools/r06_hostpath_sweep.sh (GPU box): the pipelined host path under chunk size x ring depth x copy threads x copy mode -> gpurun_out/r06_hostpath_sweep.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_hostpath_sweep.txt; : > $OUT
echo "cpus $(nproc), cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" >> $OUT
run() { echo -n "$* -> " >> $OUT; env "$@" timeout -k 5 300 python tools/hostpath_once.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT; }
for mode in 2 0 1; do
  run LH_COPY_MODE=$mode LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=3 LH_COPY_THREADS=7
  run LH_COPY_MODE=$mode LH_PIPE_CHUNK=1048576 LH_PIPE_DEPTH=4 LH_COPY_THREADS=7
done
run LH_COPY_MODE=1 LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=4 LH_COPY_THREADS=7
run LH_COPY_MODE=1 LH_PIPE_CHUNK=4194304 LH_PIPE_DEPTH=3 LH_COPY_THREADS=7
run LH_COPY_MODE=1 LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=3 LH_COPY_THREADS=3
run LH_COPY_MODE=1 LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=3 LH_COPY_THREADS=12
run LH_COPY_MODE=1 LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=3 LH_COPY_THREADS=0
cat $OUT
