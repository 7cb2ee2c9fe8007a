# tools/r06_hostpath_sweep.sh (GPU box): the pipelined host path (lh_accel_intersect_host, 20 M rays) under chunk size x ring depth x copy threads x the
# priority pool its streams come from -> gpurun_out/r06_hostpath_sweep.txt   (profiles/r06_hostpath.txt holds the round's runs of it, also of the
# variants that were removed again: streaming-store copies, spare workgroups, a D2H stream of its own)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_hostpath_sweep.txt; : > $OUT
echo "cpus $(nproc), cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" >> $OUT
run() { echo -n "$* -> " >> $OUT; env "$@" timeout -k 5 300 python tools/hostpath_once.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT; }
run LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=3
run LH_PIPE_CHUNK=2097152 LH_PIPE_DEPTH=4
run LH_PIPE_CHUNK=1048576 LH_PIPE_DEPTH=4
run LH_PIPE_CHUNK=4194304 LH_PIPE_DEPTH=3
run LH_PIPE_PRIORITY=0
run LH_PIPE_PRIORITY=-1
run LH_COPY_THREADS=3
run LH_COPY_THREADS=12
run LH_COPY_THREADS=0
cat $OUT
