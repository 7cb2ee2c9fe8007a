cd $GRAFT_REPO_ROOT
timeout -k 5 900 python bench.py > gpurun_out/r06_bench_line_full.json 2> gpurun_out/r06_bench_line_full.err
tail -3 gpurun_out/r06_bench_line_full.err | cut -c1-300
head -c 400 gpurun_out/r06_bench_line_full.json; echo
timeout -k 5 600 python tools/experiments/two_halves_probe.py 0 2>&1 | grep -v amdgpu | tee gpurun_out/r06_two_halves_probe.txt
