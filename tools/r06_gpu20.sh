cd $GRAFT_REPO_ROOT
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest4.log
tail -4 gpurun_out/r06_gputest4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 900 python bench.py > gpurun_out/r06_bench_line_full.json 2> gpurun_out/r06_bench_line_full.err; tail -2 gpurun_out/r06_bench_line_full.err | cut -c1-200; head -c 200 gpurun_out/r06_bench_line_full.json
