# tools/r06_profiles.sh (GPU box): round 6's evidence at the round's final kernels -- per bench leg rocprofv3 --kernel-trace --stats and separate
# FETCH_SIZE / WRITE_SIZE passes (tools/profile_round2.sh), SQ / TCC counter passes over the config-5 frame as one batch and over the path-traced
# frame (tools/pmc_cmd.sh), the kernel timeline of a path-traced frame, and the full default bench line
cd $GRAFT_REPO_ROOT
bash tools/profile_round2.sh r06 main hbm ao pt > gpurun_out/r06_profile_round.log 2>&1
bash tools/pmc_cmd.sh r06_pmc_ao_dense python $GRAFT_REPO_ROOT/tools/rank_stage_probe.py 1 4096 device > /dev/null 2>&1
cp gpurun_out/r06_pmc_ao_dense/pmc_summary.txt gpurun_out/r06_pmc_ao_dense.txt
timeout -k 5 900 python bench.py > gpurun_out/r06_bench_line_full.json 2> gpurun_out/r06_bench_line_full.err
tail -c 600 gpurun_out/r06_bench_line_full.json
ls gpurun_out/r06/*
