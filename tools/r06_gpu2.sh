cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest1.log
tail -15 gpurun_out/r06_gputest1.log
export LH_SOUP_NTRI=10000000 LH_SOUP_HALF=0.002 LH_SOUP_BUILD=device LH_VARIANT_COUNT=1
timeout 200 python tools/variant_once.py 4 20000000 wide8=0 > gpurun_out/r06_soup10m_q4_q8.txt 2>&1
timeout 200 python tools/variant_once.py 4 20000000 wide8=1 >> gpurun_out/r06_soup10m_q4_q8.txt 2>&1
cat gpurun_out/r06_soup10m_q4_q8.txt
bash tools/pmc_cmd.sh r06_pmc_soup10m_q4 python tools/variant_once.py 4 20000000 wide8=0 > /dev/null 2>&1
bash tools/pmc_cmd.sh r06_pmc_soup10m_q8 python tools/variant_once.py 4 20000000 wide8=1 > /dev/null 2>&1
cat gpurun_out/r06_pmc_soup10m_q4/pmc_summary.txt gpurun_out/r06_pmc_soup10m_q8/pmc_summary.txt | grep -v "^   SQ_\|GRBM" 
