cd $GRAFT_REPO_ROOT
bash tools/r06_share_probe.sh > gpurun_out/r06_share_probe.txt 2>&1
cat gpurun_out/r06_share_probe.txt
export LH_SOUP_NTRI=10000000 LH_SOUP_HALF=0.002 LH_SOUP_BUILD=device LH_VARIANT_COUNT=1
bash tools/pmc_cmd.sh r06_pmc_soup10m_q4 python $GRAFT_REPO_ROOT/tools/variant_once.py 4 20000000 wide8=0 > /dev/null 2>&1
bash tools/pmc_cmd.sh r06_pmc_soup10m_q8 python $GRAFT_REPO_ROOT/tools/variant_once.py 4 20000000 wide8=1 > /dev/null 2>&1
cat gpurun_out/r06_pmc_soup10m_q4/pmc_summary.txt gpurun_out/r06_pmc_soup10m_q8/pmc_summary.txt | grep -v "^   SQ_\|GRBM" 
