#!/bin/bash
# usage (GPU box): tools/pmc_cmd.sh <tag> <command...>
# Separate rocprofv3 --pmc passes (no tracing flags) of <command>: SQ wave/instruction counters, TCC hit/miss/EA.
# (TA_* / TCP_* stall and latency counters HANG rocprofv3 on this pool -- each pass ran into its timeout, r02b -- so they are not collected.)
# Per kernel, the LARGEST dispatch's counters are summarised into gpurun_out/<tag>/pmc_summary.txt.
R=$GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES TCP_TCC_READ_REQ_sum
PASSES
cd $R
python $R/tools/pmc_cmd_summarize.py $OUT > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/p[0-9]
cat $OUT/pmc_summary.txt
