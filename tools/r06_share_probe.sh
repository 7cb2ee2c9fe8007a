# tools/r06_share_probe.sh (GPU box, round 6): a rank's share of the config-5 frame (rank 0 and rank 7 of 8, tools/rank_stage_probe.py) against
# the knobs that could shorten its fixed part: guided ranges (LH_GSS), the fused AO stage's visit budget and range length, the sweep's grid
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python tools/rank_stage_probe.py 8 2>&1 | grep "world"; }
run LH_GSS=0
run LH_GSS=1
run LH_GSS=2
run LH_GSS=4
run LH_GSS=2 LH_AO_CHUNK=2048
run LH_AO_BUDGET=128
run LH_AO_BUDGET=256
run LH_SWEEP_MULT=4
run LH_SWEEP_MULT=2
run LH_GSS=2 LH_SWEEP_MULT=4 LH_AO_BUDGET=256
