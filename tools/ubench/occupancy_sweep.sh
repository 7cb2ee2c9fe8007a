# tools/ubench/occupancy_sweep.sh (GPU box): dependent 64-byte / 128-byte gathers at 64 MB, one chain per lane (modes 0, 8, 5)
# and per quad (modes 1, 6), resident waves capped through dynamic LDS; GATHER_PERTURB=0 reproduces round 1's figures for the
# per-quad modes (unperturbed links: the chains merge and the L2 hit rate climbs)
cd $GRAFT_REPO_ROOT/tools/ubench
echo "== links perturbed by the chain's own running sum (every mode)"
for lds in 0 26000 40000 53000 80000; do
  GATHER_LDS=$lds timeout 120 ./gather 64 200 4096 0 8 1 5 6 2>&1 | grep -v amdgpu
done
echo "== round 1's per-quad modes: links followed unmodified"
GATHER_PERTURB=0 timeout 120 ./gather 64 200 4096 1 6 2>&1 | grep -v amdgpu
GATHER_PERTURB=0 GATHER_LDS=40000 timeout 120 ./gather 64 200 4096 1 6 2>&1 | grep -v amdgpu
