cd $GRAFT_REPO_ROOT/tools/ubench
for lds in 0 20000 26000 32000 40000 53000 80000; do
  GATHER_LDS=$lds timeout 120 ./gather 64 200 4096 0 1 8 2>&1 | grep -v amdgpu
done
