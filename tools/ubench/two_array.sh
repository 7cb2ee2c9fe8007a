# tools/ubench/two_array.sh (GPU box; VERDICT r05 "next" 2): does a 64-byte 8-wide node pay on S-soup-10M?  Node records and triangle
# records in arrays of their OWN size -- 8-wide nodes of 128 B at 240 MB against 64 B at 120 MB, triangles at 480 MB, eight node
# steps per triangle step (the walk's 39.7 : 5.1), three workgroups per CU (the 8-wide walk's occupancy), then four.
# Round 5's log (r05_ubench_gather_alu.log) gathered both record sizes out of the SAME 717 MB.
cd $GRAFT_REPO_ROOT/tools/ubench
run() { echo "-- $*"; env "$@" 2>&1 | grep -v amdgpu; }
for lds in 53000 40000; do
  echo "== GATHER_LDS=$lds ($((163840 / lds)) workgroups per CU)"
  for alu in 0 256; do
    run GATHER_LDS=$lds GATHER_ALU=$alu GATHER_NODE_MB=240 GATHER_TRI_MB=480 timeout 120 ./gather 0 200 4096 11
    run GATHER_LDS=$lds GATHER_ALU=$alu GATHER_NODE_MB=120 GATHER_TRI_MB=480 timeout 120 ./gather 0 200 4096 10
  done
  # what the footprints alone are worth: the node array by itself, the 4-wide tree's sizes (64-byte 4-wide nodes: 285 MB), one array of everything
  run GATHER_LDS=$lds GATHER_NODE_MB=285 GATHER_TRI_MB=480 timeout 120 ./gather 0 200 4096 10
  run GATHER_LDS=$lds GATHER_NODE_MB=60 GATHER_TRI_MB=480 timeout 120 ./gather 0 200 4096 10
  run GATHER_LDS=$lds timeout 120 ./gather 120 200 4096 0
  run GATHER_LDS=$lds timeout 120 ./gather 240 200 4096 5
  run GATHER_LDS=$lds timeout 120 ./gather 717 200 4096 0 5
done
