#!/bin/bash
# per-kernel time of the device commit of the config-5 scene: tools/refbuild_prof.sh  (on the GPU box; writes gpurun_out/refbuild_prof/)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/refbuild_prof; mkdir -p gpurun_out/refbuild_prof
cat > /tmp/rb_one.py <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import lucille_amd as la
from lucille_amd import scenes
g = np.load("tests/golden/ao_c1.npz")
acc = la.HipAccel(0)
for k in range(int(g["ngeoms"])):
    P, I = scenes.tessellate(g["pos%d" % k], g["idx%d" % k], 8); acc.add_mesh(P, I)
acc.commit(on_device=True); acc.close()
P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/refbuild_prof -- python /tmp/rb_one.py < /dev/null > gpurun_out/refbuild_prof/run.log 2>&1
f=$(find gpurun_out/refbuild_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'P'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-60s calls %5s total %9.2f ms avg %8.3f ms" % (re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
P
