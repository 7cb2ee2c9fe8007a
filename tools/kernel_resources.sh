#!/bin/bash
# round 5: VGPRs / scratch / occupancy of every instantiation of the traversal kernels, from the compiler's own remarks (no GPU
# needed).  The persistent walk must stay at <= 128 VGPRs (four waves per SIMD) with its 16 bytes of scratch: a change that
# looks free in the source can cost 80 bytes of spills per lane (r05: a `continue` in the refill; the path tracer lost 12 %).
#   tools/kernel_resources.sh [file.hip]
cd "$(dirname "$0")/../lucille_amd/csrc" || exit 1
f=${1:-lh_kernels.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - \
  | sed 's/Function Name: _ZN12_GLOBAL__N_1[0-9]*//; s/EEEv12lh_dev_scene[A-Za-z0-9_]*//; s/ILb/</; s/ELb/,/g; s/ELi/,/g; s/EEv[A-Za-z0-9_]*//' | sort
