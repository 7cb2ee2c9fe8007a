#!/bin/bash
# AO-frame sensitivity to the regroup threshold / triangle batch (run on the GPU box)
for ma in 24 32 40 48; do for tb in 4 8 16; do
  echo -n "min_active $ma tri_batch $tb: "
  LH_MIN_ACTIVE=$ma LH_TRI_BATCH=$tb timeout 200 python tools/ao_probe.py 2048 64 6 2>&1 | grep -o "'frame_ms': [0-9.]*, 'value': [0-9.]*"
done; done
