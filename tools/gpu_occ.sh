#!/bin/bash
# block map: how much is there to gain?  default size (8 slots per voxel block), 64 per block (conflicts ~0.5 %), off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "1 0" "1 536870912" "0 0" "1 0" "1 536870912" "0 0"; do
  set -- $cfg
  if [ "$2" = 0 ]; then unset DSR_OCC_ENTRIES; else export DSR_OCC_ENTRIES=$2; fi
  DSR_OCC=$1 timeout 200 python tools/bench_variants.py "" 2>/dev/null | sed "s/^/map=$1 slots=$2 /" | tee -a gpurun_out/map_size_ab.log
done
