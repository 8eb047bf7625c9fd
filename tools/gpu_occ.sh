#!/bin/bash
# same-box A/B: the library as of the previous commit (libdsr_hip_prev.so) against the current one with the block map on / off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
  DSR_HIP_LIB=$PWD/dynslam_amd/csrc/libdsr_hip_prev.so timeout 200 python tools/bench_variants.py "" 2>/dev/null | sed "s/^/prev /" | tee -a gpurun_out/prev_ab.log
  DSR_OCC=1 timeout 200 python tools/bench_variants.py "" 2>/dev/null | sed "s/^/cur map=1 /" | tee -a gpurun_out/prev_ab.log
  DSR_OCC=0 timeout 200 python tools/bench_variants.py "" 2>/dev/null | sed "s/^/cur map=0 /" | tee -a gpurun_out/prev_ab.log
done
