# round 4, third GPU call (every step under a hard timeout): diagnostics + A/Bs of the new host / stream structure
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DSR_BENCH_NO_POOL=1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04c}
timeout -k 5 90 python tools/diag/oob_vistype.py > $O/${T}_oob_vistype.log 2>&1; tail -n 30 $O/${T}_oob_vistype.log
timeout -k 5 240 python tools/ab_engine_env.py --repeat 1 'DSR_RAYCAST_SPLIT=0' 'DSR_RAYCAST_SPLIT=64,DSR_RAYCAST_TAIL_MODE=8' 'DSR_RAYCAST_SPLIT=64,DSR_RAYCAST_TAIL_MODE=1' 'DSR_RAYCAST_SPLIT=64,DSR_RAYCAST_TAIL_MODE=4' 'DSR_RAYCAST_SPLIT=48,DSR_RAYCAST_TAIL_MODE=1' 'DSR_RAYCAST_SPLIT=32,DSR_RAYCAST_TAIL_MODE=1' 'DSR_RAYCAST_SPLIT=96,DSR_RAYCAST_TAIL_MODE=1' 'DSR_RAYCAST_SPLIT=24,DSR_RAYCAST_TAIL_MODE=4' > $O/${T}_raycast_split_kernels.log 2>&1; tail -n 10 $O/${T}_raycast_split_kernels.log
timeout -k 5 200 python tools/ab_engine_env.py 'DSR_OVERLAP_PREPARE=0' 'DSR_OVERLAP_PREPARE=1' 'DSR_OVERLAP_PREPARE=1,DSR_INTEGRATE_XLDS=0' > $O/${T}_overlap_prepare_ab.log 2>&1; tail -n 8 $O/${T}_overlap_prepare_ab.log
for pv in 1 0; do
  DSR_PIPELINED_VIEW=$pv timeout -k 5 120 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_pv${pv}.log 2>&1; echo "cfg2 pipelined=$pv $(tail -n 1 $O/${T}_shim_cfg2_pv${pv}.log | cut -c1-200)"
  DSR_PIPELINED_VIEW=$pv timeout -k 5 120 python tools/bench_through_shim.py --steps 20 --warmup 5 > $O/${T}_shim_cfg1_pv${pv}.log 2>&1; echo "cfg1 pipelined=$pv $(tail -n 1 $O/${T}_shim_cfg1_pv${pv}.log | cut -c1-200)"
done
DSR_HOST_PAGEABLE_PREVIEWS=1 timeout -k 5 120 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_pageable.log 2>&1; echo "cfg2 pageable previews $(tail -n 1 $O/${T}_shim_cfg2_pageable.log | cut -c1-200)"
timeout -k 5 120 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 --preset 5cm > $O/${T}_shim_cfg2_5cm.log 2>&1; echo "cfg2 5cm $(tail -n 1 $O/${T}_shim_cfg2_5cm.log | cut -c1-200)"
timeout -k 5 420 python -m pytest -m gpu -q -x --timeout 200 tests/test_gpu_parity.py tests/test_reference_compiles.py tests/test_edges.py tests/test_gpu_composite.py tests/test_multigpu_gloo.py tests/test_shim.py > $O/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $O/${T}_gpu_subset.log; tail -n 12 $O/${T}_gpu_subset.log
