# round 4, fourth GPU call (hard timeouts): parity of the fixes, hardware-queue experiments for configs[2] through the host
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DSR_BENCH_NO_POOL=1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04d}
run_cfg2() { # name, env...
  name=$1; shift
  env "$@" timeout -k 5 100 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_${name}.log 2>&1
  echo "cfg2 $name: $(tail -n 1 $O/${T}_shim_cfg2_${name}.log | cut -c1-130)"
}
run_cfg2 pv0 DSR_PIPELINED_VIEW=0
run_cfg2 pv1 DSR_PIPELINED_VIEW=1
run_cfg2 pv1_prio DSR_PIPELINED_VIEW=1 DSR_STREAM_PRIORITY=1
run_cfg2 pv0_prio DSR_PIPELINED_VIEW=0 DSR_STREAM_PRIORITY=1
run_cfg2 pv1_q8 DSR_PIPELINED_VIEW=1 GPU_MAX_HW_QUEUES=8
run_cfg2 pv1_q16 DSR_PIPELINED_VIEW=1 GPU_MAX_HW_QUEUES=16
run_cfg2 pv1_q16_prio DSR_PIPELINED_VIEW=1 GPU_MAX_HW_QUEUES=16 DSR_STREAM_PRIORITY=1
run_cfg2 pv0_q8 DSR_PIPELINED_VIEW=0 GPU_MAX_HW_QUEUES=8
timeout -k 5 300 python -m pytest -m gpu -q -x --timeout 200 tests/test_gpu_parity.py tests/test_reference_compiles.py "tests/test_gpu_fullsize.py::test_byte_tallies_of_k_integrate_equal_an_independent_count" > $O/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $O/${T}_gpu_subset.log; tail -n 12 $O/${T}_gpu_subset.log
