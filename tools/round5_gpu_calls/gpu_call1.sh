# round 5, call 1: the one-workgroup kernels of instance-sized volumes (k_small.h), the fused view split, stream sharing, the pruned
# library — whole GPU suite, then the instance frame at round 4's library / this one / this one on a shared stream, then a bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05a}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 600 python -m pytest tests -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 30 $G/${T}_gpu_suite.log
{
  echo "== round 4 library (0c5fbef), two-step split"
  DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r04/libdsr_hip.so DSR_HIP_LIB_OLDER_ABI=1 timeout -k 5 120 python tools/bench_instance_frame.py --two-step-split
  echo "== this library, two-step split"
  timeout -k 5 120 python tools/bench_instance_frame.py --two-step-split
  echo "== this library, one-call split"
  timeout -k 5 120 python tools/bench_instance_frame.py
  echo "== this library, one-call split, instance on the view engine's stream"
  timeout -k 5 120 python tools/bench_instance_frame.py --share-stream
  echo "== round 4 library again (box drift)"
  DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r04/libdsr_hip.so DSR_HIP_LIB_OLDER_ABI=1 timeout -k 5 120 python tools/bench_instance_frame.py --two-step-split
} > $G/${T}_instance_frame_ab.log 2>&1
grep -o '"free_running": {"us_per_frame": [0-9.]*\|"sync_per_frame": {"us_per_frame": [0-9.]*\|"launches_per_frame": [0-9.]*\|^==.*' $G/${T}_instance_frame_ab.log
timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim > $G/${T}_bench_line.json 2> $G/${T}_bench.err; echo "bench rc=$?"
head -c 1500 $G/${T}_bench_line.json; echo
tail -n 5 $G/${T}_bench.err
