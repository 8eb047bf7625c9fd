# round 5, call 2: the mark's visBits atomics deduplicated per wave (call 1: 2.9 ms of same-word atomics) — the parity tests that
# touch instance-sized volumes, the instance frame A/B again, the 8-volume workload on one GPU, and where the C++ host runs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05b}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 400 python -m pytest tests/test_edges.py tests/test_gpu_parity.py tests/test_gpu_fullsize_golden.py tests/test_gpu_composite.py -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 12 $G/${T}_gpu_subset.log
{
  echo "== round 4 library (0c5fbef), two-step split"
  DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r04/libdsr_hip.so DSR_HIP_LIB_OLDER_ABI=1 timeout -k 5 120 python tools/bench_instance_frame.py --two-step-split
  echo "== this library, one-call split"
  timeout -k 5 120 python tools/bench_instance_frame.py
  echo "== this library, one-call split, instance on the view engine's stream"
  timeout -k 5 120 python tools/bench_instance_frame.py --share-stream
  echo "== this library, two-step split, own streams"
  timeout -k 5 120 python tools/bench_instance_frame.py --two-step-split
} > $G/${T}_instance_frame_ab.log 2>&1
grep -o '"free_running": {"us_per_frame": [0-9.]*\|"sync_per_frame": {"us_per_frame": [0-9.]*\|"launches_per_frame": [0-9.]*}\|^==.*' $G/${T}_instance_frame_ab.log | grep -v '"launches_per_frame": [0-9.]*}$' 
timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8.json 2> $G/${T}_bench_instvol8.err; echo "instvol8 rc=$?"
head -c 400 $G/${T}_bench_instvol8.json; echo
DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r04/libdsr_hip.so DSR_HIP_LIB_OLDER_ABI=1 timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_r04lib.json 2> $G/${T}_bench_instvol8_r04lib.err; echo "instvol8 (r04 lib) rc=$?"
head -c 400 $G/${T}_bench_instvol8_r04lib.json; echo
timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --volumes 8 > $G/${T}_bench_volumes8.json 2> $G/${T}_bench_volumes8.err; echo "volumes8 rc=$?"
head -c 300 $G/${T}_bench_volumes8.json; echo
timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg --instances 4 > $G/${T}_bench_inst4.json 2> $G/${T}_bench_inst4.err; echo "inst4 rc=$?"
head -c 300 $G/${T}_bench_inst4.json; echo
bash tools/next_round/where_does_the_host_run.sh $T 2>&1 | tail -n 25
