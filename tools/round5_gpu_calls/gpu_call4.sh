# round 5, call 4: no atomics in the mark at all (touched groups as plain byte stores, bits built by the list kernel) — the parity
# tests that touch instance-sized volumes, the instance frame A/B, the 8-volume workload on one GPU at 4 / 8 / 16 hardware queues,
# configs[2] / configs[3], and configs[2] through the C++ host pinned / not pinned from two kinds of parent process
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05d}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 400 python -m pytest tests/test_edges.py tests/test_gpu_parity.py tests/test_gpu_fullsize_golden.py tests/test_gpu_composite.py tests/test_reference_pipeline.py -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 6 $G/${T}_gpu_subset.log
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
python - <<PY
import json
for line in open("$G/${T}_instance_frame_ab.log"):
    line = line.strip()
    if line.startswith("=="): print(line)
    elif line.startswith("{"):
        d = json.loads(line)
        print("   free", d["free_running"]["us_per_frame"], "enqueue", d["free_running"]["host_enqueue_us_per_frame"], "sync", d["sync_per_frame"]["us_per_frame"],
              "launches", d["launches_per_frame"], {k.split(":")[1]: v["us_per_frame"] for k, v in d["gpu_kernels"].items()})
PY
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for Q in default; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  timeout -k 5 200 $B --instance-volumes 8 > $G/${T}_bench_instvol8_q$Q.json 2> $G/${T}_bench_instvol8_q$Q.err; echo "instvol8 queues=$Q rc=$?"
  python -c "
import json,sys
d=json.loads(open('$G/${T}_bench_instvol8_q$Q.json').read().strip().splitlines()[-1])
print('   ', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'host_enqueue_ms', d['config'].get('host_enqueue_ms_per_step_rank0'))"
done
unset GPU_MAX_HW_QUEUES
timeout -k 5 200 $B --volumes 8 > $G/${T}_bench_volumes8.json 2> $G/${T}_bench_volumes8.err; echo "volumes8 rc=$?"; head -c 260 $G/${T}_bench_volumes8.json; echo
timeout -k 5 200 $B --no-through-shim --no-scaling-leg --instances 4 > $G/${T}_bench_inst4.json 2> $G/${T}_bench_inst4.err; echo "inst4 rc=$?"; head -c 260 $G/${T}_bench_inst4.json; echo
timeout -k 5 200 $B --no-through-shim --no-scaling-leg --preset 5cm > $G/${T}_bench_5cm.json 2> $G/${T}_bench_5cm.err; echo "5cm rc=$?"; head -c 260 $G/${T}_bench_5cm.json; echo
timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1 | cut -c1-200
DSR_PIPELINED_VIEW=2 timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1 | cut -c1-200
timeout -k 5 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg > $G/${T}_bench_line.json 2> $G/${T}_bench_line.err; head -c 300 $G/${T}_bench_line.json; echo
