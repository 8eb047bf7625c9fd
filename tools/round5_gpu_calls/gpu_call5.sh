# round 5, call 5: D0 / B of k_small_alloc_visible loop over SET bytes only (call 4: 80 us of sixteen serial byte reloads per row);
# poses handed over as PoseArg; configs[4] sustained on this round's code; the whole suite under DSR_PIPELINED_VIEW=2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05e}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 300 python -m pytest tests/test_edges.py tests/test_gpu_parity.py -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 4 $G/${T}_gpu_subset.log
{
  echo "== this library, one-call split"
  timeout -k 5 120 python tools/bench_instance_frame.py
  echo "== this library, one-call split, instance on the view engine's stream"
  timeout -k 5 120 python tools/bench_instance_frame.py --share-stream
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
timeout -k 5 200 $B --instance-volumes 8 > $G/${T}_bench_instvol8.json 2> $G/${T}_bench_instvol8.err; echo "instvol8 rc=$?"
python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8.json').read().strip().splitlines()[-1])
c=d['config']
print('   ', d['value'], d['unit'], 'ms/step', d['ms_per_step'], {k: c.get(k) for k in ('host_enqueue_ms_per_step_rank0','chain_us_max_rank','gather_us','composite_us','rccl_ranks')})"
timeout -k 5 240 python tools/bench_cfg5_sustained.py > $G/${T}_cfg5_sustained_4541frames.json 2> $G/${T}_cfg5_sustained.err; echo "cfg5 rc=$?"; head -c 700 $G/${T}_cfg5_sustained_4541frames.json; echo
DSR_PIPELINED_VIEW=2 timeout -k 5 500 python -m pytest tests -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_suite_pv2.log 2>&1; echo "suite(pv2) rc=$?" >> $G/${T}_gpu_suite_pv2.log
tail -n 6 $G/${T}_gpu_suite_pv2.log
