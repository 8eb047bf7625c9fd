# round 5, call 7: batch integrate takes the specialisation every active volume qualifies for (call 6: a -0 in a pose made the volumes differ); D0 requests the first group of all nine rows together
# with shared streams as the default of host-driven engines; the instance frame with a lighter Python host
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r05g}
G=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $G
timeout -k 5 500 python -m pytest tests/test_gpu_batch.py tests/test_edges.py tests/test_multigpu_gloo.py tests/test_gpu_parity.py -m gpu -q --timeout 240 --maxfail=12 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 25 $G/${T}_gpu_subset.log | cut -c1-300
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
        print("   free", d["free_running"], "untimed", d.get("free_running_untimed_calls"), "sync", d["sync_per_frame"]["us_per_frame"],
              "launches", d["launches_per_frame"], {k.split(":")[1]: v["us_per_frame"] for k, v in d["gpu_kernels"].items()})
PY
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
show() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1])
c=d['config']
print('   ', d['value'], d['unit'], 'ms/step', d['ms_per_step'], {k: c.get(k) for k in ('host_enqueue_ms_per_step_rank0','chain_us_max_rank','gather_us','composite_us','rccl_ranks')}, {k: v.get('avg_us') for k, v in (d.get('kernels') or {}).items()})"; }
timeout -k 5 200 $B --instance-volumes 8 > $G/${T}_bench_instvol8.json 2> $G/${T}_bench_instvol8.err; echo "instvol8 rc=$?"; show $G/${T}_bench_instvol8.json; tail -n 3 $G/${T}_bench_instvol8.err
timeout -k 5 200 $B --volumes 8 > $G/${T}_bench_volumes8.json 2> $G/${T}_bench_volumes8.err; echo "volumes8 rc=$?"; show $G/${T}_bench_volumes8.json; tail -n 3 $G/${T}_bench_volumes8.err
timeout -k 5 200 $B --no-through-shim --no-scaling-leg --instances 4 > $G/${T}_bench_inst4.json 2> $G/${T}_bench_inst4.err; echo "inst4 rc=$?"; head -c 260 $G/${T}_bench_inst4.json; echo
timeout -k 5 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_torchrun1.json 2> $G/${T}_bench_instvol8_torchrun1.err; echo "instvol8 torchrun rc=$?"; show $G/${T}_bench_instvol8_torchrun1.json
timeout -k 5 160 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim > $G/${T}_bench_line.json 2> $G/${T}_bench_line.err; python -c "
import json
d=json.loads(open('$G/${T}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v.get('avg_us') for k, v in d['kernels'].items()}, 'instvol8_1gpu', (d.get('instance_volumes8_1gpu') or {}).get('value'), ((d.get('instance_volumes8_1gpu') or {}).get('config') or {}).get('host_enqueue_ms_per_step_rank0'))"
