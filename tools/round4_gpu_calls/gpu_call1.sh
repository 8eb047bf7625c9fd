# round 4, first GPU call: the suite on the new code paths, the raycast A/B, the standing lines through the host
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04a}
timeout 900 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
tail -n 15 $O/${T}_gpu_suite.log
timeout 300 python tools/ab_raycast_split.py > $O/${T}_raycast_split_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_raycast_split_ab.log
cat $O/${T}_raycast_split_ab.log | tail -n 14
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/${T}_bench_line.json") if l.startswith("{")][-1])
    print("value", d["value"], "shim", d.get("through_shim"), "k", {k:v["avg_us"] for k,v in d["kernels"].items()}, "iv8", (d.get("instance_volumes8_1gpu") or {}).get("value"))
except Exception as ex: print("bench parse failed", ex)
PY
tail -n 5 $O/${T}_bench.err
for two in 0 1; do
  if [ $two = 1 ]; then export DSR_HOST_TWO_STEP_UPDATE=1; fi
  timeout 200 python tools/bench_through_shim.py --steps 20 --warmup 5 > $O/${T}_shim_cfg1_two${two}.log 2>&1; tail -n 1 $O/${T}_shim_cfg1_two${two}.log
  timeout 200 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_two${two}.log 2>&1; tail -n 1 $O/${T}_shim_cfg2_two${two}.log
done
unset DSR_HOST_TWO_STEP_UPDATE
DSR_NO_PUBLISHED_STATUS=1 timeout 200 python tools/bench_through_shim.py --steps 20 --warmup 5 > $O/${T}_shim_cfg1_nopub.log 2>&1; tail -n 1 $O/${T}_shim_cfg1_nopub.log
timeout 200 python tools/bench_through_shim.py --steps 20 --warmup 5 --preset 5cm --instances 4 > $O/${T}_shim_cfg2_5cm.log 2>&1; tail -n 1 $O/${T}_shim_cfg2_5cm.log
timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --instance-volumes 8 --volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_torchrun1_both_legs.json 2>> $O/${T}_bench.err
for f in $O/${T}_bench_instvol8.json $O/${T}_bench_torchrun1_both_legs.json; do echo $f; grep '^{' $f | head -c 300 | tail -c 220; echo; done
tail -n 5 $O/${T}_bench.err
