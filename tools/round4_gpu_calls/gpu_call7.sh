# round 4, seventh GPU call: device-scope ordering events + lazy view event — the WHOLE suite again, then the lines they should move
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=${1:-r04g}
timeout -k 5 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2> $O/${T}_bench.err
DSR_EVENT_SYSTEM_SCOPE=1 timeout -k 5 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame_sysscope.json 2>> $O/${T}_bench.err
timeout -k 5 200 python bench.py --preset 5cm --steps 45 --warmup 5 --no-cpu-baseline > $O/${T}_bench_5cm.json 2>> $O/${T}_bench.err
timeout -k 5 200 python bench.py --instance-volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
DSR_EVENT_SYSTEM_SCOPE=1 timeout -k 5 200 python bench.py --instance-volumes 8 --steps 40 --warmup 10 --no-cpu-baseline > $O/${T}_bench_instvol8_sysscope.json 2>> $O/${T}_bench.err
for f in $O/${T}_instance_frame.json $O/${T}_instance_frame_sysscope.json; do python - <<PY
import json
d=json.loads([l for l in open("$f") if l.startswith("{")][-1])
print("$f".split("/")[-1], d["free_running"]["us_per_frame"], d["free_running"]["host_enqueue_us_per_frame"], d["sync_per_frame"], d["gpu_us_per_frame"])
PY
done
for f in $O/${T}_bench_5cm.json $O/${T}_bench_instvol8.json $O/${T}_bench_instvol8_sysscope.json; do echo $f; grep '^{' $f | head -c 230 | tail -c 140; echo; done
DSR_BENCH_NO_POOL=1 timeout -k 5 100 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2.log 2>&1; echo "cfg2: $(tail -n 1 $O/${T}_shim_cfg2.log | cut -c1-130)"
DSR_BENCH_NO_POOL=1 DSR_PIPELINED_VIEW=2 timeout -k 5 100 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_pv2.log 2>&1; echo "cfg2 pv2: $(tail -n 1 $O/${T}_shim_cfg2_pv2.log | cut -c1-130)"
DSR_BENCH_NO_POOL=1 DSR_PIPELINED_VIEW=2 timeout -k 5 100 python tools/bench_through_shim.py --steps 20 --warmup 5 > $O/${T}_shim_cfg1_pv2.log 2>&1; echo "cfg1 pv2: $(tail -n 1 $O/${T}_shim_cfg1_pv2.log | cut -c1-130)"
timeout -k 5 600 python -m pytest tests -m gpu -q --timeout 240 > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
tail -n 8 $O/${T}_gpu_suite.log
