# the driver's command with the through-shim legs measured before the parent process opens the device
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r04u}
timeout -k 5 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_cmd.json 2> $O/${T}_bench.err
python - $O/${T}_bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ts=d.get('through_shim') or {}
print('value', d['value'], 'shim', ts.get('frames_per_s'), 'configs2', (ts.get('configs2') or {}).get('frames_per_s'), 'instvol8', (d.get('instance_volumes8_1gpu') or {}).get('value'), 'cpu', d['cpu_baseline']['value'], 'frac', d['roofline']['frac'], d['roofline']['traffic_frac'], d['roofline']['traffic_source'])
PY
