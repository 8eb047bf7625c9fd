# configs[2] through the host inside the driver's 20-step command: behind the configs[1] leg (as shipped) and before it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r04v}
pr() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ts=d.get('through_shim') or {}
print(sys.argv[1].split('/')[-1], 'value', d['value'], 'shim', ts.get('frames_per_s'), 'configs2', (ts.get('configs2') or {}).get('frames_per_s'), 'instvol8', (d.get('instance_volumes8_1gpu') or {}).get('value'))
PY
}
DSR_BENCH_SHIM_ORDER=configs2-first timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_cfg2first.json 2> $O/${T}_bench.err; pr $O/${T}_bench_cfg2first.json
timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_asshipped.json 2>> $O/${T}_bench.err; pr $O/${T}_bench_asshipped.json
