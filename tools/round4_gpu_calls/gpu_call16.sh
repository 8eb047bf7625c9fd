# configs[2] through the host: stand-alone tool vs inside bench.py, with and without the frame-generation worker pool
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r04t}
L=$O/${T}_cfg2_through_host_where.log
: > $L
echo "tool, no pool:   $(DSR_BENCH_NO_POOL=1 timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1 | cut -c1-110)" >> $L
echo "tool, pool:      $(timeout -k 5 60 python tools/bench_through_shim.py --steps 20 --warmup 5 --instances 4 2>&1 | tail -n 1 | cut -c1-110)" >> $L
pr() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ts=d.get('through_shim') or {}
print('value', d['value'], 'shim', ts.get('frames_per_s'), 'configs2', (ts.get('configs2') or {}).get('frames_per_s'))
PY
}
DSR_BENCH_NO_POOL=1 timeout -k 5 90 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_b1.json 2> $O/${T}_bench.err; echo "bench, no pool:  $(pr $O/${T}_b1.json)" >> $L
cat $L
