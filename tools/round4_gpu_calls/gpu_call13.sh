# (a) the no-flag bench with the collector paused in the timed region; (b) which through-shim leg runs behind which; (c) configs[2]
# through the host, stand-alone, 45 timed frames
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r04w}
export DSR_BENCH_STEP_TIMES=1
pr() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e=d.get('step_enqueued_ms') or [0]
ts=d.get('through_shim') or {}
print(sys.argv[1].split('/')[-1], 'value', d['value'], 'steps', d['steps'], 'max enqueue gap ms', round(max(b-a for a,b in zip([0]+e,e)),2), 'shim', ts.get('frames_per_s'), 'configs2', (ts.get('configs2') or {}).get('frames_per_s'))
PY
}
timeout -k 5 120 python bench.py --no-cpu-baseline --no-scaling-leg > $O/${T}_bench_default_nogc.json 2> $O/${T}_bench.err; pr $O/${T}_bench_default_nogc.json
DSR_BENCH_SHIM_ORDER=configs2-first timeout -k 5 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-leg > $O/${T}_bench_cfg2first.json 2>> $O/${T}_bench.err; pr $O/${T}_bench_cfg2first.json
DSR_BENCH_NO_POOL=1 timeout -k 5 90 python tools/bench_through_shim.py --steps 45 --warmup 5 --instances 4 > $O/${T}_shim_cfg2_45.log 2>&1; echo "cfg2 stand-alone 45: $(tail -n 1 $O/${T}_shim_cfg2_45.log | cut -c1-130)"
