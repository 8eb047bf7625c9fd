# round 6: the driver's three round-end commands once more on the final commit (another box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r06final}
timeout -k 5 700 python -m pytest tests -x -q -m gpu > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log; tail -n 3 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 2 $G/${T}_smoke.log
SECONDS=0; timeout -k 5 400 python bench.py > $G/${T}_bench_default.json 2> $G/${T}_bench_default.err; echo "bench (no flags) rc=$? wall ${SECONDS}s"
SECONDS=0; timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $G/${T}_bench_line.json 2> $G/${T}_bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<P
import json
for n in ('bench_default','bench_line'):
    d=json.loads(open('$G/${T}_'+n+'.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['roofline']['measured_copy_spread_GBps'], (d.get('instance_volumes8_1gpu') or {}).get('value'), [(k, (d.get(k) or {}).get('value'), (d.get(k) or {}).get('status')) for k in ('configs2','configs3_1gpu','configs4_short')])
P
