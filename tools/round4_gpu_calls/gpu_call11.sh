# why the default 45-step run reads 2.0 ms per step when 20 steps read 1.13: step count, the legs before it, the round-3 library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r04y}
L=$O/${T}_steps45.log
: > $L
export DSR_BENCH_STEP_TIMES=1
run() { local label=$1; shift; echo "== $label" >> $L; (timeout -k 5 100 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value':d['value'],'steps':d['steps'],'ms_per_step':d['ms_per_step'],'kernels':{k:v['avg_us'] for k,v in d['kernels'].items()},'enq':d.get('step_enqueued_ms')}))") >> $L; }
F="--no-cpu-baseline --no-through-shim"
run "45 steps, no scaling leg" python bench.py --steps 45 --warmup 5 $F --no-scaling-leg
run "20 steps, no scaling leg" python bench.py --steps 20 --warmup 5 $F --no-scaling-leg
run "45 steps, with scaling leg" python bench.py --steps 45 --warmup 5 $F
run "45 steps, no profile events" python bench.py --steps 45 --warmup 5 $F --no-scaling-leg --no-profile
(cd build_variants/wt_0841960 && echo "== round 3 library and bench, 45 steps" >> $L && timeout -k 5 100 python bench.py --steps 45 --warmup 5 $F 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value':d['value'],'steps':d['steps'],'ms_per_step':d['ms_per_step']}))" >> $L)
cut -c1-700 $L
