# round 6, call 8: single-pass commit / visible list (decoupled look-back): whole GPU suite, then configs[1] A/B against the
# three-launch forms (DSR_LOOKBACK=0) with --profile-all, configs[4]-like and 5 cm legs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06h
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 8 $G/${T}_gpu_suite.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
for L in 1 0 1 0; do
  DSR_LOOKBACK=$L timeout -k 5 120 $B > $G/${T}_bench_lookback$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_lookback$L.json').read().strip().splitlines()[-1]); print('lookback $L', d['value'], d['ms_per_step'], d['config']['status'])"
done
for L in 1 0; do
  DSR_LOOKBACK=$L timeout -k 5 120 $B --profile-all > $G/${T}_bench_profile_all_lookback$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_profile_all_lookback$L.json').read().strip().splitlines()[-1]); print('lookback $L', d['value'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  DSR_LOOKBACK=$L timeout -k 5 120 $B --preset 5cm --steps 45 > $G/${T}_bench_5cm_lookback$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_5cm_lookback$L.json').read().strip().splitlines()[-1]); print('5cm lookback $L', d['value'], d['ms_per_step'])"
  DSR_LOOKBACK=$L timeout -k 5 160 $B --preset 4mm --decay --swap --steps 60 > $G/${T}_bench_4mm_gc_swap_lookback$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_4mm_gc_swap_lookback$L.json').read().strip().splitlines()[-1]); print('4mm gc swap lookback $L', d['value'], d['ms_per_step'], d['config']['status'])"
done
