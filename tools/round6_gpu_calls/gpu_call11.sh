# round 6, call 11: after the dsr_view.hip split: whole GPU suite, smoke, the driver's command (twice: is the probe's denominator
# stable?), through the C++ host
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06k
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 8 $G/${T}_gpu_suite.log
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > $G/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $G/${T}_smoke.log; tail -n 2 $G/${T}_smoke.log
for R in 1 2; do
SECONDS=0; timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $G/${T}_bench_line_$R.json 2> $G/${T}_bench_$R.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<P
import json
d=json.loads(open('$G/${T}_bench_line_$R.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], {k: r.get(k) for k in ('frac','frac_of_measured_copy','measured_copy_spread_GBps','raycast_frac','composite_frac')}, r['target_60pct_of_measured'])
print('shim', d['through_shim'] and (d['through_shim'].get('frames_per_s'), (d['through_shim'].get('configs2') or {}).get('frames_per_s')))
for k in ('instance_volumes8_1gpu','configs2','configs3_1gpu','configs4_short'):
    v=d.get(k) or {}
    print(k, v.get('value'), v.get('unit'), v.get('ms_per_step'), v.get('status'), {x: (v.get('config') or {}).get(x) for x in ('chain_us_max_rank','composite_us','structural_invariants')})
P
done
timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps 45 --warmup 5 --width 1242 --height 375 --instances 4 2>/dev/null | tail -n 1
timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps 20 --warmup 5 --width 1242 --height 375 --instances 4 2>/dev/null | tail -n 1
