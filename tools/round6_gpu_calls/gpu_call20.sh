# round 6, call 20: the exchange on the engines' stream when there is nothing to gather (dsr_exchange_share_stream): tests, then A/B of
# the 8-volume leg (throughput and chain) and of the default line's nested legs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 5 900 python -m pytest tests/test_gpu_composite.py tests/test_multigpu_gloo.py tests/test_gpu_batch.py tests/test_bench_contract.py -m gpu -x -q > $G/r06s_subset.log 2>&1; echo "subset rc=$?"; tail -n 3 $G/r06s_subset.log
for REP in 0 1; do
for SH in 0 1; do
  DSR_EXCHANGE_SHARE_STREAM=$SH timeout -k 5 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/r06s_instvol8_share$SH.$REP.json 2>> $G/r06s_bench.err
  python - <<P
import json
d=json.loads([l for l in open('$G/r06s_instvol8_share$SH.$REP.json') if l.startswith('{')][-1])
print('share', $SH, d['value'], d['ms_per_step'], {k: d.get(k) for k in ('chain_us_max_rank','composite_us')}, d['config'].get('preview_hit_fraction'))
P
done
done
for SH in 0 1; do
  DSR_EXCHANGE_SHARE_STREAM=$SH timeout -k 5 300 python bench.py --no-cpu-baseline --no-through-shim > $G/r06s_bench_share$SH.json 2>> $G/r06s_bench.err
  python - <<P
import json
d=json.loads([l for l in open('$G/r06s_bench_share$SH.json') if l.startswith('{')][-1])
print('share', $SH, d['value'])
for k in ('configs2','configs3_1gpu','configs4_short','instance_volumes8_1gpu'):
    v=d.get(k) or {}
    print('   ', k, v.get('value'), v.get('status'), v.get('chain_us_max_rank'), v.get('composite_us'))
P
done
tail -n 5 $G/r06s_bench.err
