# round 6, call 21: a step's LATENCY (host waits for every step) with the exchange on its own stream / on the engines' stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for REP in 0 1 2; do
for SH in 0 1; do
  DSR_EXCHANGE_SHARE_STREAM=$SH timeout -k 5 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 --sync-every-step > $G/r06t_instvol8_latency_share$SH.$REP.json 2>> $G/r06t_bench.err
  python - <<P
import json
d=json.loads([l for l in open('$G/r06t_instvol8_latency_share$SH.$REP.json') if l.startswith('{')][-1])
print('share', $SH, 'latency us/step', round(1e3*d['ms_per_step'],1), 'chain', d['config']['chain_us_max_rank'], 'composite', d['config']['composite_us'])
P
done
done
