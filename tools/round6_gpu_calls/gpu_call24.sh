# round 6, call 24: what do the HIP events around k_integrate / k_raycast (the roofline's live measurement) cost the timed step?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
for REP in 0 1 2; do
for FL in "" "--no-profile"; do
  timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg $FL > $G/r06w_bench$REP$FL.json 2>> $G/r06w_bench.err
  python - <<P
import json
d=json.loads([l for l in open('$G/r06w_bench$REP$FL.json') if l.startswith('{')][-1])
print('events' if '$FL'=='' else 'no events', d['value'], d['ms_per_step'], (d.get('kernels') or {}).get('k_integrate'), (d.get('kernels') or {}).get('k_raycast'))
P
done
done
