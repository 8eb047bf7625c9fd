# round 6, call 19: one steady-state frame of configs[1] (the static map) as the GPU ran it: kernel start / end per queue
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
export DSR_BENCH_NO_POOL=1
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/kt -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg --no-profile > $G/r06r_kt.log 2>&1
python tools/profile_summary.py timeline $G/kt k_view_ingest 5 > $G/r06r_map_frame_timeline.json
rm -rf $G/kt
python - <<P
import json
d=json.load(open('$G/r06r_map_frame_timeline.json'))
print('map frame', d.get('step_us'), d.get('error'))
prev=None
for k in d.get('kernels', []):
    print('  %-32s q%-3s %8.1f %8.1f %7.1f' % (k['name'], k['queue'], k['start_us'], k['end_us'], k['us']))
P
