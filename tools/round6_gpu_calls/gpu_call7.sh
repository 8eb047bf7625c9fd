# round 6, call 7: the driver's command with the nested legs (how long does it run?), and one steady-state step of the 8-volume
# batch / of the instance frame as the GPU ran it (kernel start / end per queue)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06g
SECONDS=0; timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $G/${T}_bench_line.json 2> $G/${T}_bench.err; echo "bench rc=$? wall ${SECONDS}s"; tail -n 3 $G/${T}_bench.err
python - <<P
import json
d=json.loads(open('$G/${T}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: d['roofline'].get(k) for k in ('frac','frac_of_measured_copy','measured_copy_spread_GBps','raycast_frac','composite_frac','target_60pct_of_measured')})
print('shim', d['through_shim'] and (d['through_shim'].get('frames_per_s'), (d['through_shim'].get('configs2') or {}).get('frames_per_s')))
for k in ('instance_volumes8_1gpu','configs2','configs3_1gpu','configs4_short'):
    v=d.get(k) or {}
    print(k, v.get('value'), v.get('unit'), v.get('ms_per_step'), v.get('status'), {x: (v.get('config') or {}).get(x) for x in ('chain_us_max_rank','composite_us','structural_invariants','engine_status','invariants_check_s')})
print('composite', d['roofline'].get('composite'))
P
export DSR_BENCH_NO_POOL=1
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/ktb -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $G/${T}_ktb.log 2>&1
python tools/profile_summary.py timeline $G/ktb k_batch_split 35 > $G/${T}_batch_step_timeline.json
python tools/profile_summary.py stats $G/ktb 20 > $G/${T}_batch_kernel_stats.json
rm -rf $G/ktb
python - <<P
import json
d=json.load(open('$G/${T}_batch_step_timeline.json'))
print('batch step', d.get('step_us'))
for k in d.get('kernels', []): print('  %-32s q%-3s %8.1f %8.1f %7.1f' % (k['name'], k['queue'], k['start_us'], k['end_us'], k['us']))
P
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/kti -o kt --output-format csv -- python tools/bench_instance_frame.py --share-stream --frames 60 > $G/${T}_kti.log 2>&1
python tools/profile_summary.py timeline $G/kti k_view_ingest 150 > $G/${T}_instance_frame_timeline.json
rm -rf $G/kti
python - <<P
import json
d=json.load(open('$G/${T}_instance_frame_timeline.json'))
print('instance frame', d.get('step_us'), d.get('error'))
for k in d.get('kernels', []): print('  %-32s q%-3s %8.1f %8.1f %7.1f' % (k['name'], k['queue'], k['start_us'], k['end_us'], k['us']))
P
