# round 6, call 14: the cut-out writes only its own box and the previous one (dsr_engine::blankBox): parity of everything that
# splits views, then the batch and the instance frame against the numbers of the final set
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06n
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 6 $G/${T}_gpu_suite.log
timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8.json 2>> $G/${T}_bench.err
python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8.json').read().strip().splitlines()[-1]); c=d['config']; print('instvol8', d['value'], d['unit'], d['ms_per_step'], c['chain_us_max_rank'], c['composite_us'])"
timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --volumes 8 > $G/${T}_bench_volumes8.json 2>> $G/${T}_bench.err
python -c "
import json
d=json.loads(open('$G/${T}_bench_volumes8.json').read().strip().splitlines()[-1]); print('configs3', d['value'], d['unit'], d['ms_per_step'])"
timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/${T}_instance_frame.json 2>> $G/${T}_bench.err
python -c "
import json
d=json.loads(open('$G/${T}_instance_frame.json').read().strip().splitlines()[-1]); print('instance frame', d['free_running']['us_per_frame'], d['sync_per_frame'], {k: v['us_per_frame'] for k, v in d['gpu_kernels'].items()})"
export DSR_BENCH_NO_POOL=1
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/ktb -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $G/${T}_ktb.log 2>&1
python tools/profile_summary.py timeline $G/ktb k_batch_split 35 > $G/${T}_batch_step_timeline.json
rm -rf $G/ktb
python - <<P
import json
d=json.load(open('$G/${T}_batch_step_timeline.json'))
print('batch step', d.get('step_us'))
for k in d.get('kernels', []): print('  %-32s q%-3s %8.1f %8.1f %7.1f' % (k['name'], k['queue'], k['start_us'], k['end_us'], k['us']))
P
timeout -k 5 300 python tools/bench_through_shim.py --preset 5mm --steps 20 --warmup 5 --width 1242 --height 375 --instances 4 2>/dev/null | tail -n 1 | cut -c1-160
