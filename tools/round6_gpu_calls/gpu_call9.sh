# round 6, call 9: the paired render (deferred tracking render + preview raycast as one launch) and the target clear folded into
# the composite: whole GPU suite, then A/B (DSR_PAIR_RENDER=0 / 1, and the preview branch as it was)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06i
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 12 $G/${T}_gpu_suite.log
for V in "1 0" "0 0" "0 1"; do
  set -- $V
  export DSR_PAIR_RENDER=$1 DSR_PREVIEW_BRANCH=$2
  N=pair$1_branch$2
  timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/${T}_instance_frame_$N.json 2>> $G/${T}_if.err
  python -c "
import json
d=json.loads(open('$G/${T}_instance_frame_$N.json').read().strip().splitlines()[-1]); print('$N', d['free_running']['us_per_frame'], d['sync_per_frame'], d['launches_per_frame'], {k: v['us_per_frame'] for k, v in d['gpu_kernels'].items()})"
  timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_$N.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8_$N.json').read().strip().splitlines()[-1]); c=d['config']; print('$N', d['value'], d['unit'], d['ms_per_step'], c['chain_us_max_rank'], c['composite_us'])"
  timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --volumes 8 > $G/${T}_bench_volumes8_$N.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_volumes8_$N.json').read().strip().splitlines()[-1]); print('$N configs3', d['value'], d['unit'], d['ms_per_step'])"
done
unset DSR_PAIR_RENDER DSR_PREVIEW_BRANCH
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
