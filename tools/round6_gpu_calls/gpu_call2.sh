# round 6, call 2: the range image's box (k_raycast.h RB_*) + the batch's per-call table as kernel arguments: whole GPU suite,
# then the instance frame and the 8-volume batch against the same library with DSR_RAY_BOX=0
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06b
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 15 $G/${T}_gpu_suite.log
for RB in 1 0; do
  DSR_RAY_BOX=$RB timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/${T}_instance_frame_raybox$RB.json 2>> $G/${T}_if.err
  python -c "
import json
d=json.loads(open('$G/${T}_instance_frame_raybox$RB.json').read().strip().splitlines()[-1]); print('raybox $RB', d['free_running']['us_per_frame'], d['sync_per_frame'], {k: v['us_per_frame'] for k, v in d['gpu_kernels'].items()})"
  DSR_RAY_BOX=$RB timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_raybox$RB.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8_raybox$RB.json').read().strip().splitlines()[-1]); print('raybox $RB', d['value'], d['unit'], d['ms_per_step'], {k: d[k] for k in d if 'us' in k})"
done
