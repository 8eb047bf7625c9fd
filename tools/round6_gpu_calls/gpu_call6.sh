# round 6, call 6: the preview branch (free-view list + raycast on a stream of their own, next to the tracking render):
# parity of everything that renders through a batch / a shared stream, then A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06f
timeout -k 5 600 python -m pytest tests/test_gpu_batch.py tests/test_multigpu_gloo.py tests/test_gpu_fullsize_golden.py tests/test_bench_contract.py tests/test_reference_pipeline.py tests/test_shim.py -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 5 $G/${T}_gpu_subset.log
for B in 1 0; do
  DSR_PREVIEW_BRANCH=$B timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/${T}_instance_frame_branch$B.json 2>> $G/${T}_if.err
  python -c "
import json
d=json.loads(open('$G/${T}_instance_frame_branch$B.json').read().strip().splitlines()[-1]); print('branch $B', d['free_running']['us_per_frame'], d['sync_per_frame'], {k: v['us_per_frame'] for k, v in d['gpu_kernels'].items()})"
  DSR_PREVIEW_BRANCH=$B timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_branch$B.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8_branch$B.json').read().strip().splitlines()[-1]); c=d['config']; print('branch $B', d['value'], d['unit'], d['ms_per_step'], c['chain_us_max_rank'], c['composite_us'])"
  DSR_PREVIEW_BRANCH=$B timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --volumes 8 > $G/${T}_bench_volumes8_branch$B.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_volumes8_branch$B.json').read().strip().splitlines()[-1]); c=d['config']; print('branch $B configs3', d['value'], d['unit'], d['ms_per_step'])"
done
export DSR_BENCH_NO_POOL=1
for PX in 2 4; do
DSR_COMPOSITE_PX=$PX timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/ktb -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $G/${T}_ktb.log 2>&1
python tools/profile_summary.py stats $G/ktb 20 > $G/${T}_batch_kernel_stats_px$PX.json
rm -rf $G/ktb
python - <<P
import json
d=json.load(open('$G/${T}_batch_kernel_stats_px$PX.json'))
print('px $PX', {k: v.get('avg_us_last_20', v['avg_us']) for k, v in d.items() if k.startswith('k_') or 'omposite' in k})
P
done
