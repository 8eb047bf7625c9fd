# round 6, call 5: the sorted list of allocated entries (k_small.h list path): whole GPU suite with it, the small-volume tests
# again with DSR_SMALL_LISTS=0 (sweeps only), phase clocks, instance frame and 8-volume batch A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06e
timeout -k 5 700 python -m pytest tests -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $G/${T}_gpu_suite.log
tail -n 12 $G/${T}_gpu_suite.log
DSR_SMALL_LISTS=0 timeout -k 5 400 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_multigpu_gloo.py -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_subset_lists0.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset_lists0.log
tail -n 4 $G/${T}_gpu_subset_lists0.log
timeout -k 5 150 python tools/small_kernel_clocks.py --frames 64 > $G/${T}_small_kernel_clocks.json 2> $G/${T}_clk.err; cat $G/${T}_small_kernel_clocks.json
for L in 1 0; do
  DSR_SMALL_LISTS=$L timeout -k 5 120 python tools/bench_instance_frame.py --share-stream > $G/${T}_instance_frame_lists$L.json 2>> $G/${T}_if.err
  python -c "
import json
d=json.loads(open('$G/${T}_instance_frame_lists$L.json').read().strip().splitlines()[-1]); print('lists $L', d['free_running']['us_per_frame'], d['sync_per_frame'], {k: v['us_per_frame'] for k, v in d['gpu_kernels'].items()})"
  DSR_SMALL_LISTS=$L timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_lists$L.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8_lists$L.json').read().strip().splitlines()[-1]); c=d['config']; print('lists $L', d['value'], d['unit'], d['ms_per_step'], c['chain_us_max_rank'], c['composite_us'])"
done
