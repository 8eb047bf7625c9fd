# round 6, call 3: the vectorised composite (parity + timing), the batch's kernel table with / without the ray box, the
# one-workgroup kernels launched twice (is it the cold instruction cache?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06c
timeout -k 5 400 python -m pytest tests/test_gpu_composite.py tests/test_reference_edges.py tests/test_gpu_batch.py tests/test_gpu_fullsize_golden.py tests/test_multigpu_gloo.py -m gpu -q -x --timeout 240 -p no:cacheprovider > $G/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $G/${T}_gpu_subset.log
tail -n 6 $G/${T}_gpu_subset.log
timeout -k 5 150 python tools/small_kernel_clocks.py --frames 64 --twice > $G/${T}_small_kernel_clocks_twice.json 2> $G/${T}_clk.err; cat $G/${T}_small_kernel_clocks_twice.json
export DSR_BENCH_NO_POOL=1
for RB in 1 0; do
  DSR_RAY_BOX=$RB timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/ktb$RB -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes 8 --no-profile > $G/${T}_ktb$RB.log 2>&1
  python tools/profile_summary.py stats $G/ktb$RB 20 > $G/${T}_batch_kernel_stats_raybox$RB.json
  rm -rf $G/ktb$RB
  python - <<P
import json
d=json.load(open('$G/${T}_batch_kernel_stats_raybox$RB.json'))
print('raybox $RB', {k: v.get('avg_us_last_20', v['avg_us']) for k, v in d.items() if k.startswith('k_')})
P
done
unset DSR_BENCH_NO_POOL
for RB in 1 0; do
  DSR_RAY_BOX=$RB timeout -k 5 160 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --instance-volumes 8 > $G/${T}_bench_instvol8_raybox$RB.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_instvol8_raybox$RB.json').read().strip().splitlines()[-1]); print('raybox $RB', d['value'], d['unit'], d['ms_per_step'], {k: d[k] for k in d if 'us' in k})"
done
