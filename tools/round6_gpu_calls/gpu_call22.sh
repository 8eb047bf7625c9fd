# round 6, call 22: the batch's kernels against the number of volumes in it (2..8): what grows with the volumes, what is the slowest one's chain
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
export DSR_BENCH_NO_POOL=1
for V in 2 3 4 5 6 7 8; do
  timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $G/ktv$V -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --instance-volumes $V --no-profile > $G/r06u_ktv$V.log 2>&1
  python tools/profile_summary.py stats $G/ktv$V 20 > $G/r06u_batch_kernel_stats_v$V.json
  rm -rf $G/ktv$V
  python - <<P
import json
d=json.load(open('$G/r06u_batch_kernel_stats_v$V.json'))
print($V, {k.replace('k_batch_',''): v.get('avg_us_last_20', v['avg_us']) for k, v in d.items() if k.startswith('k_batch') or k=='k_composite'})
P
done
