# r03b: visible-block stream, small-volume paths, direct render outputs, copy probe.  bash tools/gpu_r03b.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03b
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/${T}_gpu_suite.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/${T}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --profile-all --no-cpu-baseline --no-through-shim > $O/${T}_bench_line_profile_all.json 2>> $O/${T}_bench.err
timeout 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2>> $O/${T}_bench.err
for q in 2 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8_q$q.json 2>> $O/${T}_bench.err
done
timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_volumes8.json 2>> $O/${T}_bench.err
tail -n 4 $O/${T}_gpu_suite.log
head -c 300 $O/${T}_bench_line.json; echo
grep -v "amdgpu.ids\|hostname of the client" $O/${T}_bench.err | tail -n 5
