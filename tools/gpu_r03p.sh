# r03p: rocprofv3 timeline of the instance-volume workload; bench.py --gpus 2 on a one-GPU box must fail cleanly.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DSR_BENCH_NO_POOL=1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03p
( timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-configs3 > $O/${T}_gpus2_on_1gpu.out 2> $O/${T}_gpus2_on_1gpu.err; echo "rc=$?" >> $O/${T}_gpus2_on_1gpu.out )
rm -rf /tmp/kt_inst; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_inst -o kt --output-format csv -- python bench.py --instance-volumes 8 --steps 40 --warmup 10 --no-profile > $O/${T}_instvol8_under_rocprof.log 2>&1
python tools/profile_summary.py stats /tmp/kt_inst 0 > $O/${T}_instvol8_kernel_stats.json
cp /tmp/kt_inst/*/*kernel_stats.csv $O/${T}_instvol8_kernel_stats.csv 2>/dev/null || cp /tmp/kt_inst/*kernel_stats.csv $O/${T}_instvol8_kernel_stats.csv 2>/dev/null
timeout 300 python -m pytest tests/test_edges.py tests/test_gpu_parity.py -m gpu -x -q -k "silhouette or instance or sequence" > $O/${T}_tests.log 2>&1; tail -n 1 $O/${T}_tests.log
cat $O/${T}_gpus2_on_1gpu.out; grep -v "amdgpu.ids" $O/${T}_gpus2_on_1gpu.err | tail -n 6
grep '^{' $O/${T}_instvol8_under_rocprof.log | head -c 250; echo
head -c 1200 $O/${T}_instvol8_kernel_stats.json
