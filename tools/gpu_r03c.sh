# r03c: visible_write gathers hoisted, LDS-atomic filter, pinned mask ring, previews from HBM.  bash tools/gpu_r03c.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03c
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edges.py tests/test_gpu_composite.py tests/test_shim.py tests/test_reference_compiles.py tests/test_swapping.py -m gpu -x -q > $O/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $O/${T}_gpu_subset.log
timeout 600 python -m pytest tests/test_gpu_fullsize_golden.py -m gpu -x -q -k "bench_5mm or gc_defaults or cfg2" > $O/${T}_gpu_fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/${T}_gpu_fullsize.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_line.json 2> $O/${T}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --profile-all --no-cpu-baseline --no-through-shim > $O/${T}_bench_line_profile_all.json 2>> $O/${T}_bench.err
DSR_EXPECTED_FILTER=1 timeout 300 python bench.py --steps 20 --warmup 5 --profile-all --no-cpu-baseline --no-through-shim > $O/${T}_bench_line_profile_all_filter.json 2>> $O/${T}_bench.err
timeout 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2>> $O/${T}_bench.err
timeout 300 python tools/bench_through_shim.py --instances 4 > $O/${T}_through_shim_configs2.log 2>> $O/${T}_bench.err
timeout 300 python tools/bench_through_shim.py --instances 4 --preset 5cm >> $O/${T}_through_shim_configs2.log 2>> $O/${T}_bench.err
timeout 300 python bench.py --instances 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_inst4.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_volumes8.json 2>> $O/${T}_bench.err
tail -n 3 $O/${T}_gpu_subset.log $O/${T}_gpu_fullsize.log
cat $O/${T}_through_shim_configs2.log
grep -v "amdgpu.ids\|hostname of the client" $O/${T}_bench.err | tail -n 5
