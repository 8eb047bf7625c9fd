# r03d: range image under the integration (side stream), one-WG range image threshold, expected-depth shape A/B.  bash tools/gpu_r03d.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03d
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edges.py tests/test_gpu_composite.py tests/test_shim.py tests/test_reference_compiles.py tests/test_swapping.py tests/test_meshing.py tests/test_driver_mirror.py -m gpu -x -q > $O/${T}_gpu_subset.log 2>&1; echo "subset rc=$?" >> $O/${T}_gpu_subset.log
timeout 600 python -m pytest tests/test_gpu_fullsize_golden.py -m gpu -x -q -k "bench_5mm or gc_defaults or cfg5" > $O/${T}_gpu_fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/${T}_gpu_fullsize.log
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim"
$B > $O/${T}_bench_overlap1.json 2> $O/${T}_bench.err
DSR_OVERLAP_EXPECTED=0 $B > $O/${T}_bench_overlap0.json 2>> $O/${T}_bench.err
$B > $O/${T}_bench_overlap1_b.json 2>> $O/${T}_bench.err
DSR_OVERLAP_EXPECTED=0 DSR_GRID_EXPECTED=256 DSR_EXPECTED_THREADS=512 $B --profile-all > $O/${T}_bench_exp_256x512.json 2>> $O/${T}_bench.err
DSR_OVERLAP_EXPECTED=0 DSR_GRID_EXPECTED=256 DSR_EXPECTED_THREADS=256 $B --profile-all > $O/${T}_bench_exp_256x256.json 2>> $O/${T}_bench.err
DSR_OVERLAP_EXPECTED=0 DSR_GRID_EXPECTED=128 DSR_EXPECTED_THREADS=512 $B --profile-all > $O/${T}_bench_exp_128x512.json 2>> $O/${T}_bench.err
DSR_GRID_EXPECTED=256 DSR_EXPECTED_THREADS=256 $B > $O/${T}_bench_overlap1_256x256.json 2>> $O/${T}_bench.err
timeout 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --instance-volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_instvol8.json 2>> $O/${T}_bench.err
timeout 300 python bench.py --volumes 8 --steps 40 --warmup 10 > $O/${T}_bench_volumes8.json 2>> $O/${T}_bench.err
tail -n 3 $O/${T}_gpu_subset.log $O/${T}_gpu_fullsize.log
grep -v "amdgpu.ids\|hostname of the client" $O/${T}_bench.err | tail -n 5
