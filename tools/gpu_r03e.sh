# r03e: speculative raycast variants (cast_ray<SPEC>): parity under each, then timing.  bash tools/gpu_r03e.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03e
for sp in 1 3; do
  DSR_RAYCAST_SPEC=$sp timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/${T}_parity_spec$sp.log 2>&1; echo "spec $sp rc=$?" >> $O/${T}_parity_spec$sp.log
done
DSR_RAYCAST_SPEC=1 timeout 600 python -m pytest tests/test_gpu_fullsize_golden.py -m gpu -x -q -k "bench_5mm or cfg2" > $O/${T}_fullsize_spec1.log 2>&1; echo "rc=$?" >> $O/${T}_fullsize_spec1.log
timeout 600 python tools/bench_variants.py "RSPEC=0" "RSPEC=1" "RSPEC=2" "RSPEC=3" "RSPEC=0" "RSPEC=1" > $O/${T}_variants.log 2> $O/${T}_variants.err
DSR_RAYCAST_SPEC=1 timeout 200 python tools/bench_instance_frame.py > $O/${T}_instance_frame_spec1.json 2>> $O/${T}_variants.err
tail -n 2 $O/${T}_parity_spec1.log $O/${T}_parity_spec3.log $O/${T}_fullsize_spec1.log
cat $O/${T}_variants.log
