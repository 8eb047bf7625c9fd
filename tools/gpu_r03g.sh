# r03g: k_integrate with its hot uniforms in vector registers (VREG), 6 waves per SIMD.  bash tools/gpu_r03g.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03g
DSR_INTEGRATE_VREG=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/${T}_parity_vreg.log 2>&1; echo "rc=$?" >> $O/${T}_parity_vreg.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sequence or geometry or instance or weighting" > $O/${T}_parity.log 2>&1; echo "rc=$?" >> $O/${T}_parity.log
DSR_INTEGRATE_VREG=1 timeout 600 python -m pytest tests/test_gpu_fullsize_golden.py -m gpu -x -q -k "bench_5mm or cfg2 or cfg5" > $O/${T}_fullsize_vreg.log 2>&1; echo "rc=$?" >> $O/${T}_fullsize_vreg.log
timeout 600 python tools/bench_variants.py "VREG=0" "VREG=1" "VREG=0" "VREG=1" "VREG=1 GRID=8192" "VREG=1 GRID=4096" > $O/${T}_variants.log 2> $O/${T}_variants.err
DSR_INTEGRATE_VREG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim > $O/${T}_bench_vreg1.json 2> $O/${T}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim > $O/${T}_bench_vreg0.json 2>> $O/${T}_bench.err
tail -n 2 $O/${T}_parity_vreg.log $O/${T}_parity.log $O/${T}_fullsize_vreg.log
cat $O/${T}_variants.log
head -c 250 $O/${T}_bench_vreg1.json; echo; head -c 250 $O/${T}_bench_vreg0.json; echo
