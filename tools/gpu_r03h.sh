# r03h: k_integrate at 8 waves per SIMD (64 VGPRs since the row products are hoisted by hand).  bash tools/gpu_r03h.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03h
DSR_INTEGRATE_OCC8=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sequence or geometry or instance or weighting or odd" > $O/${T}_parity_occ8.log 2>&1; echo "rc=$?" >> $O/${T}_parity_occ8.log
timeout 600 python tools/bench_variants.py "OCC8=0" "OCC8=1" "OCC8=0" "OCC8=1" "OCC8=1 GRID=8192" "OCC8=1 GRID=32768" "OCC8=0 GRID=8192" > $O/${T}_variants.log 2> $O/${T}_variants.err
tail -n 2 $O/${T}_parity_occ8.log
cat $O/${T}_variants.log
