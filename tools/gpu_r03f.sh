# r03f: k_integrate without the held eta / ok arrays; -fno-slp-vectorize measured on the GPU.  bash tools/gpu_r03f.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03f
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/${T}_parity.log 2>&1; echo "rc=$?" >> $O/${T}_parity.log
timeout 600 python -m pytest tests/test_gpu_fullsize_golden.py -m gpu -x -q -k "bench_5mm or cfg2" > $O/${T}_fullsize.log 2>&1; echo "rc=$?" >> $O/${T}_fullsize.log
: > $O/${T}_variants.log
for lib in base new noslp new_noslp base new; do
  L=$GRAFT_REPO_ROOT/build_variants/libdsr_$lib.so
  [ $lib = new ] && L=$GRAFT_REPO_ROOT/dynslam_amd/csrc/libdsr_hip.so
  echo "lib=$lib" >> $O/${T}_variants.log
  DSR_HIP_LIB=$L timeout 300 python tools/bench_variants.py "" "" >> $O/${T}_variants.log 2>> $O/${T}_variants.err
done
timeout 900 python tools/bench_cfg5_sustained.py > $O/${T}_cfg5_sustained.json 2> $O/${T}_cfg5.err
tail -n 2 $O/${T}_parity.log $O/${T}_fullsize.log
cat $O/${T}_variants.log
tail -c 1500 $O/${T}_cfg5_sustained.json
