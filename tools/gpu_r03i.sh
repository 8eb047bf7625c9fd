# r03i: LLVM AMDGPU scheduler strategies for the whole library (same source, same results).  bash tools/gpu_r03i.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
T=r03i
: > $O/${T}_variants.log
for lib in default sched_max-ilp sched_max-memory-clause sched_iterative-minreg default; do
  L=$GRAFT_REPO_ROOT/build_variants/libdsr_$lib.so
  [ $lib = default ] && L=$GRAFT_REPO_ROOT/dynslam_amd/csrc/libdsr_hip.so
  echo "lib=$lib" >> $O/${T}_variants.log
  DSR_VARIANTS_PROFILE_ALL=1 DSR_HIP_LIB=$L timeout 300 python tools/bench_variants.py "" "" >> $O/${T}_variants.log 2>> $O/${T}_variants.err
done
cut -c1-400 $O/${T}_variants.log
