# round 6, call 13: VERDICT r5 item 7a — the uniforms of k_integrate's per-voxel projection in vector registers (opaque copies):
# 4 / 6 (64 VGPRs, still 8 waves per SIMD) / 7 / 9 (65-66 VGPRs: 7 waves) against the library as it is; parity of each on the
# full-size fixture first
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT/gpurun_out
T=r06m
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-through-shim --no-scaling-leg"
for V in 4 6 7 9; do
  DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r06_vuni$V/libdsr_hip.so timeout -k 5 200 python -m pytest tests/test_gpu_fullsize_golden.py -m gpu -q -x --timeout 240 -p no:cacheprovider -k "bench_5mm" 2>&1 | tail -n 1
done
for R in 1 2 3; do
for V in 0 4 6 7 9; do
  if [ $V = 0 ]; then unset DSR_HIP_LIB; else export DSR_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/r06_vuni$V/libdsr_hip.so; fi
  timeout -k 5 120 $B > $G/${T}_bench_vuni${V}_$R.json 2>> $G/${T}_bench.err
  python -c "
import json
d=json.loads(open('$G/${T}_bench_vuni${V}_$R.json').read().strip().splitlines()[-1]); print('vuni $V run $R', d['value'], d['ms_per_step'], d['kernels']['integrate']['avg_us'], d['kernels']['raycast']['avg_us'])"
done
done
