cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
  DSR_HIP_LIB=$PWD/dynslam_amd/csrc/libdsr_hip_prev.so timeout 200 python tools/bench_variants.py "" 2>/dev/null | sed "s/^/base /" | tee -a gpurun_out/rc_lean_ab.log
  timeout 200 python tools/bench_variants.py "" 2>/dev/null | sed "s/^/lean /" | tee -a gpurun_out/rc_lean_ab.log
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sequence_bit_exact or free_view_render_types or hash_collisions or odd_image_sizes or instance_volume" 2>&1 | tail -2
