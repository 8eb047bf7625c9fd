// dsr_profile.hip — measurement side of the library: the HIP-event profile of an engine's kernels (dsr_profile_*), the division
// self-tests behind DESIGN.md "bit-exactness rules" (dsr_selftest_division; an engine checks its own mu at creation), and the HBM
// copy probe the roofline fraction is quoted against (dsr_measure_copy_bandwidth).
#include "dsr_internal.h"

// div_short(a, b, RN(1/b)) against a / b for every numerator mantissa (a in [1, 2): division is
// scale invariant while nothing under- or overflows, and symmetric in the signs)
__global__ __launch_bounds__(256) void k_check_short_division(float b, unsigned long long *mismatches) {
  const float y = 1.0f / b;
  unsigned long long bad = 0;
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
    const float a = __uint_as_float(0x3f800000u | m);
    if (__float_as_uint(div_short(a, b, y)) != __float_as_uint(a / b)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

// true when the one-correction division is exact for this divisor (k_integrate.h div_short)
int dsr_internal::short_division_exact(hipStream_t stream, float b, bool *exact) {
  unsigned long long *d = nullptr, h = 1;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 8));
  (void)hipMemsetAsync(d, 0, 8, stream);
  hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, stream, b, d);
  hipError_t err = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, stream);
  if (err == hipSuccess) err = hipStreamSynchronize(stream);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(DSR_E_DEVICE, "short-division check failed to run");
  *exact = (h == 0);
  return DSR_OK;
}

// ---- HBM ceiling probe kernel (dsr_measure_copy_bandwidth)
typedef float copy_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy16(const copy_v4f *__restrict__ in, copy_v4f *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);  // streaming: no reuse to keep in L2
    else out[i] = in[i];
  }
}

extern "C" {

// ---- self-test

__global__ __launch_bounds__(256) void k_selftest_division(unsigned long long n, unsigned long long seed,
                                                           unsigned long long *mismatches) {
  const float y32767 = rcp_refined(32767.0f), y255 = rcp_refined(255.0f);
  unsigned long long bad = 0;
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  // exhaustive small domains
  if (tid < 65536) {
    const float a = (float)(short)(int)(tid - 32768);
    if (__float_as_uint(div_with_rcp(a, 32767.0f, y32767)) != __float_as_uint(a / 32767.0f)) bad++;
  }
  if (tid < 256) {
    const float a = (float)(int)tid;
    if (__float_as_uint(div_with_rcp(a, 255.0f, y255)) != __float_as_uint(a / 255.0f)) bad++;
  }
  for (unsigned long long i = tid; i < n; i += stride) {
    // splitmix64
    unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    // a: sign, exponent in [-40, 40]; b: sign, exponent in [-34, 40] (>= 1e-10: the smallest divisor a
    // call site lets through is the frustum test's),
    // random mantissas; every 16th pair uses small integers (weights) as divisor
    const unsigned ma = (unsigned)(z & 0x7fffffu), mb = (unsigned)((z >> 23) & 0x7fffffu);
    const int ea = (int)((z >> 46) % 81) - 40, eb = (int)((z >> 53) % 75) - 34;
    float a = __uint_as_float(((unsigned)(ea + 127) << 23) | ma);
    float b = __uint_as_float(((unsigned)(eb + 127) << 23) | mb);
    if (z >> 63) a = -a;
    if ((z >> 62) & 1) b = -b;
    if ((i & 15) == 0) b = (float)(1 + (int)((z >> 23) & 0x1ff));
    if ((i & 255) == 1) a = 0.0f;
    const float q = a / b;
    if (!(fabsf(q) == 0.0f || (fabsf(q) >= 1.17549435e-38f && fabsf(q) < 3.0e38f))) continue;  // not tame
    if (__float_as_uint(fdiv_tame(a, b)) != __float_as_uint(q)) bad++;
    const float yb = rcp_refined(b);
    if (__float_as_uint(div_with_rcp(a, b, yb)) != __float_as_uint(q)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

int dsr_selftest_division(int device, uint64_t n, uint64_t seed, uint64_t *mismatches) {
  if (!mismatches) return fail(DSR_E_ARG, "null");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  unsigned long long *d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 8));
  HIP_TRY(hipMemset(d, 0, 8));
  hipLaunchKernelGGL(k_selftest_division, dim3(4096), dim3(256), 0, 0, (unsigned long long)n, (unsigned long long)seed, d);
  // the divisors the one-correction form is used with: the constants, every integer weight, and
  // the truncation bands of the presets (an engine checks its own mu at creation)
  hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, 32767.0f, d);
  hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, 255.0f, d);
  for (int w = 1; w <= 256; ++w) hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, (float)w, d);
  for (float mu : {0.02f, 0.016f, 0.2f, 0.14f, 0.1f, 0.3f, 0.05f, 0.04f, 0.08f, 0.5f, 1.0f, 4.0f})
    hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, mu, d);
  unsigned long long h = 0;
  hipError_t err = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(DSR_E_DEVICE, "selftest failed to run");
  *mismatches = h;
  return DSR_OK;
}

// ---- HBM ceiling probe (roofline harness)

int dsr_measure_copy_bandwidth(int device, uint64_t bytes, int iters, double *gbps_out) {
  if (!gbps_out || bytes < 16 || iters <= 0) return fail(DSR_E_ARG, "bad bandwidth probe arguments");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  float4 *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a), bytes));
  if (hipMalloc(reinterpret_cast<void **>(&b), bytes) != hipSuccess) { (void)hipFree(a); return fail(DSR_E_NOMEM, "probe buffers"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMemset(a, 1, bytes);
  if (err == hipSuccess) err = hipMemset(b, 2, bytes);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  // The ceiling a copy kernel reaches depends on its launch shape (4.57 TB/s with one fixed shape where the
  // guide's float4 copy reaches 6.29): three grids x plain / non-temporal accesses, `iters` passes each, the BEST is reported.
  float ms = 0.0f;
  if (err == hipSuccess) {
    const size_t n = bytes / 16;
    float best = 0.0f;
    for (int variant = 0; variant < 6 && err == hipSuccess; ++variant) {
      const int grid = 256 * (variant % 3 == 0 ? 4 : variant % 3 == 1 ? 8 : 16);  // 4 / 8 / 16 workgroups per CU, grid-stride
      const bool nt = variant >= 3;
      auto launch = [&]() {
        if (nt) hipLaunchKernelGGL((k_copy16<true>), dim3(grid), dim3(256), 0, 0, (const copy_v4f *)a, (copy_v4f *)b, n);
        else hipLaunchKernelGGL((k_copy16<false>), dim3(grid), dim3(256), 0, 0, (const copy_v4f *)a, (copy_v4f *)b, n);
      };
      launch();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < iters; ++i) launch();
      (void)hipEventRecord(e1, 0);
      err = hipEventSynchronize(e1);
      float t = 0.0f;
      if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
      if (err == hipSuccess && t > 0.0f && (best == 0.0f || t < best)) best = t;
    }
    ms = best;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (err != hipSuccess || !(ms > 0.0f)) return fail(DSR_E_DEVICE, "bandwidth probe failed");
  *gbps_out = 2.0 * (double)(bytes / 16 * 16) * iters / ((double)ms * 1e-3) / 1e9;
  return DSR_OK;
}

// The same probe as a DENOMINATOR (five driver runs read 4.6-6.2 TB/s from a 1 GiB, 10-pass probe): `bytes` per direction
// (>= 4 GiB asked for by bench.py: far beyond the 256 MB of the last-level cache), the clocks warmed by ~50 ms of copies first,
// every launch shape (4 / 8 / 16 workgroups per CU x plain / non-temporal) timed launch by launch, `repeats` rounds; a round's
// figure is its best launch, out[] = {max, median, min} over the rounds in GB/s — the spread says how far to trust the max.
int dsr_measure_copy_bandwidth_spread(int device, uint64_t bytes, int repeats, double out[3]) {
  if (!out || bytes < (1u << 20) || repeats <= 0 || repeats > 64) return fail(DSR_E_ARG, "bad bandwidth probe arguments");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  float4 *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a), bytes));
  if (hipMalloc(reinterpret_cast<void **>(&b), bytes) != hipSuccess) { (void)hipFree(a); return fail(DSR_E_NOMEM, "probe buffers"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMemset(a, 1, bytes);
  if (err == hipSuccess) err = hipMemset(b, 2, bytes);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  const size_t n = bytes / 16;
  auto launch = [&](int variant) {
    const int grid = 256 * (variant % 3 == 0 ? 4 : variant % 3 == 1 ? 8 : 16);
    if (variant >= 3) hipLaunchKernelGGL((k_copy16<true>), dim3(grid), dim3(256), 0, 0, (const copy_v4f *)a, (copy_v4f *)b, n);
    else hipLaunchKernelGGL((k_copy16<false>), dim3(grid), dim3(256), 0, 0, (const copy_v4f *)a, (copy_v4f *)b, n);
  };
  std::vector<double> rounds;
  if (err == hipSuccess) {
    // warm-up: copies until ~50 ms of GPU time have passed (power state and clocks settle; the first launches of a process read low)
    float warm = 0.0f;
    for (int i = 0; i < 64 && warm < 50.0f && err == hipSuccess; ++i) {
      (void)hipEventRecord(e0, 0);
      launch(i % 6);
      (void)hipEventRecord(e1, 0);
      err = hipEventSynchronize(e1);
      float t = 0.0f;
      if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
      warm += t;
    }
    for (int r = 0; r < repeats && err == hipSuccess; ++r) {
      float best = 0.0f;
      for (int variant = 0; variant < 6 && err == hipSuccess; ++variant)
        for (int k = 0; k < 2 && err == hipSuccess; ++k) {
          (void)hipEventRecord(e0, 0);
          launch(variant);
          (void)hipEventRecord(e1, 0);
          err = hipEventSynchronize(e1);
          float t = 0.0f;
          if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
          if (err == hipSuccess && t > 0.0f && (best == 0.0f || t < best)) best = t;
        }
      if (best > 0.0f) rounds.push_back(2.0 * (double)(n * 16) / ((double)best * 1e-3) / 1e9);
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (err != hipSuccess || rounds.empty()) return fail(DSR_E_DEVICE, "bandwidth probe failed");
  std::sort(rounds.begin(), rounds.end());
  out[0] = rounds.back(); out[1] = rounds[rounds.size() / 2]; out[2] = rounds.front();
  return DSR_OK;
}

// ---- profiling

int dsr_profile_enable(dsr_engine *e, int enable) {
  CHECK_E(e);
  if (!enable) dsr_internal::engine_prof_resolve(e);
  e->profiling = enable == 2 ? 2 : (enable != 0);
  return DSR_OK;
}

int dsr_profile_reset(dsr_engine *e) {
  CHECK_E(e);
  dsr_internal::engine_prof_resolve(e);
  for (auto &r : e->profRecs) { r.ms = 0; r.launches = 0; }
  // work counters restart as well (decayed-block count is kept)
  unsigned long long zero = 0;
  HIP_TRY(hipMemcpyAsync(e->scene.work + WORK_V_INTEGRATED, &zero, 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->scene.work + WORK_V_EXPECTED, &zero, 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->scene.work + WORK_V_DECAY, &zero, 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(e->integrateStats, 0, e->integrateStatsCount * sizeof(uint2), e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

int dsr_profile_get(dsr_engine *e, dsr_kernel_time *out, int cap) {
  if (!e || !out || cap <= 0) return 0;
  if (dsr_internal::engine_set_device(e)) return 0;
  dsr_internal::engine_prof_resolve(e);
  unsigned long long work[WORK_COUNT];
  if (hipMemcpy(work, e->scene.work, sizeof work, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  const double P = (double)e->P, E = (double)e->E, B = (double)kBlockBytes;
  // k_integrate's own tallies: lanes that stored their 24 B of depth planes, voxels that got colour
  double storeLanes = 0.0, colourVoxels = 0.0;
  {
    std::vector<uint2> ws(e->integrateStatsCount);
    if (hipMemcpy(ws.data(), e->integrateStats, ws.size() * sizeof(uint2), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (const uint2 &w : ws) { storeLanes += (double)w.x; colourVoxels += (double)w.y; }
  }
  int n = 0;
  for (auto &r : e->profRecs) {
    if (n >= cap) break;
    if (r.launches == 0) continue;
    dsr_kernel_time &k = out[n++];
    memset(&k, 0, sizeof k);
    strncpy(k.name, r.name.c_str(), sizeof k.name - 1);
    k.total_ms = r.ms; k.launches = r.launches;
    const double L = (double)r.launches;
    // algorithmic bytes, SURVEY.md 8(d) / DESIGN.md "byte model"
    if (r.name == "integrate") {
      const double V = (double)work[WORK_V_INTEGRATED];
      k.bytes = V * (16.0 + 2.0 * B) + L * 8.0 * P;  // the reference's AoS formulation
      // what THIS layout has to move (DESIGN.md "byte model"): per visible block its list id (4 B), hash
      // entry (16 B) and the sdf + w_depth planes (1536 B) read; 24 B written back per lane that updated
      // a voxel; per colour voxel ONE 4-byte word (r, g, b, w_color) read and written; the depth and RGB frames (8 B per pixel)
      k.bytes_layout = V * (4.0 + 16.0 + 1536.0) + storeLanes * 24.0 + colourVoxels * 8.0 + L * 8.0 * P;
      k.units = V;
      k.store_lanes = storeLanes; k.colour_voxels = colourVoxels;
    }
    else if (r.name == "depth_to_float") k.bytes = L * 6.0 * P;
    else if (r.name == "expected_depth") k.bytes = (double)work[WORK_V_EXPECTED] * 16.0 + L * 8.0 * std::ceil(e->W / 8.0) * std::ceil(e->H / 8.0);
    else if (r.name == "icp_maps") k.bytes = L * P * (16.0 + 16.0 + 16.0 + 4.0);
    else if (r.name == "alloc_commit") k.bytes = L * E / 8.0;
    else if (r.name == "visible_count") k.bytes = L * E * 1.0;
    else if (r.name == "visible_write") k.bytes = L * E * 1.0;
    else if (r.name == "decay_blocks") k.bytes = (double)work[WORK_V_DECAY] * (16.0 + 2.0 * B);
    else k.bytes = 0.0;
  }
  return n;
}

}  // extern "C"
