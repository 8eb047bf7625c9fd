// ubench_gather.hip — what does MI355X deliver for SCATTERED reads?  (round 3)
//
// k_raycast fetches 1.45 GB per launch in ~405 us = 3.6 TB/s, in 128-byte lines picked by a hash table and by rays through a
// voxel block array (L2 hit rate 23 %, profiles/r02f_bench5mm_pmc_cache.json).  Against the 8 TB/s of the data sheet, or the
// 5.5-6.4 TB/s a streaming copy reaches, that reads as "0.38 of peak".  This tool measures the roof for THAT access pattern:
// every lane of every wave loads 4 bytes from a pseudo-random 128-byte line of a large buffer, K independent loads in flight
// per lane (memory-level parallelism as a parameter), all CUs busy.  Reported: lines per second and the equivalent GB/s at
// 128 B per line, for buffers of 64 MB (fits the 256 MB Infinity Cache), 1 GiB and 8 GiB, and — with K = 1 and each address
// depending on the previous value — the latency of a dependent scattered read under that load.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o ubench_gather ubench_gather.hip     Run: ./ubench_gather > log
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                            \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {  // a cheap integer hash (lowbias32)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// K independent scattered loads per lane and iteration; lineMask + 1 = number of 128-byte lines of the buffer (a power of two)
template <int K>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ buf, uint32_t lineMask, int iters, uint32_t *__restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0, seed = tid * 0x9e3779b9u + 1u;
  for (int it = 0; it < iters; ++it) {
    uint32_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      seed = mix(seed + (uint32_t)k);
      v[k] = buf[(size_t)(seed & lineMask) * 32u + (tid & 31u)];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k];
  }
  if (acc == 0x12345678u) sink[0] = acc;  // never true for the data below; keeps the loads alive
}

// one DEPENDENT scattered load per lane and iteration: the next line is a function of the value just read
__global__ __launch_bounds__(256) void k_chase(const uint32_t *__restrict__ buf, uint32_t lineMask, int iters, uint32_t *__restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = mix(tid + 7u);
  for (int it = 0; it < iters; ++it) x = mix(x + buf[(size_t)(x & lineMask) * 32u + (tid & 31u)]);
  if (x == 0x12345678u) sink[0] = x;
}

// the same chain with only the first `lanes` lanes of every wave loading: the latency of a scattered read as a function of the load
__global__ __launch_bounds__(64) void k_chase_lanes(const uint32_t *__restrict__ buf, uint32_t lineMask, int iters, int lanes, uint32_t salt, uint32_t *__restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if ((int)(threadIdx.x & 63u) >= lanes) return;
  uint32_t x = mix(tid + salt);  // a new chain per launch: a short chain repeated would sit in the L2
  for (int it = 0; it < iters; ++it) x = mix(x + buf[(size_t)(x & lineMask) * 32u + (tid & 31u)]);
  if (x == 0x12345678u) sink[0] = x;
}

template <class F>
static float time_ms(F launch, int reps) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipEventRecord(a, 0));
  for (int r = 0; r < reps; ++r) launch();
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const size_t maxBytes = 8ull << 30;
  uint32_t *buf = nullptr, *sink = nullptr;
  CHECK(hipMalloc((void **)&buf, maxBytes));
  CHECK(hipMalloc((void **)&sink, 64));
  CHECK(hipMemset(buf, 0x5a, maxBytes));
  CHECK(hipMemset(sink, 0, 64));
  const int reps = 3;
  printf("# scattered 4-byte loads, one per lane from a pseudo-random 128-byte line; waves/SIMD = resident waves per SIMD\n");
  printf("%-10s %-10s %-4s %12s %14s %12s\n", "buffer", "waves/SIMD", "K", "ms", "G lines/s", "GB/s @128B");
  for (size_t bytes : {64ull << 20, 1ull << 30, 8ull << 30}) {
    const uint32_t lineMask = (uint32_t)(bytes / 128 - 1);
    for (int wavesPerSimd : {2, 4, 8}) {
      const int grid = prop.multiProcessorCount * wavesPerSimd;  // 256-thread workgroups = 4 waves = one per SIMD
      const int iters = 512;
      auto report = [&](int K, float ms) {
        const double lines = (double)grid * 256.0 * iters * K;
        printf("%-10zu %-10d %-4d %12.4f %14.2f %12.1f\n", bytes >> 20, wavesPerSimd, K, ms, lines / ms / 1e6, lines * 128.0 / ms / 1e6);
      };
      report(1, time_ms([&] { hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, buf, lineMask, iters, sink); }, reps));
      report(2, time_ms([&] { hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, buf, lineMask, iters, sink); }, reps));
      report(4, time_ms([&] { hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, buf, lineMask, iters, sink); }, reps));
      report(8, time_ms([&] { hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, buf, lineMask, iters, sink); }, reps));
    }
  }
  printf("# dependent chain: one scattered load per lane and iteration, address from the previous value\n");
  printf("%-10s %-10s %12s %16s %14s\n", "buffer", "waves/SIMD", "ms", "us per step", "G lines/s");
  for (size_t bytes : {64ull << 20, 1ull << 30, 8ull << 30}) {
    const uint32_t lineMask = (uint32_t)(bytes / 128 - 1);
    for (int wavesPerSimd : {1, 4, 8}) {
      const int grid = prop.multiProcessorCount * wavesPerSimd;
      const int iters = 256;
      const float ms = time_ms([&] { hipLaunchKernelGGL(k_chase, dim3(grid), dim3(256), 0, 0, buf, lineMask, iters, sink); }, reps);
      printf("%-10zu %-10d %12.4f %16.3f %14.2f\n", bytes >> 20, wavesPerSimd, ms, 1e3 * ms / iters, (double)grid * 256.0 * iters / ms / 1e6);
    }
  }
  printf("# latency against load: 64-thread workgroups, `lanes` lanes of each wave chase; lines in flight = waves x lanes\n");
  printf("%-10s %-8s %-6s %12s %16s %14s\n", "buffer", "waves", "lanes", "in flight", "us per step", "G lines/s");
  for (size_t bytes : {1ull << 30, 8ull << 30}) {
    const uint32_t lineMask = (uint32_t)(bytes / 128 - 1);
    const int cu = prop.multiProcessorCount;
    const int cfg[][2] = {{1, 1}, {cu, 1}, {cu, 8}, {cu, 64}, {cu * 4, 16}, {cu * 4, 64}, {cu * 8, 64}, {cu * 16, 16}, {cu * 16, 64}, {cu * 32, 16}, {cu * 32, 64}};
    for (auto &c : cfg) {
      const int iters = 2048;
      uint32_t salt = 7u;
      const float ms = time_ms([&] { salt += 0x10001u; hipLaunchKernelGGL(k_chase_lanes, dim3(c[0]), dim3(64), 0, 0, buf, lineMask, iters, c[1], salt, sink); }, reps);
      printf("%-10zu %-8d %-6d %12d %16.3f %14.2f\n", bytes >> 20, c[0], c[1], c[0] * c[1], 1e3 * ms / iters, (double)c[0] * c[1] * iters / ms / 1e6);
    }
  }
  CHECK(hipFree(buf)); CHECK(hipFree(sink));
  return 0;
}
