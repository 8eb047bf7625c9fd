// k_edges.h — the per-pixel steps right before the hot path, moved to the GPU:
//   * disparity -> int16-mm depth      (DepthProvider::DepthFromDisparityMap, DepthProvider.h:94-137)
//   * instance view split / blanking   (ProcessSilhouette_CPU / RemoveSilhouette_CPU,
//                                       InstanceReconstructor.cpp:59-170)
// so that a frame never leaves HBM between depth ingest and fusion.
#pragma once
#include "dsr_device.h"

namespace dsr {

// Each kernel below is "one thread per element" around a per-element function; those are __host__ __device__ templates over
// Ops (dsr_device.h) so that tests/test_reference_edges.py can run them on the CPU against the REFERENCE'S OWN functions
// (oracle/_ref) as well as on the GPU.

template <class Ops = DeviceOps>
__host__ __device__ __forceinline__ void depth_from_disparity_px(int i, const float *__restrict__ disp, short *__restrict__ out, float baseline,
                                                                 float focal, float scale, int minDepthMm, int maxDepthMm) {
  const float d = disp[i];
  // kMetersToMillimeters * scale * DepthFromDisparity(disp): ((1000*scale) * ((b*f)/d)), float
  int depth_mm = Ops::f2i(1000.0f * scale * ((baseline * focal) / d));
  if ((double)fabsf(d) < 1e-5) depth_mm = 0;
  if (depth_mm > maxDepthMm || depth_mm < minDepthMm) depth_mm = 0;
  out[i] = (short)depth_mm;
}
__global__ __launch_bounds__(256) void k_depth_from_disparity(const float *__restrict__ disp, short *__restrict__ out,
                                                              int n, float baseline, float focal, float scale,
                                                              int minDepthMm, int maxDepthMm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  depth_from_disparity_px(i, disp, out, baseline, focal, scale, minDepthMm, maxDepthMm);
}

// InfiniTamDriver.cpp CvToItm / ItmToCv / FloatDepthmapToShort: the host's per-pixel layout loops
__host__ __device__ __forceinline__ void bgr_to_rgba_px(int i, const uint8_t *__restrict__ bgr, uchar4 *__restrict__ rgba) {
  rgba[i] = make_uchar4(bgr[3 * i + 2], bgr[3 * i + 1], bgr[3 * i], 255);  // .r = col[2], .g = col[1], .b = col[0]
}
__host__ __device__ __forceinline__ void rgba_to_bgr_px(int i, const uchar4 *__restrict__ rgba, uint8_t *__restrict__ bgr) {
  const uchar4 c = rgba[i];
  bgr[3 * i] = c.z; bgr[3 * i + 1] = c.y; bgr[3 * i + 2] = c.x;
}
template <class Ops = DeviceOps>
__host__ __device__ __forceinline__ void depth_m_to_mm_px(int i, const float *__restrict__ m, short *__restrict__ mm) {
  mm[i] = (short)Ops::f2i(m[i] * (float)1000);  // pixels[..] * kMetersToMillimeters (int promoted to float)
}
__global__ __launch_bounds__(256) void k_bgr_to_rgba(const uint8_t *__restrict__ bgr, uchar4 *__restrict__ rgba, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bgr_to_rgba_px(i, bgr, rgba);
}
__global__ __launch_bounds__(256) void k_rgba_to_bgr(const uchar4 *__restrict__ rgba, uint8_t *__restrict__ bgr, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rgba_to_bgr_px(i, rgba, bgr);
}
__global__ __launch_bounds__(256) void k_depth_m_to_mm(const float *__restrict__ m, short *__restrict__ mm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  depth_m_to_mm_px(i, m, mm);
}

// dest (instance view) := default everywhere, source pixel where the bbox-local mask is 1
__host__ __device__ __forceinline__ void extract_silhouette_px(int x, int y, const uchar4 *__restrict__ srcRgb, const float *__restrict__ srcDepth,
                                                               uchar4 *__restrict__ dstRgb, float *__restrict__ dstDepth, int W,
                                                               const uint8_t *__restrict__ mask, int x0, int y0, int bw, int bh) {
  const int idx = x + y * W;
  const int col = x - x0, row = y - y0;
  const bool inBox = col >= 0 && col < bw && row >= 0 && row < bh;
  if (inBox && mask[row * bw + col] == 1) {
    dstRgb[idx] = srcRgb[idx];
    dstDepth[idx] = srcDepth[idx];
  } else {
    dstRgb[idx] = make_uchar4(255, 255, 255, 255);
    dstDepth[idx] = 0.0f;
  }
}
__global__ __launch_bounds__(256) void k_extract_silhouette(const uchar4 *__restrict__ srcRgb,
                                                            const float *__restrict__ srcDepth,
                                                            uchar4 *__restrict__ dstRgb, float *__restrict__ dstDepth,
                                                            int W, int H, const uint8_t *__restrict__ mask, int x0,
                                                            int y0, int bw, int bh) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  extract_silhouette_px(x, y, srcRgb, srcDepth, dstRgb, dstDepth, W, mask, x0, y0, bw, bh);
}

// (col, row): a cell of the mask's box
__host__ __device__ __forceinline__ void remove_silhouette_px(int col, int row, uchar4 *__restrict__ rgb, float *__restrict__ depth, int W, int H,
                                                              const uint8_t *__restrict__ mask, int x0, int y0, int bw) {
  const int x = col + x0, y = row + y0;
  if (x < 0 || x >= W || y < 0 || y >= H) return;
  if (mask[row * bw + col] == 1) {
    rgb[x + y * W] = make_uchar4(0, 0, 0, 0);
    depth[x + y * W] = 0.0f;
  }
}
__global__ __launch_bounds__(256) void k_remove_silhouette(uchar4 *__restrict__ rgb, float *__restrict__ depth, int W,
                                                           int H, const uint8_t *__restrict__ mask, int x0, int y0,
                                                           int bw, int bh) {
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), row = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (col >= bw || row >= bh) return;
  remove_silhouette_px(col, row, rgb, depth, W, H, mask, x0, y0, bw);
}

// Float views handed over by the host (dsr_set_view_float[_dev]) may hold +inf.  The reference's arithmetic treats
// such a pixel like any depth far beyond the volume: computeUpdatedVoxelDepthInfo fuses MIN(1, eta / mu) = 1 and the
// colour gate (eta > mu) rejects it.  The integrate kernel's division-by-reciprocal sequence would turn eta = inf
// into a NaN quotient instead, so the view is stored with values above 1e30 replaced by 1e30 — every result is the one
// the reference computes for +inf (eta >= 1e30 - z: quotient > 1, gate rejects; the allocation rejects the pixel
// through depth + mu > viewFrustum_max either way).  NaN and -inf pass through (compare false / rejected by d <= 0).
__global__ __launch_bounds__(256) void k_copy_depth_finite(const float *__restrict__ in, float *__restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float d = in[i]; out[i] = d > 1e30f ? 1e30f : d; }
}

// PrecomputedDepthProvider::ReadPrecomputed, the input_is_depth_ clamp for int16 maps
// (PrecomputedDepthProvider.cpp:55-74): depth > max_depth_mm_s -> 0
__global__ __launch_bounds__(256) void k_clip_depth_mm(short *__restrict__ depth, int n, short maxMm) {  // in place
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const short d = depth[i]; if (d > maxMm) depth[i] = (short)0; }
}

}  // namespace dsr
