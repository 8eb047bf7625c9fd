// k_edges.h — the per-pixel steps right before the hot path, moved to the GPU:
//   * disparity -> int16-mm depth      (DepthProvider::DepthFromDisparityMap, DepthProvider.h:94-137)
//   * instance view split / blanking   (ProcessSilhouette_CPU / RemoveSilhouette_CPU,
//                                       InstanceReconstructor.cpp:59-170)
// so that a frame never leaves HBM between depth ingest and fusion.
#pragma once
#include "dsr_device.h"

namespace dsr {

// ---------------------------------------------------------------------- K0: view

// ITMViewBuilder.h convertDepthAffineToFloat: int16 mm -> float m, <=0 or >32000 -> -1.
__global__ __launch_bounds__(256) void k_depth_to_float(const short *__restrict__ in, float *__restrict__ out, int n,
                                                        float a, float b) {
  // 4 pixels per thread: 8 B load, 16 B store
  int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    short4 d = *reinterpret_cast<const short4 *>(in + i4);
    float4 o;
    o.x = (d.x <= 0 || d.x > 32000) ? -1.0f : (float)d.x * a + b;
    o.y = (d.y <= 0 || d.y > 32000) ? -1.0f : (float)d.y * a + b;
    o.z = (d.z <= 0 || d.z > 32000) ? -1.0f : (float)d.z * a + b;
    o.w = (d.w <= 0 || d.w > 32000) ? -1.0f : (float)d.w * a + b;
    *reinterpret_cast<float4 *>(out + i4) = o;
  } else {
    for (int i = i4; i < n; ++i) {
      short d = in[i];
      out[i] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
    }
  }
}

// UpdateView from device-resident inputs in ONE launch: copy the RGBA frame (16 B per thread) and
// convert the depth (4 pixels per thread) — two hipMemcpyAsync D2D + a kernel cost ~60 us of
// launch latency per frame, this one ~5 us.  Both input pointers must be 16-byte aligned.
__global__ __launch_bounds__(256) void k_view_ingest(const uint4 *__restrict__ rgbIn, uint4 *__restrict__ rgbOut, int nRgbVec,
                                                     int nRgbPixels, const short *__restrict__ depthIn,
                                                     float *__restrict__ depthOut, int n, float a, float b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nRgbVec) rgbOut[i] = rgbIn[i];
  if (i < nRgbPixels - nRgbVec * 4)  // the 0..3 pixels after the last whole 16-byte vector
    reinterpret_cast<uint32_t *>(rgbOut)[nRgbVec * 4 + i] = reinterpret_cast<const uint32_t *>(rgbIn)[nRgbVec * 4 + i];
  const int i4 = i * 4;
  if (i4 + 3 < n) {
    short4 d = *reinterpret_cast<const short4 *>(depthIn + i4);
    float4 o;
    o.x = (d.x <= 0 || d.x > 32000) ? -1.0f : (float)d.x * a + b;
    o.y = (d.y <= 0 || d.y > 32000) ? -1.0f : (float)d.y * a + b;
    o.z = (d.z <= 0 || d.z > 32000) ? -1.0f : (float)d.z * a + b;
    o.w = (d.w <= 0 || d.w > 32000) ? -1.0f : (float)d.w * a + b;
    *reinterpret_cast<float4 *>(depthOut + i4) = o;
  } else {
    for (int k = i4; k < n; ++k) {
      short d = depthIn[k];
      depthOut[k] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
    }
  }
}

// ITMViewBuilder.h filterDepth (one bilateral pass); borders keep their old value.
__global__ __launch_bounds__(256) void k_filter_depth(const float *__restrict__ in, float *__restrict__ out, int W, int H) {
  const float MEAN_SIGMA_L = 1.2232f;
  int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x < 2 || x >= W - 2 || y < 2 || y >= H - 2) return;
  float z = in[x + y * W];
  if (z < 0.0f) { out[x + y * W] = -1.0f; return; }
  float final_depth = 0.0f, w_sum = 0.0f;
  float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / sqrtf(z) * 0.25f);
  for (int i = -2; i <= 2; i++)
    for (int j = -2; j <= 2; j++) {
      float tmpz = in[(x + j) + (y + i) * W];
      if (tmpz < 0.0f) continue;
      float dz = (tmpz - z); dz *= dz;
      float w = expf(-0.5f * ((abs(i) + abs(j)) * MEAN_SIGMA_L * MEAN_SIGMA_L + dz * sigma_z * sigma_z));
      w_sum += w;
      final_depth += w * tmpz;
    }
  final_depth /= w_sum;
  out[x + y * W] = final_depth;
}


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

// PrepareNextStep's two previews (ItmToCv + ItmDepthToCv, InfiniTamDriver.h:154-156) in ONE kernel that may store straight into
// page-locked HOST memory: four pixels per thread — 12 bytes of packed BGR as three dwords, four int16 millimetres as one 8-byte
// store — so that a wave writes 768 + 512 consecutive bytes over the host link instead of byte-sized stores, and no copy command
// (with its hand-over between the compute queue and the copy engine) follows.  Same per-pixel functions as the two kernels above.
__global__ __launch_bounds__(256) void k_previews(const uchar4 *__restrict__ rgba, const float *__restrict__ depth,
                                                  uint32_t *__restrict__ bgrOut, short *__restrict__ mmOut, int n) {
  const int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    if (bgrOut) {
      const uint4 c = *reinterpret_cast<const uint4 *>(rgba + i4);  // four RGBA pixels (r = low byte)
      auto B = [](uint32_t px) { return (px >> 16) & 0xffu; };
      auto G = [](uint32_t px) { return (px >> 8) & 0xffu; };
      auto R = [](uint32_t px) { return px & 0xffu; };
      uint32_t *o = bgrOut + (i4 >> 2) * 3;  // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
      o[0] = B(c.x) | (G(c.x) << 8) | (R(c.x) << 16) | (B(c.y) << 24);
      o[1] = G(c.y) | (R(c.y) << 8) | (B(c.z) << 16) | (G(c.z) << 24);
      o[2] = R(c.z) | (B(c.w) << 8) | (G(c.w) << 16) | (R(c.w) << 24);
    }
    if (mmOut) {
      const float4 d = *reinterpret_cast<const float4 *>(depth + i4);
      short4 m;
      m.x = (short)DeviceOps::f2i(d.x * (float)1000); m.y = (short)DeviceOps::f2i(d.y * (float)1000);
      m.z = (short)DeviceOps::f2i(d.z * (float)1000); m.w = (short)DeviceOps::f2i(d.w * (float)1000);
      *reinterpret_cast<short4 *>(mmOut + i4) = m;
    }
  } else {
    for (int i = i4; i < n; ++i) {
      if (bgrOut) rgba_to_bgr_px(i, rgba, reinterpret_cast<uint8_t *>(bgrOut));
      if (mmOut) depth_m_to_mm_px(i, depth, mmOut);
    }
  }
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

// ProcessSilhouette_CPU + RemoveSilhouette_CPU of one instance in ONE pass over the frame (dsr_view_split_silhouette): the
// cut-out is taken from the pixel as it is BEFORE this instance's blanking, which is the order of the two host loops
// (InstanceReconstructor.cpp:238-263).  The two masks differ in the reference (copy mask x1.0, delete mask x1.2: Utils/Mask.cpp).
// wr (round 6): the pixels of the cut-out that have to be WRITTEN — the instance's view is known to hold the blank constants
// outside the box of its previous cut-out (dsr_engine::blankBox), so only the new box and the old one are; the whole image when
// nothing is known about the buffer.  (An instance covers a few per cent of the frame: 3.7 MB of constants per instance and frame.)
__global__ __launch_bounds__(256) void k_split_silhouette(uchar4 *srcRgb, float *srcDepth, uchar4 *__restrict__ dstRgb,
                                                          float *__restrict__ dstDepth, int W, int H,
                                                          const uint8_t *__restrict__ mask, int x0, int y0, int bw, int bh,
                                                          const uint8_t *__restrict__ rmask, int rx0, int ry0, int rbw, int rbh,
                                                          int4 wr) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const int idx = x + y * W;
  const int col = x - x0, row = y - y0;
  if (x >= wr.x && x < wr.z && y >= wr.y && y < wr.w) {
    if (col >= 0 && col < bw && row >= 0 && row < bh && mask[row * bw + col] == 1) {
      dstRgb[idx] = srcRgb[idx];
      dstDepth[idx] = srcDepth[idx];
    } else {
      dstRgb[idx] = make_uchar4(255, 255, 255, 255);
      dstDepth[idx] = 0.0f;
    }
  }
  const int rcol = x - rx0, rrow = y - ry0;
  if (rcol >= 0 && rcol < rbw && rrow >= 0 && rrow < rbh && rmask[rrow * rbw + rcol] == 1) {
    srcRgb[idx] = make_uchar4(0, 0, 0, 0);
    srcDepth[idx] = 0.0f;
  }
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

// InfiniTamDriver::UpdateView from the host's own buffers (dsr_update_view_bgr): CvToItm's BGR -> RGBA loop
// (InfiniTamDriver.cpp:81-100, a = 255) and convertDepthAffineToFloat in one launch, 4 pixels per thread (12 B of BGR in,
// 16 B of RGBA out, 8 B of int16 depth in, 16 B of float depth out).  The staging buffers are 16-byte aligned.
__global__ __launch_bounds__(256) void k_view_ingest_bgr(const uint32_t *__restrict__ bgr, uint4 *__restrict__ rgbaOut, int n,
                                                         const short *__restrict__ depthIn, float *__restrict__ depthOut,
                                                         float a, float b) {
  const int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const uint32_t w0 = bgr[(i4 >> 2) * 3], w1 = bgr[(i4 >> 2) * 3 + 1], w2 = bgr[(i4 >> 2) * 3 + 2];  // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
    auto px = [](uint32_t bb, uint32_t gg, uint32_t rr) -> uint32_t { return (rr & 0xffu) | ((gg & 0xffu) << 8) | ((bb & 0xffu) << 16) | 0xff000000u; };
    uint4 o;
    o.x = px(w0, w0 >> 8, w0 >> 16);
    o.y = px(w0 >> 24, w1, w1 >> 8);
    o.z = px(w1 >> 16, w1 >> 24, w2);
    o.w = px(w2 >> 8, w2 >> 16, w2 >> 24);
    rgbaOut[i4 >> 2] = o;
    const short4 d = *reinterpret_cast<const short4 *>(depthIn + i4);
    float4 f;
    f.x = (d.x <= 0 || d.x > 32000) ? -1.0f : (float)d.x * a + b;
    f.y = (d.y <= 0 || d.y > 32000) ? -1.0f : (float)d.y * a + b;
    f.z = (d.z <= 0 || d.z > 32000) ? -1.0f : (float)d.z * a + b;
    f.w = (d.w <= 0 || d.w > 32000) ? -1.0f : (float)d.w * a + b;
    *reinterpret_cast<float4 *>(depthOut + i4) = f;
  } else {
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(bgr);
    for (int k = i4; k < n; ++k) {
      bgr_to_rgba_px(k, bytes, reinterpret_cast<uchar4 *>(rgbaOut));
      const short d = depthIn[k];
      depthOut[k] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
    }
  }
}

// SetView from host buffers (dsr_set_view_float): RGBA copy + the finite clamp above, out of the upload staging
__global__ __launch_bounds__(256) void k_set_view_ingest(const uchar4 *__restrict__ rgbaIn, uchar4 *__restrict__ rgbaOut, int nRgb,
                                                         const float *__restrict__ depthIn, float *__restrict__ depthOut, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nRgb) rgbaOut[i] = rgbaIn[i];
  if (i < n) { const float d = depthIn[i]; depthOut[i] = d > 1e30f ? 1e30f : d; }
}

// PrecomputedDepthProvider::ReadPrecomputed, the input_is_depth_ clamp for int16 maps
// (PrecomputedDepthProvider.cpp:55-74): depth > max_depth_mm_s -> 0
__global__ __launch_bounds__(256) void k_clip_depth_mm(short *__restrict__ depth, int n, short maxMm) {  // in place
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const short d = depth[i]; if (d > maxMm) depth[i] = (short)0; }
}

}  // namespace dsr
