// k_composite.h — on-GPU instance compositing.
//
// Replaces the host loops CompositeInstances / CompositeColor / CompositeDepth
// (InstanceReconstructor.cpp:851-990): software z-buffer over the per-instance raycast
// renders, in the host's iteration order.  One thread per pixel walks the layers
// sequentially, so the strict "t > s" rule is applied exactly as the serial code does.
// Colour arithmetic follows the reference's C++ promotions: uchar * double for the colour
// term, int * float (then widened) for the tint term.
#pragma once
#include <cstring>

#include "dsr_device.h"

namespace dsr {

constexpr int kMaxCompositeLayers = 64;

struct CompositeP {
  int nLayers, nPixels, dimBackground;
  float tintStrength;
  int clearTarget;  // the target counts as empty (colour 0, depth 0) and is not read: dsr_exchange_clear_target folded into the composite
  uchar4 tint[kMaxCompositeLayers];  // kMatplotlib2Palette[track_id % 10], resolved on the host
};
struct CompositeLayers {  // one (colour, depth) pointer pair per layer: the layers stay where the all-gather left them
  const uchar4 *rgba[kMaxCompositeLayers];
  const float *depth[kMaxCompositeLayers];
};

// kMatplotlib2Palette (InstanceReconstructor.cpp:44-55) and the parameter block of a composite
static const unsigned char kMatplotlib2Palette[10][3] = {
    {0x1f, 0x77, 0xb4}, {0xff, 0x7f, 0x0e}, {0x2c, 0xa0, 0x2c}, {0xd6, 0x27, 0x28}, {0x94, 0x67, 0xbd},
    {0x8c, 0x56, 0x4b}, {0xe3, 0x77, 0xc2}, {0x71, 0x71, 0x71}, {0xbc, 0xbd, 0x22}, {0x17, 0xbe, 0xcf}};
inline CompositeP composite_params(const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength, int dim_background) {
  CompositeP c;
  memset(&c, 0, sizeof c);
  c.nLayers = n_layers; c.nPixels = n_pixels; c.dimBackground = dim_background; c.tintStrength = tint_strength;
  for (int l = 0; l < n_layers; ++l) {
    const unsigned char *t = kMatplotlib2Palette[((track_ids[l] % 10) + 10) % 10];
    c.tint[l] = make_uchar4(t[0], t[1], t[2], 255);
  }
  return c;
}

// ---- the per-pixel rules, shared by the one-pixel function (which tests/test_reference_edges.py also runs on the CPU against
// the reference's own CompositeColor / CompositeDepth) and the four-pixels-per-lane kernel
__host__ __device__ __forceinline__ uchar4 composite_dim(uchar4 col) {  // the background pre-dimmed by 10 % (:945-954)
  const double f = 1.0 - (double)0.10f;
  col.x = (unsigned char)((double)col.x * f);
  col.y = (unsigned char)((double)col.y * f);
  col.z = (unsigned char)((double)col.z * f);
  return col;
}
// a layer's depth s against the running depth t: strictly in front, 0 = nothing rendered (:861-867, :896-897)
__host__ __device__ __forceinline__ void composite_depth_step(float &t, int &winner, float s, int layer) {
  const bool onTop = (s != 0.0f) && (t == 0.0f || t > s);
  if (onTop) { t = s; winner = layer; }
}
// the winner's colour, tinted (:898-906): uchar * double for the colour term, int * float (then widened) for the tint term
__host__ __device__ __forceinline__ uchar4 composite_tinted(uchar4 col, uchar4 sc, uchar4 tint, float tintStrength) {
  const double colStrength = 1.0 + (double)0.50f - (double)tintStrength;
  const double r = fmin(255.0, (double)sc.x * colStrength + (double)((float)tint.x * tintStrength));
  const double g = fmin(255.0, (double)sc.y * colStrength + (double)((float)tint.y * tintStrength));
  const double b = fmin(255.0, (double)sc.z * colStrength + (double)((float)tint.z * tintStrength));
  col.x = (unsigned char)r; col.y = (unsigned char)g; col.z = (unsigned char)b;
  return col;
}

// one pixel of the composite.  The serial loop overwrites the colour every time a layer wins the pixel, so the result is the
// colour of the LAST winner: the depth walk only has to remember which layer that was, and ONE colour read follows it.
template <bool PTRS>
__host__ __device__ __forceinline__ void composite_px(int i, const CompositeP &c, uchar4 *__restrict__ tRgba, float *__restrict__ tDepth,
                                                      const uchar4 *__restrict__ lRgba, const float *__restrict__ lDepth,
                                                      const CompositeLayers &lp) {
  float t = c.clearTarget ? 0.0f : tDepth[i];
  uchar4 col = make_uchar4(0, 0, 0, 0);
  if (tRgba) {
    if (!c.clearTarget) col = tRgba[i];
    if (c.dimBackground) col = composite_dim(col);
  }
  int winner = -1;
  for (int l = 0; l < c.nLayers; ++l) composite_depth_step(t, winner, PTRS ? lp.depth[l][i] : lDepth[(size_t)l * c.nPixels + i], l);
  if (tRgba && winner >= 0)
    col = composite_tinted(col, PTRS ? lp.rgba[winner][i] : lRgba[(size_t)winner * c.nPixels + i], c.tint[winner], c.tintStrength);
  tDepth[i] = t;
  if (tRgba) tRgba[i] = col;
}

// PX (2 or 4) pixels per lane (round 6).  The composite is a stream: per pixel 4 B of depth per layer + the target — 23 MB for
// eight layers at 1242x375 — and ran at 0.4-0.6 TB/s as one pixel per thread with 4-byte loads (37-56 us).  Here a
// lane reads 4 * PX bytes of each layer's depth plane, eight layers in flight, keeps the running depth and the last winner of its
// pixels in registers, reads colour only where some pixel of the wave has a winner, and stores only what changed.
// (A layer's planes are 8-byte aligned only — P * 4 bytes is not a multiple of 16 at 1242x375 —, hence the packed vector type:
//  gfx950 performs unaligned 16-byte global loads.  The per-layer colour pointers and tints are looked up PER LANE (the winner
//  differs from pixel to pixel): they are staged in LDS once per workgroup instead of being gathered from the kernel-argument
//  segment by every wave.)
template <int PX>
struct __attribute__((packed, aligned(4))) CompVec { float v[PX]; };
constexpr int kCompositeLayersInFlight = 8;

template <bool PTRS, int PX>
__global__ __launch_bounds__(256) void k_composite(CompositeP c, uchar4 *__restrict__ tRgba, float *__restrict__ tDepth,
                                                   const uchar4 *__restrict__ lRgba, const float *__restrict__ lDepth,
                                                   CompositeLayers lp) {
  __shared__ const uchar4 *sRgba[kMaxCompositeLayers];
  __shared__ uchar4 sTint[kMaxCompositeLayers];
  if (tRgba) {  // (uniform)
    if ((int)threadIdx.x < c.nLayers) {
      sRgba[threadIdx.x] = PTRS ? lp.rgba[threadIdx.x] : lRgba + (size_t)threadIdx.x * c.nPixels;
      sTint[threadIdx.x] = c.tint[threadIdx.x];
    }
    __syncthreads();
  }
  const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
  if (i0 >= c.nPixels) return;
  if (i0 + PX > c.nPixels) {  // the last 1..PX-1 pixels
    for (int i = i0; i < c.nPixels; ++i) composite_px<PTRS>(i, c, tRgba, tDepth, lRgba, lDepth, lp);
    return;
  }
  typedef CompVec<PX> Vec;
  auto depth_of = [&](int l) { return PTRS ? lp.depth[l] + i0 : lDepth + (size_t)l * c.nPixels + i0; };
  const bool clr = c.clearTarget != 0;  // (uniform) nothing of the target is read, every pixel of it is written
  float t[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) t[k] = 0.0f;
  if (!clr) {
    const Vec t4 = *reinterpret_cast<const Vec *>(tDepth + i0);
#pragma unroll
    for (int k = 0; k < PX; ++k) t[k] = t4.v[k];
  }
  uint32_t colRaw[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) colRaw[k] = 0u;
  if (tRgba && !clr) {
    const CompVec<PX> raw = *reinterpret_cast<const CompVec<PX> *>(tRgba + i0);  // (bit pattern only)
#pragma unroll
    for (int k = 0; k < PX; ++k) colRaw[k] = __float_as_uint(raw.v[k]);
  }
  int winner[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) winner[k] = -1;
  for (int l0 = 0; l0 < c.nLayers; l0 += kCompositeLayersInFlight) {
    Vec s[kCompositeLayersInFlight];
#pragma unroll
    for (int j = 0; j < kCompositeLayersInFlight; ++j) {
      const int l = l0 + j < c.nLayers ? l0 + j : c.nLayers - 1;
      s[j] = *reinterpret_cast<const Vec *>(depth_of(l));
    }
    // (no branch in here: a `break` on the layer count makes every load a basic block of its own, waited for before the next one
    //  is issued — eight serial round trips; a layer beyond the count contributes depth 0, which never wins)
#pragma unroll
    for (int j = 0; j < kCompositeLayersInFlight; ++j) {
      const bool valid = l0 + j < c.nLayers;
#pragma unroll
      for (int k = 0; k < PX; ++k) composite_depth_step(t[k], winner[k], valid ? s[j].v[k] : 0.0f, l0 + j);
    }
  }
  int andW = -1;
#pragma unroll
  for (int k = 0; k < PX; ++k) andW &= winner[k];
  const bool won = andW >= 0;  // some pixel of this lane changed
  if (won || clr) {
    Vec o;
#pragma unroll
    for (int k = 0; k < PX; ++k) o.v[k] = t[k];
    *reinterpret_cast<Vec *>(tDepth + i0) = o;
  }
  if (!tRgba) return;
  uchar4 col[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const uint32_t w = colRaw[k];
    col[k] = make_uchar4((unsigned char)(w & 0xffu), (unsigned char)((w >> 8) & 0xffu), (unsigned char)((w >> 16) & 0xffu),
                         (unsigned char)(w >> 24));
    if (c.dimBackground) col[k] = composite_dim(col[k]);
  }
  if (__any(won)) {  // (wave-uniform)
    uchar4 sc[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k)  // the colour reads go out together; a pixel without a winner re-reads layer 0's (dropped)
      sc[k] = sRgba[winner[k] >= 0 ? winner[k] : 0][i0 + k];
#pragma unroll
    for (int k = 0; k < PX; ++k)
      if (winner[k] >= 0) col[k] = composite_tinted(col[k], sc[k], sTint[winner[k]], c.tintStrength);
  }
  if (!c.dimBackground && !won && !clr) return;  // nothing of this lane's colours changed
  CompVec<PX> out;
#pragma unroll
  for (int k = 0; k < PX; ++k)
    out.v[k] = __uint_as_float((uint32_t)col[k].x | ((uint32_t)col[k].y << 8) | ((uint32_t)col[k].z << 16) | ((uint32_t)col[k].w << 24));
  *reinterpret_cast<CompVec<PX> *>(tRgba + i0) = out;
}

}  // namespace dsr
