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

// one pixel of the composite (a __host__ __device__ function: tests/test_reference_edges.py also runs it on the CPU against the
// reference's own CompositeColor / CompositeDepth)
template <bool PTRS>
__host__ __device__ __forceinline__ void composite_px(int i, const CompositeP &c, uchar4 *__restrict__ tRgba, float *__restrict__ tDepth,
                                                      const uchar4 *__restrict__ lRgba, const float *__restrict__ lDepth,
                                                      const CompositeLayers &lp) {
  float t = tDepth[i];
  uchar4 col = make_uchar4(0, 0, 0, 0);
  if (tRgba) {
    col = tRgba[i];
    if (c.dimBackground) {
      const double f = 1.0 - (double)0.10f;
      col.x = (unsigned char)((double)col.x * f);
      col.y = (unsigned char)((double)col.y * f);
      col.z = (unsigned char)((double)col.z * f);
    }
  }
  const double colStrength = 1.0 + (double)0.50f - (double)c.tintStrength;
  // The serial loop overwrites the colour every time a layer wins the pixel, so the result is the colour of the LAST winner:
  // the depth walk (four layers' depths requested together: a load per loop iteration is waited for before the next is issued)
  // only has to remember which layer that was, and ONE colour read follows it.
  int winner = -1;
  for (int l0 = 0; l0 < c.nLayers; l0 += 4) {
    float s4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int l = l0 + k < c.nLayers ? l0 + k : c.nLayers - 1;
      s4[k] = PTRS ? lp.depth[l][i] : lDepth[(size_t)l * c.nPixels + i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (l0 + k >= c.nLayers) break;
      const float s = s4[k];
      const bool onTop = (s != 0.0f) && (t == 0.0f || t > s);
      if (onTop) { t = s; winner = l0 + k; }
    }
  }
  if (tRgba && winner >= 0) {
    const uchar4 sc = PTRS ? lp.rgba[winner][i] : lRgba[(size_t)winner * c.nPixels + i];
    const uchar4 tint = c.tint[winner];
    const double r = fmin(255.0, (double)sc.x * colStrength + (double)((float)tint.x * c.tintStrength));
    const double g = fmin(255.0, (double)sc.y * colStrength + (double)((float)tint.y * c.tintStrength));
    const double b = fmin(255.0, (double)sc.z * colStrength + (double)((float)tint.z * c.tintStrength));
    col.x = (unsigned char)r; col.y = (unsigned char)g; col.z = (unsigned char)b;
  }
  tDepth[i] = t;
  if (tRgba) tRgba[i] = col;
}

template <bool PTRS>
__global__ __launch_bounds__(256) void k_composite(CompositeP c, uchar4 *__restrict__ tRgba, float *__restrict__ tDepth,
                                                   const uchar4 *__restrict__ lRgba, const float *__restrict__ lDepth,
                                                   CompositeLayers lp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.nPixels) return;
  composite_px<PTRS>(i, c, tRgba, tDepth, lRgba, lDepth, lp);
}

}  // namespace dsr
