// Host stand-ins for ONE lane of the kernels: their per-pixel / per-block device functions — alloc_ray, check_block_visibility
// (k_alloc.h), project_single_block, cast_ray, icp_pixel, render_pixel (k_raycast.h), the per-element functions of k_edges.h and
// k_composite.h (checked against the REFERENCE'S OWN code by tests/test_reference_edges.py), templates over an Ops policy —
// compiled for the CPU with a one-ray Ops, so that tests/test_device_functions_host.py can check the allocation ray walk, the
// frustum test, the range image, the march (table walk, look-ahead slot, trilinear reads with their block rounds) and the
// shading (image-space normals, SDF-gradient normals, interpolated colours, the depth-weight map) against the oracle
// WITHOUT a GPU.
// Test infrastructure: built by the test with hipcc (host code only is run), never part of libdsr_hip.so.
#include <cmath>
#include <cstring>
#include <set>
#include <tuple>

#include "../../dynslam_amd/csrc/k_alloc.h"
#include "../../dynslam_amd/csrc/k_composite.h"
#include "../../dynslam_amd/csrc/k_edges.h"
#include "../../dynslam_amd/csrc/k_raycast.h"

namespace {
struct HostOps {
  static int f2i(float f) {  // v_cvt_i32_f32: toward zero, saturating, NaN -> 0
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
  }
  static bool any(bool b) { return b; }
  static float sqrt(float f) { return sqrtf(f); }
  static float floor(float f) { return floorf(f); }
  static float ceil(float f) { return ceilf(f); }
  static float rcp(float) { return 0.0f; }                       // the device shares a refined reciprocal between two divisions ...
  static float div(float a, float b, float) { return a / b; }    // ... whose results are the correctly rounded quotients (dsr_selftest_division)
};
}  // namespace

// table: noTotalEntries entries; vba: noBlocks * 4096 bytes in the library's block layout (only the sdf plane is read)
extern "C" int rr_cast_all(const float *invM, const float *proj, float voxelSize, float mu, int W, int H, int noBuckets, int noTotalEntries,
                           const dsr_hash_entry *table, const unsigned char *vba, const float *minmax, float *out) {
  using namespace dsr;
  FrameP p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.invM.m, invM, sizeof p.invM.m);
  p.proj = make_float4(proj[0], proj[1], proj[2], proj[3]);
  p.voxelSize = voxelSize; p.mu = mu; p.W = W; p.H = H;
  p.noBuckets = noBuckets; p.noTotalEntries = noTotalEntries; p.hashMask = (uint32_t)noBuckets - 1u;
  SceneP s;
  std::memset(&s, 0, sizeof s);
  s.table = const_cast<dsr_hash_entry *>(table);
  s.vba = const_cast<uint8_t *>(vba);
  const int mw = (W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const float *mmp = minmax + 2 * ((x >> 3) + (y >> 3) * mw);
      const float4 r = cast_ray<HostOps>(p, s, x, y, make_float2(mmp[0], mmp[1]));
      float *o = out + 4 * ((size_t)x + (size_t)y * W);
      o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
  return 0;
}

// k_icp_maps: points / normals / grey image of the tracking view from its raycast result (pointsRay: W*H float4, voxel units)
extern "C" int rr_icp_all(const float *invM, float voxelSize, int W, int H, const float *pointsRay, float *pointsOut, float *normalsOut,
                          unsigned char *greyOut) {
  using namespace dsr;
  FrameP p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.invM.m, invM, sizeof p.invM.m);
  p.voxelSize = voxelSize; p.W = W; p.H = H;
  const float4 *pr = reinterpret_cast<const float4 *>(pointsRay);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float4 pt, nm;
      uchar4 g;
      icp_pixel<HostOps>(p, pr, x, y, pt, nm, g);
      const size_t i = (size_t)x + (size_t)y * W;
      std::memcpy(pointsOut + 4 * i, &pt, 16); std::memcpy(normalsOut + 4 * i, &nm, 16); std::memcpy(greyOut + 4 * i, &g, 4);
    }
  return 0;
}

// k_render: one free-view image type from that view's raycast result; vba in the library's full block layout
extern "C" int rr_render_all(int type, const float *M, const float *invM, float voxelSize, int maxW, int W, int H, int noBuckets,
                             int noTotalEntries, const dsr_hash_entry *table, const unsigned char *vba, const float *pointsRay,
                             unsigned char *rgbaOut, float *depthOut) {
  using namespace dsr;
  FrameP p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.M.m, M, sizeof p.M.m);
  std::memcpy(p.invM.m, invM, sizeof p.invM.m);
  p.voxelSize = voxelSize; p.maxW = maxW; p.W = W; p.H = H;
  p.noBuckets = noBuckets; p.noTotalEntries = noTotalEntries; p.hashMask = (uint32_t)noBuckets - 1u;
  SceneP s;
  std::memset(&s, 0, sizeof s);
  s.table = const_cast<dsr_hash_entry *>(table);
  s.vba = const_cast<uint8_t *>(vba);
  const float4 *pr = reinterpret_cast<const float4 *>(pointsRay);
  int scratch[9];
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const size_t i = (size_t)x + (size_t)y * W;
      const uchar4 c = render_pixel<HostOps>(p, s, type, pr[i], scratch);
      std::memcpy(rgbaOut + 4 * i, &c, 4);
      depthOut[i] = render_depth(p, pr[i]);
    }
  return 0;
}

// K6 (k_expected_depth*): the range image of a view from the blocks of its visible list (pos: n x 3 int16, block coordinates)
extern "C" int rr_range_image(const float *M, const float *proj, float voxelSize, int W, int H, const short *pos, int n, float *minmaxOut) {
  using namespace dsr;
  FrameP p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.M.m, M, sizeof p.M.m);
  p.proj = make_float4(proj[0], proj[1], proj[2], proj[3]);
  p.voxelSize = voxelSize; p.W = W; p.H = H;
  const int mw = (W + kMinmaxSubsample - 1) / kMinmaxSubsample, mh = (H + kMinmaxSubsample - 1) / kMinmaxSubsample;
  for (int c = 0; c < mw * mh; ++c) { minmaxOut[2 * c] = kFarAway; minmaxOut[2 * c + 1] = kVeryClose; }
  for (int i = 0; i < n; ++i) {
    int2 ul, lr;
    float2 zr;
    if (!project_single_block<HostOps>(pos + 3 * i, p, mw, mh, ul, lr, zr)) continue;
    for (int y = ul.y; y <= lr.y; ++y)
      for (int x = ul.x; x <= lr.x; ++x) {
        float *c = minmaxOut + 2 * (x + y * mw);
        if (zr.x < c[0]) c[0] = zr.x;
        if (zr.y > c[1]) c[1] = zr.y;
      }
  }
  return 0;
}

// K5 (FindVisibleBlocks of a free view): the entries with a block whose corners reach into the image, ascending
extern "C" int rr_freeview_visible(const float *M, const float *proj, float voxelSize, int W, int H, const dsr_hash_entry *table, int noTotalEntries,
                                   int *idsOut) {
  using namespace dsr;
  Mat4 m;
  std::memcpy(m.m, M, sizeof m.m);
  const float4 pr = make_float4(proj[0], proj[1], proj[2], proj[3]);
  int n = 0;
  for (int t = 0; t < noTotalEntries; ++t) {
    if (table[t].ptr < 0) continue;
    bool vis, visEnlarged;
    check_block_visibility<false, HostOps>(vis, visEnlarged, table[t].pos, m, pr, voxelSize, W, H);
    if (vis) idsOut[n++] = t;
  }
  return n;
}

// K1 (buildHashAllocAndVisibleTypePP): the blocks the depth rays of a frame ask for — alloc_ray's segment [d - mu, d + mu] walked
// with the kernel's running additions.  out: up to cap (x, y, z) int16 triples, sorted; returns the number of distinct blocks.
extern "C" int rr_alloc_blocks(const float *invM, const float *proj, float voxelSize, float mu, float vfMin, float vfMax, int W, int H,
                               const float *depth, short *out, int cap) {
  using namespace dsr;
  FrameP p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.invM.m, invM, sizeof p.invM.m);
  p.proj = make_float4(proj[0], proj[1], proj[2], proj[3]);
  p.voxelSize = voxelSize; p.mu = mu; p.vfMin = vfMin; p.vfMax = vfMax; p.W = W; p.H = H;
  std::set<std::tuple<short, short, short>> blocks;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      AllocRay r;
      if (!alloc_ray<HostOps>(p, depth, x, y, r)) continue;
      float px = r.px, py = r.py, pz = r.pz;
      for (int i = 0; i < r.noSteps; ++i) {
        blocks.emplace((short)HostOps::f2i(floorf(px)), (short)HostOps::f2i(floorf(py)), (short)HostOps::f2i(floorf(pz)));
        px += r.dx; py += r.dy; pz += r.dz;
      }
    }
  int n = 0;
  for (const auto &b : blocks) {
    if (n < cap) { out[3 * n] = std::get<0>(b); out[3 * n + 1] = std::get<1>(b); out[3 * n + 2] = std::get<2>(b); }
    ++n;
  }
  return n;
}

// ---- the edges of the path (k_edges.h, k_composite.h): same arguments as the dsr_* entry points of include/dsr.h

extern "C" int hs_depth_from_disparity(const float *disparity, int16_t *depth_mm_out, int n, float baseline_m, float focal_px, float scale,
                                       float min_depth_m, float max_depth_m) {
  const int minMm = (int)(min_depth_m * 1000.0f), maxMm = (int)(max_depth_m * 1000.0f);  // dsr_depth_from_disparity_dev
  if (maxMm >= 32767) return 1;
  for (int i = 0; i < n; ++i) dsr::depth_from_disparity_px<HostOps>(i, disparity, depth_mm_out, baseline_m, focal_px, scale, minMm, maxMm);
  return 0;
}
extern "C" int hs_bgr_to_rgba(const uint8_t *bgr, uint8_t *rgba_out, int n) {
  for (int i = 0; i < n; ++i) dsr::bgr_to_rgba_px(i, bgr, reinterpret_cast<uchar4 *>(rgba_out));
  return 0;
}
extern "C" int hs_rgba_to_bgr(const uint8_t *rgba, uint8_t *bgr_out, int n) {
  for (int i = 0; i < n; ++i) dsr::rgba_to_bgr_px(i, reinterpret_cast<const uchar4 *>(rgba), bgr_out);
  return 0;
}
extern "C" int hs_depth_m_to_mm(const float *depth_m, int16_t *depth_mm_out, int n) {
  for (int i = 0; i < n; ++i) dsr::depth_m_to_mm_px<HostOps>(i, depth_m, depth_mm_out);
  return 0;
}
extern "C" int hs_composite_instances(uint8_t *target_rgba, float *target_depth, const uint8_t *layers_rgba, const float *layers_depth,
                                      const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength, int dim_background) {
  using namespace dsr;
  if (n_layers > kMaxCompositeLayers) return 1;
  const CompositeP c = composite_params(track_ids, n_layers, n_pixels, tint_strength, dim_background);
  CompositeLayers none;
  std::memset(&none, 0, sizeof none);
  for (int i = 0; i < n_pixels; ++i)
    composite_px<false>(i, c, reinterpret_cast<uchar4 *>(target_rgba), target_depth, reinterpret_cast<const uchar4 *>(layers_rgba), layers_depth, none);
  return 0;
}
// ProcessSilhouette_CPU / RemoveSilhouette_CPU on plain buffers (the dsr_view_* entry points apply them to two engines' views)
extern "C" int hs_extract_silhouette(const uint8_t *src_rgba, const float *src_depth, uint8_t *dst_rgba, float *dst_depth, int W, int H,
                                     const uint8_t *mask, int x0, int y0, int bw, int bh) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      dsr::extract_silhouette_px(x, y, reinterpret_cast<const uchar4 *>(src_rgba), src_depth, reinterpret_cast<uchar4 *>(dst_rgba), dst_depth, W, mask,
                                 x0, y0, bw, bh);
  return 0;
}
extern "C" int hs_remove_silhouette(uint8_t *rgba, float *depth, int W, int H, const uint8_t *mask, int x0, int y0, int bw, int bh) {
  for (int row = 0; row < bh; ++row)
    for (int col = 0; col < bw; ++col) dsr::remove_silhouette_px(col, row, reinterpret_cast<uchar4 *>(rgba), depth, W, H, mask, x0, y0, bw);
  return 0;
}

