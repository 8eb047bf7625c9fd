// k_integrate.h — K4: SDF / colour integration of every visible voxel block.
//
// Replaces integrateIntoScene_device (upstream: one 512-thread CUDA block per voxel block,
// 8 B array-of-structs voxels).  CDNA4 formulation: ONE WAVE64 PER VOXEL BLOCK, lane = (y,z)
// row of the 8^3 block, 8 voxels along x per lane.  With the plane-wise block layout
// (dsr_device.h) every lane moves 16 B of sdf, 8 B of w_depth, 8 B of w_color and 32 B of
// colour with fully coalesced dwordx4/dwordx2 accesses (1 KiB per wave instruction on the
// sdf plane).  A fixed persistent grid strides over the visible list whose length is read
// from device memory, so the host never synchronises to learn noVisibleBlocks.
// Planes a lane did not change are not written back.
//
// Arithmetic follows ITMSceneReconstructionEngine.h computeUpdatedVoxelDepthInfo /
// computeUpdatedVoxelColorInfo / ComputeUpdatedVoxelInfo<true> expression by expression.
#pragma once
#include "dsr_device.h"

namespace dsr {

// ITMPixelUtils.h interpolateBilinear on the RGBA frame (uchar4 gathers)
__device__ __forceinline__ float3 bilinear_rgb(const uchar4 *__restrict__ src, float px, float py, int W) {
  const int ix = f2i(floorf(px)), iy = f2i(floorf(py));
  const float dx = px - (float)ix, dy = py - (float)iy;
  uchar4 a = src[ix + iy * W];
  uchar4 b = make_uchar4(0, 0, 0, 0), c = b, d = b;
  if (dx != 0) b = src[(ix + 1) + iy * W];
  if (dy != 0) c = src[ix + (iy + 1) * W];
  if (dx != 0 && dy != 0) d = src[(ix + 1) + (iy + 1) * W];
  float3 r;
  r.x = ((float)a.x * (1.0f - dx) * (1.0f - dy) + (float)b.x * dx * (1.0f - dy) + (float)c.x * (1.0f - dx) * dy + (float)d.x * dx * dy);
  r.y = ((float)a.y * (1.0f - dx) * (1.0f - dy) + (float)b.y * dx * (1.0f - dy) + (float)c.y * (1.0f - dx) * dy + (float)d.y * dx * dy);
  r.z = ((float)a.z * (1.0f - dx) * (1.0f - dy) + (float)b.z * dx * (1.0f - dy) + (float)c.z * (1.0f - dx) * dy + (float)d.z * dx * dy);
  return r;
}

// fork: WeightParams.depthWeighting; adopted definition max(1, round(10/z)) (see oracle)
__device__ __forceinline__ int depth_weight(float depth_measure) {
  int w = f2i(10.0f / depth_measure + 0.5f);
  return w < 1 ? 1 : w;
}

template <bool RGB_SAME>
__global__ __launch_bounds__(256) void k_integrate(FrameP p, SceneP s, const float *__restrict__ depth,
                                                   const uchar4 *__restrict__ rgb,
                                                   const int32_t *__restrict__ visibleIDs) {
  const int noVisible = s.ctr[CTR_NO_VISIBLE_LIVE];
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&s.work[WORK_V_INTEGRATED], (unsigned long long)noVisible);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wavesInGrid = gridDim.x * 4;
  const int ly = lane & 7, lz = lane >> 3;
  const Mat4 &Mr = RGB_SAME ? p.M : p.M_rgb;
  const float4 projr = RGB_SAME ? p.proj : p.proj_rgb;
  const int Wc = RGB_SAME ? p.W : p.Wr, Hc = RGB_SAME ? p.H : p.Hr;

  for (int b = blockIdx.x * 4 + wave; b < noVisible; b += wavesInGrid) {
    const int entryId = __builtin_amdgcn_readfirstlane(visibleIDs[b]);
    const dsr_hash_entry he = load_entry(s.table, entryId);
    if (he.ptr < 0) continue;
    uint8_t *blk = s.vba + (size_t)he.ptr * kBlockBytes;

    uint4 sdfRaw = *reinterpret_cast<const uint4 *>(blk + kOffSdf + lane * 16);
    uint2 wdRaw = *reinterpret_cast<const uint2 *>(blk + kOffWDepth + lane * 8);
    uint2 wcRaw = *reinterpret_cast<const uint2 *>(blk + kOffWColor + lane * 8);
    uint4 c0 = *reinterpret_cast<const uint4 *>(blk + kOffClr + lane * 32);
    uint4 c1 = *reinterpret_cast<const uint4 *>(blk + kOffClr + lane * 32 + 16);

    uint32_t sdfW[4] = {sdfRaw.x, sdfRaw.y, sdfRaw.z, sdfRaw.w};
    uint32_t wdW[2] = {wdRaw.x, wdRaw.y};
    uint32_t wcW[2] = {wcRaw.x, wcRaw.y};
    uint32_t clrW[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    bool dirtyDepth = false, dirtyColor = false;

    const int gx = he.pos[0] * kBlockSize, gy = he.pos[1] * kBlockSize, gz = he.pos[2] * kBlockSize;
    const float my = (float)(gy + ly) * p.voxelSize;
    const float mz = (float)(gz + lz) * p.voxelSize;

#pragma unroll
    for (int x = 0; x < 8; ++x) {
      short sdf = (short)((sdfW[x >> 1] >> ((x & 1) * 16)) & 0xffffu);
      int wDepth = (int)((wdW[x >> 2] >> ((x & 3) * 8)) & 0xffu);
      if (p.stopAtMaxW && wDepth == p.maxW) continue;
      const float mx = (float)(gx + x) * p.voxelSize;

      // ---- computeUpdatedVoxelDepthInfo
      float eta;
      float3 pc = mat_mul3(p.M, mx, my, mz, 1.0f);
      float u = 0.0f, v = 0.0f;
      bool projected = false;
      if (pc.z <= 0) eta = -1.0f;
      else {
        u = p.proj.x * pc.x / pc.z + p.proj.z;
        v = p.proj.y * pc.y / pc.z + p.proj.w;
        projected = true;
        if ((u < 1) || (u > (float)(p.W - 2)) || (v < 1) || (v > (float)(p.H - 2))) eta = -1.0f;
        else {
          float depth_measure = depth[f2i(u + 0.5f) + f2i(v + 0.5f) * p.W];
          if (depth_measure <= 0.0f) eta = -1.0f;
          else {
            eta = depth_measure - pc.z;
            if (!(eta < -p.mu)) {
              float oldF = sdf_to_float((float)sdf);
              int oldW = wDepth;
              const float q = eta / p.mu;
              float newF = (1.0f < q) ? 1.0f : q;  // MIN(1.0f, eta / mu)
              int newW = p.depthWeighting ? depth_weight(depth_measure) : 1;
              newF = (float)oldW * oldF + (float)newW * newF;
              newW = oldW + newW;
              newF /= (float)newW;
              newW = newW < p.maxW ? newW : p.maxW;
              sdf = sdf_from_float(newF);
              sdfW[x >> 1] = (sdfW[x >> 1] & ~(0xffffu << ((x & 1) * 16))) | ((uint32_t)(uint16_t)sdf << ((x & 1) * 16));
              wdW[x >> 2] = (wdW[x >> 2] & ~(0xffu << ((x & 3) * 8))) | ((uint32_t)(newW & 0xff) << ((x & 3) * 8));
              dirtyDepth = true;
            }
          }
        }
      }
      // ---- ComputeUpdatedVoxelInfo<true>::compute gate
      if ((eta > p.mu) || (fabsf(eta / p.mu) > 0.25f)) continue;

      // ---- computeUpdatedVoxelColorInfo
      float uc, vc;
      if (RGB_SAME && projected) { uc = u; vc = v; }
      else {
        float3 pr = mat_mul3(Mr, mx, my, mz, 1.0f);
        uc = projr.x * pr.x / pr.z + projr.z;
        vc = projr.y * pr.y / pr.z + projr.w;
      }
      if ((uc < 1) || (uc > (float)(Wc - 2)) || (vc < 1) || (vc > (float)(Hc - 2))) continue;
      const uint32_t cw = clrW[x];
      const float oldWc = (float)((wcW[x >> 2] >> ((x & 3) * 8)) & 0xffu);
      float ocx = (float)(cw & 0xffu) / 255.0f, ocy = (float)((cw >> 8) & 0xffu) / 255.0f, ocz = (float)((cw >> 16) & 0xffu) / 255.0f;
      float3 m = bilinear_rgb(rgb, uc, vc, Wc);
      float rx = m.x / 255.0f, ry = m.y / 255.0f, rz = m.z / 255.0f;
      float newWc = 1.0f;
      float ncx = ocx * oldWc + rx * newWc, ncy = ocy * oldWc + ry * newWc, ncz = ocz * oldWc + rz * newWc;
      newWc = oldWc + newWc;
      ncx /= newWc; ncy /= newWc; ncz /= newWc;
      newWc = (newWc < (float)p.maxW) ? newWc : (float)p.maxW;  // MIN(newW, maxW)
      uint32_t r8 = (uint32_t)f2i(ncx * 255.0f) & 0xffu, g8 = (uint32_t)f2i(ncy * 255.0f) & 0xffu, b8 = (uint32_t)f2i(ncz * 255.0f) & 0xffu;
      clrW[x] = r8 | (g8 << 8) | (b8 << 16);
      wcW[x >> 2] = (wcW[x >> 2] & ~(0xffu << ((x & 3) * 8))) | (((uint32_t)f2i(newWc) & 0xffu) << ((x & 3) * 8));
      dirtyColor = true;
    }

    if (dirtyDepth) {
      *reinterpret_cast<uint4 *>(blk + kOffSdf + lane * 16) = make_uint4(sdfW[0], sdfW[1], sdfW[2], sdfW[3]);
      *reinterpret_cast<uint2 *>(blk + kOffWDepth + lane * 8) = make_uint2(wdW[0], wdW[1]);
    }
    if (dirtyColor) {
      *reinterpret_cast<uint2 *>(blk + kOffWColor + lane * 8) = make_uint2(wcW[0], wcW[1]);
      *reinterpret_cast<uint4 *>(blk + kOffClr + lane * 32) = make_uint4(clrW[0], clrW[1], clrW[2], clrW[3]);
      *reinterpret_cast<uint4 *>(blk + kOffClr + lane * 32 + 16) = make_uint4(clrW[4], clrW[5], clrW[6], clrW[7]);
    }
  }
}

}  // namespace dsr
