// k_raycast_lds.h — K7 with a wave-cooperative LDS cache of sdf planes.
//
// Measurements of the per-lane formulation (k_raycast.h): 79 % of wave cycles waiting on memory,
// bound by gather-REQUEST throughput — every march step of every lane is a scattered 2-byte load
// (plus 8 more near the surface), although the 64 rays of an 8x8 pixel tile walk through the same
// few voxel blocks.  Here a wave keeps the 1 KiB sdf planes of the blocks its rays are in resident
// in LDS: a plane is fetched ONCE per wave by one fully coalesced 1 KiB load (64 lanes x 16 B) and
// then serves all lanes and all their steps from LDS.  Hash-table probes stay per lane (they are
// needed only when a lane enters a block that is not resident).
//
// STATUS (round 1): bit-exact (tests/test_gpu_parity.py::test_lds_raycast_variant) but SLOWER than the
// per-lane kernel — 1.49 ms vs 0.66 ms at the 5 mm bench, for 2, 4 and 8 slots alike: rays enter a new
// block every 1-2 steps, so fills are compulsory, and each fill is one more DRAM round trip that the
// whole wave waits for.  The march is latency-bound per wave (all waves are co-resident), not
// request-bound.  Not used by default (engine: DSR_RAYCAST_SLOTS); next step would be batching the
// fills of a step and prefetching along the ray.
//
// Exactness: a voxel lookup is a pure function of (hash table, voxel array); how the value reaches
// the lane cannot change it.  The march itself — sample points, step lengths, refinement — is
// castRay (ITMVisualisationEngine.h) expression by expression, as in k_raycast.h; only the control
// flow is rewritten in wave-convergent form (every lane stays in the loop until the wave is done)
// so that all 64 lanes can take part in a fill.
#pragma once
#include "k_raycast.h"

namespace dsr {

template <int K>
struct WaveSdfCache {
  int4 *tags;          // [K] (bx, by, bz, ptr) of the resident planes; ptr < 0 = empty slot
  unsigned short *data;  // [K][512]
  int next;            // round-robin victim (identical in all lanes)
  // per-lane memo of the last probed block (positive or negative)
  int lbx, lby, lbz, lptr;
};

// Raw sdf (float(short), 32767 for a missing block) of voxel (vx,vy,vz) for lanes with want==true.
// MUST be called by all lanes of the wave (convergent); lanes with want==false only help filling.
template <int K>
__device__ __forceinline__ float wave_read_sdf(const SceneP &s, const FrameP &p, WaveSdfCache<K> &c, bool want, int vx,
                                               int vy, int vz, bool &found) {
  const int lane = threadIdx.x & 63;
  const int bx = vx >> 3, by = vy >> 3, bz = vz >> 3;
  const int lin = (vx & 7) + ((vy & 7) << 3) + ((vz & 7) << 6);
  float val = 32767.0f;
  found = false;
  bool needFill = false;
  int ptr = -1;
  if (want) {
    int slot = -1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int4 t = c.tags[k];
      if (t.x == bx && t.y == by && t.z == bz && t.w >= 0) slot = k;
    }
    if (slot >= 0) {  // resident: read before any fill can evict it
      val = (float)(short)c.data[slot * kBlockSize3 + lin];
      found = true;
    } else {
      if (bx == c.lbx && by == c.lby && bz == c.lbz) ptr = c.lptr;
      else {
        // ITMRepresentationAccess.h readVoxel: bucket head, then the excess chain
        uint32_t hashIdx = hash_index(bx, by, bz, p.hashMask);
        while (true) {
          const dsr_hash_entry he = load_entry(s.table, hashIdx);
          if (he.pos[0] == bx && he.pos[1] == by && he.pos[2] == bz && he.ptr >= 0) { ptr = he.ptr; break; }
          if (he.offset < 1) break;
          hashIdx = (uint32_t)(p.noBuckets + he.offset - 1);
        }
        c.lbx = bx; c.lby = by; c.lbz = bz; c.lptr = ptr;
      }
      needFill = ptr >= 0;
    }
  }
  unsigned long long m = __ballot(needFill);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    const int P = __shfl(ptr, src), Bx = __shfl(bx, src), By = __shfl(by, src), Bz = __shfl(bz, src);
    const int k = c.next;
    c.next = (k + 1 == K) ? 0 : k + 1;
    // one coalesced 1 KiB load of the plane, 16 B per lane, into slot k
    const uint4 v = *reinterpret_cast<const uint4 *>(s.vba + (size_t)P * kBlockBytes + kOffSdf + lane * 16);
    *reinterpret_cast<uint4 *>(c.data + k * kBlockSize3 + lane * 8) = v;
    if (lane == 0) c.tags[k] = make_int4(Bx, By, Bz, P);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (needFill && bx == Bx && by == By && bz == Bz) {
      val = (float)(short)c.data[k * kBlockSize3 + lin];
      found = true;
      needFill = false;
    }
    m = __ballot(needFill);
  }
  return val;
}

// readFromSDF_float_interpolated for lanes with want==true (convergent).  Cells inside one block
// (2/3 of the samples) take one cooperative lookup and 8 LDS reads; cells straddling blocks use the
// per-lane global path of k_raycast.h.
template <int K>
__device__ __forceinline__ float wave_read_sdf_interpolated(const SceneP &s, const FrameP &p, WaveSdfCache<K> &c, bool want,
                                                            float x, float y, float z, VoxCache &gcache, VoxCache &gcache2) {
  const int ix = f2i(floorf(x)), iy = f2i(floorf(y)), iz = f2i(floorf(z));
  const bool single = ((ix & 7) != 7) && ((iy & 7) != 7) && ((iz & 7) != 7);
  bool f;
  // make the base block resident (and fetch corner 0) cooperatively
  const float v0 = wave_read_sdf<K>(s, p, c, want && single, ix, iy, iz, f);
  float result = 0.0f;
  if (want) {
    if (single) {
      const float cx = x - (float)ix, cy = y - (float)iy, cz = z - (float)iz;
      float v[8];
      v[0] = v0;
#pragma unroll
      for (int k = 1; k < 8; ++k) v[k] = 32767.0f;
      if (f) {
        // the block is resident in the slot that served corner 0: find it again and read the rest
        const int bx = ix >> 3, by = iy >> 3, bz = iz >> 3;
        const int lin = (ix & 7) + ((iy & 7) << 3) + ((iz & 7) << 6);
        int slot = -1;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int4 t = c.tags[k];
          if (t.x == bx && t.y == by && t.z == bz && t.w >= 0) slot = k;
        }
        if (slot >= 0) {
          const unsigned short *b = c.data + slot * kBlockSize3 + lin;
          v[1] = (float)(short)b[1]; v[2] = (float)(short)b[8]; v[3] = (float)(short)b[9];
          v[4] = (float)(short)b[64]; v[5] = (float)(short)b[65]; v[6] = (float)(short)b[72]; v[7] = (float)(short)b[73];
        } else {
          // evicted by a fill for another lane in the same call: read the corners from HBM
          int l2;
          const int ptr = find_block(s, p, ix, iy, iz, l2, gcache);
          const short *b = reinterpret_cast<const short *>(s.vba + (size_t)ptr * kBlockBytes + kOffSdf) + l2;
          v[1] = (float)b[1]; v[2] = (float)b[8]; v[3] = (float)b[9];
          v[4] = (float)b[64]; v[5] = (float)b[65]; v[6] = (float)b[72]; v[7] = (float)b[73];
        }
      }
      float res1 = (1.0f - cx) * v[0] + cx * v[1];
      res1 = (1.0f - cy) * res1 + cy * ((1.0f - cx) * v[2] + cx * v[3]);
      float res2 = (1.0f - cx) * v[4] + cx * v[5];
      res2 = (1.0f - cy) * res2 + cy * ((1.0f - cx) * v[6] + cx * v[7]);
      result = sdf_to_float((1.0f - cz) * res1 + cz * res2);
    } else {
      result = read_sdf_interpolated(s, p, x, y, z, gcache, gcache2);
    }
  }
  return result;
}

template <int K>
__global__ __launch_bounds__(256) void k_raycast_lds(FrameP p, SceneP s, int ctrIdx, const float2 *__restrict__ minmax,
                                                     float4 *__restrict__ raycastResult) {
  __shared__ int4 s_tags[4][K];
  __shared__ unsigned short s_data[4][K * kBlockSize3];
  if (s.ctr[ctrIdx] <= 0 && ctrIdx == CTR_NO_VISIBLE_LIVE) return;  // Prepare() is skipped without visible blocks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool inImage = x < p.W && y < p.H;
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample;

  WaveSdfCache<K> c;
  c.tags = s_tags[wave];
  c.data = s_data[wave];
  c.next = 0;
  c.lbx = c.lby = c.lbz = 0x7fffffff; c.lptr = -1;
  if (lane < K) c.tags[lane] = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, -1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  VoxCache gcache; cache_init(gcache);
  VoxCache gcache2; cache_init(gcache2);

  // ---- castRay set-up (ITMVisualisationEngine.h), identical to cast_ray() in k_raycast.h
  const float2 mm = inImage ? minmax[(x >> 3) + (y >> 3) * mw] : make_float2(kFarAway, kVeryClose);
  const float oneOverVoxelSize = 1.0f / p.voxelSize;
  const float invFx = 1.0f / p.proj.x, invFy = 1.0f / p.proj.y;
  const float stepScale = p.mu * oneOverVoxelSize;
  float cz = mm.x;
  float cx = cz * (((float)x - p.proj.z) * invFx);
  float cy = cz * (((float)y - p.proj.w) * invFy);
  float totalLength = sqrtf(cx * cx + cy * cy + cz * cz) * oneOverVoxelSize;
  float3 t = mat_mul3(p.invM, cx, cy, cz, 1.0f);
  const float sx = t.x * oneOverVoxelSize, sy = t.y * oneOverVoxelSize, sz = t.z * oneOverVoxelSize;
  cz = mm.y;
  cx = cz * (((float)x - p.proj.z) * invFx);
  cy = cz * (((float)y - p.proj.w) * invFy);
  const float totalLengthMax = sqrtf(cx * cx + cy * cy + cz * cz) * oneOverVoxelSize;
  t = mat_mul3(p.invM, cx, cy, cz, 1.0f);
  const float ex = t.x * oneOverVoxelSize, ey = t.y * oneOverVoxelSize, ez = t.z * oneOverVoxelSize;
  float dx = ex - sx, dy = ey - sy, dz = ez - sz;
  const float direction_norm = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= direction_norm; dy *= direction_norm; dz *= direction_norm;

  float rx = sx, ry = sy, rz = sz;
  float sdfValue = 1.0f, stepLength;
  bool active = inImage && (totalLength < totalLengthMax);

  // ---- the march, wave-convergent: `while (totalLength < totalLengthMax) {...}` per lane
  while (__any(active)) {
    bool hash_found;
    const float raw = wave_read_sdf<K>(s, p, c, active, f2i(roundf_itm(rx)), f2i(roundf_itm(ry)), f2i(roundf_itm(rz)), hash_found);
    float sv = sdf_to_float(raw);
    const bool needTri = active && hash_found && (sv <= 0.1f) && (sv >= -0.5f);
    const float tri = wave_read_sdf_interpolated<K>(s, p, c, needTri, rx, ry, rz, gcache, gcache2);
    if (active) {
      if (needTri) sv = tri;
      sdfValue = sv;
      if (!hash_found) {
        stepLength = (float)kBlockSize;
      } else if (sdfValue <= 0.0f) {
        active = false;  // break
      } else {
        const float ss = sdfValue * stepScale;
        stepLength = (ss > 1.0f) ? ss : 1.0f;  // MAX(sdfValue * stepScale, 1.0f)
      }
      if (active) {
        rx += stepLength * dx; ry += stepLength * dy; rz += stepLength * dz;
        totalLength += stepLength;
        active = totalLength < totalLengthMax;
      }
    }
  }

  // ---- refinement of a hit: two interpolated reads (convergent as well)
  const bool hit = inImage && (sdfValue <= 0.0f);
  if (hit) {
    stepLength = sdfValue * stepScale;
    rx += stepLength * dx; ry += stepLength * dy; rz += stepLength * dz;
  }
  const float refined = wave_read_sdf_interpolated<K>(s, p, c, hit, rx, ry, rz, gcache, gcache2);
  if (inImage) {
    float4 out;
    if (hit) {
      sdfValue = refined;
      stepLength = sdfValue * stepScale;
      rx += stepLength * dx; ry += stepLength * dy; rz += stepLength * dz;
      out.w = 1.0f;
    } else out.w = 0.0f;
    out.x = rx; out.y = ry; out.z = rz;
    raycastResult[x + y * p.W] = out;
  }
}

}  // namespace dsr
