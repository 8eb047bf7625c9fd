// k_mesh.h — marching cubes over the allocated blocks.
//
// Replaces ITMMeshingEngine_{CPU,CUDA}<TVoxel,ITMVoxelBlockHash>::MeshScene.  Upstream's CUDA engine
// runs one 8x8x8 thread block per hash entry and appends triangles with atomicAdd (arbitrary
// order); here the output order is the serial engine's (entries ascending, voxels z/y/x,
// triangles in table order) so that the mesh is reproducible and comparable bit for bit:
//   pass 1  k_mesh_blocks<false>: a wave64 per allocated block (list built by k_allocated_*) stages
//           the 9x9x9 corner lattice of the block — its own sdf plane plus the first layer of the 7
//           neighbours in +x/+y/+z, found with 8 table lookups done by 8 lanes — in LDS, classifies
//           the 512 cells and counts the triangles of the block;
//   scan    block counts -> block offsets (tile sums + one-workgroup scan + tile pass);
//   pass 2  k_mesh_blocks<true>: same staging, a wave prefix over the lanes' counts gives every cell
//           its slot; vertices by sdfInterp on the cell edges, scaled to metres.
// Arithmetic follows ITMMeshingEngine.h (findPointNeighbors, sdfInterp, buildVertList).
#pragma once
#include "dsr_device.h"

#define MC_TABLE_ATTR __constant__ static const
#include "mc_tables.h"

namespace dsr {

struct MeshP {
  float voxelSize;
  uint32_t hashMask;
  int noBuckets;
};

constexpr int kMeshWaves = 4;
constexpr int kMissingCorner = 0x7fffffff;  // lattice value of a corner whose block is not allocated

// ITMMeshingEngine.h sdfInterp on one coordinate triple
__device__ __forceinline__ float3 sdf_interp(float3 p1, float3 p2, float v1, float v2) {
  if (fabsf(0.0f - v1) < 0.00001f) return p1;
  if (fabsf(0.0f - v2) < 0.00001f) return p2;
  if (fabsf(v1 - v2) < 0.00001f) return p1;
  const float t = (0.0f - v1) / (v2 - v1);
  return make_float3(p1.x + t * (p2.x - p1.x), p1.y + t * (p2.y - p1.y), p1.z + t * (p2.z - p1.z));
}

template <bool WRITE>
__global__ __launch_bounds__(64 * kMeshWaves) void k_mesh_blocks(SceneP s, MeshP mp, const int32_t *__restrict__ blockList,
                                                                 const int32_t *__restrict__ nPtr,
                                                                 uint32_t *__restrict__ blockCount,
                                                                 const uint32_t *__restrict__ blockOffset,
                                                                 dsr_triangle *__restrict__ out, unsigned long long cap) {
  __shared__ int s_lat[kMeshWaves][9 * 9 * 9];
  __shared__ int s_nbr[kMeshWaves][8];
  const int n = *nPtr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int *lat = s_lat[wave];
  int *nbr = s_nbr[wave];
  for (int i = blockIdx.x * kMeshWaves + wave; i < n; i += gridDim.x * kMeshWaves) {
    const dsr_hash_entry he = load_entry(s.table, (uint32_t)blockList[i]);
    // ---- the 8 blocks the lattice touches: lane k looks up block pos + (k&1, k>>1&1, k>>2)
    if (lane < 8) {
      const int bx = he.pos[0] + (lane & 1), by = he.pos[1] + ((lane >> 1) & 1), bz = he.pos[2] + (lane >> 2);
      int ptr = -1;
      uint32_t h = hash_index(bx, by, bz, mp.hashMask);
      while (true) {  // ITMRepresentationAccess.h findVoxel
        const dsr_hash_entry q = load_entry(s.table, h);
        if (q.pos[0] == bx && q.pos[1] == by && q.pos[2] == bz && q.ptr >= 0) { ptr = q.ptr; break; }
        if (q.offset < 1) break;
        h = (uint32_t)(mp.noBuckets + q.offset - 1);
      }
      nbr[lane] = ptr;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c = lane; c < 729; c += 64) {
      const int cx = c % 9, cy = (c / 9) % 9, cz = c / 81;
      const int ptr = nbr[(cx >> 3) | ((cy >> 3) << 1) | ((cz >> 3) << 2)];
      int v = kMissingCorner;
      if (ptr >= 0) {
        const int lin = (cx & 7) + ((cy & 7) << 3) + ((cz & 7) << 6);
        v = (int)*reinterpret_cast<const short *>(s.vba + (size_t)ptr * kBlockBytes + kOffSdf + lin * 2);
      }
      lat[c] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- lane l owns cells locId = 8 l .. 8 l + 7: x = 0..7 of row (y, z) = (l & 7, l >> 3)
    const int y = lane & 7, z = lane >> 3;
    int cube[8];
    int nTri = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      // corners in the cube numbering of the tables
      const int o = x + y * 9 + z * 81;
      const int c0 = lat[o], c1 = lat[o + 1], c2 = lat[o + 10], c3 = lat[o + 9];
      const int c4 = lat[o + 81], c5 = lat[o + 82], c6 = lat[o + 91], c7 = lat[o + 90];
      // findPointNeighbors: every corner present and not at the initial value (sdf == 1.0f <=> 32767)
      auto usable = [](int c) { return c != kMissingCorner && c != 32767; };
      const bool ok = usable(c0) && usable(c1) && usable(c2) && usable(c3) && usable(c4) && usable(c5) && usable(c6) && usable(c7);
      // sdf < 0 <=> short < 0 (the division by 32767 keeps the sign)
      int ci = (c0 < 0 ? 1 : 0) | (c1 < 0 ? 2 : 0) | (c2 < 0 ? 4 : 0) | (c3 < 0 ? 8 : 0) | (c4 < 0 ? 16 : 0) |
               (c5 < 0 ? 32 : 0) | (c6 < 0 ? 64 : 0) | (c7 < 0 ? 128 : 0);
      if (!ok || kMcEdgeTable[ci] == 0) ci = -1;
      cube[x] = ci;
      if (ci >= 0)
        for (int k = 0; kMcTriTable[ci][k] != -1; k += 3) nTri++;
    }
    // wave prefix over the lanes (cells ascend with the lane)
    int inc = nTri;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    const int total = __shfl(inc, 63);
    if (!WRITE) {
      if (lane == 0) blockCount[i] = (uint32_t)total;
    } else if (total > 0) {
      unsigned long long slot = (unsigned long long)blockOffset[i] + (unsigned long long)(inc - nTri);
      const int gx = he.pos[0] * kBlockSize, gy = he.pos[1] * kBlockSize + y, gz = he.pos[2] * kBlockSize + z;
      for (int x = 0; x < 8; ++x) {
        const int ci = cube[x];
        if (ci < 0) continue;
        const int o = x + y * 9 + z * 81;
        const int off[8] = {o, o + 1, o + 10, o + 9, o + 81, o + 82, o + 91, o + 90};
        for (int k = 0; kMcTriTable[ci][k] != -1; k += 3) {
          float3 v[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int e = kMcTriTable[ci][k + j];
            // edge e joins corners a and b: 0-1 1-2 2-3 3-0 4-5 5-6 6-7 7-4 0-4 1-5 2-6 3-7
            const int a = e < 8 ? e : e - 8, b = e < 8 ? ((e & 4) | ((e + 1) & 3)) : e - 4;
            const int ax = (a == 1 || a == 2 || a == 5 || a == 6), ay = (a == 2 || a == 3 || a == 6 || a == 7), az = a >> 2;
            const int bx = (b == 1 || b == 2 || b == 5 || b == 6), by = (b == 2 || b == 3 || b == 6 || b == 7), bz = b >> 2;
            const float3 pa = make_float3((float)(gx + x + ax), (float)(gy + ay), (float)(gz + az));
            const float3 pb = make_float3((float)(gx + x + bx), (float)(gy + by), (float)(gz + bz));
            const float va = sdf_to_float((float)lat[off[a]]), vb = sdf_to_float((float)lat[off[b]]);
            const float3 q = sdf_interp(pa, pb, va, vb);
            v[j] = make_float3(q.x * mp.voxelSize, q.y * mp.voxelSize, q.z * mp.voxelSize);
          }
          // triangles[n] = t; if (n < noMaxTriangles - 1) n++;  => the first cap triangles survive
          if (slot < cap) {
            float *t = reinterpret_cast<float *>(out + slot);
            t[0] = v[0].x; t[1] = v[0].y; t[2] = v[0].z; t[3] = v[1].x; t[4] = v[1].y; t[5] = v[1].z;
            t[6] = v[2].x; t[7] = v[2].y; t[8] = v[2].z;
          }
          slot++;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- exclusive scan of the per-block triangle counts (tiles of kTile blocks)
__global__ __launch_bounds__(kTileThreads) void k_u32_tile_sums(const uint32_t *__restrict__ v, const int32_t *__restrict__ nPtr,
                                                                int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int n = *nPtr;
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);
  for (int j = 0; j < kTileItems; ++j)
    if (base + j < n) c.x += (int)v[base + j];
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kTileThreads) void k_u32_tile_offsets(const uint32_t *__restrict__ v, const int32_t *__restrict__ nPtr,
                                                                   const int2 *__restrict__ tileOffsets,
                                                                   uint32_t *__restrict__ out) {
  __shared__ int2 lds[kTileThreads / 64];
  const int n = *nPtr;
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  uint32_t a[kTileItems];
  int2 c = make_int2(0, 0);
  for (int j = 0; j < kTileItems; ++j) {
    a[j] = base + j < n ? v[base + j] : 0u;
    c.x += (int)a[j];
  }
  int2 total;
  const int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  uint32_t run = (uint32_t)(tileOffsets[blockIdx.x].x + ex.x);
  for (int j = 0; j < kTileItems; ++j)
    if (base + j < n) { out[base + j] = run; run += a[j]; }
}

}  // namespace dsr
