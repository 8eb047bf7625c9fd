// k_raycast.h — K6 expected depths, K7 raycast, K8 ICP maps / shading kernels.
//
// Replaces projectAndSplitBlocks/fillBlocks, genericRaycast, renderICP/renderGrey/renderColour/
// renderColourFromNormal (+ the fork's depth / depth-weight shaders) of upstream's
// ITMVisualisationEngine_CUDA.  Arithmetic follows ITMVisualisationEngine.h and
// ITMRepresentationAccess.h expression by expression.
//
// CDNA4 notes: a wave64 is an 8x8 pixel tile = exactly one cell of the 8x-subsampled
// range image, so all lanes of a wave share one [min,max] range (scalar load, uniform trip
// count bound); the raycast touches only the 1 KiB sdf plane of a block.
#pragma once
#include "dsr_device.h"

namespace dsr {

// ------------------------------------------------------------------ voxel access

struct VoxCache {  // ITMVoxelBlockHash::IndexCache
  int bx, by, bz;
  int ptr;  // block index
};
__host__ __device__ __forceinline__ void cache_init(VoxCache &c) { c.bx = c.by = c.bz = 0x7fffffff; c.ptr = -1; }

// ITMRepresentationAccess.h readVoxel (with per-thread cache): returns the block index
// holding voxel (x,y,z) or -1; linearIdx is the offset inside the block.
__host__ __device__ __forceinline__ int find_block(const SceneP &s, const FrameP &p, int x, int y, int z, int &linearIdx,
                                          VoxCache &cache) {
  const int bx = x >> 3, by = y >> 3, bz = z >> 3;  // == pointToVoxelBlockPos for negatives too
  linearIdx = (x & 7) + ((y & 7) << 3) + ((z & 7) << 6);
  if (bx == cache.bx && by == cache.by && bz == cache.bz) return cache.ptr;
  uint32_t hashIdx = hash_index(bx, by, bz, p.hashMask);
  while (true) {
    dsr_hash_entry he = load_entry(s.table, hashIdx);
    if (he.pos[0] == bx && he.pos[1] == by && he.pos[2] == bz && he.ptr >= 0) {
      cache.bx = bx; cache.by = by; cache.bz = bz; cache.ptr = he.ptr;
      return he.ptr;
    }
    if (he.offset < 1) break;
    hashIdx = (uint32_t)(p.noBuckets + he.offset - 1);
  }
  return -1;
}

// two neighbouring shorts of an sdf plane with ONE load (the address is only 2-byte aligned: unaligned dword
// access is supported for global memory on gfx9 and the compiler emits a single global_load_dword)
__host__ __device__ __forceinline__ uint32_t load_pair(const short *p) {
  uint32_t w;
  __builtin_memcpy(&w, p, 4);
  return w;
}

// ROUND() = (x < 0) ? (x - 0.5f) : (x + 0.5f), always followed by the conversion to int.  x + copysign(0.5, x) is the same float
// except for x = -0 (-0.5 instead of +0.5: both convert to 0) and saves a compare + select per coordinate (six per iteration).
__host__ __device__ __forceinline__ float roundf_itm(float x) { return x + __builtin_copysignf(0.5f, x); }

// The blocks a 2x2x2 voxel cell with base block (bx0, by0, bz0) touches: slot c = (ox, oy, oz) in {0,1}^3 is needed iff the
// cell straddles (f*) in every axis where o = 1.  Blocks the caller already knows (cache, cache2 — which also remembers
// absent blocks) cost nothing; the others are looked up in ROUNDS of one bucket head per ray for all rays of the wave
// together: a round is one gather instruction and one wait for the whole wave, and the number of rounds is the largest
// number of unknown blocks any ray has (0 or 1 for nearly all of them).  bp[c] = block index or -1.
template <class Ops>
__host__ __device__ __forceinline__ void resolve_cell_blocks(const SceneP &s, const FrameP &p, int bx0, int by0, int bz0, bool fx, bool fy,
                                                             bool fz, const VoxCache &cache, VoxCache &cache2, int (&bp)[8]) {
  uint32_t need = 0;  // bit c: slot c has to be looked up
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    bp[c] = -1;
    const bool needed = (!(c & 1) || fx) && (!(c & 2) || fy) && (!(c & 4) || fz);
    const int bx = bx0 + (c & 1), by = by0 + ((c >> 1) & 1), bz = bz0 + (c >> 2);
    if (needed) {
      if (bx == cache.bx && by == cache.by && bz == cache.bz) bp[c] = cache.ptr;
      else if (bx == cache2.bx && by == cache2.by && bz == cache2.bz) bp[c] = cache2.ptr;
      else need |= 1u << c;
    }
  }
  while (Ops::any(need != 0)) {
    if (need != 0) {
      const int c = __builtin_ctz(need);
      need &= need - 1;
      const int bx = bx0 + (c & 1), by = by0 + ((c >> 1) & 1), bz = bz0 + (c >> 2);
      int4 raw = *reinterpret_cast<const int4 *>(s.table + hash_index(bx, by, bz, p.hashMask));
      int found = -1;
      while (true) {  // ITMRepresentationAccess.h findVoxel
        const int hx = (short)(raw.x & 0xffff), hy = (short)((uint32_t)raw.x >> 16), hz = (short)(raw.y & 0xffff);
        if (hx == bx && hy == by && hz == bz && raw.w >= 0) { found = raw.w; break; }
        if (raw.z < 1) break;
        raw = *reinterpret_cast<const int4 *>(s.table + (uint32_t)(p.noBuckets + raw.z - 1));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) bp[k] = (k == c) ? found : bp[k];
      cache2.bx = bx; cache2.by = by; cache2.bz = bz; cache2.ptr = found;
    }
  }
}

// returns the trilinear combination BEFORE SDF_valueToFloat (the division by 32767)
//
// readFromSDF_float_interpolated reads the 8 corners of the cell one after the other — lookup, voxel, lookup, voxel ...: up to
// 16 dependent round trips.  A lookup is a pure function of the table and a voxel read of the block array, so HOW the corners
// are fetched cannot change a value; the combination at the end is the reference's expression order.
//
// Round 3 (found in the ISA, confirmed by SQ counters: the kernel's duration = VMEM instructions per wave x ~1.1 us, i.e.
// every load of a wave is waited for before the next is issued): rounds 1-2 had three separate code paths here — cell inside
// one block (2/3 of the samples), straddling two (29 %), four or eight (4 %) — which a wave with rays in all three executes ONE
// AFTER THE OTHER, each with its own lookups and its own waits (a band iteration cost ~10 serialised round trips, with the
// corner loads in their own `if (ptr >= 0)` blocks ~25).  Now every ray of the wave goes through the SAME two phases:
//   1. resolve the blocks the cell touches (resolve_cell_blocks above: rounds of one bucket head per ray, all rays together);
//   2. all corner loads of all rays issued back to back, unconditionally (a missing block reads block 0 and the value is
//      replaced afterwards), ONE wait.
template <class Ops>
__host__ __device__ __forceinline__ float read_sdf_interpolated_raw(const SceneP &s, const FrameP &p, float x, float y, float z,
                                                                    VoxCache &cache, VoxCache &cache2) {
  const int ix = Ops::f2i(Ops::floor(x)), iy = Ops::f2i(Ops::floor(y)), iz = Ops::f2i(Ops::floor(z));
  const float cx = x - (float)ix, cy = y - (float)iy, cz = z - (float)iz;
  const bool fx = (ix & 7) == 7, fy = (iy & 7) == 7, fz = (iz & 7) == 7;
  const int bx0 = ix >> 3, by0 = iy >> 3, bz0 = iz >> 3;
  int bp[8];  // block index per slot (-1: no such block); only the needed slots are meaningful
  resolve_cell_blocks<Ops>(s, p, bx0, by0, bz0, fx, fy, fz, cache, cache2, bp);
  // corners: the two corners of an x-pair are neighbouring shorts of one block's sdf plane, so FOUR (possibly 2-byte aligned)
  // dword loads fetch the 8 corners (half the gather instructions of 8 short loads); a ray that straddles in x takes its four
  // +x corners from the x-neighbour blocks with four short loads more.  All loads are issued before the first is used.
  const uint8_t *vb = s.vba + kOffSdf;
  uint32_t w[4];
  short hi[4] = {0, 0, 0, 0};
  int pLo[4], pHi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dy = k & 1, dz = k >> 1;
    const int slotYZ = ((fy && dy) ? 2 : 0) | ((fz && dz) ? 4 : 0);
    pLo[k] = slotYZ == 0 ? bp[0] : (slotYZ == 2 ? bp[2] : (slotYZ == 4 ? bp[4] : bp[6]));
    pHi[k] = slotYZ == 0 ? bp[1] : (slotYZ == 2 ? bp[3] : (slotYZ == 4 ? bp[5] : bp[7]));
    const int linYZ = (((iy + dy) & 7) << 3) + (((iz + dz) & 7) << 6);
    w[k] = load_pair(reinterpret_cast<const short *>(vb + (size_t)(pLo[k] >= 0 ? pLo[k] : 0) * kBlockBytes) + ((ix & 7) + linYZ));
  }
  if (fx) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int dy = k & 1, dz = k >> 1;
      const int linYZ = (((iy + dy) & 7) << 3) + (((iz + dz) & 7) << 6);  // x = 0 of the neighbour block
      hi[k] = *reinterpret_cast<const short *>(vb + (size_t)(pHi[k] >= 0 ? pHi[k] : 0) * kBlockBytes + linYZ * 2);
    }
  }
  float v[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k * 2] = pLo[k] >= 0 ? (float)(short)(w[k] & 0xffffu) : 32767.0f;
    const float inBlock = pLo[k] >= 0 ? (float)(short)(w[k] >> 16) : 32767.0f;
    const float neighbour = pHi[k] >= 0 ? (float)hi[k] : 32767.0f;
    v[k * 2 + 1] = fx ? neighbour : inBlock;
  }
  float res1 = (1.0f - cx) * v[0] + cx * v[1];
  res1 = (1.0f - cy) * res1 + cy * ((1.0f - cx) * v[2] + cx * v[3]);
  float res2 = (1.0f - cx) * v[4] + cx * v[5];
  res2 = (1.0f - cy) * res2 + cy * ((1.0f - cx) * v[6] + cx * v[7]);
  return (1.0f - cz) * res1 + cz * res2;
}
// --------------------------------------------------------- K6: expected depths

__global__ __launch_bounds__(256) void k_minmax_init(float2 *__restrict__ minmax, int n, const int32_t *__restrict__ ctr,
                                                     int skipIfZeroIdx) {
  if (skipIfZeroIdx >= 0 && ctr[skipIfZeroIdx] <= 0) return;  // Prepare() is skipped without visible blocks
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) minmax[i] = make_float2(kFarAway, kVeryClose);
}

// ITMVisualisationEngine.h ProjectSingleBlock on the compact ceil(W/8) x ceil(H/8) image
template <class Ops>
__host__ __device__ __forceinline__ bool project_single_block(const short pos[3], const FrameP &p, int imgW, int imgH, int2 &ul,
                                                              int2 &lr, float2 &zr) {
  ul = make_int2(imgW, imgH);
  lr = make_int2(-1, -1);
  zr = make_float2(kFarAway, kVeryClose);
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    short tx = (short)(pos[0] + ((corner & 1) ? 1 : 0));
    short ty = (short)(pos[1] + ((corner & 2) ? 1 : 0));
    short tz = (short)(pos[2] + ((corner & 4) ? 1 : 0));
    float3 q = mat_mul3(p.M, (float)tx * (float)kBlockSize * p.voxelSize, (float)ty * (float)kBlockSize * p.voxelSize,
                        (float)tz * (float)kBlockSize * p.voxelSize, 1.0f);
    if (q.z < 1e-6f) continue;
    // q.z >= 1e-6: tame divisor, the two divisions share the refined reciprocal (dsr_device.h)
    const float yz = Ops::rcp(q.z);
    float px = (Ops::div(p.proj.x * q.x, q.z, yz) + p.proj.z) / (float)kMinmaxSubsample;
    float py = (Ops::div(p.proj.y * q.y, q.z, yz) + p.proj.w) / (float)kMinmaxSubsample;
    if ((float)ul.x > Ops::floor(px)) ul.x = Ops::f2i(Ops::floor(px));
    if ((float)lr.x < Ops::ceil(px)) lr.x = Ops::f2i(Ops::ceil(px));
    if ((float)ul.y > Ops::floor(py)) ul.y = Ops::f2i(Ops::floor(py));
    if ((float)lr.y < Ops::ceil(py)) lr.y = Ops::f2i(Ops::ceil(py));
    if (zr.x > q.z) zr.x = q.z;
    if (zr.y < q.z) zr.y = q.z;
  }
  if (ul.x < 0) ul.x = 0;
  if (ul.y < 0) ul.y = 0;
  if (lr.x >= imgW) lr.x = imgW - 1;
  if (lr.y >= imgH) lr.y = imgH - 1;
  if (ul.x > lr.x) return false;
  if (ul.y > lr.y) return false;
  if (zr.x < kVeryClose) zr.x = kVeryClose;
  if (zr.y < kVeryClose) return false;
  return true;
}

// One thread per visible block: project, then min/max into the range image.  All values are
// positive floats, so integer atomicMin/atomicMax on the bit patterns order them correctly and
// the result is independent of the order of arrival.
__global__ __launch_bounds__(256) void k_expected_depth(FrameP p, SceneP s, const int4 *__restrict__ visBlocks,
                                                        int ctrIdx, int2 *__restrict__ minmax) {
  const int n = s.ctr[ctrIdx];
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&s.work[WORK_V_EXPECTED], (unsigned long long)n);
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample, mh = (p.H + kMinmaxSubsample - 1) / kMinmaxSubsample;
  // min only decreases / max only increases, so a (possibly stale) plain read can only
  // over-estimate the need for an atomic: skipping on it is safe and removes almost all of the
  // same-address atomic traffic (a cell settles after O(log n) updates).
  auto update_cell = [&](int x, int y, int zmin, int zmax) {
    int2 *px = minmax + x + y * mw;
    // (agent-scope sc1 loads for this filter were measured: 443 us vs 216 us with plain loads)
    const int2 cur = *px;
    if (zmin < cur.x) atomicMin(&px->x, zmin);
    if (zmax > cur.y) atomicMax(&px->y, zmax);
  };
  const int lane = threadIdx.x & 63;
  const int stride = gridDim.x * blockDim.x;
  // wave-uniform trip count (the cooperative part needs every lane of the wave)
  for (int base = blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n; base += stride) {
    const int i = base + lane;
    bool valid = false;
    int2 ul = make_int2(0, 0), lr = make_int2(-1, -1);
    float2 zr = make_float2(0.f, 0.f);
    if (i < n) {
      const dsr_hash_entry he = entry_of_record(visBlocks[i]);  // the visible-block stream: 16 B per lane, coalesced
      if (he.ptr >= 0) valid = project_single_block<DeviceOps>(he.pos, p, mw, mh, ul, lr, zr);
    }
    const int zmin = __float_as_int(zr.x), zmax = __float_as_int(zr.y);
    const int bw = lr.x - ul.x + 1, bh = lr.y - ul.y + 1;
    // small boxes (the 5 mm regime: 1-4 cells): the owning lane fills them;
    // large boxes (coarse voxels / near blocks: up to thousands of cells): the whole wave does
    const bool big = valid && bw * bh > 16;
    if (valid && !big)
      for (int y = ul.y; y <= lr.y; ++y)
        for (int x = ul.x; x <= lr.x; ++x) update_cell(x, y, zmin, zmax);
    unsigned long long m = __ballot(big);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int x0 = __shfl(ul.x, src), y0 = __shfl(ul.y, src), w = __shfl(bw, src), cells = w * __shfl(bh, src);
      const int zmn = __shfl(zmin, src), zmx = __shfl(zmax, src);
      for (int c = lane; c < cells; c += 64) update_cell(x0 + c % w, y0 + c / w, zmn, zmx);
    }
  }
}

// K6, LDS-privatised form (used when the range image fits: ceil(W/8)*ceil(H/8)*8 B <= 64 KiB,
// i.e. 58.7 KB at 1242x375).  Each of a few workgroups keeps a private copy of the WHOLE range
// image in LDS, folds its share of the visible blocks into it with ds_min/ds_max (no global
// same-address atomic traffic at all during the fold), then flushes the cells it touched with
// filtered global atomics.  min/max are order independent, so the image is identical.
__global__ __launch_bounds__(1024) void k_expected_depth_lds(FrameP p, SceneP s, const int4 *__restrict__ visBlocks,
                                                             int ctrIdx, int2 *__restrict__ minmax) {
  extern __shared__ int2 cellsLds[];
  const int n = s.ctr[ctrIdx];
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&s.work[WORK_V_EXPECTED], (unsigned long long)n);
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample, mh = (p.H + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const int nCells = mw * mh;
  const int farBits = __float_as_int(kFarAway), closeBits = __float_as_int(kVeryClose);
  if ((int)(blockIdx.x * blockDim.x) >= n) return;  // nothing for this workgroup
  for (int c = threadIdx.x; c < nCells; c += blockDim.x) cellsLds[c] = make_int2(farBits, closeBits);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int stride = gridDim.x * blockDim.x;
  for (int base = blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n; base += stride) {
    const int i = base + lane;
    bool valid = false;
    int2 ul = make_int2(0, 0), lr = make_int2(-1, -1);
    float2 zr = make_float2(0.f, 0.f);
    if (i < n) {
      const dsr_hash_entry he = entry_of_record(visBlocks[i]);  // the visible-block stream: 16 B per lane, coalesced
      if (he.ptr >= 0) valid = project_single_block<DeviceOps>(he.pos, p, mw, mh, ul, lr, zr);
    }
    const int zmin = __float_as_int(zr.x), zmax = __float_as_int(zr.y);
    const int bw = lr.x - ul.x + 1, bh = lr.y - ul.y + 1;
    // boxes of up to 144 cells are filled by the owning lane (fire-and-forget LDS atomics); the wave-cooperative path below
    // costs a round of seven cross-lane broadcasts per box, SEQUENTIAL within the wave — with the 16-cell threshold this
    // kernel shared with the global-atomics one, coarse voxels (5 cm: a block covers ~5 x 5 cells at 10 m) sent nearly
    // every block down that path (32 us for ~5 k blocks at the reference's own operating point)
    const bool big = valid && bw * bh > 144;
    auto fold = [&](int idx, int zmn, int zmx) {  // (reading the cell first to skip atomics that cannot change it: slower)
      atomicMin(&cellsLds[idx].x, zmn);
      atomicMax(&cellsLds[idx].y, zmx);
    };
    if (valid && !big)
      for (int y = ul.y; y <= lr.y; ++y)
        for (int x = ul.x; x <= lr.x; ++x) fold(x + y * mw, zmin, zmax);
    unsigned long long m = __ballot(big);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int x0 = __shfl(ul.x, src), y0 = __shfl(ul.y, src), w = __shfl(bw, src), cells = w * __shfl(bh, src);
      const int zmn = __shfl(zmin, src), zmx = __shfl(zmax, src);
      for (int c = lane; c < cells; c += 64) fold((x0 + c % w) + (y0 + c / w) * mw, zmn, zmx);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nCells; c += blockDim.x) {
    const int2 v = cellsLds[c];
    if (v.x == farBits && v.y == closeBits) continue;  // untouched by this workgroup
    const int2 cur = minmax[c];
    if (v.x < cur.x) atomicMin(&minmax[c].x, v.x);
    if (v.y > cur.y) atomicMax(&minmax[c].y, v.y);
  }
}

// The boxes of a wave's (up to) 64 projected blocks folded into a range image in LDS; call with all lanes of the wave.
// An instance is seen from close: a 0.28 m block at 8 m covers ~5 x 5 cells, so with the global-atomics kernel's threshold
// (16 cells) nearly every block took the wave-cooperative path below — 64 SEQUENTIAL rounds of seven cross-lane
// broadcasts per wave (measured: ~28 us for 539 blocks).  Here the owning lane fills boxes of up to 144 cells itself
// (fire-and-forget LDS atomics, ~10 cycles a cell) and only a block right in front of the camera is shared.
// (Reading the cell first to skip atomics that cannot change it was measured too: slower, 35 us — the read's latency.
//  Eight lanes per block, every eighth cell of the box each — 128 blocks per pass of the workgroup — was measured in round 4:
//  28 us per launch against 19.6, the passes and the index arithmetic cost more than the short boxes save.)
__device__ __forceinline__ void fold_wave_boxes(int2 *cellsLds, int mw, bool valid, int2 ul, int2 lr, float2 zr, int lane) {
  const int zmin = __float_as_int(zr.x), zmax = __float_as_int(zr.y);
  const int bw = lr.x - ul.x + 1, bh = lr.y - ul.y + 1;
  const bool big = valid && bw * bh > 144;
  auto fold = [&](int idx, int zmn, int zmx) {
    atomicMin(&cellsLds[idx].x, zmn);
    atomicMax(&cellsLds[idx].y, zmx);
  };
  if (valid && !big)
    for (int y = ul.y; y <= lr.y; ++y)
      for (int x = ul.x; x <= lr.x; ++x) fold(x + y * mw, zmin, zmax);
  unsigned long long m = __ballot(big);
  while (m) {  // large boxes (coarse voxels seen from close: hundreds of cells) are filled by the whole wave
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int x0 = __shfl(ul.x, src), y0 = __shfl(ul.y, src), w = __shfl(bw, src), cells = w * __shfl(bh, src);
    const int zmn = __shfl(zmin, src), zmx = __shfl(zmax, src);
    for (int c = lane; c < cells; c += 64) fold((x0 + c % w) + (y0 + c / w) * mw, zmn, zmx);
  }
}

// ---- the BOX of the range image (instance-sized volumes, round 6) -----------------------------------------------------------
// An instance volume covers a few per cent of the frame: its range image is empty outside the bounding box of a few hundred
// projected blocks, and a ray through an empty cell is a miss before its first step.  The full-frame kernels behind the range
// image — raycast, ICP maps, the free-view raycast + shading — spent their time writing that miss into 465 k pixels per volume
// (k_batch_icp_maps: 194 MB per launch for eight volumes; profiles/r05z_batch_kernel_stats.json).  The one-workgroup kernel that
// builds the image (k_small.h, k_expected_depth_one) now also keeps, per render state, a small device record `rb`:
//   CUR    box of the non-empty cells of the current image (cell units, end exclusive; empty: x0 >= x1)
//   DIRTY  the cells whose pixels may hold anything but a miss, or must be recomputed: what the next raycast has to cover
//   RAN    a raycast has consumed DIRTY since it was last extended
//   LAST   the DIRTY box of the last raycast that ran, EVER: one has run, POSE: its FrameP
// and the pixel kernels skip every 16x16 tile outside DIRTY.  Invariant: outside DIRTY every pixel of raycastResult has w == 0,
// the ICP maps hold their miss constants and the images are 0 — which is all their consumers ever look at (icp_pixel,
// render_pixel, render_depth test w first).  The xyz of a miss (the ray's far-plane start point: pose dependent, never read by
// the path) goes stale outside DIRTY; dsr_dump_render_state — the parity tests' view of the buffer — completes it with
// k_raycast_fill_outside from LAST + POSE, so a dump equals the serial engine's buffer bit for bit as before.
// After creation / a reset DIRTY is the whole image: the first raycast is a full-frame one (a reset keeps LAST / EVER / POSE: the
// buffers still hold the previous scene's renders until then).
constexpr int RB_CUR = 0, RB_DIRTY = 4, RB_RAN = 8, RB_LAST = 9, RB_EVER = 13, RB_POSE = 16, RB_WORDS = 128;
static_assert(RB_POSE * 4 + sizeof(FrameP) <= RB_WORDS * 4, "the pose record must fit");
static_assert(sizeof(FrameP) % 4 == 0, "FrameP is copied word by word");

// keepLast (a ResetScene, not the creation): LAST / EVER / POSE stay — the render buffers keep what the previous scene's raycasts
// left in them (as the serial engine's do), and until the next raycast runs a dump still needs that record to complete them
__global__ void k_raybox_reset(int32_t *rb, int mw, int mh, int keepLast) {
  if (threadIdx.x < (keepLast ? RB_LAST : RB_WORDS)) rb[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) { rb[RB_DIRTY + 2] = mw; rb[RB_DIRTY + 3] = mh; }
}
static_assert(RB_CUR < RB_LAST && RB_DIRTY < RB_LAST && RB_RAN < RB_LAST && RB_EVER > RB_LAST && RB_POSE > RB_LAST, "k_raybox_reset's split");

// All threads of the ONE workgroup that holds the finished range image in LDS: store it, find the box of its non-empty cells,
// update the record.  boxLds: 4 ints of LDS set to {INT_MAX, INT_MAX, -1, -1} before the barrier that precedes this call.
__device__ __forceinline__ void store_range_image(const int2 *cells, int2 *__restrict__ minmax, int nCells, int mw,
                                                  int32_t *__restrict__ rb, int *boxLds) {
  const int farBits = __float_as_int(kFarAway);
  int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
  for (int c = threadIdx.x; c < nCells; c += blockDim.x) {
    const int2 v = cells[c];
    minmax[c] = v;
    if (v.x != farBits) {  // a block was folded into this cell (every z is below FAR_AWAY)
      const int cy = c / mw, cx = c - cy * mw;
      x0 = cx < x0 ? cx : x0; y0 = cy < y0 ? cy : y0; x1 = cx > x1 ? cx : x1; y1 = cy > y1 ? cy : y1;
    }
  }
  if (!rb) return;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int a = __shfl_xor(x0, d), b = __shfl_xor(y0, d), c = __shfl_xor(x1, d), e = __shfl_xor(y1, d);
    x0 = a < x0 ? a : x0; y0 = b < y0 ? b : y0; x1 = c > x1 ? c : x1; y1 = e > y1 ? e : y1;
  }
  if ((threadIdx.x & 63) == 0 && x1 >= 0) {
    atomicMin(&boxLds[0], x0); atomicMin(&boxLds[1], y0); atomicMax(&boxLds[2], x1); atomicMax(&boxLds[3], y1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // the new box; empty: all zero
    if (boxLds[2] >= 0) { n0 = boxLds[0]; n1 = boxLds[1]; n2 = boxLds[2] + 1; n3 = boxLds[3] + 1; }
    // what may hold hits: the box of the image the last raycast saw (if one ran since), else everything still pending
    const int b = rb[RB_RAN] ? RB_CUR : RB_DIRTY;
    int d0 = rb[b], d1 = rb[b + 1], d2 = rb[b + 2], d3 = rb[b + 3];
    if (d0 >= d2 || d1 >= d3) { d0 = n0; d1 = n1; d2 = n2; d3 = n3; }
    else if (n0 < n2 && n1 < n3) { d0 = n0 < d0 ? n0 : d0; d1 = n1 < d1 ? n1 : d1; d2 = n2 > d2 ? n2 : d2; d3 = n3 > d3 ? n3 : d3; }
    rb[RB_DIRTY] = d0; rb[RB_DIRTY + 1] = d1; rb[RB_DIRTY + 2] = d2; rb[RB_DIRTY + 3] = d3;
    rb[RB_CUR] = n0; rb[RB_CUR + 1] = n1; rb[RB_CUR + 2] = n2; rb[RB_CUR + 3] = n3;
    rb[RB_RAN] = 0;
  }
}

// does the 16x16 pixel tile (wgx, wgy) — cells [2 wgx, 2 wgx + 2) x [2 wgy, 2 wgy + 2) — touch the box at rb[which]?
__device__ __forceinline__ bool raybox_tile(const int32_t *__restrict__ rb, int which, int wgx, int wgy) {
  const int x0 = rb[which], y0 = rb[which + 1], x1 = rb[which + 2], y1 = rb[which + 3];
  return 2 * wgx < x1 && 2 * wgx + 2 > x0 && 2 * wgy < y1 && 2 * wgy + 2 > y0;
}
// the raycast that runs records it (one thread of its first workgroup; nobody reads these words before the next launch)
__device__ __forceinline__ void raybox_mark_ran(int32_t *__restrict__ rb, const FrameP &p) {
  rb[RB_RAN] = 1; rb[RB_EVER] = 1;
  rb[RB_LAST] = rb[RB_DIRTY]; rb[RB_LAST + 1] = rb[RB_DIRTY + 1]; rb[RB_LAST + 2] = rb[RB_DIRTY + 2]; rb[RB_LAST + 3] = rb[RB_DIRTY + 3];
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&p);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(FrameP) / 4); ++i) reinterpret_cast<uint32_t *>(rb + RB_POSE)[i] = src[i];
}

// K6 for SMALL volumes (an instance volume: a few hundred visible blocks): ONE workgroup owns the whole range image —
// initialise it in LDS, fold every visible block, store every cell.  No global atomics, no separate initialisation
// launch (k_minmax_init), no 128 workgroups that each clear and flush a 58 KB image for nothing: on an instance volume
// K6 was 2 launches and ~25 us of a ~300 us frame, twice per frame (tracking view + preview).  min / max are order
// independent: the image is the one the other two kernels give.  `keepIfEmpty`: the live view's Prepare() is skipped
// without visible blocks (the image keeps its previous contents).
__global__ __launch_bounds__(1024) void k_expected_depth_one(FrameP p, SceneP s, const int4 *__restrict__ visBlocks, int ctrIdx,
                                                             int2 *__restrict__ minmax, int keepIfEmpty, int32_t *__restrict__ rb) {
  extern __shared__ int2 cellsLds[];
  __shared__ int boxLds[4];
  const int n = s.ctr[ctrIdx];
  if (n <= 0 && keepIfEmpty) return;
  if (threadIdx.x == 0) {
    atomicAdd(&s.work[WORK_V_EXPECTED], (unsigned long long)n);
    boxLds[0] = boxLds[1] = 0x7fffffff; boxLds[2] = boxLds[3] = -1;
  }
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample, mh = (p.H + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const int nCells = mw * mh;
  const int farBits = __float_as_int(kFarAway), closeBits = __float_as_int(kVeryClose);
  for (int c = threadIdx.x; c < nCells; c += blockDim.x) cellsLds[c] = make_int2(farBits, closeBits);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int base = threadIdx.x & ~63; base < n; base += blockDim.x) {  // wave-uniform trip count
    const int i = base + lane;
    bool valid = false;
    int2 ul = make_int2(0, 0), lr = make_int2(-1, -1);
    float2 zr = make_float2(0.f, 0.f);
    if (i < n) {
      const dsr_hash_entry he = entry_of_record(visBlocks[i]);
      if (he.ptr >= 0) valid = project_single_block<DeviceOps>(he.pos, p, mw, mh, ul, lr, zr);
    }
    fold_wave_boxes(cellsLds, mw, valid, ul, lr, zr, lane);
  }
  __syncthreads();
  store_range_image(cellsLds, minmax, nCells, mw, rb, boxLds);
}

// ----------------------------------------------------------------- K7: raycast

// -DDSR_RAYCAST_STATS (tools/raycast_wave_stats.py, never the product build): per-wave clocks and stage counts of the march
#ifdef DSR_RAYCAST_STATS
__device__ unsigned int *g_rcStats;  // 12 words per wave
struct RcStats { unsigned nIter = 0, nLook = 0, nVox = 0, nBand = 0, wLook = 0, wBand = 0, wHead = 0, wChain = 0; };
#define RC_STAT(...) __VA_ARGS__
#else
#define RC_STAT(...)
#endif

// ITMVisualisationEngine.h castRay, in three parts — the ray's segment, the march, the refinement of a hit; cast_ray<Ops> composes
// them into the reference's function.  (Templates over Ops only so that tests/ can run these very functions on the CPU.)
struct RayState {
  float rx, ry, rz;       // current sample position, voxel units
  float dx, dy, dz;       // unit direction
  float totalLength, totalLengthMax;
};

template <class Ops>
__host__ __device__ __forceinline__ void ray_setup(const FrameP &p, int x, int y, float2 mm, RayState &r) {
  const float oneOverVoxelSize = 1.0f / p.voxelSize;
  const float invFx = 1.0f / p.proj.x, invFy = 1.0f / p.proj.y;
  float cz = mm.x;
  float cx = cz * (((float)x - p.proj.z) * invFx);
  float cy = cz * (((float)y - p.proj.w) * invFy);
  r.totalLength = Ops::sqrt(cx * cx + cy * cy + cz * cz) * oneOverVoxelSize;
  float3 t = mat_mul3(p.invM, cx, cy, cz, 1.0f);
  float sx = t.x * oneOverVoxelSize, sy = t.y * oneOverVoxelSize, sz = t.z * oneOverVoxelSize;
  cz = mm.y;
  cx = cz * (((float)x - p.proj.z) * invFx);
  cy = cz * (((float)y - p.proj.w) * invFy);
  r.totalLengthMax = Ops::sqrt(cx * cx + cy * cy + cz * cz) * oneOverVoxelSize;
  t = mat_mul3(p.invM, cx, cy, cz, 1.0f);
  float ex = t.x * oneOverVoxelSize, ey = t.y * oneOverVoxelSize, ez = t.z * oneOverVoxelSize;
  float dx = ex - sx, dy = ey - sy, dz = ez - sz;
  float direction_norm = 1.0f / Ops::sqrt(dx * dx + dy * dy + dz * dz);
  r.dx = dx * direction_norm; r.dy = dy * direction_norm; r.dz = dz * direction_norm;
  r.rx = sx; r.ry = sy; r.rz = sz;
}

// The while loop of castRay.  Returns the sdf value of the last sample (1.0 before the first).
template <class Ops>
__host__ __device__ __forceinline__ float ray_march(const FrameP &p, const SceneP &s, RayState &r, VoxCache &cache, VoxCache &cache2
                                                    RC_STAT(, RcStats &st)) {
  const float stepScale = p.mu * (1.0f / p.voxelSize);
  const float dx = r.dx, dy = r.dy, dz = r.dz;
  float rx = r.rx, ry = r.ry, rz = r.rz, totalLength = r.totalLength;
  const float totalLengthMax = r.totalLengthMax;
  float sdfValue = 1.0f, stepLength;
  bool hash_found;
  uint32_t pfIdx = 0xffffffffu;  // table index of the prefetched entry
  int4 pfRaw = make_int4(0, 0, 0, -2);
  while (totalLength < totalLengthMax) {
    // (sample_sdf_march — one lookup + the 8 corner loads for every step — was measured: 937 us vs
    //  666 us.  The march is bound by gather-request throughput, not by the number of dependent
    //  phases, so the single uninterpolated load per far step stays.)
    {
      // readFromSDF_float_uninterpolated with a one-step look-ahead on the hash table: the bucket
      // head the NEXT sample will need (guessing the usual step: 8 voxels after a miss, mu/voxel
      // when the block is in front of the surface, sdf = 1) is requested together with this
      // sample's voxel, so a correct guess turns lookup -> voxel into one round trip per step.
      // A wrong guess costs one unused 16-byte read; values are never affected.
      const int vx = Ops::f2i(roundf_itm(rx)), vy = Ops::f2i(roundf_itm(ry)), vz = Ops::f2i(roundf_itm(rz));
      const int bx = vx >> 3, by = vy >> 3, bz = vz >> 3;
      int ptr;
      RC_STAT(++st.nIter; bool didHead = false, didChain = false;)
      if (bx == cache.bx && by == cache.by && bz == cache.bz) ptr = cache.ptr;
      else {
        uint32_t h = hash_index(bx, by, bz, p.hashMask);
        RC_STAT(if (h != pfIdx) { ++st.nLook; didHead = true; })
        // (`(h == pfIdx) ? pfRaw : *entry` compiles to four predicated one-word loads; and the cache is written after the walk,
        //  not inside it, where it costs the loop ten register copies per entry)
        int4 raw = pfRaw;
        if (h != pfIdx) raw = *reinterpret_cast<const int4 *>(s.table + h);
        ptr = -1;
        while (true) {
          const int hx = (short)(raw.x & 0xffff), hy = (short)((uint32_t)raw.x >> 16), hz = (short)(raw.y & 0xffff);
          if (hx == bx && hy == by && hz == bz && raw.w >= 0) { ptr = raw.w; break; }
          if (raw.z < 1) break;
          h = (uint32_t)(p.noBuckets + raw.z - 1);
          raw = *reinterpret_cast<const int4 *>(s.table + h);
          RC_STAT(++st.nLook; didChain = true;)
        }
        if (ptr >= 0) { cache.bx = bx; cache.by = by; cache.bz = bz; cache.ptr = ptr; }
      }
      hash_found = ptr >= 0;
      RC_STAT(st.wLook += __any(didHead || didChain) ? 1u : 0u; st.wHead += __any(didHead) ? 1u : 0u; st.wChain += __any(didChain) ? 1u : 0u;
              st.nVox += hash_found ? 1u : 0u;)
      {
        const float g = hash_found ? stepScale : (float)kBlockSize;
        const int nx = Ops::f2i(roundf_itm(rx + g * dx)) >> 3, ny = Ops::f2i(roundf_itm(ry + g * dy)) >> 3,
                  nz = Ops::f2i(roundf_itm(rz + g * dz)) >> 3;
        if (nx != bx || ny != by || nz != bz) {
          pfIdx = hash_index(nx, ny, nz, p.hashMask);
          pfRaw = *reinterpret_cast<const int4 *>(s.table + pfIdx);
        }
        // (measured and rejected: a second look-ahead slot during runs of misses, 580 us vs 515 us;
        //  four bucket heads at once after 8 misses in a row, 631 us; a fire-and-forget prefetch
        //  of the head 3..10 miss steps ahead into an LDS sink, 560 us; parking lanes that need a
        //  trilinear sample until 1..48 of them can take it together, 607..896 us; giving each XCD a
        //  band of tile columns (workgroup i -> XCD i % 8), 533 vs 512 us; inside runs of misses
        //  prefetching the word of a 1 MB bucket-occupancy bitmap instead of the 16 B entry (a clear
        //  bit is a sure miss: no table read at all), 634 us; workgroups of 64 / 512 / 1024 threads
        //  (1 / 8 / 16 neighbouring tiles) instead of 256: 601 / 536 / 519 vs 488 us; round 2: a per-wave
        //  direct-mapped LDS cache of lookup OUTCOMES (found block / no block) shared by the 64 rays of a
        //  tile, 572 vs 435 us — also when compiled for 7 or 6 waves per SIMD, which by themselves change
        //  nothing (439 / 438 us); round 3: issuing the four corner-pair loads of the trilinear cell TOGETHER with the
        //  step's own voxel load whenever the previous step was inside the interpolation band (one round trip per band
        //  step instead of two, no extra loads when the guess holds), 441-443 vs 429 us, and requesting the NEXT step's
        //  voxel one iteration ahead in saturated space, 518 us (7 spilled registers at the 64-VGPR limit), both 543 us
        //  (profiles/r03e_raycast_speculation_variants.log; all bit-exact).
        //  Round 3, after measuring what a launch is made of (tools/raycast_wave_stats.py, profiles/r03_raycast_wave_stats*.json:
        //  all 7.3 k waves are resident at once; a wave runs 105 dependent memory stages where its neediest single ray needs 35;
        //  while all waves run, a stage costs its round trip plus the wave's share of a half-busy SIMD (~160 instructions),
        //  and the launch's last third runs with a few per cent of the waves, the ones with 200+ iterations of absent blocks), all
        //  verified bit-exact on the CPU first (tests/test_device_functions_host.py) and then on the GPU:
        //   * an occupancy BITMAP per 4x4x4-block cell in front of the table (clear bit = no such block, no read at all):
        //     -21 % bytes fetched, one more dependent round for every block that exists: 536 vs 461 us;
        //   * rays advancing independently through a per-ray state machine, one read per ray and round (a wave needs as many
        //     rounds as its neediest ray has reads) — and ~4x the instructions of this loop: 798 vs 461 us
        //     (profiles/r03_raycast_rounds_variant.h.txt, r03_raycast_rounds_ab.log);
        //   * this table walk as wave-wide rounds (`while (__any(...))`, every ray issues the entry it needs next from ONE
        //     load instruction per round): 500-507 vs 405-411 us (profiles/r03_raycast_lookup_rounds_ab.log);
        //   * a direct-mapped BLOCK MAP (position -> ptr, one read instead of head + chain, conflicts fall back to the table)
        //     maintained by the allocation / GC / swap kernels: chain rounds 18 -> 6 per wave, but every conflict is one
        //     more round, and a wave has 64 rays: 1 % (profiles/r03_block_map_variant.diff, r03_block_map_*_ab.log);
        //   * on top of it, probing runs of absent blocks four blocks at a time from iteration N of a wave on: 717-835 vs
        //     650 us with the code present and never taken, ~500 without it (profiles/r03_raycast_probe_variant.h.txt).
        //  Every variant that adds requests, instructions or iterations loses.)
      }
      float raw16 = 32767.0f;
      if (hash_found) {
        const int lin = (vx & 7) + ((vy & 7) << 3) + ((vz & 7) << 6);
        raw16 = (float)*reinterpret_cast<const short *>(s.vba + (size_t)ptr * kBlockBytes + kOffSdf + lin * 2);
      }
      sdfValue = sdf_to_float_short(raw16);
    }
    RC_STAT(const bool inBand = hash_found && (sdfValue <= 0.1f) && (sdfValue >= -0.5f); st.nBand += inBand ? 1u : 0u; st.wBand += __any(inBand) ? 1u : 0u;)
    if (!hash_found) {
      stepLength = (float)kBlockSize;
    } else {
      // (experiment, round 3: without this interpolated read the kernel takes 331 instead of 404 us — the band phases are 18 %
      //  of it; the rest is the plain march: one or two dependent gathers per step at ~3.6 TB/s of scattered 128-byte fetches)
      if ((sdfValue <= 0.1f) && (sdfValue >= -0.5f)) sdfValue = sdf_to_float_short(read_sdf_interpolated_raw<Ops>(s, p, rx, ry, rz, cache, cache2));
      if (sdfValue <= 0.0f) break;
      float ss = sdfValue * stepScale;
      stepLength = (ss > 1.0f) ? ss : 1.0f;  // MAX(sdfValue * stepScale, 1.0f)
    }
    rx += stepLength * dx; ry += stepLength * dy; rz += stepLength * dz;
    totalLength += stepLength;
  }
  r.rx = rx; r.ry = ry; r.rz = rz; r.totalLength = totalLength;
  return sdfValue;
}

// the two refinement steps of a hit / the miss result
template <class Ops>
__host__ __device__ __forceinline__ float4 ray_finish(const FrameP &p, const SceneP &s, const RayState &r, float sdfValue, VoxCache &cache,
                                                      VoxCache &cache2) {
  const float stepScale = p.mu * (1.0f / p.voxelSize);
  float rx = r.rx, ry = r.ry, rz = r.rz;
  float4 out;
  if (sdfValue <= 0.0f) {
    float stepLength = sdfValue * stepScale;
    rx += stepLength * r.dx; ry += stepLength * r.dy; rz += stepLength * r.dz;
    sdfValue = sdf_to_float_short(read_sdf_interpolated_raw<Ops>(s, p, rx, ry, rz, cache, cache2));
    stepLength = sdfValue * stepScale;
    rx += stepLength * r.dx; ry += stepLength * r.dy; rz += stepLength * r.dz;
    out.w = 1.0f;
  } else out.w = 0.0f;
  out.x = rx; out.y = ry; out.z = rz;
  return out;
}

template <class Ops>
__host__ __device__ __forceinline__ float4 cast_ray(const FrameP &p, const SceneP &s, int x, int y, float2 mm RC_STAT(, RcStats &st)) {
  RayState r;
  ray_setup<Ops>(p, x, y, mm, r);
  VoxCache cache; cache_init(cache);
  VoxCache cache2; cache_init(cache2);  // neighbour block of two-block trilinear cells
  const float sdfValue = ray_march<Ops>(p, s, r, cache, cache2 RC_STAT(, st));
  return ray_finish<Ops>(p, s, r, sdfValue, cache, cache2);
}

// 8x8 pixel tile per wave (4 tiles per 256-thread workgroup, laid out 2x2 => 16x16 pixels)
// (Round 3, measured and dropped: mapping the workgroup ids that land on one XCD (id % 8) to 2x2 / 4x4 / 8x8 patches of
//  neighbouring workgroups so that rays sharing voxel and table lines share an L2: 427 / 413 / 443 vs 403 us,
//  profiles/r03w_raycast_xcd_patches.log — the working set of a patch is far beyond 4 MB either way.)
//
// Round 4 measured the tail of the launch taken OUT of the kernel (a wave leaves the march after K trips, the rays still under way go
// to a compact list and a second kernel resumes them with eight lanes per ray): bit-identical at every K and no faster at any
// (K = 64: 285 + 160 us against 444 in one kernel).  The code is archived as profiles/r05_pruned_raycast_split.diff with its logs
// (profiles/r04c_raycast_split_kernels_ab.log, r04b_raycast_split_tail_rays.log); DESIGN.md 6.3.
__global__ __launch_bounds__(256, 8) void k_raycast(FrameP p, SceneP s, int ctrIdx, const float2 *__restrict__ minmax,
                                                 float4 *__restrict__ raycastResult) {
  if (s.ctr[ctrIdx] <= 0 && ctrIdx == CTR_NO_VISIBLE_LIVE) return;  // Prepare() is skipped without visible blocks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wgx = blockIdx.x, wgy = blockIdx.y;
  const int x = wgx * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = wgy * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (x >= p.W || y >= p.H) return;
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const float2 mm = minmax[(x >> 3) + (y >> 3) * mw];
#ifdef DSR_RAYCAST_STATS
  const unsigned long long t0 = wall_clock64();
  RcStats st;
  raycastResult[x + y * p.W] = cast_ray<DeviceOps>(p, s, x, y, mm, st);
  const unsigned long long t1 = wall_clock64();
  // own stages of a ray, were the rays decoupled: one per table read, voxel read and band read
  unsigned own = st.nLook + st.nVox + st.nBand, iter = st.nIter, look = st.nLook, sumIter = st.nIter;
  for (int d = 1; d < 64; d <<= 1) {
    own = max(own, (unsigned)__shfl_xor((int)own, d)); iter = max(iter, (unsigned)__shfl_xor((int)iter, d));
    st.wLook = max(st.wLook, (unsigned)__shfl_xor((int)st.wLook, d)); st.wBand = max(st.wBand, (unsigned)__shfl_xor((int)st.wBand, d));
    st.wHead = max(st.wHead, (unsigned)__shfl_xor((int)st.wHead, d)); st.wChain = max(st.wChain, (unsigned)__shfl_xor((int)st.wChain, d));
    look = max(look, (unsigned)__shfl_xor((int)look, d)); sumIter += (unsigned)__shfl_xor((int)sumIter, d);
  }
  const unsigned nLanes = (unsigned)__popcll(__ballot(1));
  if (lane == 0) {
    unsigned int *o = g_rcStats + 12u * ((blockIdx.x + blockIdx.y * gridDim.x) * 4u + wave);
    o[0] = (unsigned)t0; o[1] = (unsigned)(t0 >> 32); o[2] = (unsigned)t1; o[3] = (unsigned)(t1 >> 32);
    o[4] = iter; o[5] = st.wLook; o[6] = own; o[7] = look; o[8] = st.wHead; o[9] = st.wChain; o[10] = sumIter; o[11] = nLanes | (st.wBand << 8);
  }
#else
  raycastResult[x + y * p.W] = cast_ray<DeviceOps>(p, s, x, y, mm);
#endif
}

// k_raycast for a volume whose range image carries a box (see RB_*): tiles outside DIRTY keep their miss.
// (the body, by one 16x16-pixel workgroup (wgx, wgy): the kernel below, or the live half of k_raycast_pair)
__device__ __forceinline__ void raycast_box_tile(const FrameP &p, const SceneP &s, int ctrIdx, const float2 *__restrict__ minmax,
                                                 float4 *__restrict__ raycastResult, int32_t *__restrict__ rb, int wgx, int wgy) {
  if (s.ctr[ctrIdx] <= 0 && ctrIdx == CTR_NO_VISIBLE_LIVE) return;  // Prepare() is skipped without visible blocks
  if (wgx == 0 && wgy == 0 && threadIdx.x == 0) raybox_mark_ran(rb, p);
  if (!raybox_tile(rb, RB_DIRTY, wgx, wgy)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = wgx * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = wgy * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (x >= p.W || y >= p.H) return;
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const float2 mm = minmax[(x >> 3) + (y >> 3) * mw];
  RC_STAT(RcStats st;)
  raycastResult[x + y * p.W] = cast_ray<DeviceOps>(p, s, x, y, mm RC_STAT(, st));
}
__global__ __launch_bounds__(256, 8) void k_raycast_box(FrameP p, SceneP s, int ctrIdx, const float2 *__restrict__ minmax,
                                                        float4 *__restrict__ raycastResult, int32_t *__restrict__ rb) {
  raycast_box_tile(p, s, ctrIdx, minmax, raycastResult, rb, blockIdx.x, blockIdx.y);
}

// dsr_dump_render_state: the miss pixels outside the last raycast's box get the value that raycast would have written — the
// start point of a ray through an empty cell, for the pose it ran with (rb's POSE record); pixels inside are left alone
__global__ __launch_bounds__(256) void k_raycast_fill_outside(SceneP s, const int32_t *__restrict__ rb, float4 *__restrict__ raycastResult) {
  if (!rb[RB_EVER] || raybox_tile(rb, RB_LAST, blockIdx.x, blockIdx.y)) return;
  const FrameP &p = *reinterpret_cast<const FrameP *>(rb + RB_POSE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (x >= p.W || y >= p.H) return;
  RC_STAT(RcStats st;)
  raycastResult[x + y * p.W] = cast_ray<DeviceOps>(p, s, x, y, make_float2(kFarAway, kVeryClose) RC_STAT(, st));
}

// ---------------------------------------------------------------- K8: ICP maps

template <class Ops>
__host__ __device__ __forceinline__ uchar4 grey_px(float angle) {  // drawPixelGrey
  float outRes = (0.8f * angle + 0.2f) * 255.0f;
  uint8_t g = (uint8_t)Ops::f2i(outRes);
  return make_uchar4(g, g, g, g);
}

// ITMVisualisationEngine.h processPixelICP<true> + computeNormalAndAngle<true> (image space), one pixel.  (Like cast_ray a
// template over Ops so that tests/test_device_functions_host.py runs it on the CPU against the oracle.)
template <class Ops>
__host__ __device__ __forceinline__ void icp_pixel(const FrameP &p, const float4 *__restrict__ pointsRay, int x, int y,
                                                   float4 &pointOut, float4 &normalOut, uchar4 &greyOut) {
  const int W = p.W, H = p.H;
  const int locId = x + y * W;
  const float lsx = -p.invM.m[8], lsy = -p.invM.m[9], lsz = -p.invM.m[10];
  const float4 point = pointsRay[locId];
  bool foundPoint = point.w > 0.0f;
  float nx = 0, ny = 0, nz = 0, angle = 0;
  if (foundPoint) {
    if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) foundPoint = false;
    else {
      float4 xp1_y = pointsRay[(x + 2) + y * W], x_yp1 = pointsRay[x + (y + 2) * W];
      float4 xm1_y = pointsRay[(x - 2) + y * W], x_ym1 = pointsRay[x + (y - 2) * W];
      float4 diff_x = make_float4(0, 0, 0, 0), diff_y = make_float4(0, 0, 0, 0);
      bool doPlus1 = false;
      if (xp1_y.w <= 0 || x_yp1.w <= 0 || xm1_y.w <= 0 || x_ym1.w <= 0) doPlus1 = true;
      else {
        diff_x = make_float4(xp1_y.x - xm1_y.x, xp1_y.y - xm1_y.y, xp1_y.z - xm1_y.z, 0);
        diff_y = make_float4(x_yp1.x - x_ym1.x, x_yp1.y - x_ym1.y, x_yp1.z - x_ym1.z, 0);
        float a = diff_x.x * diff_x.x + diff_x.y * diff_x.y + diff_x.z * diff_x.z;
        float b = diff_y.x * diff_y.x + diff_y.y * diff_y.y + diff_y.z * diff_y.z;
        float length_diff = (a > b) ? a : b;  // MAX
        if (length_diff * p.voxelSize * p.voxelSize > (0.15f * 0.15f)) doPlus1 = true;
      }
      if (doPlus1) {
        xp1_y = pointsRay[(x + 1) + y * W]; x_yp1 = pointsRay[x + (y + 1) * W];
        xm1_y = pointsRay[(x - 1) + y * W]; x_ym1 = pointsRay[x + (y - 1) * W];
        diff_x = make_float4(xp1_y.x - xm1_y.x, xp1_y.y - xm1_y.y, xp1_y.z - xm1_y.z, 0);
        diff_y = make_float4(x_yp1.x - x_ym1.x, x_yp1.y - x_ym1.y, x_yp1.z - x_ym1.z, 0);
        if (xp1_y.w <= 0 || x_yp1.w <= 0 || xm1_y.w <= 0 || x_ym1.w <= 0) foundPoint = false;
      }
      if (foundPoint) {
        nx = -(diff_x.y * diff_y.z - diff_x.z * diff_y.y);
        ny = -(diff_x.z * diff_y.x - diff_x.x * diff_y.z);
        nz = -(diff_x.x * diff_y.y - diff_x.y * diff_y.x);
        float normScale = 1.0f / Ops::sqrt(nx * nx + ny * ny + nz * nz);
        nx *= normScale; ny *= normScale; nz *= normScale;
        angle = nx * lsx + ny * lsy + nz * lsz;
        if (!(angle > 0.0f)) foundPoint = false;
      }
    }
  }
  if (foundPoint) {
    greyOut = grey_px<Ops>(angle);
    pointOut = make_float4(point.x * p.voxelSize, point.y * p.voxelSize, point.z * p.voxelSize, 1.0f);
    normalOut = make_float4(nx, ny, nz, 0.0f);
  } else {
    pointOut = normalOut = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    greyOut = make_uchar4(0, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void k_icp_maps(FrameP p, SceneP s, const float4 *__restrict__ pointsRay,
                                                  float4 *__restrict__ pointsMap, float4 *__restrict__ normalsMap,
                                                  uchar4 *__restrict__ outRendering) {
  if (s.ctr[CTR_NO_VISIBLE_LIVE] <= 0) return;
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= p.W || y >= p.H) return;
  float4 point, normal;
  uchar4 grey;
  icp_pixel<DeviceOps>(p, pointsRay, x, y, point, normal, grey);
  const int locId = x + y * p.W;
  outRendering[locId] = grey;
  pointsMap[locId] = point;
  normalsMap[locId] = normal;
}
// ... for a volume whose range image carries a box: outside DIRTY the maps already hold the miss constants
__global__ __launch_bounds__(256) void k_icp_maps_box(FrameP p, SceneP s, const float4 *__restrict__ pointsRay,
                                                      float4 *__restrict__ pointsMap, float4 *__restrict__ normalsMap,
                                                      uchar4 *__restrict__ outRendering, const int32_t *__restrict__ rb) {
  if (s.ctr[CTR_NO_VISIBLE_LIVE] <= 0 || !raybox_tile(rb, RB_DIRTY, blockIdx.x, blockIdx.y)) return;
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= p.W || y >= p.H) return;
  float4 point, normal;
  uchar4 grey;
  icp_pixel<DeviceOps>(p, pointsRay, x, y, point, normal, grey);
  const int locId = x + y * p.W;
  outRendering[locId] = grey;
  pointsMap[locId] = point;
  normalsMap[locId] = normal;
}

// ------------------------------------------------------- free-view shading (K8)

// ITMRepresentationAccess.h computeSingleNormalFromSDF
// The 32 voxels it reads lie in the 4x4x4 neighbourhood [i-1, i+2]^3, i.e. in at most 2x2x2 blocks.
// The reference reads them one after the other through a one-entry cache that a neighbourhood
// across a block boundary keeps evicting (76 % of the pixels): up to 32 dependent lookups.  Here
// the <= 8 blocks are resolved first (bucket heads requested two at a time, chains then walked;
// their indices kept in a per-thread LDS row), after which the 32 loads are independent.
template <class Ops>
__host__ __device__ __forceinline__ float3 normal_from_sdf(const SceneP &s, const FrameP &p, float x, float y, float z,
                                                           int *__restrict__ bptr /* [8], per thread */) {
  const int ix = Ops::f2i(Ops::floor(x)), iy = Ops::f2i(Ops::floor(y)), iz = Ops::f2i(Ops::floor(z));
  const float cx = x - (float)ix, cy = y - (float)iy, cz = z - (float)iz;
  const float nx = 1.0f - cx, ny = 1.0f - cy, nz = 1.0f - cz;
  const int b0x = (ix - 1) >> 3, b0y = (iy - 1) >> 3, b0z = (iz - 1) >> 3;
  const bool fx = ((ix + 2) >> 3) != b0x, fy = ((iy + 2) >> 3) != b0y, fz = ((iz + 2) >> 3) != b0z;
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) {  // blocks (ox, oy, oz) = (0|1, pair & 1, pair >> 1)
    const int oy = pair & 1, oz = pair >> 1;
    int r0 = -1, r1 = -1;
    if (!((oy && !fy) || (oz && !fz))) {
      const int by = b0y + oy, bz = b0z + oz;
      int4 h0 = *reinterpret_cast<const int4 *>(s.table + hash_index(b0x, by, bz, p.hashMask)), h1 = make_int4(0, 0, 0, -2);
      if (fx) h1 = *reinterpret_cast<const int4 *>(s.table + hash_index(b0x + 1, by, bz, p.hashMask));
      auto resolve = [&](int4 raw, int bx) -> int {  // ITMRepresentationAccess.h findVoxel
        while (true) {
          const int hx = (short)(raw.x & 0xffff), hy = (short)((uint32_t)raw.x >> 16), hz = (short)(raw.y & 0xffff);
          if (hx == bx && hy == by && hz == bz && raw.w >= 0) return raw.w;
          if (raw.z < 1) return -1;
          raw = *reinterpret_cast<const int4 *>(s.table + (uint32_t)(p.noBuckets + raw.z - 1));
        }
      };
      r0 = resolve(h0, b0x);
      if (fx) r1 = resolve(h1, b0x + 1);
    }
    bptr[pair * 2] = r0; bptr[pair * 2 + 1] = r1;
  }
  const uint8_t *vb = s.vba + kOffSdf;
  auto rd = [&](int dx, int dy, int dz) -> float {  // readVoxel(...).sdf as float; missing voxel: TVoxel() => 32767
    const int vx = ix + dx, vy = iy + dy, vz = iz + dz;
    const int ptr = bptr[((vx >> 3) - b0x) | (((vy >> 3) - b0y) << 1) | (((vz >> 3) - b0z) << 2)];
    // unconditional load (a missing block reads block 0, replaced below): a load per `if` is a basic block with its own wait,
    // and the 32 reads of this function would be 32 serialised round trips
    const short v = *reinterpret_cast<const short *>(vb + (size_t)(ptr >= 0 ? ptr : 0) * kBlockBytes + (((vx & 7) + ((vy & 7) << 3) + ((vz & 7) << 6)) * 2));
    return ptr >= 0 ? (float)v : 32767.0f;
  };
#define RD(dx, dy, dz) rd((dx), (dy), (dz))
  float4 front, back, tmp;
  front.x = RD(0, 0, 0); front.y = RD(1, 0, 0); front.z = RD(0, 1, 0); front.w = RD(1, 1, 0);
  back.x = RD(0, 0, 1); back.y = RD(1, 0, 1); back.z = RD(0, 1, 1); back.w = RD(1, 1, 1);
  float p1, p2, v1;
  float3 ret;
  // gradient x
  p1 = front.x * ny * nz + front.z * cy * nz + back.x * ny * cz + back.z * cy * cz;
  tmp.x = RD(-1, 0, 0); tmp.y = RD(-1, 1, 0); tmp.z = RD(-1, 0, 1); tmp.w = RD(-1, 1, 1);
  p2 = tmp.x * ny * nz + tmp.y * cy * nz + tmp.z * ny * cz + tmp.w * cy * cz;
  v1 = p1 * cx + p2 * nx;
  p1 = front.y * ny * nz + front.w * cy * nz + back.y * ny * cz + back.w * cy * cz;
  tmp.x = RD(2, 0, 0); tmp.y = RD(2, 1, 0); tmp.z = RD(2, 0, 1); tmp.w = RD(2, 1, 1);
  p2 = tmp.x * ny * nz + tmp.y * cy * nz + tmp.z * ny * cz + tmp.w * cy * cz;
  ret.x = sdf_to_float(p1 * nx + p2 * cx - v1);
  // gradient y
  p1 = front.x * nx * nz + front.y * cx * nz + back.x * nx * cz + back.y * cx * cz;
  tmp.x = RD(0, -1, 0); tmp.y = RD(1, -1, 0); tmp.z = RD(0, -1, 1); tmp.w = RD(1, -1, 1);
  p2 = tmp.x * nx * nz + tmp.y * cx * nz + tmp.z * nx * cz + tmp.w * cx * cz;
  v1 = p1 * cy + p2 * ny;
  p1 = front.z * nx * nz + front.w * cx * nz + back.z * nx * cz + back.w * cx * cz;
  tmp.x = RD(0, 2, 0); tmp.y = RD(1, 2, 0); tmp.z = RD(0, 2, 1); tmp.w = RD(1, 2, 1);
  p2 = tmp.x * nx * nz + tmp.y * cx * nz + tmp.z * nx * cz + tmp.w * cx * cz;
  ret.y = sdf_to_float(p1 * ny + p2 * cy - v1);
  // gradient z
  p1 = front.x * nx * ny + front.y * cx * ny + front.z * nx * cy + front.w * cx * cy;
  tmp.x = RD(0, 0, -1); tmp.y = RD(1, 0, -1); tmp.z = RD(0, 1, -1); tmp.w = RD(1, 1, -1);
  p2 = tmp.x * nx * ny + tmp.y * cx * ny + tmp.z * nx * cy + tmp.w * cx * cy;
  v1 = p1 * cz + p2 * nz;
  p1 = back.x * nx * ny + back.y * cx * ny + back.z * nx * cy + back.w * cx * cy;
  tmp.x = RD(0, 0, 2); tmp.y = RD(1, 0, 2); tmp.z = RD(0, 1, 2); tmp.w = RD(1, 1, 2);
  p2 = tmp.x * nx * ny + tmp.y * cx * ny + tmp.z * nx * cy + tmp.w * cx * cy;
  ret.z = sdf_to_float(p1 * nz + p2 * cz - v1);
#undef RD
  return ret;
}

// ITMRepresentationAccess.h readFromSDF_color4u_interpolated (missing voxels: colour 0).  The reference reads the 8 corners
// one after the other (8 dependent lookup + load pairs); here the cell's blocks are resolved first (resolve_cell_blocks) and
// the 8 colour words are loaded together, unconditionally (a missing block reads block 0 and contributes 0): the sums below
// are the reference's, in its order.
template <class Ops>
__host__ __device__ __forceinline__ float3 color_interpolated(const SceneP &s, const FrameP &p, float x, float y, float z) {
  VoxCache cache; cache_init(cache);
  VoxCache cache2; cache_init(cache2);
  const int ix = Ops::f2i(Ops::floor(x)), iy = Ops::f2i(Ops::floor(y)), iz = Ops::f2i(Ops::floor(z));
  const float cx = x - (float)ix, cy = y - (float)iy, cz = z - (float)iz;
  const bool fx = (ix & 7) == 7, fy = (iy & 7) == 7, fz = (iz & 7) == 7;
  int bp[8];
  resolve_cell_blocks<Ops>(s, p, ix >> 3, iy >> 3, iz >> 3, fx, fy, fz, cache, cache2, bp);
  uchar4 c[8];
  bool have[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
    const int slot = ((fx && dx) ? 1 : 0) | ((fy && dy) ? 2 : 0) | ((fz && dz) ? 4 : 0);
    int ptr = bp[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) ptr = (slot == q) ? bp[q] : ptr;
    const int lin = ((ix + dx) & 7) + (((iy + dy) & 7) << 3) + (((iz + dz) & 7) << 6);
    have[k] = ptr >= 0;
    c[k] = *reinterpret_cast<const uchar4 *>(s.vba + (size_t)(ptr >= 0 ? ptr : 0) * kBlockBytes + kOffClr + lin * 4);
  }
  float rx = 0.0f, ry = 0.0f, rz = 0.0f;
  auto acc = [&](int k, float w) {
    const uchar4 v = have[k] ? c[k] : make_uchar4(0, 0, 0, 0);
    rx += w * (float)v.x; ry += w * (float)v.y; rz += w * (float)v.z;
  };
  acc(0, (1.0f - cx) * (1.0f - cy) * (1.0f - cz));
  acc(1, (cx) * (1.0f - cy) * (1.0f - cz));
  acc(2, (1.0f - cx) * (cy) * (1.0f - cz));
  acc(3, (cx) * (cy) * (1.0f - cz));
  acc(4, (1.0f - cx) * (1.0f - cy) * cz);
  acc(5, (cx) * (1.0f - cy) * cz);
  acc(6, (1.0f - cx) * (cy) * cz);
  acc(7, (cx) * (cy) * cz);
  return make_float3(rx / 255.0f, ry / 255.0f, rz / 255.0f);
}

// RenderImage_common shading + the fork's FREECAMERA_DEPTH / COLOUR_FROM_DEPTH_WEIGHT
// (definitions adopted in oracle/dsr_oracle.cpp render_image()), one pixel: pt = its raycast result, bptr = 8 ints of scratch
template <class Ops>
__host__ __device__ __forceinline__ uchar4 render_pixel(const FrameP &p, const SceneP &s, int type, const float4 &pt, int *__restrict__ bptr) {
  const float lsx = -p.invM.m[8], lsy = -p.invM.m[9], lsz = -p.invM.m[10];
  bool foundPoint = pt.w > 0;
  uchar4 out = make_uchar4(0, 0, 0, 0);
  switch (type) {
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME:
      if (foundPoint) {
        float3 c = color_interpolated<Ops>(s, p, pt.x, pt.y, pt.z);
        out.x = (uint8_t)Ops::f2i(c.x * 255.0f); out.y = (uint8_t)Ops::f2i(c.y * 255.0f); out.z = (uint8_t)Ops::f2i(c.z * 255.0f);
        out.w = 255;
      }
      break;
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL:
    case DSR_IMAGE_FREECAMERA_SHADED: {
      float3 n = make_float3(0, 0, 0);
      float angle = 0;
      if (foundPoint) {
        n = normal_from_sdf<Ops>(s, p, pt.x, pt.y, pt.z, bptr);
        float normScale = 1.0f / Ops::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
        n.x *= normScale; n.y *= normScale; n.z *= normScale;
        angle = n.x * lsx + n.y * lsy + n.z * lsz;
        if (!(angle > 0.0f)) foundPoint = false;
      }
      if (foundPoint) {
        if (type == DSR_IMAGE_FREECAMERA_SHADED) out = grey_px<Ops>(angle);
        else {
          out.x = (uint8_t)Ops::f2i((0.3f + (-n.x + 1.0f) * 0.35f) * 255.0f);
          out.y = (uint8_t)Ops::f2i((0.3f + (-n.y + 1.0f) * 0.35f) * 255.0f);
          out.z = (uint8_t)Ops::f2i((0.3f + (-n.z + 1.0f) * 0.35f) * 255.0f);
          out.w = 0;
        }
      }
    } break;
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT:
      if (foundPoint) {
        VoxCache cache; cache_init(cache);
        int lin;
        int ptr = find_block(s, p, Ops::f2i(roundf_itm(pt.x)), Ops::f2i(roundf_itm(pt.y)), Ops::f2i(roundf_itm(pt.z)), lin, cache);
        float w = 0.0f;
        if (ptr >= 0) w = (float)s.vba[(size_t)ptr * kBlockBytes + kOffWDepth + lin];
        float t = w / (float)p.maxW;
        t = (t > 0.0f) ? t : 0.0f;   // max(0, t)
        t = (1.0f < t) ? 1.0f : t;   // min(1, .)
        out.x = (uint8_t)Ops::f2i(255.0f * (1.0f - t)); out.y = 0; out.z = (uint8_t)Ops::f2i(255.0f * t); out.w = 255;
      }
      break;
    default: break;
  }
  return out;
}
// ... and its depth along the camera's z axis (FREECAMERA_DEPTH; 0 where the ray found nothing)
__host__ __device__ __forceinline__ float render_depth(const FrameP &p, const float4 &pt) {
  float d = 0.0f;
  if (pt.w > 0) {
    const float mx = pt.x * p.voxelSize, my = pt.y * p.voxelSize, mz = pt.z * p.voxelSize;
    d = p.M.m[2] * mx + p.M.m[6] * my + p.M.m[10] * mz + p.M.m[14] * 1.0f;
  }
  return d;
}

// outRgba2: a second destination of the colour image (the caller's HBM buffer: no copy kernel after the shading)
__global__ __launch_bounds__(256) void k_render(FrameP p, SceneP s, int type, const float4 *__restrict__ pointsRay,
                                                uchar4 *__restrict__ outRgba, float *__restrict__ outDepth,
                                                uchar4 *__restrict__ outRgba2) {
  __shared__ int s_blocks[256][9];  // normal_from_sdf: the <= 8 blocks of a pixel's neighbourhood (row padded)
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= p.W || y >= p.H) return;
  const int locId = x + y * p.W;
  const float4 pt = pointsRay[locId];
  const uchar4 out = render_pixel<DeviceOps>(p, s, type, pt, s_blocks[threadIdx.x]);
  if (outRgba) outRgba[locId] = out;
  if (outRgba2) outRgba2[locId] = out;
  if (outDepth) outDepth[locId] = render_depth(p, pt);
}

// A free-view raycast that shades its own pixels (small volumes, round 4): an instance's preview was raycast + render, two
// launches of a frame that is a chain of ~20 launches of a few microseconds — the render only ever looks at its own pixel's
// ray.  The ray is k_raycast's plain path (same functions, same result, still written to raycastResult for later renders of
// the same pose), the shading is k_render's.  Measured on an instance volume (profiles/r04i_instance_frame_fold_fuse_ab.log):
// 34.4 + 12.2 us as two launches, 38.1 us as one.
// rb (may be null): the range image's box record — outside DIRTY the engine's own buffers (raycastResult, outRgba) keep their
// miss, the CALLER's buffers (outDepth, outRgba2) get it written.
// (the body, by one 16x16-pixel workgroup: the kernel below, or the preview half of k_raycast_pair)
__device__ __forceinline__ void raycast_render_tile(const FrameP &p, const SceneP &s, const float2 *__restrict__ minmax,
                                                    float4 *__restrict__ raycastResult, int type, uchar4 *__restrict__ outRgba,
                                                    float *__restrict__ outDepth, uchar4 *__restrict__ outRgba2,
                                                    int32_t *__restrict__ rb, int wgx, int wgy, int *__restrict__ bptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = wgx * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = wgy * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (rb) {
    if (wgx == 0 && wgy == 0 && threadIdx.x == 0) raybox_mark_ran(rb, p);
    if (!raybox_tile(rb, RB_DIRTY, wgx, wgy)) {
      if (x >= p.W || y >= p.H) return;
      if (outRgba2) outRgba2[x + y * p.W] = make_uchar4(0, 0, 0, 0);
      if (outDepth) outDepth[x + y * p.W] = 0.0f;
      return;
    }
  }
  if (x >= p.W || y >= p.H) return;
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const float2 mm = minmax[(x >> 3) + (y >> 3) * mw];
  RC_STAT(RcStats st;)
  const float4 pt = cast_ray<DeviceOps>(p, s, x, y, mm RC_STAT(, st));
  const int locId = x + y * p.W;
  raycastResult[locId] = pt;
  const uchar4 out = render_pixel<DeviceOps>(p, s, type, pt, bptr);
  if (outRgba) outRgba[locId] = out;
  if (outRgba2) outRgba2[locId] = out;
  if (outDepth) outDepth[locId] = render_depth(p, pt);
}
__global__ __launch_bounds__(256) void k_raycast_render(FrameP p, SceneP s, const float2 *__restrict__ minmax,
                                                        float4 *__restrict__ raycastResult, int type, uchar4 *__restrict__ outRgba,
                                                        float *__restrict__ outDepth, uchar4 *__restrict__ outRgba2,
                                                        int32_t *__restrict__ rb) {
  __shared__ int s_blocks[256][9];
  raycast_render_tile(p, s, minmax, raycastResult, type, outRgba, outDepth, outRgba2, rb, blockIdx.x, blockIdx.y, s_blocks[threadIdx.x]);
}

// The tracking render and the preview render of ONE frame of an instance-sized volume in ONE launch (round 6).  Both are chains
// of dependent gathers for a few hundred live rays in a frame of 465 k pixels: one after the other they cost 25 + 27 us of an
// instance frame, side by side the time of one.  Two streams do not get there — a cross-queue dependency costs 13-20 us each
// way on this part (profiles/r06g_*timeline.json) —, one grid does: blockIdx.z = 0 is k_raycast_box for the live render state,
// 1 is k_raycast_render for the free camera.  The engine defers the tracking render of dsr_prepare until the next call; when
// that call is the preview render, both go out together (dsr_engine.hip "paired render").
struct RenderHalfP {  // the free-camera half's buffers
  const float2 *minmax;
  float4 *raycastResult;
  uchar4 *outRgba;
  float *outDepth;
  uchar4 *outRgba2;
  int32_t *rb;
  int type;
};
__global__ __launch_bounds__(256) void k_raycast_pair(FrameP pLive, FrameP pFree, SceneP s, const float2 *__restrict__ minmaxLive,
                                                      float4 *__restrict__ raycastResultLive, int32_t *__restrict__ rbLive, RenderHalfP fv) {
  __shared__ int s_blocks[256][9];
  if (blockIdx.z == 0) raycast_box_tile(pLive, s, CTR_NO_VISIBLE_LIVE, minmaxLive, raycastResultLive, rbLive, blockIdx.x, blockIdx.y);
  else raycast_render_tile(pFree, s, fv.minmax, fv.raycastResult, fv.type, fv.outRgba, fv.outDepth, fv.outRgba2, fv.rb, blockIdx.x,
                           blockIdx.y, s_blocks[threadIdx.x]);
}

}  // namespace dsr
