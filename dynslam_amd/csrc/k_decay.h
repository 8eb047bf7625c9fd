// k_decay.h — scene reset, voxel GC (fork's Decay/Reap), AoS<->planar conversion kernels.
//
// Voxel GC replaces the fork's decay_device family.  The fork has no
// CPU implementation (InfiniTamDriver.h:198-206); the specification is DESIGN.md "voxel GC"
// and is restated serially in oracle/dsr_oracle.cpp decay().  The GPU formulation keeps the
// serial semantics: blocks are freed onto the VBA free list in CANDIDATE ORDER through an
// ordered prefix sum, entries become tombstones (no chain surgery, hence no bucket locks).
#pragma once
#include "dsr_device.h"

namespace dsr {

// ------------------------------------------------------------------------ reset

__global__ __launch_bounds__(256) void k_reset_table(dsr_hash_entry *__restrict__ table, int n,
                                                     uint32_t *__restrict__ allocKey) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    *reinterpret_cast<int4 *>(table + i) = make_int4(0, 0, 0, -2);  // pos 0, offset 0, ptr -2
    allocKey[i] = 0u;
  }
}
__global__ __launch_bounds__(256) void k_iota(int32_t *__restrict__ a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
// default voxels: sdf = 32767, everything else 0.  16 B per thread, grid-stride.
__global__ __launch_bounds__(256) void k_reset_vba(uint4 *__restrict__ vba, size_t nVec16) {
  const uint4 sdfPat = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nVec16; i += (size_t)gridDim.x * blockDim.x) {
    // 256 x 16 B per block; the first 64 vectors are the sdf plane
    vba[i] = ((i & 255) < 64) ? sdfPat : zero;
  }
}
__global__ void k_reset_counters(int32_t *ctr, unsigned long long *work, int noBlocks, int noExcess) {
  if (threadIdx.x < CTR_COUNT) ctr[threadIdx.x] = 0;
  if (threadIdx.x < WORK_COUNT) work[threadIdx.x] = 0ull;
  __syncthreads();
  if (threadIdx.x == 0) {
    ctr[CTR_LAST_FREE_BLOCK] = noBlocks - 1;
    ctr[CTR_LAST_FREE_EXCESS] = noExcess - 1;
  }
}

// ----------------------------------------------------- planar <-> AoS exchange

// one wave per block: planar HBM layout -> dsr_voxel[512] (8 B each)
__global__ __launch_bounds__(256) void k_blocks_to_aos(const uint8_t *__restrict__ vba, int firstBlock, int nBlocks,
                                                       dsr_voxel *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = blockIdx.x * 4 + wave; b < nBlocks; b += gridDim.x * 4) {
    const uint8_t *blk = vba + (size_t)(firstBlock + b) * kBlockBytes;
    for (int v = lane; v < kBlockSize3; v += 64) {
      dsr_voxel o;
      o.sdf = *reinterpret_cast<const short *>(blk + kOffSdf + v * 2);
      o.w_depth = blk[kOffWDepth + v];
      uchar4 c = *reinterpret_cast<const uchar4 *>(blk + kOffClr + v * 4);
      o.clr[0] = c.x; o.clr[1] = c.y; o.clr[2] = c.z;
      o.w_color = c.w;
      o._pad = 0;
      out[(size_t)b * kBlockSize3 + v] = o;
    }
  }
}

// ------------------------------------------------------------------- voxel GC

// copy the live visible list into a FIFO slot (ids + count)
// The voxel-GC FIFO holds the visible list of each of the last min_age + 1 Decay calls.  A list is
// ASCENDING in entry index (ordered compaction), so it is stored as a BIT PER HASH ENTRY: a plane of
// E / 8 bytes per queued frame (2.6 MB at 21 M entries) instead of a slot sized for the worst case
// (noBlocks ints: 13.5 GB for min_age 200 at 2^24 blocks) — fixed size, independent of how many blocks
// are visible, and popping a plane (k_bits_count / k_bits_write: ordered compaction again) gives back
// exactly the list that was pushed.
__global__ __launch_bounds__(256) void k_fifo_push_bits(const int32_t *__restrict__ visibleIDs, const int32_t *__restrict__ ctr,
                                                        uint32_t *__restrict__ plane) {
  const int n = ctr[CTR_NO_VISIBLE_LIVE];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t id = (uint32_t)visibleIDs[i];
    atomicOr(&plane[id >> 5], 1u << (id & 31u));
  }
}
// a thread owns kTileItems = 8 consecutive entries = one byte of the plane
static_assert(kTileItems == 8, "one plane byte per thread");
__global__ __launch_bounds__(kTileThreads) void k_bits_count(const uint8_t *__restrict__ plane, int noTotalEntries,
                                                             int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);
  if (base < noTotalEntries) c.x = __popc((uint32_t)plane[base >> 3]);
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kTileThreads) void k_bits_write(const uint8_t *__restrict__ plane, int noTotalEntries,
                                                             const int2 *__restrict__ tileOffsets, int32_t *__restrict__ out,
                                                             int capacity) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  const uint32_t bits = base < noTotalEntries ? (uint32_t)plane[base >> 3] : 0u;
  int2 c = make_int2(__popc(bits), 0);
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  int rank = tileOffsets[blockIdx.x].x + ex.x;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (bits & (1u << j)) { if (rank < capacity) out[rank] = base + j; rank++; }
}

// One wave per candidate block: reset voxels with w_depth <= maxWeight, flag the block when
// all 512 voxels end up with w_depth == 0.
// zeroIsReset: the engine guarantees that a voxel with w_depth == 0 is in the reset state already
// (sdf 32767, no colour) — true whenever colour can only be fused together with depth (mu < 4 m:
// voxels the depth step rejects never pass the colour gate) and max_w >= 1 —, so resetting it
// again would only rewrite the same bytes: such voxels count as empty but are not touched.
__global__ __launch_bounds__(256) void k_decay_blocks(SceneP s, const int32_t *__restrict__ cand,
                                                      const int32_t *__restrict__ nCandPtr, int maxWeight,
                                                      uint8_t *__restrict__ freedFlag, int zeroIsReset) {
  const int n = *nCandPtr;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&s.work[WORK_V_DECAY], (unsigned long long)n);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const int t = __builtin_amdgcn_readfirstlane(cand[i]);
    const int ptr = s.table[t].ptr;
    if (ptr < 0) { if (lane == 0) freedFlag[i] = 0; continue; }
    uint8_t *blk = s.vba + (size_t)ptr * kBlockBytes;
    // the weights decide everything: only lanes that reset a voxel touch the other two planes
    const uint2 wdRaw = *reinterpret_cast<const uint2 *>(blk + kOffWDepth + lane * 8);
    uint32_t wdW[2] = {wdRaw.x, wdRaw.y};
    uint32_t resetMask = 0;  // bit x: voxel x of this lane is reset
    int empty = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      int w = (int)((wdW[x >> 2] >> ((x & 3) * 8)) & 0xffu);
      if (w <= maxWeight) {
        if (!(zeroIsReset && w == 0)) resetMask |= 1u << x;
        wdW[x >> 2] &= ~(0xffu << ((x & 3) * 8));
        w = 0;
      }
      if (w == 0) empty++;
    }
    if (resetMask) {
      // the read-modify-writes of the lane are independent: loads first, then the stores
      const uint4 sdfRaw = *reinterpret_cast<const uint4 *>(blk + kOffSdf + lane * 16);
      const uint4 c0 = *reinterpret_cast<const uint4 *>(blk + kOffClr + lane * 32);
      const uint4 c1 = *reinterpret_cast<const uint4 *>(blk + kOffClr + lane * 32 + 16);
      uint32_t sdfW[4] = {sdfRaw.x, sdfRaw.y, sdfRaw.z, sdfRaw.w};
      uint32_t clrW[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (resetMask & (1u << x)) {
          sdfW[x >> 1] = (sdfW[x >> 1] & ~(0xffffu << ((x & 1) * 16))) | (0x7fffu << ((x & 1) * 16));
          clrW[x] = 0u;  // colour and w_color
        }
      *reinterpret_cast<uint4 *>(blk + kOffSdf + lane * 16) = make_uint4(sdfW[0], sdfW[1], sdfW[2], sdfW[3]);
      *reinterpret_cast<uint2 *>(blk + kOffWDepth + lane * 8) = make_uint2(wdW[0], wdW[1]);
      *reinterpret_cast<uint4 *>(blk + kOffClr + lane * 32) = make_uint4(clrW[0], clrW[1], clrW[2], clrW[3]);
      *reinterpret_cast<uint4 *>(blk + kOffClr + lane * 32 + 16) = make_uint4(clrW[4], clrW[5], clrW[6], clrW[7]);
    }
    // wave64 reduction of the empty count
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) empty += __shfl_xor(empty, d);
    if (lane == 0) freedFlag[i] = (empty == kBlockSize3) ? 1 : 0;
  }
}

// ordered prefix over the candidate list's freed flags
__global__ __launch_bounds__(kTileThreads) void k_flag_count(const uint8_t *__restrict__ flags,
                                                             const int32_t *__restrict__ nPtr,
                                                             int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int n = *nPtr;
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (base + j < n && flags[base + j]) c.x++;
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}

// push freed blocks onto the VBA free list in candidate order; entries become tombstones
__global__ __launch_bounds__(kTileThreads) void k_decay_commit(SceneP s, const int32_t *__restrict__ cand,
                                                               const int32_t *__restrict__ nPtr,
                                                               const uint8_t *__restrict__ flags,
                                                               const int2 *__restrict__ tileOffsets,
                                                               uint8_t *__restrict__ visType) {
  __shared__ int2 lds[kTileThreads / 64];
  const int n = *nPtr;
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  uint8_t f[kTileItems];
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    f[j] = (base + j < n) ? flags[base + j] : 0;
    if (f[j]) c.x++;
  }
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  int rank = tileOffsets[blockIdx.x].x + ex.x;
  const int oldHead = s.ctr[CTR_ALLOC_OLD_HEAD_VBA];
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (f[j]) {
      const int t = cand[base + j];
      dsr_hash_entry *he = s.table + t;
      s.voxelAllocList[oldHead + 1 + rank] = he->ptr;
      he->ptr = -2;
      visType[t] = 0;
      if (s.allocBits) {  // (k_small.h: the allocated set of an instance-sized volume, as bits and as a sorted list)
        atomicAnd(&s.allocBits[t >> 5], ~(1u << (t & 31)));
        s.ctr[CTR_ALLOC_IDS_VALID] = 0;  // the list is rebuilt from the bits by the next allocation
      }
      if (s.swapState) { s.swapState[t] = 0; s.swapStored[t] = 0; }
      rank++;
    }
}

// forceAllVoxels: candidates = every entry with ptr >= 0, ascending (ordered compaction)
__global__ __launch_bounds__(kTileThreads) void k_allocated_count(SceneP s, int noTotalEntries,
                                                                  int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (base + j < noTotalEntries && s.table[base + j].ptr >= 0) c.x++;
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kTileThreads) void k_allocated_write(SceneP s, int noTotalEntries,
                                                                  const int2 *__restrict__ tileOffsets,
                                                                  int32_t *__restrict__ out, int capacity) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  bool a[kTileItems];
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    a[j] = base + j < noTotalEntries && s.table[base + j].ptr >= 0;
    if (a[j]) c.x++;
  }
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  int rank = tileOffsets[blockIdx.x].x + ex.x;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (a[j]) { if (rank < capacity) out[rank] = base + j; rank++; }
}

// ITMVisualisationEngine::FindVisibleBlocks for a free camera, DENSE: a thread per ALLOCATED entry
// (ascending list, rebuilt only when the scene changes) runs the frustum test; the flagged ones
// are then compacted in order.  (A sweep over all entries with the test inside costs 4x as much:
// 2/3 of the entries are empty and the test diverges.)
__global__ __launch_bounds__(256) void k_freeview_test(FrameP p, SceneP s, const int32_t *__restrict__ list,
                                                       const int32_t *__restrict__ nPtr, uint8_t *__restrict__ flags) {
  const int n = *nPtr;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const dsr_hash_entry he = load_entry(s.table, (uint32_t)list[i]);
    bool isVisible = false, isVisibleEnlarged;
    if (he.ptr >= 0) check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
    flags[i] = isVisible ? 1 : 0;
  }
}
__global__ __launch_bounds__(kTileThreads) void k_flag_write(const int32_t *__restrict__ list, const uint8_t *__restrict__ flags,
                                                             const int32_t *__restrict__ nPtr,
                                                             const int2 *__restrict__ tileOffsets,
                                                             int32_t *__restrict__ out, int capacity,
                                                             const dsr_hash_entry *__restrict__ table, int4 *__restrict__ outBlocks) {
  __shared__ int2 lds[kTileThreads / 64];
  const int n = *nPtr;
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  bool f[kTileItems];
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    f[j] = base + j < n && flags[base + j];
    if (f[j]) c.x++;
  }
  int2 total;
  const int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  int rank = tileOffsets[blockIdx.x].x + ex.x;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (f[j]) {
      if (rank < capacity) {
        const int t = list[base + j];
        out[rank] = t;
        outBlocks[rank] = make_vis_record(*reinterpret_cast<const int4 *>(table + t), t);  // the visible-block stream (dsr_device.h)
      }
      rank++;
    }
}

// live visible list compaction after decay: keep ids whose visType is still != 0
__global__ __launch_bounds__(kTileThreads) void k_live_keep_count(const int32_t *__restrict__ ids,
                                                                  const int32_t *__restrict__ ctr,
                                                                  const uint8_t *__restrict__ visType,
                                                                  int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int n = ctr[CTR_NO_VISIBLE_LIVE];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (base + j < n && visType[ids[base + j]] != 0) c.x++;
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kTileThreads) void k_live_keep_write(const int32_t *__restrict__ ids,
                                                                  const int32_t *__restrict__ ctr,
                                                                  const uint8_t *__restrict__ visType,
                                                                  const int2 *__restrict__ tileOffsets,
                                                                  int32_t *__restrict__ out, const int4 *__restrict__ inBlocks,
                                                                  int4 *__restrict__ outBlocks) {
  __shared__ int2 lds[kTileThreads / 64];
  const int nOld = ctr[CTR_TMP_OLD_NVIS];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int32_t id[kTileItems];
  bool keep[kTileItems];
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    keep[j] = false;
    if (base + j < nOld) { id[j] = ids[base + j]; keep[j] = visType[id[j]] != 0; }
    if (keep[j]) c.x++;
  }
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  int rank = tileOffsets[blockIdx.x].x + ex.x;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (keep[j]) { out[rank] = id[j]; outBlocks[rank] = inBlocks[base + j]; rank++; }  // kept entries: pos and ptr unchanged
}

}  // namespace dsr
