// k_swap.h — host swapping of voxel blocks (settings.use_swapping): ITMSwappingEngine +
// ITMGlobalCache of upstream InfiniTAM (Engine/ITMSwappingEngine*.{h,tpp,cu}, Objects/ITMGlobalCache.h),
// restated serially in oracle/dsr_oracle.cpp swap_in()/swap_out().
//
// Per frame, after integration:
//   swap-in  (IntegrateGlobalIntoLocal): the first <= 4096 entries in ascending order whose
//            swap state is 1 are merged (combineVoxelInformation) with their copy in the host
//            store, if there is one, and become state 2;
//   swap-out (SaveToGlobalMemory): the first <= 4096 entries in ascending order that are in
//            state 2, resident and not visible are copied to the host store, their block is
//            reset and returned to the free list IN ENTRY ORDER, ptr = -1, state 0.
// Both candidate lists are ordered compactions with a cap (tile counts -> scan -> write), so the
// free-list order equals the serial loop's.
//
// The host store is a pool of PINNED host slabs (16384 plane-wise 4 KiB blocks each by default) that the
// GPU addresses directly: swap-out kernels write blocks into it over the host link, swap-in
// kernels read them back.  As in upstream's ITMGlobalCache an entry OWNS ONE SLOT for the life of the
// scene (swapSlot[entry], -1 until its first swap-out): it is handed out once, by a device counter in
// list order, and every later swap-out of that entry — also after voxel GC dropped the copy or a
// tombstone was reused for another block — overwrites it, so the store is bounded by the number of
// entries that were ever swapped out instead of growing by up to 16 MiB per frame.  The host never learns the lists: no round trip, no synchronisation —
// ITMDenseMapper::ProcessFrame stays asynchronous with swapping on.  (Upstream copies counts and
// id lists to the host and memcpys every block through a staging buffer each frame.)
#pragma once
#include "dsr_device.h"

namespace dsr {

constexpr int kTransferBlocks = DSR_TRANSFER_BLOCK_NUM;  // SDF_TRANSFER_BLOCK_NUM
constexpr int kSlabBlocksDefault = 16384;                 // 64 MiB of pinned host memory per slab

__device__ __forceinline__ uint8_t *host_block(const SceneP &s, int slot) {
  return s.hostSlabs[slot / s.slabBlocks] + (size_t)(slot % s.slabBlocks) * kBlockBytes;
}

template <bool OUT>
__device__ __forceinline__ bool swap_candidate(const SceneP &s, int t, const uint8_t *__restrict__ visType) {
  if (OUT) return s.swapState[t] == 2 && s.table[t].ptr >= 0 && visType[t] == 0;
  return s.swapState[t] == 1;
}

template <bool OUT>
__global__ __launch_bounds__(kTileThreads) void k_swap_count(SceneP s, int noTotalEntries,
                                                             const uint8_t *__restrict__ visType,
                                                             int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);  // x: candidates, y (swap-out): those that do not own a host slot yet
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (base + j < noTotalEntries && swap_candidate<OUT>(s, base + j, visType)) {
      c.x++;
      if (OUT && s.swapSlot[base + j] < 0) c.y++;
    }
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}

template <bool OUT>
__global__ __launch_bounds__(kTileThreads) void k_swap_write(SceneP s, int noTotalEntries,
                                                             const uint8_t *__restrict__ visType,
                                                             const int2 *__restrict__ tileOffsets,
                                                             int32_t *__restrict__ ids, uint8_t *__restrict__ storedFlags) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  bool cand[kTileItems], fresh[kTileItems];
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    cand[j] = base + j < noTotalEntries && swap_candidate<OUT>(s, base + j, visType);
    fresh[j] = OUT && cand[j] && s.swapSlot[base + j] < 0;
    if (cand[j]) c.x++;
    if (fresh[j]) c.y++;
  }
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  const int2 off = tileOffsets[blockIdx.x];
  int rank = off.x + ex.x, freshRank = off.y + ex.y;
  const int firstSlot = OUT ? s.ctr[CTR_SWAP_FIRST_SLOT] : 0;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (cand[j]) {
      if (rank < kTransferBlocks) {
        ids[rank] = base + j;
        if (!OUT) storedFlags[rank] = s.swapStored[base + j];
        if (fresh[j]) {
          // first swap-out of this entry: it takes the next host slot, in list order (every candidate
          // before it is inside the cap too, so freshRank counts exactly the slots handed out before)
          s.swapSlot[base + j] = firstSlot + freshRank;
          atomicMax(&s.ctr[CTR_HOST_USED], firstSlot + freshRank + 1);
        }
      }
      rank++;
      if (fresh[j]) freshRank++;
    }
}

// swap-in, step 1: the stored copies of the listed entries, host store -> device staging buffer
// (16 B per lane, coalesced over the host link; one wave per block)
__global__ __launch_bounds__(256) void k_swapin_fetch(SceneP s, const int32_t *__restrict__ ids,
                                                      const uint8_t *__restrict__ storedFlags, uint8_t *__restrict__ staging) {
  const int n = s.ctr[CTR_SWAP_COUNT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    if (!storedFlags[i]) continue;
    const int id = __builtin_amdgcn_readfirstlane(ids[i]);
    const uint4 *src = reinterpret_cast<const uint4 *>(host_block(s, s.swapSlot[id]));
    uint4 *dst = reinterpret_cast<uint4 *>(staging + (size_t)i * kBlockBytes);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k * 64 + lane] = src[k * 64 + lane];
  }
}

// ITMSwappingEngine.h combineVoxelDepthInformation / combineVoxelColorInformation on the plane-wise
// layout.  One wave per transferred block, lane = 8 voxels.
__global__ __launch_bounds__(256) void k_swapin_combine(SceneP s, int maxW, const int32_t *__restrict__ ids,
                                                        const uint8_t *__restrict__ storedFlags,
                                                        const uint8_t *__restrict__ staging) {
  const int n = s.ctr[CTR_SWAP_COUNT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const int id = __builtin_amdgcn_readfirstlane(ids[i]);
    const int ptr = s.table[id].ptr;
    const bool stored = storedFlags[i] != 0;
    if (stored && ptr < 0) continue;  // no block could be given to it: stays in state 1
    if (stored) {
      const uint8_t *src = staging + (size_t)i * kBlockBytes;
      uint8_t *dst = s.vba + (size_t)ptr * kBlockBytes;
#pragma unroll 1
      for (int x = 0; x < 8; ++x) {
        const int v = lane * 8 + x;
        {  // depth
          int newW = dst[kOffWDepth + v];
          const int oldW = src[kOffWDepth + v];
          if (oldW != 0) {
            float newF = sdf_to_float((float)*reinterpret_cast<const short *>(dst + kOffSdf + v * 2));
            const float oldF = sdf_to_float((float)*reinterpret_cast<const short *>(src + kOffSdf + v * 2));
            newF = (float)oldW * oldF + (float)newW * newF;
            newW = oldW + newW;
            newF /= (float)newW;
            newW = newW < maxW ? newW : maxW;
            dst[kOffWDepth + v] = (uint8_t)newW;
            *reinterpret_cast<short *>(dst + kOffSdf + v * 2) = sdf_from_float(newF);
          }
        }
        {  // colour
          const uchar4 dc = *reinterpret_cast<const uchar4 *>(dst + kOffClr + v * 4);  // (r, g, b, w_color)
          const uchar4 sc = *reinterpret_cast<const uchar4 *>(src + kOffClr + v * 4);
          int newW = dc.w;
          const int oldW = sc.w;
          if (oldW != 0) {
            float nx = (float)dc.x / 255.0f, ny = (float)dc.y / 255.0f, nz = (float)dc.z / 255.0f;
            const float ox = (float)sc.x / 255.0f, oy = (float)sc.y / 255.0f, oz = (float)sc.z / 255.0f;
            nx = ox * (float)oldW + nx * (float)newW;
            ny = oy * (float)oldW + ny * (float)newW;
            nz = oz * (float)oldW + nz * (float)newW;
            newW = oldW + newW;
            nx /= (float)newW; ny /= (float)newW; nz /= (float)newW;
            newW = newW < maxW ? newW : maxW;
            *reinterpret_cast<uchar4 *>(dst + kOffClr + v * 4) =
                make_uchar4((uint8_t)f2i(nx * 255.0f), (uint8_t)f2i(ny * 255.0f), (uint8_t)f2i(nz * 255.0f), (uint8_t)newW);
          }
        }
      }
    }
    if (lane == 0) s.swapState[id] = 2;
  }
}

// copy the block into the entry's host slot (k_swap_write<true> gave it one if it had none), reset it,
// return it to the free list (list order = entry order: slot oldHead + 1 + i), ptr = -1, state 0,
// mark the host store as holding it
__global__ __launch_bounds__(256) void k_swapout_move(SceneP s, const int32_t *__restrict__ ids) {
  const int n = s.ctr[CTR_SWAP_COUNT];
  const int oldHead = s.ctr[CTR_ALLOC_OLD_HEAD_VBA];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint4 sdfPat = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const int id = __builtin_amdgcn_readfirstlane(ids[i]);
    const int ptr = s.table[id].ptr;
    uint4 *blk = reinterpret_cast<uint4 *>(s.vba + (size_t)ptr * kBlockBytes);
    uint4 *dst = reinterpret_cast<uint4 *>(host_block(s, s.swapSlot[id]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // 256 x 16 B per block
      const int v = k * 64 + lane;
      dst[v] = blk[v];
      blk[v] = (v < 64) ? sdfPat : zero;
    }
    if (lane == 0) {
      s.voxelAllocList[oldHead + 1 + i] = ptr;
      s.table[id].ptr = -1;
      s.swapState[id] = 0;
      s.swapStored[id] = 1;
    }
  }
}

}  // namespace dsr
