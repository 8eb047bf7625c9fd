// k_alloc.h — voxel-block allocation and visible-list kernels (the view conversion kernels: k_edges.h).
//
// Replaces: convertDepthAffineToFloat_device, buildHashAllocAndVisibleType,
// allocateVoxelBlocksList, buildVisibleList, setToType3 of upstream's *_CUDA engines — with a
// DETERMINISTIC formulation whose result equals the serial _CPU engine:
//   * K0b re-tests the previous frame's visible list against the frustum DENSELY (a thread per list
//     entry) instead of inside the sweep over all entries;
//   * K1 marks per pixel; for every target entry the LAST writer in (raster pixel, step)
//     order wins, selected with an atomicMax on a 32-bit order key; the first writer of an
//     entry also counts it (a byte per 8 entries, a word per sweep tile);
//   * the commit pops the voxel/excess free lists in ASCENDING ENTRY ORDER through ordered tile
//     prefix sums (wave64 shuffle scans + a single-workgroup scan of tile sums), reading one
//     byte per 8 entries, compacts the marked entries into an ordered work list and lets a dense
//     kernel recompute each winner's block position and write the table;
//   * the visible list is an ordered compaction, so visibleEntryIDs is ascending.
#pragma once
#include "dsr_device.h"

namespace dsr {

// ------------------------------------------------------------- allocation ray

struct AllocRay {
  float px, py, pz;  // start point, block units
  float dx, dy, dz;  // per-step increment
  int noSteps;
};

// First half of buildHashAllocAndVisibleTypePP: the ray segment [d-mu, d+mu] in block units.
template <class Ops = DeviceOps>
__host__ __device__ __forceinline__ bool alloc_ray(const FrameP &p, const float *__restrict__ depth, int x, int y, AllocRay &r) {
  float depth_measure = depth[x + y * p.W];
  if (depth_measure <= 0 || (depth_measure - p.mu) < 0 || (depth_measure - p.mu) < p.vfMin ||
      (depth_measure + p.mu) > p.vfMax)
    return false;
  const float oneOverVoxelSize = 1.0f / (p.voxelSize * (float)kBlockSize);
  const float invFx = 1.0f / p.proj.x, invFy = 1.0f / p.proj.y;
  float cz = depth_measure;
  float cx = cz * (((float)x - p.proj.z) * invFx);
  float cy = cz * (((float)y - p.proj.w) * invFy);
  float norm = Ops::sqrt(cx * cx + cy * cy + cz * cz);
  float f1 = 1.0f - p.mu / norm;
  float3 t = mat_mul3(p.invM, cx * f1, cy * f1, cz * f1, 1.0f);
  r.px = t.x * oneOverVoxelSize; r.py = t.y * oneOverVoxelSize; r.pz = t.z * oneOverVoxelSize;
  float f2 = 1.0f + p.mu / norm;
  t = mat_mul3(p.invM, cx * f2, cy * f2, cz * f2, 1.0f);
  float ex = t.x * oneOverVoxelSize, ey = t.y * oneOverVoxelSize, ez = t.z * oneOverVoxelSize;
  float dx = ex - r.px, dy = ey - r.py, dz = ez - r.pz;
  norm = Ops::sqrt(dx * dx + dy * dy + dz * dz);
  r.noSteps = Ops::f2i(Ops::ceil(2.0f * norm));
  float denom = (float)(r.noSteps - 1);
  r.dx = dx / denom; r.dy = dy / denom; r.dz = dz / denom;
  return true;
}

// K1: per-pixel mark.  16x16 pixel tiles (a wave covers 16x4 pixels: neighbouring rays
// probe the same buckets, which keeps the 16-byte entry gathers in L2).
// BITS (instance-sized volumes, k_small.h): the one-workgroup list kernel that follows must find the entries this kernel
// touched without sweeping 1.18 MB of types.  Atomics are out: the 64 rays of a wave walk through the same few blocks, a bit per
// step and lane is ~300 atomic ORs on each of a few hundred words — 2.9 ms for a kernel of 20 us, still 122 us with one atomic per
// distinct entry of a wave (profiles/r05a_, r05c_instance_frame_ab.log).  So only PLAIN byte stores here: the type is written as
// kTouchedNow (1 with the top bit set: "set by THIS frame's mark", which a type left from an earlier frame never carries) and
// the byte of the entry's group of 8 in s.visGrp is set; the list kernel reads the few marked groups' types, turns them into
// bits and puts the plain 1 back (k_small.h phase D0).
constexpr uint8_t kTouchedNow = 0x81;
// Called by EVERY lane of the wave (inImage: the lane's pixel exists) — round 6: the order keys of a wave are de-duplicated before
// they go out.  An instance volume's blocks are large on the image (0.28 m at 5 m: 40 pixels across) and a ray names ~15 of
// them: hundreds of lanes raise the key of the same entry, serialised at L2 — in frames that allocate, k_batch_alloc_mark took
// 73 us for eight volumes where the same launch takes 21 us once every block exists (profiles/r06i_batch_step_timeline.json).
// Only the LARGEST key of an entry matters (atomicMax) and exactly one writer must see the key still 0 and count the entry; so a
// lane drops a target when a lane whose key is certainly larger names the same entry in the same batch of steps: its own later
// step, its right neighbour in the row (pixel + 1) or the pixel below (pixel + W) — the 16x4 pixels of a wave.  Whoever is not
// covered issues; the chain of coverers ends at a lane that does, with a key at least as large.  The result — every key, every
// count — is the one all writers together produce.
template <bool BITS>
__device__ __forceinline__ void alloc_mark_pixel(const FrameP &p, const SceneP &s, const float *__restrict__ depth,
                                                 uint8_t *__restrict__ visType, int x, int y, bool inImage) {
  AllocRay r;
  r.px = r.py = r.pz = r.dx = r.dy = r.dz = 0.0f; r.noSteps = 0;
  bool ok = inImage && alloc_ray(p, depth, x, y, r);
  if (ok && (uint32_t)(r.noSteps > 0 ? r.noSteps : 0) > p.maxSteps) {
    // the order key holds maxSteps steps per pixel (bound derived for a rigid pose, dsr_engine.hip):
    // a scaled / non-orthonormal pose would make the commit replay the wrong step — report it
    // instead of writing a wrong block position
    s.ctr[CTR_STATUS] = DSR_E_ARG;
    ok = false;
  }
  const int noSteps = ok ? r.noSteps : 0;
  // (BITS = the instance-sized volumes: there the de-duplication pays — 73 -> 30 us for eight volumes; a map's 4 cm blocks are 2-3
  //  pixels across and its mark read 34.5 -> 35.6 us with it, profiles/r06j_bench_profile_all_*.json: not used there)
  int waveSteps = noSteps;  // the wave runs as many batches as its longest ray needs (the shuffles below need every lane)
  if (BITS) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(waveSteps, d); waveSteps = o > waveSteps ? o : waveSteps; }
  }
  const int lane = threadIdx.x & 63;
  const uint32_t keyBase = (uint32_t)(x + y * p.W) * p.maxSteps + 1u;
  float px = r.px, py = r.py, pz = r.pz;
  // The steps of a ray are independent (stores of the same value, atomicMax, exactly-once counting),
  // so they are taken four at a time: positions first (the same running additions as the serial
  // loop), then the four bucket heads are requested together — one round trip instead of four.
  for (int i0 = 0; i0 < waveSteps; i0 += 4) {
    short cb[4][3];
    uint32_t ch[4];
    dsr_hash_entry head[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cb[k][0] = f2s(floorf(px)); cb[k][1] = f2s(floorf(py)); cb[k][2] = f2s(floorf(pz));
      ch[k] = hash_index(cb[k][0], cb[k][1], cb[k][2], p.hashMask);
      if (i0 + k < noSteps) head[k] = load_entry(s.table, ch[k]);
      px += r.dx; py += r.dy; pz += r.dz;
    }
    // what each of the four steps marks for allocation: entry index (kNoTarget: nothing) and whether it is a chain append
    constexpr uint32_t kNoTarget = 0xffffffffu;
    uint32_t tgt[4] = {kNoTarget, kNoTarget, kNoTarget, kNoTarget};
    bool exc[4] = {false, false, false, false};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k;
      if (i >= noSteps) break;
      const short bx = cb[k][0], by = cb[k][1], bz = cb[k][2];
      uint32_t hashIdx = ch[k];
      bool isFound = false;
      int firstFree = -1;
      dsr_hash_entry he = head[k];
      if (he.pos[0] == bx && he.pos[1] == by && he.pos[2] == bz && he.ptr >= -1) {
        if (BITS) { visType[hashIdx] = kTouchedNow; s.visGrp[hashIdx >> 3] = 1; }  // (no swapped-out entries on this path)
        else visType[hashIdx] = (he.ptr == -1) ? (uint8_t)2 : (uint8_t)1;
        isFound = true;
      }
      if (!isFound) {
        if (he.ptr < -1) firstFree = (int)hashIdx;
        while (he.offset >= 1) {
          hashIdx = (uint32_t)(p.noBuckets + he.offset - 1);
          he = load_entry(s.table, hashIdx);
          if (he.pos[0] == bx && he.pos[1] == by && he.pos[2] == bz && he.ptr >= -1) {
            if (BITS) { visType[hashIdx] = kTouchedNow; s.visGrp[hashIdx >> 3] = 1; }
            else visType[hashIdx] = (he.ptr == -1) ? (uint8_t)2 : (uint8_t)1;
            isFound = true;
            break;
          }
          if (he.ptr < -1 && firstFree < 0) firstFree = (int)hashIdx;
        }
        if (!isFound) {
          const bool isExcess = firstFree < 0;
          const uint32_t target = isExcess ? hashIdx : (uint32_t)firstFree;
          if (!isExcess) {
            if (BITS) { visType[target] = kTouchedNow; s.visGrp[target >> 3] = 1; }
            else visType[target] = 1;
          }
          tgt[k] = target; exc[k] = isExcess;
        }
      }
    }
    // de-duplication (see above): targets of the right neighbour in the row and of the pixel below, this batch
    bool covered[4] = {false, false, false, false};
    if (BITS) {
      const bool hasRight = (lane & 15) != 15, hasBelow = lane < 48;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint32_t right = (uint32_t)__shfl_down((int)tgt[kk], 1), below = (uint32_t)__shfl_down((int)tgt[kk], 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (hasRight && right == tgt[k]) covered[k] = true;
          if (hasBelow && below == tgt[k]) covered[k] = true;
          if (kk > k && tgt[kk] == tgt[k]) covered[k] = true;  // this ray's own later step
        }
      }
    }
    // The order keys of the (up to) four steps go out TOGETHER and their results are looked at afterwards: an atomic whose
    // result is used inside its own `if` is a basic block with its own wait — four serialised round trips per batch for the
    // wave.  (Sending the idle steps to one spare word instead of predicating them was tried: 16 ms of same-address contention.)
    uint32_t old[4] = {1u, 1u, 1u, 1u};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (tgt[k] != kNoTarget && !covered[k]) old[k] = atomicMax(&s.allocKey[tgt[k]], keyBase + (uint32_t)(i0 + k));  // step < maxSteps (checked above)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // the first writer of an entry in this frame (its key was still 0) also counts it: per group of 8 entries and per sweep
      // tile, so that the commit finds the ~1 % marked entries without reading all the keys
      if (tgt[k] != kNoTarget && !covered[k] && old[k] == 0u) {
        const uint32_t one = exc[k] ? 0x11u : 0x01u;
        atomicAdd(&s.allocGrp[tgt[k] >> 5], one << (((tgt[k] >> 3) & 3u) * 8u));
        atomicAdd(&s.allocTile[tgt[k] / (uint32_t)kTile], exc[k] ? 0x100000001ull : 1ull);
      }
    }
  }
}

// block position written by the winning (pixel, step) — replays the walk of that pixel
__device__ __forceinline__ void alloc_winner_pos(const FrameP &p, const float *__restrict__ depth, uint32_t key,
                                                 short &bx, short &by, short &bz) {
  const uint32_t k = key - 1u;
  const uint32_t pixel = k / p.maxSteps, step = k - pixel * p.maxSteps;
  const int y = (int)(pixel / (uint32_t)p.W), x = (int)(pixel - (uint32_t)y * (uint32_t)p.W);
  AllocRay r;
  alloc_ray(p, depth, x, y, r);
  float px = r.px, py = r.py, pz = r.pz;
  for (uint32_t i = 0; i < step; ++i) { px += r.dx; py += r.dy; pz += r.dz; }
  bx = f2s(floorf(px)); by = f2s(floorf(py)); bz = f2s(floorf(pz));
}

// Exclusive scan of the tile sums by ONE workgroup of 1024 threads; mode selects the epilogue.
enum ScanMode { SCAN_ALLOC = 0, SCAN_VISIBLE_LIVE = 1, SCAN_VISIBLE_FREE = 2, SCAN_DECAY = 3, SCAN_COMPACT_LIVE = 4, SCAN_NCAND = 5, SCAN_SWAP_IN = 6, SCAN_SWAP_OUT = 7, SCAN_MESH = 8, SCAN_ALLOCATED = 9 };
// (the body, by a workgroup of NT threads: the scan launch below, or a phase of the one-workgroup kernels of k_small.h)
template <int NT>
__device__ __forceinline__ void scan_tile_sums_body(int2 *__restrict__ tileSums, int numTiles, const SceneP &s, int mode,
                                                    int capacity, int2 *lds) {
  int2 carry = make_int2(0, 0);
  constexpr int kPer = 8;
  if (numTiles <= NT * kPer) {
    // the usual case (<= 8192 tiles = 16.8 M entries with 1024 threads): 8 consecutive sums per thread, ONE workgroup scan
    const int base = threadIdx.x * kPer;
    int2 v[kPer];
    int2 sum = make_int2(0, 0);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      v[j] = (base + j < numTiles) ? tileSums[base + j] : make_int2(0, 0);
      sum.x += v[j].x; sum.y += v[j].y;
    }
    int2 run = wg_exclusive_scan2<NT>(sum, carry, lds);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (base + j < numTiles) tileSums[base + j] = run;
      run.x += v[j].x; run.y += v[j].y;
    }
  } else {
    for (int base = 0; base < numTiles; base += NT) {
      int i = base + threadIdx.x;
      int2 v = (i < numTiles) ? tileSums[i] : make_int2(0, 0);
      int2 total;
      int2 ex = wg_exclusive_scan2<NT>(v, total, lds);
      if (i < numTiles) tileSums[i] = make_int2(carry.x + ex.x, carry.y + ex.y);
      carry.x += total.x; carry.y += total.y;
    }
  }
  if (threadIdx.x == 0) {
    int32_t *ctr = s.ctr;
    if (mode == SCAN_ALLOC) {
      const int oldV = ctr[CTR_LAST_FREE_BLOCK], oldE = ctr[CTR_LAST_FREE_EXCESS];
      ctr[CTR_ALLOC_OLD_HEAD_VBA] = oldV; ctr[CTR_ALLOC_OLD_HEAD_EXC] = oldE;
      ctr[CTR_ALLOC_TOTAL12] = carry.x; ctr[CTR_ALLOC_TOTAL2] = carry.y;
      int nv = oldV - carry.x, ne = oldE - carry.y;
      ctr[CTR_LAST_FREE_BLOCK] = nv < -1 ? -1 : nv;
      ctr[CTR_LAST_FREE_EXCESS] = ne < -1 ? -1 : ne;
      if (carry.x > oldV + 1 || carry.y > oldE + 1) ctr[CTR_STATUS] = DSR_E_OUT_OF_BLOCKS;
    } else if (mode == SCAN_VISIBLE_LIVE || mode == SCAN_VISIBLE_FREE || mode == SCAN_COMPACT_LIVE) {
      int n = carry.x < capacity ? carry.x : capacity;
      if (mode == SCAN_COMPACT_LIVE) ctr[CTR_TMP_OLD_NVIS] = ctr[CTR_NO_VISIBLE_LIVE];
      ctr[mode == SCAN_VISIBLE_FREE ? CTR_NO_VISIBLE_FREE : CTR_NO_VISIBLE_LIVE] = n;
      if (mode == SCAN_VISIBLE_LIVE) {
        ctr[CTR_VIS_OVERFLOW] = carry.x > capacity ? 1 : 0;  // more visible entries than the list holds (K0b)
        // with swapping: visible swapped-out entries (y) take fresh blocks, after the frame's
        // regular allocations, in ascending entry order
        const int oldV = ctr[CTR_LAST_FREE_BLOCK];
        ctr[CTR_ALLOC_OLD_HEAD_VBA] = oldV;
        const int nv = oldV - carry.y;
        ctr[CTR_LAST_FREE_BLOCK] = nv < -1 ? -1 : nv;
        if (carry.y > oldV + 1) ctr[CTR_STATUS] = DSR_E_OUT_OF_BLOCKS;
      }
    } else if (mode == SCAN_SWAP_IN || mode == SCAN_SWAP_OUT) {
      const int n = carry.x < capacity ? carry.x : capacity;
      ctr[CTR_SWAP_COUNT] = n;
      if (mode == SCAN_SWAP_OUT) {
        ctr[CTR_ALLOC_OLD_HEAD_VBA] = ctr[CTR_LAST_FREE_BLOCK];
        ctr[CTR_LAST_FREE_BLOCK] += n;
        // first host slot for the entries of this batch that do not own one yet (k_swap.h
        // k_swap_write<true> hands them out in list order and advances CTR_HOST_USED)
        ctr[CTR_SWAP_FIRST_SLOT] = ctr[CTR_HOST_USED];
      }
    } else if (mode == SCAN_ALLOCATED) {
      ctr[CTR_NO_ALLOCATED] = carry.x < capacity ? carry.x : capacity;
    } else if (mode == SCAN_MESH) {
      ctr[CTR_MESH_TOTAL] = carry.x;
    } else if (mode == SCAN_NCAND) {
      ctr[CTR_DECAY_NCAND] = carry.x < capacity ? carry.x : capacity;
    } else if (mode == SCAN_DECAY) {
      ctr[CTR_DECAY_FREED] = carry.x;
      ctr[CTR_ALLOC_OLD_HEAD_VBA] = ctr[CTR_LAST_FREE_BLOCK];
      ctr[CTR_LAST_FREE_BLOCK] += carry.x;
      atomicAdd(&s.work[WORK_DECAYED_BLOCKS], (unsigned long long)carry.x);
    }
  }
}
__global__ __launch_bounds__(1024) void k_scan_tile_sums(int2 *__restrict__ tileSums, int numTiles, SceneP s, int mode,
                                                         int capacity) {
  __shared__ int2 lds[1024 / 64];
  scan_tile_sums_body<1024>(tileSums, numTiles, s, mode, capacity, lds);
}

// (Round 4 measured the scan WITHOUT its own launch — the workgroups of the producing sweep draw a ticket, the last one scans:
//  three launches fewer per instance frame and slower, because every wave pays an agent-scope release, an L2 write-back
//  (k_alloc_mark 20 -> 124 us).  Archived: profiles/r05_pruned_fold_scans.diff, profiles/r04i_instance_frame_fold_fuse_ab.log.
//  Instance-sized volumes now run these steps inside ONE workgroup, where a barrier is all it takes: k_small.h.)

// K1, the kernel.  (tileX0, tileY0): first 16x16 tile of the grid — an instance's view is empty outside the box its silhouette
// was cut from (dsr_view_extract_silhouette), and the order keys are made of pixel indices, not of the launch geometry.
template <bool BITS>
__global__ __launch_bounds__(256) void k_alloc_mark(FrameP p, SceneP s, const float *__restrict__ depth,
                                                    uint8_t *__restrict__ visType, int tileX0, int tileY0) {
  const int x = (blockIdx.x + tileX0) * 16 + (threadIdx.x & 15), y = (blockIdx.y + tileY0) * 16 + (threadIdx.x >> 4);
  alloc_mark_pixel<BITS>(p, s, depth, visType, x, y, x < p.W && y < p.H);  // (every lane: the wave de-duplicates its keys)
}

// K2: commit in ascending entry order (the serial loop of AllocateSceneFromDepth), in two
// steps.  The sweep reads one BYTE per 8 entries (the counts k_alloc_mark kept), ranks the ~1 %
// marked entries and compacts them into an ordered work list {entry, key, vbaIdx, exlIdx}; the
// dense kernel below replays the winner's ray and writes the table, with every lane busy.  The
// per-tile totals the ranks start from were accumulated by k_alloc_mark too and scanned by
// k_scan_tile_sums: no pass over the 4-byte keys of all entries is left in the frame.
__global__ __launch_bounds__(kTileThreads) void k_alloc_commit(FrameP p, SceneP s, int2 *__restrict__ tileOffsets,
                                                               int4 *__restrict__ workList) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;  // this thread's group of 8 entries
  // counts of the group from the byte k_alloc_mark maintained (4 groups = 4 adjacent lanes per word)
  uint32_t *grpWord = s.allocGrp + (base >> 5);
  const uint32_t word = (base < p.noTotalEntries) ? *grpWord : 0u;
  const uint32_t byte = (word >> (((base >> 3) & 3) * 8)) & 0xffu;
  const int2 c = make_int2((int)(byte & 15u), (int)(byte >> 4));
  if (word != 0u && (threadIdx.x & 3) == 0) *grpWord = 0u;  // ready for the next frame
  // every thread reads the tile offset BEFORE the scan: the scan's __syncthreads() then orders all
  // these loads before thread 0 clears the word for the next frame's marks
  const int2 tileOff = tileOffsets[blockIdx.x];
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileOffsets[blockIdx.x] = make_int2(0, 0);  // k_alloc_mark accumulates into it again
  if (c.x == 0) return;
  int rank12 = tileOff.x + ex.x, rank2 = tileOff.y + ex.y;
  const int oldV = s.ctr[CTR_ALLOC_OLD_HEAD_VBA], oldE = s.ctr[CTR_ALLOC_OLD_HEAD_EXC];
  // the group's 8 keys and 8 table pointers are requested TOGETHER (a load per loop iteration behind a `continue` is a basic
  // block with its own wait: 16 serialised round trips for this thread, and its wave waits with it)
  uint32_t key[kTileItems];
  int ptrOf[kTileItems];
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    const int t = base + j < p.noTotalEntries ? base + j : p.noTotalEntries - 1;
    key[j] = s.allocKey[t];
    ptrOf[j] = s.table[t].ptr;
  }
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (base + j >= p.noTotalEntries) key[j] = 0u;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    const int t = base + j;
    const uint32_t k = key[j];
    if (!k) continue;
    const bool isExc = ptrOf[j] >= -1;
    s.allocKey[t] = 0u;  // replaces memset(entriesAllocType, 0) of the next frame
    const int vbaIdx = oldV - rank12;
    int exlIdx = 0;
    if (isExc) { exlIdx = oldE - rank2; rank2++; }
    // out of voxel blocks: nothing is written past the list end; out of excess entries: a hole
    if (vbaIdx >= 0) workList[rank12] = make_int4(exlIdx >= 0 ? t : -1, (int)k, vbaIdx, isExc ? exlIdx : -1);
    rank12++;
  }
}

// one item of the ordered work list: the winner's block position (its ray replayed), the table written.  BITS: k_small.h
// LISTS (with BITS, k_small.h's list path): the new child is flagged "touched now" like the mark's own hits instead of getting
// its bit in visBits — that plane stays untouched (all zero) while the visible list is built from the sorted list of entries
template <bool BITS>
__device__ __forceinline__ void alloc_apply_item(const FrameP &p, const SceneP &s, const float *__restrict__ depth, const int4 w,
                                                 uint8_t *__restrict__ visType, bool lists = false) {
  const int t = w.x;
  if (t < 0) return;
  short bx, by, bz;
  alloc_winner_pos(p, depth, (uint32_t)w.y, bx, by, bz);
  const int px = (int)((uint32_t)(uint16_t)bx | ((uint32_t)(uint16_t)by << 16));
  const int pz = (int)(uint32_t)(uint16_t)bz;
  const int ptr = s.voxelAllocList[w.z];
  if (w.w < 0) {  // type 1: in place (free head or tombstone); the chain link is kept
    dsr_hash_entry *he = s.table + t;
    *reinterpret_cast<int2 *>(he) = make_int2(px, pz);
    he->ptr = ptr;
    if (BITS) atomicOr(&s.allocBits[t >> 5], 1u << (t & 31));
  } else {  // type 2: append a child from the excess list to this chain tail
    const int exlOffset = s.excessAllocList[w.w];
    s.table[t].offset = exlOffset + 1;
    const int child = p.noBuckets + exlOffset;
    *reinterpret_cast<int4 *>(s.table + child) = make_int4(px, pz, 0, ptr);
    visType[child] = (BITS && lists) ? kTouchedNow : (uint8_t)1;
    if (BITS) {
      if (!lists) atomicOr(&s.visBits[child >> 5], 1u << (child & 31));
      atomicOr(&s.allocBits[child >> 5], 1u << (child & 31));
    }
  }
}
__global__ __launch_bounds__(256) void k_alloc_apply(FrameP p, SceneP s, const float *__restrict__ depth,
                                                     const int4 *__restrict__ workList, uint8_t *__restrict__ visType) {
  const int total = s.ctr[CTR_ALLOC_TOTAL12], avail = s.ctr[CTR_ALLOC_OLD_HEAD_VBA] + 1;
  const int n = total < avail ? total : avail;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    alloc_apply_item<false>(p, s, depth, workList[i], visType);
}

// ------------------------------------------------------------- block visibility

// ITMSceneReconstructionEngine.h checkPointVisibility / checkBlockVisibility (corner
// order and the incremental +=factor arithmetic are part of the result).
template <bool useSwapping, class Ops = DeviceOps>
__host__ __device__ __forceinline__ void check_point_visibility(bool &isVisible, bool &isVisibleEnlarged, float x, float y, float z,
                                                                const Mat4 &M, const float4 &proj, int W, int H) {
  float3 b = mat_mul3(M, x, y, z, 1.0f);
  if (b.z < 1e-10f) return;
  // b.z >= 1e-10: tame divisor, the two divisions share the refined reciprocal (dsr_device.h)
  const float yz = Ops::rcp(b.z);
  float u = Ops::div(proj.x * b.x, b.z, yz) + proj.z;
  float v = Ops::div(proj.y * b.y, b.z, yz) + proj.w;
  if (u >= 0 && u < (float)W && v >= 0 && v < (float)H) {
    isVisible = true; isVisibleEnlarged = true;
  } else if (useSwapping) {
    int lx = -W / 8, ly = W + W / 8, lz = -H / 8, lw = H + H / 8;
    if (u >= (float)lx && u < (float)ly && v >= (float)lz && v < (float)lw) isVisibleEnlarged = true;
  }
}
template <bool useSwapping, class Ops = DeviceOps>
__host__ __device__ __forceinline__ void check_block_visibility(bool &isVisible, bool &isVisibleEnlarged, const short pos[3],
                                                                const Mat4 &M, const float4 &proj, float voxelSize, int W, int H) {
  const float factor = (float)kBlockSize * voxelSize;
  isVisible = false; isVisibleEnlarged = false;
  float x = (float)pos[0] * factor, y = (float)pos[1] * factor, z = (float)pos[2] * factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  z += factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  y += factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  x += factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  z -= factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  y -= factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  x -= factor; y += factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H); if (isVisible) return;
  x += factor; y -= factor; z += factor;
  check_point_visibility<useSwapping, Ops>(isVisible, isVisibleEnlarged, x, y, z, M, proj, W, H);
}

// K0b: the "visible at the previous frame" pass.  The serial engine marks last frame's visible
// entries as type 3 before the per-pixel mark and re-tests the ones still at 3 in its final sweep
// over the whole table.  Here the frustum test runs DENSELY over last frame's visible list, before
// the mark: visible -> 3, otherwise 0.  The mark then overwrites the entries it sees with 1/2
// exactly as in the serial order, so every entry ends with the same type — and the 10-million-entry
// sweep no longer carries a divergent 8-corner test in 6 % of its lanes (55 -> ~15 us).
__global__ __launch_bounds__(256) void k_retest_previous_visible(FrameP p, SceneP s, const int4 *__restrict__ visBlocks,
                                                                 uint8_t *__restrict__ visType) {
  const int n = s.ctr[CTR_NO_VISIBLE_LIVE];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const dsr_hash_entry he = entry_of_record(visBlocks[i]);  // the stream: coalesced, no table gather (pos never changes)
    const int t = he.offset;
    bool isVisible, isVisibleEnlarged;
    if (p.useSwapping) {
      check_block_visibility<true>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
      visType[t] = isVisibleEnlarged ? 3 : 0;
    } else {
      check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
      visType[t] = isVisible ? 3 : 0;
    }
  }
  // The serial engine re-tests EVERY entry of type 3, not just the entries of the list.  The two sets differ only after a frame
  // whose visible entries did not fit the list (capacity = the block array size: a volume exhausted long before): an entry that
  // was cut off keeps its 3 and has no list slot to be re-tested through.  Then — and only then — the whole table is swept for
  // type-3 entries (re-testing a list entry again gives the same answer; found by the exhausted-volume parity test of round 4).
  if (s.ctr[CTR_VIS_OVERFLOW]) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < p.noTotalEntries; t += gridDim.x * blockDim.x) {
      if (visType[t] != 3) continue;
      const dsr_hash_entry he = load_entry(s.table, t);
      bool isVisible, isVisibleEnlarged;
      if (p.useSwapping) {
        check_block_visibility<true>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
        if (!isVisibleEnlarged) visType[t] = 0;
      } else {
        check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
        if (!isVisible) visType[t] = 0;
      }
    }
  }
}

// K3a (live view): count the visible entries per tile (types were settled by K0b / K1 / K2).
// K5a (free view): visible iff ptr >= 0 and inside the frustum (FindVisibleBlocks).
template <bool FREEVIEW>
__global__ __launch_bounds__(kTileThreads) void k_visible_count(FrameP p, SceneP s, uint8_t *__restrict__ visType,
                                                                int2 *__restrict__ tileSums) {
  __shared__ int2 lds[kTileThreads / 64];
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  int2 c = make_int2(0, 0);
  uint8_t v[kTileItems];
  if (!FREEVIEW) {
    if (base + kTileItems <= p.noTotalEntries) {
      uint2 raw = *reinterpret_cast<const uint2 *>(visType + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = (raw.x >> (8 * j)) & 0xff; v[4 + j] = (raw.y >> (8 * j)) & 0xff; }
    } else {
#pragma unroll
      for (int j = 0; j < kTileItems; ++j) v[j] = (base + j < p.noTotalEntries) ? visType[base + j] : 0;
    }
  }
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    const int t = base + j;
    if (FREEVIEW) {
      uint8_t vis = 0;
      if (t < p.noTotalEntries) {
        dsr_hash_entry he = load_entry(s.table, t);
        if (he.ptr >= 0) {
          bool isVisible, isVisibleEnlarged;
          check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
          vis = isVisible ? 1 : 0;
        }
        visType[t] = vis;
      }
      if (vis) c.x++;
    } else {
      if (v[j] > 0) {
        c.x++;
        if (p.useSwapping) {  // ITMSceneReconstructionEngine_CPU: swapStates + "reallocate deleted ones"
          if (s.swapState[t] != 2) s.swapState[t] = 1;
          if (s.table[t].ptr == -1) c.y++;
        }
      }
    }
  }
  int2 total;
  wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}

// K3b: ordered compaction -> ascending visibleEntryIDs; with swapping also the "reallocate
// deleted ones from previous swap operation" loop (visible entries with ptr == -1 get a block)
__global__ __launch_bounds__(kTileThreads) void k_visible_write(int noTotalEntries, const uint8_t *__restrict__ visType,
                                                                const int2 *__restrict__ tileOffsets,
                                                                int32_t *__restrict__ visibleIDs, int capacity,
                                                                SceneP s, int useSwapping, int4 *__restrict__ visBlocks,
                                                                int32_t *__restrict__ publish, int publishSeq) {
  __shared__ int2 lds[kTileThreads / 64];
  // The host's Integrate() must report an exhausted block array (the fork throws: InstanceReconstructor.cpp:662-671) and reads
  // noVisibleBlocks right after it (InfiniTamDriver.h:150).  Both words are final once the scan before this kernel has run —
  // long before the integration that follows — so the first thread publishes them to a pinned, device-mapped host word here:
  // the host polls that word instead of draining the stream behind k_integrate (dsr_engine.hip wait_published).
  if (publish && blockIdx.x == 0 && threadIdx.x == 0) {
    publish[0] = s.ctr[CTR_NO_VISIBLE_LIVE];
    publish[1] = s.ctr[CTR_STATUS];
    __hip_atomic_store(publish + 2, publishSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int base = blockIdx.x * kTile + threadIdx.x * kTileItems;
  uint8_t v[kTileItems];
  bool re[kTileItems];
  int2 c = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    v[j] = (base + j < noTotalEntries) ? visType[base + j] : 0;
    re[j] = false;
    if (v[j] > 0) {
      c.x++;
      if (useSwapping && s.table[base + j].ptr == -1) { re[j] = true; c.y++; }
    }
  }
  // the table entries of this thread's visible items (the visible-block stream, dsr_device.h): all requested before the
  // scan, so that the gathers overlap it and each other instead of forming a chain of up to 8 load -> store pairs
  int4 rawE[kTileItems];
#pragma unroll
  for (int j = 0; j < kTileItems; ++j) {
    rawE[j] = make_int4(0, 0, 0, -2);
    if (v[j] > 0) rawE[j] = *reinterpret_cast<const int4 *>(s.table + (base + j));
  }
  int2 total;
  int2 ex = wg_exclusive_scan2<kTileThreads>(c, total, lds);
  if (total.x == 0) return;
  const int2 off = tileOffsets[blockIdx.x];
  int rank = off.x + ex.x, rrank = off.y + ex.y;
  const int oldHead = useSwapping ? s.ctr[CTR_ALLOC_OLD_HEAD_VBA] : 0;
#pragma unroll
  for (int j = 0; j < kTileItems; ++j)
    if (v[j] > 0) {
      int4 raw = rawE[j];
      if (re[j]) {
        const int vbaIdx = oldHead - rrank;
        rrank++;
        if (vbaIdx >= 0) { raw.w = s.voxelAllocList[vbaIdx]; s.table[base + j].ptr = raw.w; }
      }
      if (rank < capacity) { visibleIDs[rank] = base + j; visBlocks[rank] = make_vis_record(raw, base + j); }
      rank++;
    }
}

}  // namespace dsr
