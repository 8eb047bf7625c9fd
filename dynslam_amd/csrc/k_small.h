// k_small.h — instance-sized volumes: the table sweeps of a frame inside ONE workgroup.
//
// An instance volume (InstanceReconstructor.cpp:363-392: 7142 blocks of 0.035 m voxels behind upstream's 1 179 648-entry
// table) holds a few hundred to a few thousand entries.  Its frame is bound by the NUMBER of launches: round 4 ran it as 21
// launches (232-295 us), of which the ordered table sweeps — scan, commit, apply, count, scan, write, range image, and again for
// the preview camera — were 12, each a grid of 576 workgroups that finds a handful of entries, separated by kernel boundaries
// because their ordered ranks need every workgroup's total.  Folding the scans into the sweeps by tickets lost (an agent-scope
// release per wave, profiles/r04i_*).  Here the dependency disappears instead: the sweeps read BIT planes (one bit per entry,
// 147 KB for the whole table — the mark sets `visBits`, the commit sets `allocBits`), which ONE workgroup of 1024 threads reads
// in nine 16-byte loads per lane, so the ordered ranks are a wave scan + sixteen wave totals in LDS and a barrier is all the
// synchronisation there is:
//   k_small_alloc_visible = k_scan_tile_sums + k_alloc_commit + k_alloc_apply + k_retest_previous_visible + k_visible_count +
//                           k_scan_tile_sums + k_visible_write + k_expected_depth_one              (8 launches -> 1)
//   k_small_freeview      = k_visible_count<FREEVIEW> + k_scan_tile_sums + k_visible_write + k_expected_depth_one   (4 -> 1)
// The arithmetic is the general path's, function for function (alloc_apply_item, check_block_visibility,
// project_single_block, fold_wave_boxes), and so is every result: hash table, free lists, visible list and stream, types,
// range image, status — the parity suite runs both paths against the oracle.
#pragma once
#include "k_alloc.h"
#include "k_raycast.h"

namespace dsr {

// -DDSR_SMALL_CLOCKS (tools/small_kernel_clocks.py, never the product build): thread 0 stamps the 100 MHz clock at the phase
// boundaries of the two one-workgroup kernels (slots 0..15: alloc_visible, 16..31: freeview; slot 15 / 31 counts launches)
#ifdef DSR_SMALL_CLOCKS
__device__ unsigned long long *g_smallClk;
#define SMALL_CLK(i) do { if (threadIdx.x == 0) g_smallClk[(i)] = wall_clock64(); } while (0)
#else
#define SMALL_CLK(i)
#endif

constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / 64;
constexpr int kSmallRows = 9;                                              // 16-byte loads per lane in a sweep of a bit plane
constexpr int kSmallBitWords = kSmallWaves * kSmallRows * 64 * 4;          // 36864 words = 1 179 648 entries: upstream's table
constexpr int kSmallMaxEntries = kSmallBitWords * 32;
constexpr int kSmallMaxTiles = kSmallThreads;                              // one allocation tile total per thread
// Round 6 — the sorted LIST of allocated entries (SceneP::allocIds).  The phase clocks of these kernels (tools/small_kernel_clocks.py,
// profiles/r06a_small_kernel_clocks.json) put 10 + 12 us of k_small_alloc_visible's 43 and 15 of k_small_freeview's 33 into the
// three sweeps of a 147 KB plane that find ~800 entries: nine 16-byte loads per lane, fifty-four shuffles, and up to nine more
// dependent reads where a lane holds bits.  An instance volume owns <= 7142 blocks, so the set of its allocated entries fits one
// list that the commit keeps sorted: a frame's new entries (a handful, ranked among themselves in LDS) are merged in by binary
// searches in LDS, and the visible list is one dense pass over the list — type and table entry of every allocated entry
// requested together, "touched by this frame's mark" / "visible last frame, re-test" decided per entry, ordered compaction.  No
// plane is swept.  The bit planes stay the ground truth: a frame that cannot take the list path — the first after a reset or a
// GC pass (list invalid), an exhausted block array, every frame whose PREVIOUS visible list holds an entry without a block (such
// an entry is re-tested and stays visible while it is in the frustum; the list does not know it), a visible list that overflowed, more than kSmallNewMax new entries — runs the sweeps as before and rebuilds the list from allocBits at its end.
constexpr int kSmallNewMax = 2048;                                         // new entries a frame may merge on the list path
// dynamic LDS of the two kernels: the range image, or — aliased with it, used before it — the merge's scratch
// (the old list + the frame's new entries, raw and sorted)
constexpr size_t small_lists_lds_bytes(int capacity) { return ((size_t)capacity + 2 * (size_t)kSmallNewMax) * sizeof(int32_t); }

struct SmallShared {  // head of the dynamic LDS; the range image follows
  int2 scan[kSmallWaves];
  int waveTotal[kSmallWaves];
  int oldV, oldE, nPrev, overflowPrev;
  int nIds, idsValid;  // SceneP::allocIds: length, validity (read once by thread 0)
  int ghost, pad1;     // the visible list of this frame holds an entry WITHOUT a block (phase G)
  int box[4];  // store_range_image (k_raycast.h): the box of the image's non-empty cells
};
static_assert(sizeof(SmallShared) % sizeof(int2) == 0, "the range image behind it is an int2 array");
constexpr size_t small_lds_bytes(int nCells) { return sizeof(SmallShared) + (size_t)nCells * sizeof(int2); }
// lower bound in an ascending LDS array: number of elements < x
__device__ __forceinline__ int small_lower_bound(const int32_t *a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// (agent scope: served by L2, never allocates a line in this CU's L1 — the planes are also updated by L2 atomics of this very
//  workgroup, and the 16-byte sweep that follows must not find a line an earlier phase left in L1)
__device__ __forceinline__ uint32_t small_bit_word(const uint32_t *plane, int entry) {
  return __hip_atomic_load(plane + (entry >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Ordered compaction of the set bits of a plane of kSmallBitWords words into `ids` (ascending entry index), by the whole
// workgroup: wave w owns words [w * 2304, (w + 1) * 2304), a lane reads 4 consecutive words of each of the 9 rows of 256 —
// every load instruction of a wave is 1 KB of consecutive bytes.  CLEAR: the words are zeroed behind the sweep (visBits is
// per-frame scratch).  Returns the number of set bits; ids beyond `capacity` are dropped.
template <bool CLEAR>
__device__ __forceinline__ int small_sweep_bits(uint32_t *plane, int32_t *ids, int capacity, SmallShared &sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 *rows = reinterpret_cast<uint4 *>(plane) + wave * (kSmallRows * 64) + lane;
  int c[kSmallRows], inc[kSmallRows];
  {
    uint4 v[kSmallRows];
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) v[j] = rows[j * 64];
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) inc[j] = c[j] = __popc(v[j].x) + __popc(v[j].y) + __popc(v[j].z) + __popc(v[j].w);
  }
  // inclusive scans over the lanes, the nine rows side by side (independent shuffles: their latencies overlap)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o[kSmallRows];
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) o[j] = __shfl_up(inc[j], d);
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) if (lane >= d) inc[j] += o[j];
  }
  int run = 0;  // inc[j] becomes the lane's first rank within the wave: rows before + lanes before in its row
#pragma unroll
  for (int j = 0; j < kSmallRows; ++j) { const int rowTotal = __shfl(inc[j], 63); inc[j] += run - c[j]; run += rowTotal; }
  if (lane == 0) sh.waveTotal[wave] = run;
  __syncthreads();
  int waveOff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kSmallWaves; ++w) {
    const int t = sh.waveTotal[w];
    if (w < wave) waveOff += t;
    total += t;
  }
  __syncthreads();  // waveTotal may be written again by the next sweep
  // the few lanes that hold bits read their words again (from L1 / L2 now) instead of keeping 36 registers alive across the scans
#pragma unroll
  for (int j = 0; j < kSmallRows; ++j) {
    if (c[j] == 0) continue;
    const uint4 v = rows[j * 64];
    if (CLEAR) rows[j * 64] = make_uint4(0u, 0u, 0u, 0u);
    int rank = waveOff + inc[j];
    const int firstEntry = ((wave * (kSmallRows * 64) + j * 64 + lane) * 4) * 32;
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t w = words[k];
      while (w) {
        const int b = __ffs((int)w) - 1;
        w &= w - 1u;
        if (rank < capacity) ids[rank] = firstEntry + k * 32 + b;
        rank++;
      }
    }
  }
  return total;
}

// AllocateSceneFromDepth behind the per-pixel mark, for an instance-sized volume, + the live view's range image.
//   A  tile totals of the marks -> offsets, free-list heads (k_scan_tile_sums SCAN_ALLOC)
//   B  commit: a thread per sweep tile that holds marks ranks them into the ordered work list (k_alloc_commit)
//   C  apply: the work list densely (k_alloc_apply)
//   D0 the mark's touched groups -> bits of visBits, the types' "touched now" flag taken off again
//   D  the previous frame's visible entries the mark did not touch: frustum test, type 3 / 0 (k_retest_previous_visible)
//   E  only after a frame whose visible entries did not fit the list: the type sweep over the whole table (K0b's rare branch)
//   F  ordered compaction of visBits -> visibleEntryIDs, counts, published status (k_visible_count / scan / k_visible_write)
//   G  the visible-block stream + the range image (k_visible_write's gather, k_expected_depth_one)
// (the body, by one workgroup of kSmallThreads: the kernel below, or one volume's workgroup of k_batch_small_alloc_visible)
__device__ __forceinline__ void small_alloc_visible_body(const FrameP &p, const SceneP &s, const float *__restrict__ depth,
                                                         uint8_t *visType, int numTiles, int4 *workList, int32_t *visibleIDs,
                                                         int4 *visBlocks, int capacity, int32_t *__restrict__ publish,
                                                         int publishSeq, int2 *__restrict__ minmax, int32_t *__restrict__ rb,
                                                         int lists) {
  extern __shared__ int2 smallLds[];
  SmallShared &sh = *reinterpret_cast<SmallShared *>(smallLds);
  int2 *cells = smallLds + sizeof(SmallShared) / sizeof(int2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample, mh = (p.H + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const int nCells = mw * mh;
  const int farBits = __float_as_int(kFarAway), closeBits = __float_as_int(kVeryClose);
  SMALL_CLK(0);
  for (int c = tid; c < nCells; c += kSmallThreads) cells[c] = make_int2(farBits, closeBits);
  if (tid == 0) {
    sh.oldV = s.ctr[CTR_LAST_FREE_BLOCK]; sh.oldE = s.ctr[CTR_LAST_FREE_EXCESS];
    sh.nPrev = s.ctr[CTR_NO_VISIBLE_LIVE]; sh.overflowPrev = s.ctr[CTR_VIS_OVERFLOW];
    sh.nIds = s.ctr[CTR_NO_ALLOC_IDS]; sh.idsValid = lists ? s.ctr[CTR_ALLOC_IDS_VALID] : 0;
    sh.box[0] = sh.box[1] = 0x7fffffff; sh.box[2] = sh.box[3] = -1;
    sh.ghost = 0;
  }
  // ---- A
  int2 *allocTile = reinterpret_cast<int2 *>(s.allocTile);
  const int2 tv = (tid < numTiles) ? allocTile[tid] : make_int2(0, 0);  // {marked entries, excess-list ones among them} of tile `tid`
  int2 total;
  const int2 tileOff = wg_exclusive_scan2<kSmallThreads>(tv, total, sh.scan);  // (its barriers also publish sh.* and the image's reset)
  const int oldV = sh.oldV, oldE = sh.oldE;
  // the list path (uniform): the list is valid, every marked entry gets its block (an entry that stays without one is visible
  // but not allocated), the previous visible list was complete, the frame's new entries fit the merge
  const int nOld = sh.nIds;
  const bool fast = lists && sh.idsValid && !sh.overflowPrev && total.x <= oldV + 1 && total.y <= oldE + 1 &&
                    total.x <= kSmallNewMax && nOld + total.x <= capacity;
  if (tid == 0) {
    int32_t *ctr = s.ctr;
    ctr[CTR_ALLOC_OLD_HEAD_VBA] = oldV; ctr[CTR_ALLOC_OLD_HEAD_EXC] = oldE;
    ctr[CTR_ALLOC_TOTAL12] = total.x; ctr[CTR_ALLOC_TOTAL2] = total.y;
    const int nv = oldV - total.x, ne = oldE - total.y;
    ctr[CTR_LAST_FREE_BLOCK] = nv < -1 ? -1 : nv;
    ctr[CTR_LAST_FREE_EXCESS] = ne < -1 ? -1 : ne;
    if (total.x > oldV + 1 || total.y > oldE + 1) ctr[CTR_STATUS] = DSR_E_OUT_OF_BLOCKS;
  }
  SMALL_CLK(1);
  // ---- B: the tile's 64 group words (a byte per 8 entries) in four rounds of four 16-byte loads
  if (tv.x != 0) {
    allocTile[tid] = make_int2(0, 0);  // k_alloc_mark accumulates into it again
    int rank12 = tileOff.x, rank2 = tileOff.y;
    const int tileBase = tid * kTile;
    uint4 *grp4 = reinterpret_cast<uint4 *>(s.allocGrp + (tileBase >> 5));
#pragma unroll 1
    for (int q = 0; q < kTile / 32 / 16; ++q) {
      uint4 g4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) g4[r] = grp4[q * 4 + r];
      uint32_t any = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) any |= g4[r].x | g4[r].y | g4[r].z | g4[r].w;
      if (any == 0u) continue;
      // which of the 16 words hold marks (from the registers), then only those are read again, one by one: the loop runs as
      // often as the busiest lane of the wave has words, and the code stays small (no sixteen unrolled copies of the body)
      uint32_t nz = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        nz |= ((g4[r].x ? 1u : 0u) | (g4[r].y ? 2u : 0u) | (g4[r].z ? 4u : 0u) | (g4[r].w ? 8u : 0u)) << (r * 4);
      const uint32_t *words = s.allocGrp + (tileBase >> 5) + q * 16;
      while (nz) {
        const int wq = __ffs((int)nz) - 1;
        nz &= nz - 1u;
        const uint32_t word = words[wq];
#pragma unroll 1
        for (int gi = 0; gi < 4; ++gi) {
          if (((word >> (gi * 8)) & 15u) == 0u) continue;
          const int base = tileBase + (q * 16 + wq) * 32 + gi * 8;  // this group of 8 entries
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
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) grp4[q * 4 + r] = make_uint4(0u, 0u, 0u, 0u);  // ready for the next frame
    }
  }
  SMALL_CLK(2);
  // ---- D0: the groups of 8 entries the mark touched (a byte each in visGrp, 147 KB, nine 16-byte loads per lane as in the
  // bit sweep): the types that carry kTouchedNow become bits of visBits — the group's 8 bits are ONE byte of that plane, nobody
  // else writes it before the barrier, so a plain byte store does — and go back to the plain type 1.  A lane rarely owns more than
  // one marked group per row: the FIRST group's types of all nine rows are requested together (one round trip for the wave, not
  // nine), the rest goes through the loop behind.  (Not on the list path: the touched entries are found through the list, and a
  // group byte left set only makes a later sweep look at eight types that carry no flag.)
  if (!fast) {
    const int wave = tid >> 6;
    uint4 *rows = reinterpret_cast<uint4 *>(s.visGrp) + wave * (kSmallRows * 64) + lane;
    uint4 g[kSmallRows];
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) g[j] = rows[j * 64];
    auto settle = [&](int grp, uint2 t8) {
      uint32_t bits = 0u;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t &w = k < 4 ? t8.x : t8.y;
        const int sh8 = (k & 3) * 8;
        if (((w >> sh8) & 0xffu) == (uint32_t)kTouchedNow) { bits |= 1u << k; w = (w & ~(0xffu << sh8)) | (1u << sh8); }
      }
      *reinterpret_cast<uint2 *>(visType + (size_t)grp * 8) = t8;
      reinterpret_cast<uint8_t *>(s.visBits)[grp] = (uint8_t)bits;
    };
    uint32_t nz[kSmallRows];
    int first[kSmallRows];
    uint2 t8[kSmallRows];
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) {
      const uint32_t gw[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
      uint32_t m = 0u;
#pragma unroll
      for (int b = 0; b < 16; ++b) m |= ((gw[b >> 2] >> ((b & 3) * 8)) & 0xffu) ? (1u << b) : 0u;
      nz[j] = m;
      first[j] = m ? (wave * (kSmallRows * 64) + j * 64 + lane) * 16 + __ffs((int)m) - 1 : -1;
      t8[j] = make_uint2(0u, 0u);
      if (m) t8[j] = *reinterpret_cast<const uint2 *>(visType + (size_t)first[j] * 8);
    }
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) {
      if (nz[j] == 0u) continue;
      settle(first[j], t8[j]);
      uint32_t m = nz[j] & (nz[j] - 1u);
      const int firstGroup = (wave * (kSmallRows * 64) + j * 64 + lane) * 16;
      while (m) {
        const int grp = firstGroup + __ffs((int)m) - 1;
        m &= m - 1u;
        settle(grp, *reinterpret_cast<const uint2 *>(visType + (size_t)grp * 8));
      }
      rows[j * 64] = make_uint4(0u, 0u, 0u, 0u);  // ready for the next frame's mark
    }
  }
  __syncthreads();
  SMALL_CLK(3);
  // ---- C
  {
    const int avail = oldV + 1;
    const int n = total.x < avail ? total.x : avail;
    for (int i = tid; i < n; i += kSmallThreads) alloc_apply_item<true>(p, s, depth, workList[i], visType, fast);
  }
  SMALL_CLK(4);
  // ---- D (independent of C: an entry C creates was not visible before, an entry the mark touched is skipped)
  const int nPrev = fast ? 0 : sh.nPrev;
  for (int i = tid; i < nPrev; i += kSmallThreads) {
    const dsr_hash_entry he = entry_of_record(visBlocks[i]);  // the previous frame's stream
    const int t = he.offset;
    const uint32_t m = 1u << (t & 31);
    if (small_bit_word(s.visBits, t) & m) continue;  // type 1 / 2 from this frame's mark
    bool isVisible, isVisibleEnlarged;
    check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
    visType[t] = isVisible ? 3 : 0;
    if (isVisible) atomicOr(&s.visBits[t >> 5], m);
  }
  SMALL_CLK(5);
  // ---- E
  if (!fast && sh.overflowPrev) {
    __syncthreads();
    auto leftover = [&](int t, uint32_t v) {
      if (v == 0u) return;
      const uint32_t m = 1u << (t & 31);
      if (small_bit_word(s.visBits, t) & m) return;
      if (v == 3u) {  // an entry the list had no room for keeps its 3 and is re-tested like every type-3 entry of the serial sweep
        const dsr_hash_entry he = load_entry(s.table, (uint32_t)t);
        bool isVisible, isVisibleEnlarged;
        check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
        if (!isVisible) { visType[t] = 0; return; }
      }
      atomicOr(&s.visBits[t >> 5], m);
    };
    const uint4 *vt4 = reinterpret_cast<const uint4 *>(visType);
    const int n16 = p.noTotalEntries >> 4;
    for (int b0 = tid; b0 < n16; b0 += kSmallThreads * 4) {  // 16 types per load, four loads in flight
      uint4 q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int idx = b0 + r * kSmallThreads; q[r] = vt4[idx < n16 ? idx : n16 - 1]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = b0 + r * kSmallThreads;
        if (idx >= n16 || (q[r].x | q[r].y | q[r].z | q[r].w) == 0u) continue;
        const uint32_t w4[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
        for (int k = 0; k < 16; ++k) leftover(idx * 16 + k, (w4[k >> 2] >> ((k & 3) * 8)) & 0xffu);
      }
    }
    for (int t = (n16 << 4) + tid; t < p.noTotalEntries; t += kSmallThreads) leftover(t, visType[t]);
  }
  __syncthreads();
  SMALL_CLK(6);
  int totalVisible = 0;
  if (fast) {
    // ---- M: this frame's new entries (the work list: in-place entries as they are, a chain append's CHILD) merged into the
    // sorted list.  LDS scratch over the range image's cells, which are reset again behind it.
    const int nNew = total.x;
    if (nNew > 0) {
      int32_t *oldL = reinterpret_cast<int32_t *>(cells), *newRaw = oldL + capacity, *newSorted = newRaw + kSmallNewMax;
      for (int i = tid; i < nOld; i += kSmallThreads) oldL[i] = s.allocIds[i];
      for (int j = tid; j < nNew; j += kSmallThreads) {
        const int4 w = workList[j];
        newRaw[j] = w.w < 0 ? w.x : p.noBuckets + s.excessAllocList[w.w];
      }
      __syncthreads();
      for (int j = tid; j < nNew; j += kSmallThreads) {  // rank among the new ones (all distinct)
        const int x = newRaw[j];
        int r = 0;
        for (int k = 0; k < nNew; ++k) r += newRaw[k] < x ? 1 : 0;
        newSorted[r] = x;
      }
      __syncthreads();
      for (int j = tid; j < nNew; j += kSmallThreads) {
        const int x = newSorted[j];
        s.allocIds[small_lower_bound(oldL, nOld, x) + j] = x;
      }
      for (int i = tid; i < nOld; i += kSmallThreads) {
        const int x = oldL[i];
        const int r = small_lower_bound(newSorted, nNew, x);
        if (r) s.allocIds[i + r] = x;
      }
      if (tid == 0) s.ctr[CTR_NO_ALLOC_IDS] = nOld + nNew;
      __syncthreads();
      for (int c = tid; c < nCells; c += kSmallThreads) cells[c] = make_int2(farBits, closeBits);
      __syncthreads();
    }
    // ---- H: one dense pass over the allocated entries, in entry order: visible iff this frame's mark touched it, or it was
    // visible in the previous frame (its type is still 1 / 3) and passes the frustum test (types as phases D0 / D leave them);
    // ordered compaction -> visibleEntryIDs + the stream, the range image folded on the way (phases F + G)
    const int nAll = nOld + nNew;
    int carry = 0;
    for (int base = 0; base < nAll; base += kSmallThreads) {  // uniform trip count
      const int i = base + tid;
      bool vis = false;
      int t = 0;
      int4 raw = make_int4(0, 0, 0, -2);
      if (i < nAll) {
        t = s.allocIds[i];
        raw = *reinterpret_cast<const int4 *>(s.table + t);
        const uint8_t ty = visType[t];
        if (ty == kTouchedNow) { vis = true; visType[t] = 1; }
        else if (ty != 0) {
          const dsr_hash_entry he = entry_of_record(raw);
          bool isVisible, isVisibleEnlarged;
          check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
          vis = isVisible;
          visType[t] = isVisible ? 3 : 0;
        }
      }
      int2 tot;
      const int2 ex = wg_exclusive_scan2<kSmallThreads>(make_int2(vis ? 1 : 0, 0), tot, sh.scan);
      const int rank = carry + ex.x;
      bool valid = false;
      int2 ul = make_int2(0, 0), lr = make_int2(-1, -1);
      float2 zr = make_float2(0.f, 0.f);
      if (vis && rank < capacity) {
        const int4 rec = make_vis_record(raw, t);
        visibleIDs[rank] = t;
        visBlocks[rank] = rec;
        const dsr_hash_entry he = entry_of_record(rec);
        if (he.ptr >= 0) valid = project_single_block<DeviceOps>(he.pos, p, mw, mh, ul, lr, zr);
      }
      fold_wave_boxes(cells, mw, valid, ul, lr, zr, lane);
      carry += tot.x;
    }
    totalVisible = carry;
  } else {
    // ---- F
    totalVisible = small_sweep_bits<true>(s.visBits, visibleIDs, capacity, sh);
    if (lists) {  // ... and the sorted list for the frames to come, from the bits (final since phase C)
      __syncthreads();
      const int totalAlloc = small_sweep_bits<false>(s.allocBits, s.allocIds, capacity, sh);
      if (tid == 0) { s.ctr[CTR_NO_ALLOC_IDS] = totalAlloc < capacity ? totalAlloc : capacity; s.ctr[CTR_ALLOC_IDS_VALID] = totalAlloc <= capacity ? 1 : 0; }
    }
  }
  const int n = totalVisible < capacity ? totalVisible : capacity;
  if (tid == 0) {
    int32_t *ctr = s.ctr;
    ctr[CTR_NO_VISIBLE_LIVE] = n;
    ctr[CTR_VIS_OVERFLOW] = totalVisible > capacity ? 1 : 0;
    ctr[CTR_ALLOC_OLD_HEAD_VBA] = ctr[CTR_LAST_FREE_BLOCK];  // (SCAN_VISIBLE_LIVE's bookkeeping; no swapped-out entries here)
    if (publish) {  // k_visible_write's hand-over of {noVisibleBlocks, status} to a host that polls
      publish[0] = n;
      publish[1] = ctr[CTR_STATUS];
      __hip_atomic_store(publish + 2, publishSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (n > 0) atomicAdd(&s.work[WORK_V_EXPECTED], (unsigned long long)n);
  }
  __syncthreads();
  SMALL_CLK(7);
  // ---- G
  if (!fast)
    for (int base = tid & ~63; base < n; base += kSmallThreads) {  // wave-uniform trip count
      const int i = base + lane;
      bool valid = false;
      int2 ul = make_int2(0, 0), lr = make_int2(-1, -1);
      float2 zr = make_float2(0.f, 0.f);
      if (i < n) {
        const int t = visibleIDs[i];
        const int4 raw = *reinterpret_cast<const int4 *>(s.table + t);
        const int4 rec = make_vis_record(raw, t);
        visBlocks[i] = rec;
        const dsr_hash_entry he = entry_of_record(rec);
        if (he.ptr >= 0) valid = project_single_block<DeviceOps>(he.pos, p, mw, mh, ul, lr, zr);
        else sh.ghost = 1;
      }
      fold_wave_boxes(cells, mw, valid, ul, lr, zr, lane);
    }
  __syncthreads();
  // An entry that is visible WITHOUT owning a block (the mark named it as an allocation target while the block array was
  // exhausted; upstream keeps it visible, and re-tests it like every other entry of the list in the frames that follow) is not in
  // the sorted list of allocated entries, so phase H would never look at it again: while the visible list holds one, the list
  // path stays off (found by tests/test_gpu_fuzz.py, seeds 188 and 398)
  if (!fast && lists && tid == 0 && sh.ghost) s.ctr[CTR_ALLOC_IDS_VALID] = 0;
  if (n <= 0) return;  // Prepare() is skipped without visible blocks: the image keeps its previous contents
  SMALL_CLK(8);
  store_range_image(cells, minmax, nCells, mw, rb, sh.box);
  SMALL_CLK(9);
}

__global__ __launch_bounds__(kSmallThreads) void k_small_alloc_visible(FrameP p, SceneP s, const float *__restrict__ depth,
                                                                       uint8_t *visType, int numTiles, int4 *workList,
                                                                       int32_t *visibleIDs, int4 *visBlocks, int capacity,
                                                                       int32_t *__restrict__ publish, int publishSeq,
                                                                       int2 *__restrict__ minmax, int32_t *__restrict__ rb, int lists) {
  small_alloc_visible_body(p, s, depth, visType, numTiles, workList, visibleIDs, visBlocks, capacity, publish, publishSeq, minmax, rb,
                           lists);
}

// FindVisibleBlocks + CreateExpectedDepths of a free camera for an instance-sized volume: the allocated entries come from
// allocBits (ascending), are tested against the frustum densely, compacted in order; the range image is folded on the way.
__device__ __forceinline__ void small_freeview_body(const FrameP &p, const SceneP &s, int32_t *stage, int32_t *__restrict__ visibleIDs,
                                                    int4 *__restrict__ visBlocks, int capacity, int2 *__restrict__ minmax,
                                                    int32_t *__restrict__ rb, int lists) {
  extern __shared__ int2 smallLds[];
  SmallShared &sh = *reinterpret_cast<SmallShared *>(smallLds);
  int2 *cells = smallLds + sizeof(SmallShared) / sizeof(int2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int mw = (p.W + kMinmaxSubsample - 1) / kMinmaxSubsample, mh = (p.H + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const int nCells = mw * mh;
  const int farBits = __float_as_int(kFarAway), closeBits = __float_as_int(kVeryClose);
  SMALL_CLK(16);
  for (int c = tid; c < nCells; c += kSmallThreads) cells[c] = make_int2(farBits, closeBits);
  if (tid == 0) {
    sh.box[0] = sh.box[1] = 0x7fffffff; sh.box[2] = sh.box[3] = -1;
    sh.nIds = s.ctr[CTR_NO_ALLOC_IDS]; sh.idsValid = lists ? s.ctr[CTR_ALLOC_IDS_VALID] : 0;
  }
  __syncthreads();
  // the allocated entries in ascending order: the list the commit keeps (k_small_alloc_visible), or — while it is not valid —
  // the bits swept into `stage`
  int nAlloc;
  if (sh.idsValid) { nAlloc = sh.nIds; stage = s.allocIds; }
  else {
    const int totalAlloc = small_sweep_bits<false>(s.allocBits, stage, capacity, sh);
    nAlloc = totalAlloc < capacity ? totalAlloc : capacity;
    __syncthreads();  // the ids
  }
  SMALL_CLK(17);
  int carry = 0;
  for (int base = 0; base < nAlloc; base += kSmallThreads) {  // uniform trip count
    const int i = base + tid;
    bool vis = false;
    int t = 0;
    int4 raw = make_int4(0, 0, 0, -2);
    if (i < nAlloc) {
      t = stage[i];
      raw = *reinterpret_cast<const int4 *>(s.table + t);
      if (raw.w >= 0) {
        const dsr_hash_entry he = entry_of_record(raw);
        bool isVisible, isVisibleEnlarged;
        check_block_visibility<false>(isVisible, isVisibleEnlarged, he.pos, p.M, p.proj, p.voxelSize, p.W, p.H);
        vis = isVisible;
      }
    }
    int2 tot;
    const int2 ex = wg_exclusive_scan2<kSmallThreads>(make_int2(vis ? 1 : 0, 0), tot, sh.scan);
    const int rank = carry + ex.x;
    bool valid = false;
    int2 ul = make_int2(0, 0), lr = make_int2(-1, -1);
    float2 zr = make_float2(0.f, 0.f);
    if (vis && rank < capacity) {
      const int4 rec = make_vis_record(raw, t);
      visibleIDs[rank] = t;
      visBlocks[rank] = rec;
      const dsr_hash_entry he = entry_of_record(rec);
      valid = project_single_block<DeviceOps>(he.pos, p, mw, mh, ul, lr, zr);
    }
    fold_wave_boxes(cells, mw, valid, ul, lr, zr, lane);
    carry += tot.x;
  }
  SMALL_CLK(18);
  if (tid == 0) {
    const int n = carry < capacity ? carry : capacity;
    s.ctr[CTR_NO_VISIBLE_FREE] = n;
    atomicAdd(&s.work[WORK_V_EXPECTED], (unsigned long long)n);
  }
  __syncthreads();
  SMALL_CLK(19);
  store_range_image(cells, minmax, nCells, mw, rb, sh.box);
  SMALL_CLK(20);
}
__global__ __launch_bounds__(kSmallThreads) void k_small_freeview(FrameP p, SceneP s, int32_t *stage, int32_t *__restrict__ visibleIDs,
                                                                  int4 *__restrict__ visBlocks, int capacity,
                                                                  int2 *__restrict__ minmax, int32_t *__restrict__ rb, int lists) {
  small_freeview_body(p, s, stage, visibleIDs, visBlocks, capacity, minmax, rb, lists);
}

}  // namespace dsr
