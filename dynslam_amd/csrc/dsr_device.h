// dsr_device.h — shared device-side types and helpers of the HIP engine (gfx950).
//
// Numerics contract (DESIGN.md "bit-exactness"): every floating point expression
// is evaluated in the order of the upstream InfiniTAM-v2 shared engine headers,
// the TU is compiled with -ffp-contract=off (no FMA contraction) and HIP's default
// correctly rounded fp32 divide/sqrt, so results equal the x86-64 _CPU engines bit
// for bit.  Out-of-range float->int conversions use the saturating hardware
// conversion (v_cvt_i32_f32), which is the adopted definition on both sides.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dsr.h"

namespace dsr {

constexpr int kBlockSize = DSR_BLOCK_SIZE;    // 8
constexpr int kBlockSize3 = DSR_BLOCK_SIZE3;  // 512
constexpr int kBlockBytes = 4096;             // one voxel block in HBM

// Plane-wise voxel block layout in HBM (4096 B per block, DESIGN.md "HBM layout"):
//   [   0,1024) int16  sdf[512]
//   [1024,1536) uint8  w_depth[512]
//   [1536,2048) unused
//   [2048,4096) uchar4 clr[512]   (r,g,b,w_color): colour and its weight in ONE word — the colour update of
//               a voxel reads and writes a single 4-byte word (one cache line per voxel instead of two:
//               the colour phase touches lines sparsely, every line it does not touch is traffic saved)
constexpr int kOffSdf = 0;
constexpr int kOffWDepth = 1024;
constexpr int kOffClr = 2048;

constexpr float kFarAway = 999999.9f;  // FAR_AWAY
constexpr float kVeryClose = 0.05f;    // VERY_CLOSE
constexpr int kMinmaxSubsample = 8;    // minmaximg_subsample

// device-resident counters (int32 each), never read by the host on the hot path
enum Ctr {
  CTR_LAST_FREE_BLOCK = 0,    // scene->localVBA.lastFreeBlockId
  CTR_LAST_FREE_EXCESS = 1,   // lastFreeExcessListId
  CTR_NO_VISIBLE_LIVE = 2,    // renderState_live->noVisibleBlocks
  CTR_NO_VISIBLE_FREE = 3,    // renderState_freeview->noVisibleBlocks
  CTR_STATUS = 4,             // sticky dsr_status
  CTR_ALLOC_OLD_HEAD_VBA = 5, // allocation context of the running frame
  CTR_ALLOC_OLD_HEAD_EXC = 6,
  CTR_ALLOC_TOTAL12 = 7,
  CTR_ALLOC_TOTAL2 = 8,
  CTR_DECAY_FREED = 9,        // blocks freed by the running decay call
  CTR_DECAY_NCAND = 10,       // candidates of the running decay call
  CTR_TMP_OLD_NVIS = 11,      // live visible count before the post-decay compaction
  CTR_SWAP_COUNT = 12,        // blocks in the running swap-in / swap-out transfer
  CTR_MESH_TOTAL = 13,        // triangles of the running MeshScene
  CTR_HOST_USED = 14,         // slots of the host store (ITMGlobalCache) handed out so far
  CTR_SWAP_FIRST_SLOT = 15,   // first host slot of the running swap-out batch
  CTR_NO_ALLOCATED = 16,      // length of the cached list of allocated entries (free-view culling)
  CTR_VIS_OVERFLOW = 17,      // the live visible list was cut at its capacity: entries of type 3 exist outside the list (k_alloc.h K0b)
  CTR_NO_ALLOC_IDS = 18,      // instance-sized volumes: length of SceneP::allocIds ...
  CTR_ALLOC_IDS_VALID = 19,   // ... and whether it lists exactly the entries that own a block (k_small.h; the voxel GC clears it)
  CTR_COUNT = 32
};
// device-resident 64-bit work counters (roofline bookkeeping + decayed count)
enum Work {
  WORK_V_INTEGRATED = 0,  // sum of noVisibleBlocks over integrate launches
  WORK_V_EXPECTED = 1,    // ... over expected-depth launches
  WORK_DECAYED_BLOCKS = 2,
  WORK_V_DECAY = 3,
  WORK_COUNT = 8
};

struct Mat4 { float m[16]; };  // ORUtils::Matrix4f, column-major

struct FrameP {
  Mat4 M;        // world->camera of the view being processed (M_d or free camera)
  Mat4 invM;     // its inverse (ORUtils cofactor inverse, computed on the host)
  Mat4 M_rgb;    // calib_inv * M_d
  float4 proj;   // fx, fy, cx, cy (depth / free camera)
  float4 proj_rgb;
  float mu, voxelSize;
  float vfMin, vfMax;
  int W, H, Wr, Hr;
  int maxW, stopAtMaxW, depthWeighting, rgbSame;
  int noBuckets, noExcess, noTotalEntries, noBlocks;
  uint32_t hashMask;
  uint32_t maxSteps;  // bound on noSteps of the allocation ray walk
  int useSwapping;
};

struct SceneP {
  dsr_hash_entry *table;
  uint8_t *vba;           // noBlocks * 4096 bytes
  int32_t *voxelAllocList;
  int32_t *excessAllocList;
  int32_t *ctr;           // Ctr
  unsigned long long *work;  // Work
  uint32_t *allocKey;     // per entry: 0 or (pixel*maxSteps + step + 1) of the winning writer
  uint32_t *allocGrp;     // per 8 entries one byte: marked entries (low nibble), of those excess-list ones (high)
  unsigned long long *allocTile;  // per sweep tile: marked entries (low 32 bits) | excess-list ones (high)
  uint8_t *swapState;     // ITMHashSwapState::state per entry (null unless use_swapping)
  uint8_t *swapStored;    // 1 = the host store (ITMGlobalCache) holds a copy of this entry's block
  int32_t *swapSlot;      // ... in this slot of it (most recent copy)
  uint8_t **hostSlabs;    // device-visible table of the pinned host slabs (slabBlocks blocks each)
  int slabBlocks;
  // instance-sized volumes only (k_small.h; null otherwise): one bit per entry, kSmallBitWords words each
  uint8_t *visGrp;        // one BYTE per 8 entries: the running frame's mark touched an entry of the group (plain stores, no atomics)
  uint32_t *visBits;      // one BIT per entry: visible in the running frame — built and cleared by the list kernel itself
  uint32_t *allocBits;    // entries that own a voxel block (ptr >= 0): set by the commit, cleared by the voxel GC
  int32_t *allocIds;      // the same set as an ASCENDING list of entry indices (ctr[CTR_NO_ALLOC_IDS] long, valid while
                          // ctr[CTR_ALLOC_IDS_VALID]): merged by the commit, rebuilt from allocBits after the GC (k_small.h, round 6)
};

// ------------------------------------------------------------------ conversions

__device__ __forceinline__ int f2i(float f) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));  // saturating, NaN -> 0
  return r;
}
__device__ __forceinline__ short f2s(float f) { return (short)f2i(f); }

// ---- correctly rounded division for "tame" operands ---------------------------------------
// hipcc lowers an IEEE fp32 a/b to: v_div_scale x2, v_rcp, 2 fma (Newton step on the
// reciprocal), mul, 4 fma (two residual corrections of the quotient), v_div_fmas, v_div_fixup.
// The scale / fmas / fixup instructions only act when the divisor or the quotient leaves the
// normal range or an operand is inf/NaN/0-divisor.  For operands that are finite with
// |b| in [2^-60, 2^60] and a quotient that is zero or normal ("tame": every division on the
// integrate / raycast paths, DESIGN.md "division"), the remaining arithmetic IS the same
// sequence, so the result is the same correctly rounded quotient — at 8 instead of 11
// instructions, and the refined reciprocal can be shared between divisions by one divisor.
// dsr_selftest_division() checks the equivalence against `/` on the GPU.
__device__ __forceinline__ float rcp_refined(float b) {
  float y = __builtin_amdgcn_rcpf(b);
  float e = __builtin_fmaf(-b, y, 1.0f);
  return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float div_with_rcp(float a, float b, float y1) {
  float q = a * y1;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y1, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y1, q);
}
__device__ __forceinline__ float fdiv_tame(float a, float b) { return div_with_rcp(a, b, rcp_refined(b)); }

// Division by a divisor whose CORRECTLY ROUNDED reciprocal y = RN(1/b) is at hand (constants
// 32767 and 255, mu, the integer weights 1..256): one residual correction gives the correctly
// rounded quotient.  Unlike the two-correction sequence this is not the compiler's own lowering
// of `/`, so it is used only for divisors for which the equivalence with `/` has been checked
// over all 2^23 numerator mantissas (division is scale invariant): the constants and the
// weight table in dsr_selftest_division, mu at engine creation (dsr_engine.hip
// short_division_exact; a failing mu selects the PLAIN = false integrate kernels).
// Precondition beyond "tame": a is not -0 (the sequence returns +0 for it; `/` returns -0).
// Every call site divides a converted integer, a difference of finite positive floats, a running
// mean of such or a trilinear combination of converted shorts, none of which can be -0.
__host__ __device__ __forceinline__ float div_short(float a, float b, float yRN) {
  const float q = a * yRN;
  const float r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, yRN, q);
}

__host__ __device__ __forceinline__ float sdf_to_float(float v) { return v / 32767.0f; }
// same value for every v that is not -0 (div_short)
__host__ __device__ __forceinline__ float sdf_to_float_short(float v) { return div_short(v, 32767.0f, 1.0f / 32767.0f); }
__device__ __forceinline__ short sdf_from_float(float f) { return (short)f2i(f * 32767.0f); }

// The per-pixel / per-block functions of the range-image, visibility, raycast and shading kernels are templates over an Ops
// policy — how to convert float -> int, round down / up, divide by a tame divisor, ask "any ray of the wave" —: the device's
// instructions here, a one-ray host stand-in in tests/hostsim, under which tests/test_device_functions_host.py runs those very functions
// on the CPU against the oracle.
struct DeviceOps {
  static __device__ __forceinline__ int f2i(float f) { return dsr::f2i(f); }
  static __device__ __forceinline__ bool any(bool b) { return __any(b) != 0; }
  static __device__ __forceinline__ float sqrt(float f) { return sqrtf(f); }
  static __device__ __forceinline__ float floor(float f) { return floorf(f); }
  static __device__ __forceinline__ float ceil(float f) { return ceilf(f); }
  // a / b for "tame" operands (dsr_device.h): the refined reciprocal of b, shared by the divisions by one divisor
  static __device__ __forceinline__ float rcp(float b) { return rcp_refined(b); }
  static __device__ __forceinline__ float div(float a, float b, float y) { return div_with_rcp(a, b, y); }
};

// ORUtils Matrix4 * Vector4, the three rows the kernels use (w explicit)
__host__ __device__ __forceinline__ float3 mat_mul3(const Mat4 &a, float x, float y, float z, float w) {
  float3 r;
  r.x = a.m[0] * x + a.m[4] * y + a.m[8] * z + a.m[12] * w;
  r.y = a.m[1] * x + a.m[5] * y + a.m[9] * z + a.m[13] * w;
  r.z = a.m[2] * x + a.m[6] * y + a.m[10] * z + a.m[14] * w;
  return r;
}

__host__ __device__ __forceinline__ uint32_t hash_index(int bx, int by, int bz, uint32_t mask) {
  return (((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349669u) ^ ((uint32_t)bz * 83492791u)) & mask;
}

// 16-byte load of one hash entry
__host__ __device__ __forceinline__ dsr_hash_entry load_entry(const dsr_hash_entry *table, uint32_t idx) {
  int4 raw = *reinterpret_cast<const int4 *>(table + idx);
  dsr_hash_entry e;
  e.pos[0] = (short)(raw.x & 0xffff);
  e.pos[1] = (short)((uint32_t)raw.x >> 16);
  e.pos[2] = (short)(raw.y & 0xffff);
  e._pad = 0;
  e.offset = raw.z;
  e.ptr = raw.w;
  return e;
}

// ---- the visible-block STREAM ------------------------------------------------------------------------------------
// The kernels that walk a visible list (integrate, expected depths, next frame's frustum re-test) used to follow every
// list id into the hash table: ~610 k scattered 16-byte reads per kernel, each costing a 64-128 B line (62 + 58 + 40 MB
// fetched for 10 MB of entries, profiles/r02f).  k_visible_write — which walks the table in ascending order anyway — now
// also writes one 16-byte record per visible entry, in list order: the entry's pos and ptr with the ENTRY INDEX in the
// slot of `offset` (no consumer of the list needs the chain link).  The consumers read this stream coalesced; the table
// is gathered once per frame instead of three times.  Kept consistent by every kernel that rewrites a visible list
// (k_visible_write, k_flag_write, the post-decay compaction k_live_keep_write).
__device__ __forceinline__ int4 make_vis_record(int4 rawEntry, int entryIdx) { return make_int4(rawEntry.x, rawEntry.y, entryIdx, rawEntry.w); }
__device__ __forceinline__ dsr_hash_entry entry_of_record(int4 r) {  // .offset holds the entry index
  dsr_hash_entry e;
  e.pos[0] = (short)(r.x & 0xffff); e.pos[1] = (short)((uint32_t)r.x >> 16); e.pos[2] = (short)(r.y & 0xffff);
  e._pad = 0; e.offset = r.z; e.ptr = r.w;
  return e;
}

// ------------------------------------------------------- workgroup-ordered scans

// Exclusive scan of an int2 over a workgroup of NT threads (NT multiple of 64).
// wave64 shuffles inside a wave, LDS across waves.  lds must hold NT/64 int2.
template <int NT>
__device__ __forceinline__ int2 wg_exclusive_scan2(int2 v, int2 &total, int2 *lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int2 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int ox = __shfl_up(inc.x, d), oy = __shfl_up(inc.y, d);
    if (lane >= d) { inc.x += ox; inc.y += oy; }
  }
  if (lane == 63) lds[wid] = inc;
  __syncthreads();
  int2 woff = make_int2(0, 0), tot = make_int2(0, 0);
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    int2 s = lds[w];
    if (w < wid) { woff.x += s.x; woff.y += s.y; }
    tot.x += s.x; tot.y += s.y;
  }
  __syncthreads();
  total = tot;
  return make_int2(woff.x + inc.x - v.x, woff.y + inc.y - v.y);
}

constexpr int kTileThreads = 256;
constexpr int kTileItems = 8;
constexpr int kTile = kTileThreads * kTileItems;  // entries per workgroup in table sweeps

}  // namespace dsr
