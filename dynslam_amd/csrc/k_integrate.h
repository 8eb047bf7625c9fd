// k_integrate.h — K4: SDF / colour integration of every visible voxel block.
//
// Replaces integrateIntoScene_device (upstream: one 512-thread CUDA block per voxel block,
// 8 B array-of-structs voxels).  CDNA4 formulation:
//   * A WAVE64 PER BLOCK, 8 voxels along x per lane (VOX = 8; 71 VGPRs, 7 waves per SIMD: the
//     default), or per HALF block (VOX = 4: 4 z-slices = 256 voxels per wave, 8 waves per SIMD).
//     With the plane-wise block layout (dsr_device.h) a lane moves 2*VOX B of sdf and VOX B of
//     w_depth in fully coalesced accesses (lane l owns voxels [VOX*l, VOX*l+VOX) of its task).
//   * PHASE A1 (project): branch-free; all depth-image gathers of a lane are issued before any is
//     consumed.  A task none of whose voxels passes the depth tests ends right after it.
//   * PHASE A2 (depth): branch-free SDF running mean.  Voxels that also pass the colour gate
//     (|eta/mu| <= 0.25: a thin sheet, ~2 % of the visible voxels) are appended to a per-wave LDS
//     list {task, voxel} — one word per voxel — (wave64 ballot + prefix popcount) that PERSISTS
//     ACROSS TASKS.
//   * PHASE B (colour): whenever the list holds 64 voxels the wave updates them DENSELY, one per
//     lane: re-reads the voxel's hash entry, re-projects it, gathers its 4 B colour + 1 B weight,
//     bilinear RGB sample, running mean, scatter back.  The divergent colour branch of the per-voxel formulation (every
//     lane paying ~150 instructions for the few that need it, once per task) is paid once per 64
//     colour voxels instead, and the colour planes of untouched voxels are never read.
//   * A persistent grid strides over the visible list whose length is read from device memory
//     (the host never synchronises to learn noVisibleBlocks); hash entries are fetched two tasks
//     ahead and voxel planes one task ahead of the arithmetic.
//   * The kernel is VALU-issue bound (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = 98 % of its duration),
//     not HBM bound, so instruction count is what is optimised: divisions use the
//     shared-reciprocal form of the IEEE sequence (dsr_device.h "correctly rounded division for
//     tame operands"), and for divisors with a correctly rounded reciprocal at hand (mu, 32767,
//     255, the integer weights) the one-correction form div_short below.
//   * The scalar unit is shared by the 4 SIMDs of a CU: per-voxel branching (exec-mask
//     bookkeeping) made it a bottleneck, hence selects instead of nested ifs.
//
// Arithmetic follows ITMSceneReconstructionEngine.h computeUpdatedVoxelDepthInfo /
// computeUpdatedVoxelColorInfo / ComputeUpdatedVoxelInfo<true> expression by expression.
#pragma once
#include <type_traits>

#include "dsr_device.h"

namespace dsr {

// ITMPixelUtils.h interpolateBilinear on the RGBA frame (uchar4 gathers)
__device__ __forceinline__ float3 bilinear_rgb(const uchar4 *__restrict__ src, float px, float py, int W) {
  const int ix = f2i(floorf(px)), iy = f2i(floorf(py));
  const float dx = px - (float)ix, dy = py - (float)iy;
  // upstream loads b/c/d only when their weight is non-zero; loading them always gives the same
  // value (weight 0 times a finite byte is +0 either way), the callers guarantee 1 <= p <= dim-2
  // so the 2x2 footprint is inside the image, and the four gathers are issued together
  const uchar4 a = src[ix + iy * W];
  const uchar4 b = src[(ix + 1) + iy * W];
  const uchar4 c = src[ix + (iy + 1) * W];
  const uchar4 d = src[(ix + 1) + (iy + 1) * W];
  float3 r;
  r.x = ((float)a.x * (1.0f - dx) * (1.0f - dy) + (float)b.x * dx * (1.0f - dy) + (float)c.x * (1.0f - dx) * dy + (float)d.x * dx * dy);
  r.y = ((float)a.y * (1.0f - dx) * (1.0f - dy) + (float)b.y * dx * (1.0f - dy) + (float)c.y * (1.0f - dx) * dy + (float)d.y * dx * dy);
  r.z = ((float)a.z * (1.0f - dx) * (1.0f - dy) + (float)b.z * dx * (1.0f - dy) + (float)c.z * (1.0f - dx) * dy + (float)d.z * dx * dy);
  return r;
}

// fork: WeightParams.depthWeighting; adopted definition max(1, round(10/z))
__device__ __forceinline__ int depth_weight(float depth_measure) {
  int w = f2i(10.0f / depth_measure + 0.5f);
  return w < 1 ? 1 : w;
}

constexpr int kIntegrateWaves = 4;  // waves (= tasks in flight) per workgroup

// the sdf / w_depth words a lane owns: VOX voxels = VOX/2 sdf words + VOX/4 weight words
template <int VOX>
struct LanePlanes {
  uint32_t sdf[VOX / 2];
  uint32_t wd[VOX / 4];
};
template <int VOX>
__device__ __forceinline__ LanePlanes<VOX> load_planes(const uint8_t *blk, int vox0) {
  LanePlanes<VOX> r;
  if (VOX == 8) {
    const uint4 a = *reinterpret_cast<const uint4 *>(blk + kOffSdf + vox0 * 2);
    const uint2 b = *reinterpret_cast<const uint2 *>(blk + kOffWDepth + vox0);
    r.sdf[0] = a.x; r.sdf[1] = a.y; r.sdf[VOX / 2 - 2] = a.z; r.sdf[VOX / 2 - 1] = a.w;
    r.wd[0] = b.x; r.wd[VOX / 4 - 1] = b.y;
  } else {
    const uint2 a = *reinterpret_cast<const uint2 *>(blk + kOffSdf + vox0 * 2);
    r.sdf[0] = a.x; r.sdf[VOX / 2 - 1] = a.y;
    r.wd[0] = *reinterpret_cast<const uint32_t *>(blk + kOffWDepth + vox0);
  }
  return r;
}
template <int VOX>
__device__ __forceinline__ void store_planes(uint8_t *blk, int vox0, const LanePlanes<VOX> &r) {
  if (VOX == 8) {
    *reinterpret_cast<uint4 *>(blk + kOffSdf + vox0 * 2) = make_uint4(r.sdf[0], r.sdf[1], r.sdf[VOX / 2 - 2], r.sdf[VOX / 2 - 1]);
    *reinterpret_cast<uint2 *>(blk + kOffWDepth + vox0) = make_uint2(r.wd[0], r.wd[VOX / 4 - 1]);
  } else {
    *reinterpret_cast<uint2 *>(blk + kOffSdf + vox0 * 2) = make_uint2(r.sdf[0], r.sdf[VOX / 2 - 1]);
    *reinterpret_cast<uint32_t *>(blk + kOffWDepth + vox0) = r.wd[0];
  }
}

// PLAIN: depth weighting and stopIntegratingAtMaxW are both off (the defaults): the flags become
// compile-time constants, their selects / the extra division disappear from the voxel loop and
// the divisions by mu, 32767 and the new weight take the one-correction form.
// OCC: waves per SIMD the register allocator must allow.
// Formulation notes (round 2; each measured alone and in combination on the bench workload with
// tools/bench_variants.py, identical results in every case — DESIGN.md "integrate variants"):
//   kept    depth gathers as RAW BUFFER loads (32-bit offsets; the hardware range check returns 0 =
//           "no depth" for lanes that are out of the image, so neither a clamped index nor a remembered
//           in-bounds mask is needed); the camera-plane fallback triggered by "some lane is not tame"
//           instead of a per-voxel (z > 0) compare; the (eta > mu) half of the colour gate dropped
//           (implied by |eta / mu| > 0.25 for a correctly rounded quotient): together 623 -> 587 us;
//   dropped no select of the divisor for lanes behind the camera plane (709 us: inf / NaN operands in
//           the division sequence are slow), image bounds as unsigned compares on the float bits
//           (no change), colour list appended once per task from a per-lane bit mask + wave prefix
//           sum (688 us: six dependent ds_bpermute), skipping x-slices without updates (no change),
//           5 / 6 instead of 7 waves per SIMD (no change).
// (Round 3, measured and dropped: copying the 13 uniform operands of the per-voxel arithmetic — first matrix column and
//  translation, projection, voxel size, mu and its reciprocal — into VECTOR registers, because tools/ubench times a full-rate
//  VALU instruction with an SGPR operand at half rate and the voxel loop holds ~110 of them per task: 78 VGPRs = 6 waves per
//  SIMD, 546 / 561 us against 559 / 559 us — nothing, profiles/r03g_integrate_vreg_variants.log.  With the row products
//  hoisted by hand (to_camera below) the kernel needs 65 VGPRs; compiled for 8 waves per SIMD (64 VGPRs, 35 SGPR spills) it
//  is slower: 576-584 vs 557-560 us, profiles/r03h_integrate_occ8_variants.log.)
// XLDS (round 4): with VOX = 8 a lane's eight voxels are the block's eight x positions, the same for every lane of the wave: the
// x terms of the camera transform — (float)(gx + x) * voxelSize and its three products with the first matrix column, five VALU
// instructions per voxel on wave-uniform values — are computed ONCE per task by lanes 0..7, parked in LDS and read back by every
// lane with one broadcast ds_read_b128 per voxel: the LDS port instead of the vector ALU this kernel is bound by (35 of ~695
// instructions per task).  The values are the same products of the same operands: bit-identical.
// (the body, as workgroup `blockId` of `numBlocks`: k_integrate below, and one volume's share of k_batch_integrate, k_batch.h)
template <bool RGB_SAME, bool PLAIN, int VOX, bool XLDS>
__device__ __forceinline__ void integrate_body(const FrameP &p, const SceneP &s, const float *__restrict__ depth,
                                               const uchar4 *__restrict__ rgb, const int4 *__restrict__ visBlocks,
                                               uint2 *__restrict__ waveStats, const int blockId, const int numBlocks) {
  constexpr int kTasksPerBlock = 8 / VOX;            // 1 (whole block per wave) or 2 (half blocks)
  constexpr int kVoxPerTask = kBlockSize3 / kTasksPerBlock;
  // per wave: the voxels waiting for their colour update, one word each:
  // (number of the task among this wave's tasks) << 9 | voxel index in the block
  constexpr int kPendCap = 64 + kVoxPerTask;
  __shared__ uint32_t s_pend[kIntegrateWaves][kPendCap];
  __shared__ float s_rcpW[257];  // RN(1/w), w = 1..256 (`/` is the correctly rounded division)
  __shared__ float4 s_xprod[XLDS ? kIntegrateWaves : 1][8];  // per wave: (M0 mx, M1 mx, M2 mx, mx) of the task's eight x positions
  static_assert(!XLDS || VOX == 8, "the x positions are wave-uniform only when a lane owns a whole x row");
  for (int i = threadIdx.x; i < 257; i += 64 * kIntegrateWaves) s_rcpW[i] = 1.0f / (float)(i > 0 ? i : 1);
  __syncthreads();

  const int noVisible = s.ctr[CTR_NO_VISIBLE_LIVE];
  const int noTasks = noVisible * kTasksPerBlock;
  if (blockId == 0 && threadIdx.x == 0) atomicAdd(&s.work[WORK_V_INTEGRATED], (unsigned long long)noVisible);
  const bool stopAtMaxW = PLAIN ? false : (p.stopAtMaxW != 0);
  const bool depthWeighting = PLAIN ? false : (p.depthWeighting != 0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stride = numBlocks * kIntegrateWaves;
  const Mat4 &Mr = RGB_SAME ? p.M : p.M_rgb;
  const float4 projr = RGB_SAME ? p.proj : p.proj_rgb;
  const int Wc = RGB_SAME ? p.W : p.Wr, Hc = RGB_SAME ? p.H : p.Hr;
  uint32_t *pend = s_pend[wave];
  const int t0 = blockId * kIntegrateWaves + wave;  // this wave's first task; its k-th is t0 + k * stride
  // Two uniforms of the per-voxel code that the register allocator keeps SPILLING (to lanes of a VGPR: a v_readlane — a half-rate
  // VALU instruction — at every use, 8 + 3 per task; ISA of round 4): held in vector registers of their own instead (an opaque
  // copy), where reading them is free.  The kernel has registers to spare since XLDS (57 of 64).
  int maxWv = p.maxW;
  asm volatile("" : "+v"(maxWv));

  // reciprocals of the constant divisors (uniform): correctly rounded for div_short, the refined
  // hardware reciprocal for the two-correction form
  const float yMu = PLAIN ? 1.0f / p.mu : rcp_refined(p.mu);
  const float y32767 = PLAIN ? 1.0f / 32767.0f : rcp_refined(32767.0f);
  const float y255 = 1.0f / 255.0f;
  // gate of ComputeUpdatedVoxelInfo<true> for voxels the depth step rejected with eta = -1
  // (true only for mu >= 4 m)
  const bool rejectedPassGate = !((-1.0f > p.mu) || (fabsf(-1.0f / p.mu) > 0.25f));
  const float wLim = (float)(p.W - 2), hLim = (float)(p.H - 2);
  const float wcLim = (float)(Wc - 2), hcLim = (float)(Hc - 2);
  // the depth image as a raw buffer: 32-bit offsets, hardware range check (gfx9 descriptor word 3: 32-bit data format)
  const __amdgpu_buffer_rsrc_t depthRsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(depth), 0, (int)((uint32_t)p.W * (uint32_t)p.H * 4u), 0x00020000);
  const int rowBytes = p.W * 4;
  const float hM0 = p.M.m[0], hM1 = p.M.m[1], hM2 = p.M.m[2], hM12 = p.M.m[12], hM13 = p.M.m[13], hM14 = p.M.m[14];
  const float hPx = p.proj.x, hPy = p.proj.y, hPz = p.proj.z, hPw = p.proj.w, hVs = p.voxelSize, hMu = p.mu, hYMu = yMu;

  // ------------------------------------------------------------ colour pass
  // computeUpdatedVoxelColorInfo for `cnt` (<= 64) pending voxels starting at list position `base`,
  // one per lane: every lane busy, which the per-voxel formulation (a divergent branch taken by
  // ~2 % of the voxels) never was.  The voxel's colour projection is recomputed from its
  // coordinates with the plain IEEE divide (for the shared-camera case it is the depth
  // projection again: same operands, same correctly rounded quotient).
  auto colour_pass = [&](int base, int cnt) {
    // the list is written and read by this wave only; LDS operations of one wave execute in
    // order, the fences keep the compiler from reordering across the boundary
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < cnt) {
      const uint32_t w = pend[base + lane];
      const int vox = (int)(w & 511u);
      const int tt = t0 + (int)(w >> 9) * stride;
      const dsr_hash_entry hc = entry_of_record(visBlocks[tt / kTasksPerBlock]);
      const uint32_t bptr = (uint32_t)hc.ptr;
      const int bx = hc.pos[0], by = hc.pos[1], bz = hc.pos[2];
      const float mx = (float)(bx * kBlockSize + (vox & 7)) * p.voxelSize;
      const float my = (float)(by * kBlockSize + ((vox >> 3) & 7)) * p.voxelSize;
      const float mz = (float)(bz * kBlockSize + (vox >> 6)) * p.voxelSize;
      const float3 pr = mat_mul3(Mr, mx, my, mz, 1.0f);
      const float u = projr.x * pr.x / pr.z + projr.z;
      const float v = projr.y * pr.y / pr.z + projr.w;
      if (!((u < 1) || (u > wcLim) || (v < 1) || (v > hcLim))) {
        uint8_t *blk = s.vba + (size_t)bptr * kBlockBytes;
        uint32_t *clrPtr = reinterpret_cast<uint32_t *>(blk + kOffClr + vox * 4);  // (r, g, b, w_color)
        const uint32_t cw = *clrPtr;
        const int oldWcI = (int)(cw >> 24);
        const float3 mm = bilinear_rgb(rgb, u, v, Wc);
        const float oldWc = (float)oldWcI;
        const float ocx = div_short((float)(cw & 0xffu), 255.0f, y255);
        const float ocy = div_short((float)((cw >> 8) & 0xffu), 255.0f, y255);
        const float ocz = div_short((float)((cw >> 16) & 0xffu), 255.0f, y255);
        const float rx = div_short(mm.x, 255.0f, y255), ry = div_short(mm.y, 255.0f, y255),
                    rz = div_short(mm.z, 255.0f, y255);
        float newWc = 1.0f;
        float ncx = ocx * oldWc + rx * newWc, ncy = ocy * oldWc + ry * newWc, ncz = ocz * oldWc + rz * newWc;
        newWc = oldWc + newWc;
        const float yW = s_rcpW[oldWcI + 1];
        ncx = div_short(ncx, newWc, yW); ncy = div_short(ncy, newWc, yW); ncz = div_short(ncz, newWc, yW);
        newWc = (newWc < (float)p.maxW) ? newWc : (float)p.maxW;  // MIN(newW, maxW)
        const uint32_t r8 = (uint32_t)f2i(ncx * 255.0f) & 0xffu, g8 = (uint32_t)f2i(ncy * 255.0f) & 0xffu,
                       b8 = (uint32_t)f2i(ncz * 255.0f) & 0xffu;
        *clrPtr = r8 | (g8 << 8) | (b8 << 16) | (((uint32_t)f2i(newWc) & 0xffu) << 24);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // task t = (visible block t / kTasksPerBlock, half t % kTasksPerBlock); a lane owns the VOX
  // consecutive voxels starting at vox0 (linear index x + 8y + 64z inside the block)
  // the block's record of the visible-block stream (dsr_device.h): ONE load where the id -> table entry chain took two
  // dependent ones, and consecutive tasks read consecutive 16-byte records instead of scattered table lines
  auto task_entry = [&](int tt) -> dsr_hash_entry { return entry_of_record(visBlocks[tt / kTasksPerBlock]); };
  auto task_vox0 = [&](int tt) -> int { return VOX * lane + kVoxPerTask * (tt % kTasksPerBlock); };

  // software pipeline over this wave's tasks: entries two ahead, voxel planes one ahead
  const dsr_hash_entry kNone = {{0, 0, 0}, 0, 0, -2};
  int nPend = 0;  // wave-uniform length of the pending colour list (< 64 between tasks)
  // byte-model bookkeeping (wave-uniform, scalar registers): lanes that stored their 24 B of
  // sdf + w_depth, voxels that took the colour update; added to this wave's own slot at the end
  uint32_t statStoreLanes = 0, statColour = 0;
  int t = t0;
  int taskNo = 0;  // t == t0 + taskNo * stride
  dsr_hash_entry heCur = (t < noTasks) ? task_entry(t) : kNone;
  dsr_hash_entry heNext = (t + stride < noTasks) ? task_entry(t + stride) : kNone;
  LanePlanes<VOX> planesNext;
#pragma unroll
  for (int k = 0; k < VOX / 2; ++k) planesNext.sdf[k] = 0;
#pragma unroll
  for (int k = 0; k < VOX / 4; ++k) planesNext.wd[k] = 0;
  if (heCur.ptr >= 0) planesNext = load_planes<VOX>(s.vba + (size_t)heCur.ptr * kBlockBytes, task_vox0(t));

  for (; t < noTasks; t += stride, ++taskNo) {
    const dsr_hash_entry he = heCur;
    const dsr_hash_entry heAfter = (t + 2 * stride < noTasks) ? task_entry(t + 2 * stride) : kNone;
    LanePlanes<VOX> pl = planesNext;
    if (heNext.ptr >= 0)  // prefetch the next task's depth planes
      planesNext = load_planes<VOX>(s.vba + (size_t)heNext.ptr * kBlockBytes, task_vox0(t + stride));
    heCur = heNext;
    heNext = heAfter;
    if (he.ptr < 0) continue;
    uint8_t *blk = s.vba + (size_t)he.ptr * kBlockBytes;

    const int vox0 = task_vox0(t);
    const int lx0 = vox0 & 7, ly = (vox0 >> 3) & 7, lz = vox0 >> 6;
    const int gx = he.pos[0] * kBlockSize + lx0, gy = he.pos[1] * kBlockSize, gz = he.pos[2] * kBlockSize;
    const float my = (float)(gy + ly) * hVs;
    const float mz = (float)(gz + lz) * hVs;
    // ORUtils Matrix4 * Vector4 (mat_mul3), row by row in its order ((m0 x + m4 y) + m8 z) + m12 w: the y and z products are
    // the same for the lane's eight voxels
    const float yx = p.M.m[4] * my, yy = p.M.m[5] * my, yz_ = p.M.m[6] * my;
    const float zx = p.M.m[8] * mz, zy = p.M.m[9] * mz, zz = p.M.m[10] * mz;
    auto to_camera = [&](float mx) -> float3 {
      float3 r;
      r.x = hM0 * mx + yx + zx + hM12 * 1.0f;
      r.y = hM1 * mx + yy + zy + hM13 * 1.0f;
      r.z = hM2 * mx + yz_ + zz + hM14 * 1.0f;
      return r;
    };
    if (XLDS) {
      // (the previous task's reads of the row are complete: its results were consumed before its stores were issued)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < 8) {
        const float mx = (float)(gx + lane) * hVs;
        s_xprod[wave][lane] = make_float4(hM0 * mx, hM1 * mx, hM2 * mx, mx);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---------------------------------------------- phase A1: project, issue the depth gathers
    float pz[VOX], dm[VOX];
    bool graze = false;
    {
      bool allTame = true;
      // the row is read one voxel AHEAD of its use (a read issued right before its use costs the wave the LDS latency eight
      // times per task); the address lives in a vector register of its own (an opaque copy: otherwise re-made per read)
      // (three words per row are read: a b128 read's unused fourth register gets re-used by the compiler at once, and the
      //  write-after-write hazard on it makes the wave wait for the read right where it was issued)
      typedef const __attribute__((address_space(3))) float *lds_row_t;
      struct xrow_v3 { float x, y, z; };
      lds_row_t xrow = (lds_row_t)&s_xprod[XLDS ? wave : 0][0];
      if (XLDS) asm volatile("" : "+v"(xrow));
      xrow_v3 xpNext = {0.f, 0.f, 0.f};
      if (XLDS) xpNext = xrow_v3{xrow[0], xrow[1], xrow[2]};
#pragma unroll
      for (int x = 0; x < VOX; ++x) {
        float3 pc;
        if (XLDS) {
          const xrow_v3 xp = xpNext;  // same address in every lane: a broadcast read
          if (x + 1 < VOX) xpNext = xrow_v3{xrow[4 * (x + 1)], xrow[4 * (x + 1) + 1], xrow[4 * (x + 1) + 2]};
          pc.x = xp.x + yx + zx + hM12 * 1.0f;
          pc.y = xp.y + yy + zy + hM13 * 1.0f;
          pc.z = xp.z + yz_ + zz + hM14 * 1.0f;
        } else {
          const float mx = (float)(gx + x) * hVs;
          pc = to_camera(mx);
        }
        const bool tame = pc.z >= 1e-4f;
        const float zs = tame ? pc.z : 1.0f;
        const float yz = rcp_refined(zs);
        const float u = div_with_rcp(hPx * pc.x, zs, yz) + hPz;
        const float v = div_with_rcp(hPy * pc.y, zs, yz) + hPw;
        const bool in = tame & !((u < 1) | (u > wLim) | (v < 1) | (v > hLim));  // bitwise: no short-circuit branches
        // byte offset of the pixel (rows and bytes per row are < 2^24); out of the image -> out of the
        // buffer's range -> the load returns 0: "no depth", rejected by (dm <= 0) like an invalid pixel
        const uint32_t off = in ? (uint32_t)(__mul24(f2i(v + 0.5f), rowBytes) + (f2i(u + 0.5f) << 2)) : 0xffffffffu;
        dm[x] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(depthRsrc, (int)off, 0, 0));
        pz[x] = pc.z;
        allTame &= tame;
      }
      graze = !allTame;  // a block that touches the camera plane: decided exactly below
    }
    if (__builtin_expect(__any(graze), 0)) {
      // voxels grazing the camera plane (0 < z < 1e-4): the divisor is not tame, redo them
      // with the plain IEEE divide
#pragma unroll
      for (int x = 0; x < VOX; ++x) {
        if (!(pz[x] > 0) || pz[x] >= 1e-4f) continue;
        const float mx = (float)(gx + x) * hVs;
        const float3 pc = to_camera(mx);
        const float u = p.proj.x * pc.x / pc.z + p.proj.z;
        const float v = p.proj.y * pc.y / pc.z + p.proj.w;
        const bool in = !((u < 1) || (u > wLim) || (v < 1) || (v > hLim));
        dm[x] = in ? depth[f2i(u + 0.5f) + f2i(v + 0.5f) * p.W] : 0.0f;
      }
    }

    // computeUpdatedVoxelDepthInfo's rejection tests; a task none of whose voxels is updated
    // (behind the surface / outside the image: ~15 % of the visible half blocks) ends here
    bool anyUpd = false;
#pragma unroll
    for (int x = 0; x < VOX; ++x) {
      const bool ok = !(dm[x] <= 0.0f);  // lanes out of the image read dm = 0
      const float eta = dm[x] - pz[x];
      anyUpd |= ok & !(eta < -hMu);
    }
    if (!rejectedPassGate && !__any(anyUpd)) continue;

    // ---------------------------------------------- phase A2: SDF running mean, colour gate
    bool dirtyDepth = false;
    const uint32_t pendWord0 = ((uint32_t)taskNo << 9) | (uint32_t)vox0;
    // REJ: voxels the depth step rejected pass the colour gate too (only for mu >= 4 m); a
    // compile-time flag of the loop so that the usual case carries no trace of it
    auto phaseA2 = [&](auto rejTag) {
      constexpr bool REJ = decltype(rejTag)::value;
#pragma unroll
      for (int x = 0; x < VOX; ++x) {
        const short sdf = (short)((pl.sdf[x >> 1] >> ((x & 1) * 16)) & 0xffffu);
        const int wDepth = (int)((pl.wd[x >> 2] >> ((x & 3) * 8)) & 0xffu);
        const bool skip = stopAtMaxW & (wDepth == p.maxW);
        // the two values of the rejection loop above are recomputed (a compare and a subtraction) rather than kept: held
        // across the task they are 8 VGPRs and 8 lane masks, which the register allocator answered with spills
        const bool okv = !(dm[x] <= 0.0f);
        const float etax = dm[x] - pz[x];
        const bool okx = !skip & okv;  // (bitwise: the compiler turns && / || chains into exec-mask branches)
        const bool upd = okx & !(etax < -hMu);
        const float q = PLAIN ? div_short(etax, hMu, hYMu) : div_with_rcp(etax, hMu, hYMu);  // eta / mu
        const float oldF = PLAIN ? div_short((float)sdf, 32767.0f, y32767)
                                 : div_with_rcp((float)sdf, 32767.0f, y32767);  // SDF_valueToFloat
        float newF = (1.0f < q) ? 1.0f : q;                              // MIN(1.0f, eta / mu)
        int newW = depthWeighting ? depth_weight(okx ? dm[x] : 1.0f) : 1;
        newF = (float)wDepth * oldF + (float)newW * newF;
        newW = wDepth + newW;
        // (PLAIN: the new weight is wDepth + 1, and the float of that sum is the sum of the floats — a full-rate add where the
        //  conversion is a half-rate instruction)
        newF = PLAIN ? div_short(newF, (float)wDepth + 1.0f, s_rcpW[newW]) : fdiv_tame(newF, (float)newW);
        newW = newW < maxWv ? newW : maxWv;
        const uint32_t sdfNew = (uint32_t)(uint16_t)sdf_from_float(newF);
        const uint32_t sw = (pl.sdf[x >> 1] & ~(0xffffu << ((x & 1) * 16))) | (sdfNew << ((x & 1) * 16));
        const uint32_t ww = (pl.wd[x >> 2] & ~(0xffu << ((x & 3) * 8))) | ((uint32_t)(newW & 0xff) << ((x & 3) * 8));
        pl.sdf[x >> 1] = upd ? sw : pl.sdf[x >> 1];
        pl.wd[x >> 2] = upd ? ww : pl.wd[x >> 2];
        dirtyDepth |= upd;
        // ---- ComputeUpdatedVoxelInfo<true>::compute gate: !(eta > mu || fabs(eta/mu) > 0.25); voxels the
        //      depth step rejected carry eta = -1.  q is the correctly rounded eta / mu and mu > 0, so
        //      eta > mu implies q >= 1 > 0.25: the first comparison never decides (a NaN fails both).  eta = +inf
        //      (the one input whose quotient this division sequence gets wrong: NaN) cannot reach this point:
        //      float views are stored with depths above 1e30 clamped to 1e30 (k_edges.h k_copy_depth_finite)
        const bool gateOk = !(fabsf(q) > 0.25f);
        const bool gate = REJ ? (!skip & (okv ? gateOk : true)) : (okx & gateOk);
        // append to the wave's pending colour list (ordered compaction across the 64 lanes)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(gate);
        if (gate)  // slot = number of gated lanes below this one (v_mbcnt)
          pend[nPend + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = pendWord0 + (uint32_t)x;
        nPend += __popcll(m);
        statColour += (uint32_t)__popcll(m);
      }
    };
    if (rejectedPassGate) phaseA2(std::true_type{});
    else phaseA2(std::false_type{});

    if (dirtyDepth) store_planes<VOX>(blk, vox0, pl);
    statStoreLanes += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(dirtyDepth));

    // ------------------------------------------------------------ phase B: colour, 64 at a time
    while (nPend >= 64) {
      nPend -= 64;
      colour_pass(nPend, 64);
    }
  }
  if (nPend > 0) colour_pass(0, nPend);
  if (lane == 0 && (statStoreLanes | statColour)) {  // one private 8-byte slot per wave: no atomics
    uint2 *slot = waveStats + (blockId * kIntegrateWaves + wave);
    const uint2 old = *slot;
    *slot = make_uint2(old.x + statStoreLanes, old.y + statColour);
  }
}

template <bool RGB_SAME, bool PLAIN, int VOX, int OCC, bool XLDS = false>
__global__ __launch_bounds__(64 * kIntegrateWaves, OCC) void k_integrate(FrameP p, SceneP s, const float *__restrict__ depth,
                                                                         const uchar4 *__restrict__ rgb,
                                                                         const int4 *__restrict__ visBlocks,
                                                                         uint2 *__restrict__ waveStats) {
  integrate_body<RGB_SAME, PLAIN, VOX, XLDS>(p, s, depth, rgb, visBlocks, waveStats, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace dsr
