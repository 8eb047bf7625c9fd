// ubench_valu.hip — issue-rate microbenchmark for gfx950 (MI355X): how many cycles does a SIMD need
// per wave64 instruction, for the instruction classes k_integrate / k_raycast are made of?
//
// DESIGN.md priced k_integrate at 4 cycles per wave64 VALU instruction
// ("98 % VALU bound"); MI355X_MICROARCH.md says a CDNA4 SIMD is 32 lanes wide and v_fma_f32 issues
// in 2.  This tool measures it on the box: every wave runs ITER x 32 INDEPENDENT instructions of one
// class (8 accumulator chains, unrolled 4x: no dependency stalls at >= 2 waves), with W waves
// resident per SIMD (grid = 256 CUs x W workgroups of 256 threads = 4 waves, one per SIMD).
//   cycles / wave-instruction / SIMD = wave's s_memtime span x (SIMD clock / memtime clock) / (instructions x W)
// The s_memtime clock is calibrated against the HIP-event wall time of the same launch and the
// SIMD clock against a dependent v_fma chain?  No: both are reported raw — the wall time of the
// launch, the span in s_memtime ticks and in s_memrealtime (100 MHz) ticks — so nothing is assumed.
// Also: a float4 device copy (the guide's 6.29 TB/s figure) to have the box's own HBM ceiling.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o ubench_valu ubench_valu.hip     Run: ./ubench_valu > log
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                            \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

enum Op {
  OP_FMA, OP_MUL, OP_ADD, OP_CNDMASK, OP_CVT_I32_F32, OP_CVT_F32_I32, OP_AND, OP_LSHL, OP_BFE, OP_PK_FMA, OP_PK_MUL, OP_RCP,
  OP_MUL_LO_U32, OP_MAD_U32_U24, OP_CVT_UBYTE, OP_CMP_VCC, OP_CMP_SGPR, OP_MBCNT, OP_FMA_SGPR, OP_MOV, OP_PERM,
  OP_MIX_SALU_1_3, OP_MIX_SALU_1_1, OP_SALU, OP_FMA_DEP,
  OP_CND_E64, OP_CND_NODEP, OP_MIN_F32, OP_MAX_F32, OP_MED3_F32, OP_BFI, OP_AND_OR, OP_LSHL_OR, OP_LSHL_ADD, OP_ADD_U32, OP_SUB_F32, OP_ASHR, OP_MAX_I32, OP_XOR, OP_OR, OP_FMAC, OP_MUL_LIT, OP_MUL_INL, OP_ADD_SGPR, OP_MUL_U24, OP_WRITELANE, OP_CVT_PKRTZ, OP_FLOOR, OP_RNDNE, OP_SQRT, OP_MAD_I32_I24, OP_ADD3, OP_SDWA_UB,
  OP_CMP_CND_PAIR, OP_READLANE, OP_DS_READ, OP_COUNT
};
static const char *kNames[OP_COUNT] = {
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_cndmask_b32", "v_cvt_i32_f32", "v_cvt_f32_i32", "v_and_b32", "v_lshlrev_b32",
    "v_bfe_u32", "v_pk_fma_f32 (2 flop-lanes)", "v_pk_mul_f32", "v_rcp_f32", "v_mul_lo_u32", "v_mad_u32_u24",
    "v_cvt_f32_ubyte1", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> sgpr pair", "v_mbcnt_lo_u32_b32", "v_fma_f32 (sgpr operand)",
    "v_mov_b32", "v_perm_b32", "3 v_fma : 1 s_add (counted: the 24 VALU)", "1 v_fma : 1 s_add (counted: the 16 VALU)", "s_add_u32 (SALU only)",
    "v_fma_f32 DEPENDENT chain (1 accumulator)",
    "v_cndmask_b32 (e64, sgpr-pair mask)",
    "v_cndmask_b32 (dst not a source)",
    "v_min_f32",
    "v_max_f32",
    "v_med3_f32",
    "v_bfi_b32",
    "v_and_or_b32",
    "v_lshl_or_b32",
    "v_lshl_add_u32",
    "v_add_u32",
    "v_sub_f32",
    "v_ashrrev_i32",
    "v_max_i32",
    "v_xor_b32",
    "v_or_b32",
    "v_fmac_f32 (VOP2)",
    "v_mul_f32 (32-bit literal operand)",
    "v_mul_f32 (inline constant 2.0)",
    "v_add_f32 (sgpr operand)",
    "v_mul_u32_u24",
    "v_writelane_b32",
    "v_cvt_pkrtz_f16_f32",
    "v_floor_f32",
    "v_rndne_f32",
    "v_sqrt_f32",
    "v_mad_i32_i24",
    "v_add3_u32",
    "v_cvt_f32_ubyte0 sdwa (byte select)",
    "v_cmp_lt_f32 vcc + v_cndmask (counted: both)", "v_readlane_b32", "ds_read_b32 (8 in flight, then waitcnt)"};

// 32 instructions per iteration on 8 independent accumulators
#define REP8(fmt)                                                                                        \
  fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define BODY4(one) one one one one

template <int OP>
__global__ __launch_bounds__(256) void k_issue(int iters, float *sink, unsigned long long *spans) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.000001f, c = 1e-7f;
  unsigned s0 = blockIdx.x, s1 = 1;
  float2 p0 = make_float2(a0, a1), p1 = make_float2(a2, a3), p2 = make_float2(a4, a5), p3 = make_float2(a6, a7);
  float2 pb = make_float2(b, b), pc = make_float2(c, c);
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a0), "v"(a4) : "vcc");
  asm volatile("v_cmp_lt_f32 s[22:23], %0, %1" ::"v"(a0), "v"(a4) : "s22", "s23");
  __shared__ float s_lds[1024];
  s_lds[threadIdx.x] = a0; s_lds[threadIdx.x + 256] = a1; s_lds[threadIdx.x + 512] = a2; s_lds[threadIdx.x + 768] = a3;
  __syncthreads();
  const unsigned ldsAddr = (unsigned)(size_t)(&s_lds[0]) + (threadIdx.x & 63) * 4;
  const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#define ONE8(INSN)                                                                                        \
  asm volatile(INSN("%0") INSN("%1") INSN("%2") INSN("%3") INSN("%4") INSN("%5") INSN("%6") INSN("%7")    \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
               : "v"(b), "v"(c), "s"(s1)                                                                  \
               : "vcc");
    if (OP == OP_FMA) {
#define I_FMA(d) "v_fma_f32 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_FMA))
    } else if (OP == OP_MUL) {
#define I_MUL(d) "v_mul_f32 " d ", " d ", %8\n"
      BODY4(ONE8(I_MUL))
    } else if (OP == OP_ADD) {
#define I_ADD(d) "v_add_f32 " d ", " d ", %9\n"
      BODY4(ONE8(I_ADD))
    } else if (OP == OP_CNDMASK) {
#define I_CND(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n"
      BODY4(ONE8(I_CND))
    } else if (OP == OP_CVT_I32_F32) {
#define I_CVTI(d) "v_cvt_i32_f32 " d ", " d "\n"
      BODY4(ONE8(I_CVTI))
    } else if (OP == OP_CVT_F32_I32) {
#define I_CVTF(d) "v_cvt_f32_i32 " d ", " d "\n"
      BODY4(ONE8(I_CVTF))
    } else if (OP == OP_AND) {
#define I_AND(d) "v_and_b32 " d ", %8, " d "\n"
      BODY4(ONE8(I_AND))
    } else if (OP == OP_LSHL) {
#define I_LSHL(d) "v_lshlrev_b32 " d ", 1, " d "\n"
      BODY4(ONE8(I_LSHL))
    } else if (OP == OP_BFE) {
#define I_BFE(d) "v_bfe_u32 " d ", " d ", 3, 9\n"
      BODY4(ONE8(I_BFE))
    } else if (OP == OP_RCP) {
#define I_RCP(d) "v_rcp_f32 " d ", " d "\n"
      BODY4(ONE8(I_RCP))
    } else if (OP == OP_MUL_LO_U32) {
#define I_MULLO(d) "v_mul_lo_u32 " d ", " d ", %8\n"
      BODY4(ONE8(I_MULLO))
    } else if (OP == OP_MAD_U32_U24) {
#define I_MAD24(d) "v_mad_u32_u24 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_MAD24))
    } else if (OP == OP_CVT_UBYTE) {
#define I_UB(d) "v_cvt_f32_ubyte1 " d ", " d "\n"
      BODY4(ONE8(I_UB))
    } else if (OP == OP_CMP_VCC) {
#define I_CMPV(d) "v_cmp_lt_f32 vcc, " d ", %8\n"
      BODY4(ONE8(I_CMPV))
    } else if (OP == OP_CMP_SGPR) {
#define I_CMPS(d) "v_cmp_lt_f32 s[20:21], " d ", %8\n"
      asm volatile(I_CMPS("%0") I_CMPS("%1") I_CMPS("%2") I_CMPS("%3") I_CMPS("%4") I_CMPS("%5") I_CMPS("%6") I_CMPS("%7")
                   I_CMPS("%0") I_CMPS("%1") I_CMPS("%2") I_CMPS("%3") I_CMPS("%4") I_CMPS("%5") I_CMPS("%6") I_CMPS("%7")
                   I_CMPS("%0") I_CMPS("%1") I_CMPS("%2") I_CMPS("%3") I_CMPS("%4") I_CMPS("%5") I_CMPS("%6") I_CMPS("%7")
                   I_CMPS("%0") I_CMPS("%1") I_CMPS("%2") I_CMPS("%3") I_CMPS("%4") I_CMPS("%5") I_CMPS("%6") I_CMPS("%7")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b)
                   : "s20", "s21");
    } else if (OP == OP_MBCNT) {
#define I_MBC(d) "v_mbcnt_lo_u32_b32 " d ", %10, " d "\n"
      BODY4(ONE8(I_MBC))
    } else if (OP == OP_FMA_SGPR) {
#define I_FMAS(d) "v_fma_f32 " d ", " d ", %10, %9\n"
      BODY4(ONE8(I_FMAS))
    } else if (OP == OP_MOV) {
#define I_MOV(d) "v_mov_b32 " d ", %8\n"
      BODY4(ONE8(I_MOV))
    } else if (OP == OP_PERM) {
#define I_PERM(d) "v_perm_b32 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_PERM))
    } else if (OP == OP_PK_FMA || OP == OP_PK_MUL) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (OP == OP_PK_FMA)
          asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                       : "v"(pb), "v"(pc));
        else
          asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n"
                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                       : "v"(pb));
      }
    } else if (OP == OP_MIX_SALU_1_3) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\ns_add_u32 %3, %3, 1\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+s"(s0)
                     : "v"(b), "v"(c)
                     : "scc");
    } else if (OP == OP_MIX_SALU_1_1) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        asm volatile("v_fma_f32 %0, %0, %2, %3\ns_add_u32 %1, %1, 1\n" : "+v"(a0), "+s"(s0) : "v"(b), "v"(c) : "scc");
    } else if (OP == OP_SALU) {
#pragma unroll
      for (int k = 0; k < 32; ++k) asm volatile("s_add_u32 %0, %0, 1\n" : "+s"(s0) : : "scc");
    } else if (OP == OP_FMA_DEP) {
#pragma unroll
      for (int k = 0; k < 32; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b), "v"(c));
    } else if (OP == OP_CND_E64) {
#define I_N0(d) "v_cndmask_b32_e64 " d ", " d ", %8, s[22:23]\n"
      BODY4(ONE8(I_N0))
    } else if (OP == OP_CND_NODEP) {
#define I_N1(d) "v_cndmask_b32 " d ", %8, %9, vcc\n"
      BODY4(ONE8(I_N1))
    } else if (OP == OP_MIN_F32) {
#define I_N2(d) "v_min_f32 " d ", " d ", %8\n"
      BODY4(ONE8(I_N2))
    } else if (OP == OP_MAX_F32) {
#define I_N3(d) "v_max_f32 " d ", " d ", %8\n"
      BODY4(ONE8(I_N3))
    } else if (OP == OP_MED3_F32) {
#define I_N4(d) "v_med3_f32 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_N4))
    } else if (OP == OP_BFI) {
#define I_N5(d) "v_bfi_b32 " d ", %8, " d ", %9\n"
      BODY4(ONE8(I_N5))
    } else if (OP == OP_AND_OR) {
#define I_N6(d) "v_and_or_b32 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_N6))
    } else if (OP == OP_LSHL_OR) {
#define I_N7(d) "v_lshl_or_b32 " d ", " d ", 1, %8\n"
      BODY4(ONE8(I_N7))
    } else if (OP == OP_LSHL_ADD) {
#define I_N8(d) "v_lshl_add_u32 " d ", " d ", 1, %8\n"
      BODY4(ONE8(I_N8))
    } else if (OP == OP_ADD_U32) {
#define I_N9(d) "v_add_u32 " d ", " d ", %8\n"
      BODY4(ONE8(I_N9))
    } else if (OP == OP_SUB_F32) {
#define I_N10(d) "v_sub_f32 " d ", " d ", %9\n"
      BODY4(ONE8(I_N10))
    } else if (OP == OP_ASHR) {
#define I_N11(d) "v_ashrrev_i32 " d ", 3, " d "\n"
      BODY4(ONE8(I_N11))
    } else if (OP == OP_MAX_I32) {
#define I_N12(d) "v_max_i32 " d ", " d ", %8\n"
      BODY4(ONE8(I_N12))
    } else if (OP == OP_XOR) {
#define I_N13(d) "v_xor_b32 " d ", %8, " d "\n"
      BODY4(ONE8(I_N13))
    } else if (OP == OP_OR) {
#define I_N14(d) "v_or_b32 " d ", %8, " d "\n"
      BODY4(ONE8(I_N14))
    } else if (OP == OP_FMAC) {
#define I_N15(d) "v_fmac_f32 " d ", %8, %9\n"
      BODY4(ONE8(I_N15))
    } else if (OP == OP_MUL_LIT) {
#define I_N16(d) "v_mul_f32 " d ", 0x3f800008, " d "\n"
      BODY4(ONE8(I_N16))
    } else if (OP == OP_MUL_INL) {
#define I_N17(d) "v_mul_f32 " d ", 2.0, " d "\n"
      BODY4(ONE8(I_N17))
    } else if (OP == OP_ADD_SGPR) {
#define I_N18(d) "v_add_f32 " d ", %10, " d "\n"
      BODY4(ONE8(I_N18))
    } else if (OP == OP_MUL_U24) {
#define I_N19(d) "v_mul_u32_u24 " d ", " d ", %8\n"
      BODY4(ONE8(I_N19))
    } else if (OP == OP_WRITELANE) {
#define I_N20(d) "v_writelane_b32 " d ", %10, 3\n"
      BODY4(ONE8(I_N20))
    } else if (OP == OP_CVT_PKRTZ) {
#define I_N21(d) "v_cvt_pkrtz_f16_f32 " d ", " d ", %8\n"
      BODY4(ONE8(I_N21))
    } else if (OP == OP_FLOOR) {
#define I_N22(d) "v_floor_f32 " d ", " d "\n"
      BODY4(ONE8(I_N22))
    } else if (OP == OP_RNDNE) {
#define I_N23(d) "v_rndne_f32 " d ", " d "\n"
      BODY4(ONE8(I_N23))
    } else if (OP == OP_SQRT) {
#define I_N24(d) "v_sqrt_f32 " d ", " d "\n"
      BODY4(ONE8(I_N24))
    } else if (OP == OP_MAD_I32_I24) {
#define I_N25(d) "v_mad_i32_i24 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_N25))
    } else if (OP == OP_ADD3) {
#define I_N26(d) "v_add3_u32 " d ", " d ", %8, %9\n"
      BODY4(ONE8(I_N26))
    } else if (OP == OP_SDWA_UB) {
#define I_N27(d) "v_cvt_f32_ubyte0_sdwa " d ", " d " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
      BODY4(ONE8(I_N27))
    } else if (OP == OP_CMP_CND_PAIR) {
#define I_PAIR(d) "v_cmp_lt_f32 vcc, " d ", %8\nv_cndmask_b32 " d ", " d ", %9, vcc\n"
      asm volatile(I_PAIR("%0") I_PAIR("%1") I_PAIR("%2") I_PAIR("%3") I_PAIR("%4") I_PAIR("%5") I_PAIR("%6") I_PAIR("%7")
                   I_PAIR("%0") I_PAIR("%1") I_PAIR("%2") I_PAIR("%3") I_PAIR("%4") I_PAIR("%5") I_PAIR("%6") I_PAIR("%7")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b), "v"(c)
                   : "vcc");
    } else if (OP == OP_READLANE) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        asm volatile("v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 3\nv_readlane_b32 s22, %2, 3\nv_readlane_b32 s23, %3, 3\n"
                     "v_readlane_b32 s24, %4, 3\nv_readlane_b32 s25, %5, 3\nv_readlane_b32 s26, %6, 3\nv_readlane_b32 s27, %7, 3\n"
                     : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (OP == OP_DS_READ) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8 offset:256\nds_read_b32 %2, %8 offset:512\nds_read_b32 %3, %8 offset:768\n"
                     "ds_read_b32 %4, %8 offset:1024\nds_read_b32 %5, %8 offset:1280\nds_read_b32 %6, %8 offset:1536\nds_read_b32 %7, %8 offset:1792\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                     : "v"(ldsAddr));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    spans[2 * w] = t1 - t0;
    spans[2 * w + 1] = r1 - r0;
  }
  // keep everything alive
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)s0;
  if (r == 123.456f) sink[threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_copy16(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read16(const float4 *__restrict__ in, float *__restrict__ out, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

typedef void (*KernelFn)(int, float *, unsigned long long *);
template <int OP>
struct Table {
  static void fill(KernelFn *t) { t[OP] = k_issue<OP>; Table<OP + 1>::fill(t); }
};
template <>
struct Table<OP_COUNT> {
  static void fill(KernelFn *) {}
};

int main(int argc, char **argv) {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  printf("# device: %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  KernelFn table[OP_COUNT];
  Table<0>::fill(table);
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float *sink;
  unsigned long long *spans;
  const int maxWaves = cus * 8 * 4;
  CHECK(hipMalloc((void **)&sink, 4096));
  CHECK(hipMalloc((void **)&spans, (size_t)maxWaves * 16));
  std::vector<unsigned long long> h((size_t)maxWaves * 2);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("# per row: W = waves resident per SIMD (grid = CUs x W workgroups of 4 waves); N = instructions counted per wave\n");
  printf("# wall_us = HIP-event time of the launch; ticks = mean s_memtime span per wave; rt = mean s_memrealtime span (100 MHz)\n");
  printf("# cyc/inst/SIMD(wall) = wall_us * 2400 MHz / (N * W)   [2.4 GHz nominal; see eff_GHz for the clock the span implies]\n");
  printf("%-46s %2s %9s %10s %12s %10s %8s %14s %14s\n", "op", "W", "N", "wall_us", "ticks", "rt", "eff_GHz", "cyc/inst(wall)", "cyc/inst(tick)");
  for (int op = 0; op < OP_COUNT; ++op) {
    for (int W : {1, 2, 4, 7, 8}) {
      const int grid = cus * W;
      const long long perIter = (op == OP_MIX_SALU_1_3) ? 24 : (op == OP_MIX_SALU_1_1) ? 16 : 32;  // (cmp+cndmask pairs: 16 pairs = 32 instructions)
      const long long N = perIter * iters;
      hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), 0, 0, 200, sink, spans);  // warm-up
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), 0, 0, iters, sink, spans);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h.data(), spans, (size_t)grid * 4 * 16, hipMemcpyDeviceToHost));
      double st = 0, sr = 0;
      for (int w = 0; w < grid * 4; ++w) { st += (double)h[2 * w]; sr += (double)h[2 * w + 1]; }
      st /= grid * 4; sr /= grid * 4;
      const double spanUs = sr / 100.0;  // 100 MHz
      const double effGHz = st / (spanUs * 1e3);
      printf("%-46s %2d %9lld %10.1f %12.0f %10.0f %8.3f %14.3f %14.3f\n", kNames[op], W, N, ms * 1e3, st, sr, effGHz,
             ms * 1e3 * 2400.0 / ((double)N * W), st / ((double)N * W));
    }
  }
  // ---- HBM ceiling: float4 copy and float4 read, 1 GiB
  const size_t bytes = 1ull << 30;
  float4 *a, *b;
  CHECK(hipMalloc((void **)&a, bytes));
  CHECK(hipMalloc((void **)&b, bytes));
  CHECK(hipMemset(a, 1, bytes));
  CHECK(hipMemset(b, 2, bytes));
  for (int grid : {cus * 4, cus * 8, cus * 16, cus * 32}) {
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy16 grid %6d : %8.1f GB/s (read+write)\n", grid, 10.0 * 2.0 * bytes / (ms * 1e-3) / 1e9);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, a, sink, bytes / 16);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("read16 grid %6d : %8.1f GB/s (read only)\n", grid, 10.0 * bytes / (ms * 1e-3) / 1e9);
  }
  return 0;
}
