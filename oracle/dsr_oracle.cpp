/*
 * dsr_oracle.cpp — CPU ORACLE for the voxel-hashed TSDF hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Nothing in the shipped product
 * (dynslam_amd/, include/, shim/) may include, link or call this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * *** PARITY UNPINNED. ***  DynSLAM reaches this code only through
 * src/InfiniTAM (git submodule github.com/AndreiBarsan/InfiniTAM, a fork of
 * victorprad/InfiniTAM v2), which is an EMPTY, UNPINNED directory in
 * /root/reference (.gitmodules:1-3) and the reference has no
 * tests or golden vectors.  This file therefore restates the
 * published upstream InfiniTAM-v2 `_CPU` engine algorithms (file names below
 * are upstream paths under src/InfiniTAM/InfiniTAM/, which the reference's
 * CMakeLists.txt:106 would build) plus the fork deltas visible from DynSLAM's
 * call sites (cited as /root/reference paths).  Function-level comments name the
 * upstream function that is restated and the DynSLAM call site that reaches it.
 * EXCEPTION — pinned by the reference's own code: the host loops at the edges of the
 * path (silhouette split, compositing, disparity -> depth, layout conversions) exist in
 * /root/reference; they are compiled from where they lie into oracle/_ref/ (ref_edges.cpp,
 * Makefile target _ref) and tests/test_reference_edges.py checks the restatements
 * below against them.
 *
 * Floating point: every expression is written in the upstream evaluation order;
 * build with -ffp-contract=off so that no FMA contraction happens (x86-64 gcc
 * -O2/-O3 default for the upstream _CPU build).  Float->int conversions that C
 * leaves undefined (out of range) are given a defined, saturating meaning
 * (f2i below) — the HIP engine uses the same definition.
 *
 * C ABI: identical to include/dsr.h with the prefix orc_.
 */
#include "../include/dsr.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <deque>
#include <new>
#include <string>
#include <unordered_map>
#include <fstream>
#include <sstream>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;
int fail(int code, const char *msg) { g_err = msg; return code; }

/* ------------------------------------------------------------------ numerics */

/* Saturating float->int (NaN -> 0).  x86 cvttss2si would give INT_MIN for out of
 * range values, GCN v_cvt_i32_f32 saturates; the saturating form is the adopted
 * definition (DESIGN.md "defined conversions"). */
static inline int f2i(float f) {
  if (!(f == f)) return 0;
  if (f >= 2147483648.0f) return INT_MAX;
  if (f <= -2147483648.0f) return INT_MIN;
  return (int)f;
}
static inline short f2s(float f) { return (short)f2i(f); } /* wraps to 16 bit */

struct V2f { float x, y; };
struct V3f { float x, y, z; };
struct V4f { float x, y, z, w; };
struct V2i { int x, y; };
struct V3i { int x, y, z; };
struct V4u { uint8_t x, y, z, w; };

/* ORUtils::Matrix4<float>: m[col*4 + row] (column-major). */
struct M4 { float m[16]; };

static inline M4 m4_identity() {
  M4 r; memset(&r, 0, sizeof r); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r;
}
/* ORUtils Matrix4 * Vector4 */
static inline V4f mul(const M4 &a, const V4f &v) {
  V4f r;
  r.x = a.m[0] * v.x + a.m[4] * v.y + a.m[8] * v.z + a.m[12] * v.w;
  r.y = a.m[1] * v.x + a.m[5] * v.y + a.m[9] * v.z + a.m[13] * v.w;
  r.z = a.m[2] * v.x + a.m[6] * v.y + a.m[10] * v.z + a.m[14] * v.w;
  r.w = a.m[3] * v.x + a.m[7] * v.y + a.m[11] * v.z + a.m[15] * v.w;
  return r;
}
/* ORUtils Matrix4 * Matrix4: r(x,y) = sum_k lhs(k,y) * rhs(x,k), at(x,y)=m[x*4+y] */
static inline M4 mul(const M4 &l, const M4 &r) {
  M4 o;
  for (int x = 0; x < 4; x++)
    for (int y = 0; y < 4; y++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += l.m[k * 4 + y] * r.m[x * 4 + k];
      o.m[x * 4 + y] = s;
    }
  return o;
}
/* ORUtils Matrix4::inv — Cramer's rule on the transposed source (the classic
 * cofactor expansion), float throughout. */
static bool m4_inv(const M4 &in, M4 &out) {
  float tmp[12], src[16], det;
  float *dst = out.m;
  for (int i = 0; i < 4; i++) {
    src[i] = in.m[i * 4];
    src[i + 4] = in.m[i * 4 + 1];
    src[i + 8] = in.m[i * 4 + 2];
    src[i + 12] = in.m[i * 4 + 3];
  }
  tmp[0] = src[10] * src[15]; tmp[1] = src[11] * src[14]; tmp[2] = src[9] * src[15];
  tmp[3] = src[11] * src[13]; tmp[4] = src[9] * src[14]; tmp[5] = src[10] * src[13];
  tmp[6] = src[8] * src[15]; tmp[7] = src[11] * src[12]; tmp[8] = src[8] * src[14];
  tmp[9] = src[10] * src[12]; tmp[10] = src[8] * src[13]; tmp[11] = src[9] * src[12];
  dst[0] = (tmp[0] * src[5] + tmp[3] * src[6] + tmp[4] * src[7]) - (tmp[1] * src[5] + tmp[2] * src[6] + tmp[5] * src[7]);
  dst[1] = (tmp[1] * src[4] + tmp[6] * src[6] + tmp[9] * src[7]) - (tmp[0] * src[4] + tmp[7] * src[6] + tmp[8] * src[7]);
  dst[2] = (tmp[2] * src[4] + tmp[7] * src[5] + tmp[10] * src[7]) - (tmp[3] * src[4] + tmp[6] * src[5] + tmp[11] * src[7]);
  dst[3] = (tmp[5] * src[4] + tmp[8] * src[5] + tmp[11] * src[6]) - (tmp[4] * src[4] + tmp[9] * src[5] + tmp[10] * src[6]);
  dst[4] = (tmp[1] * src[1] + tmp[2] * src[2] + tmp[5] * src[3]) - (tmp[0] * src[1] + tmp[3] * src[2] + tmp[4] * src[3]);
  dst[5] = (tmp[0] * src[0] + tmp[7] * src[2] + tmp[8] * src[3]) - (tmp[1] * src[0] + tmp[6] * src[2] + tmp[9] * src[3]);
  dst[6] = (tmp[3] * src[0] + tmp[6] * src[1] + tmp[11] * src[3]) - (tmp[2] * src[0] + tmp[7] * src[1] + tmp[10] * src[3]);
  dst[7] = (tmp[4] * src[0] + tmp[9] * src[1] + tmp[10] * src[2]) - (tmp[5] * src[0] + tmp[8] * src[1] + tmp[11] * src[2]);
  tmp[0] = src[2] * src[7]; tmp[1] = src[3] * src[6]; tmp[2] = src[1] * src[7];
  tmp[3] = src[3] * src[5]; tmp[4] = src[1] * src[6]; tmp[5] = src[2] * src[5];
  tmp[6] = src[0] * src[7]; tmp[7] = src[3] * src[4]; tmp[8] = src[0] * src[6];
  tmp[9] = src[2] * src[4]; tmp[10] = src[0] * src[5]; tmp[11] = src[1] * src[4];
  dst[8] = (tmp[0] * src[13] + tmp[3] * src[14] + tmp[4] * src[15]) - (tmp[1] * src[13] + tmp[2] * src[14] + tmp[5] * src[15]);
  dst[9] = (tmp[1] * src[12] + tmp[6] * src[14] + tmp[9] * src[15]) - (tmp[0] * src[12] + tmp[7] * src[14] + tmp[8] * src[15]);
  dst[10] = (tmp[2] * src[12] + tmp[7] * src[13] + tmp[10] * src[15]) - (tmp[3] * src[12] + tmp[6] * src[13] + tmp[11] * src[15]);
  dst[11] = (tmp[5] * src[12] + tmp[8] * src[13] + tmp[11] * src[14]) - (tmp[4] * src[12] + tmp[9] * src[13] + tmp[10] * src[14]);
  dst[12] = (tmp[2] * src[10] + tmp[5] * src[11] + tmp[1] * src[9]) - (tmp[4] * src[11] + tmp[0] * src[9] + tmp[3] * src[10]);
  dst[13] = (tmp[8] * src[11] + tmp[0] * src[8] + tmp[7] * src[10]) - (tmp[6] * src[10] + tmp[9] * src[11] + tmp[1] * src[8]);
  dst[14] = (tmp[6] * src[9] + tmp[11] * src[11] + tmp[3] * src[8]) - (tmp[10] * src[11] + tmp[2] * src[8] + tmp[7] * src[9]);
  dst[15] = (tmp[10] * src[10] + tmp[4] * src[8] + tmp[9] * src[9]) - (tmp[8] * src[9] + tmp[11] * src[10] + tmp[5] * src[8]);
  det = src[0] * dst[0] + src[1] * dst[1] + src[2] * dst[2] + src[3] * dst[3];
  if (det == 0.0f) return false;
  float inv = 1.0f / det;
  for (int i = 0; i < 16; i++) dst[i] *= inv;
  return true;
}

/* ---------------------------------------------------------------- constants */

const float FAR_AWAY = 999999.9f;  /* ITMVisualisationEngine.h */
const float VERY_CLOSE = 0.05f;
const int MINMAX_SUBSAMPLE = 8;    /* minmaximg_subsample */
const int16_t SDF_INITIAL = 32767; /* ITMVoxel_s_rgb::SDF_initialValue() */

static inline float sdf_to_float(float v) { return v / 32767.0f; }         /* SDF_valueToFloat */
static inline int16_t sdf_from_float(float f) { return (int16_t)f2i(f * 32767.0f); } /* SDF_floatToValue */

static inline dsr_voxel default_voxel() {
  dsr_voxel v; memset(&v, 0, sizeof v); v.sdf = SDF_INITIAL; return v;
}

/* ------------------------------------------------------------------- engine */

struct RenderState { /* ITMRenderState_VH */
  std::vector<int32_t> visibleEntryIDs;
  int noVisibleBlocks = 0;
  std::vector<uint8_t> entriesVisibleType;
  std::vector<V2f> minmax;      /* renderingRangeImage, compact ceil(W/8) x ceil(H/8) */
  std::vector<V4f> raycastResult;
  std::vector<V4u> raycastImage;
};

struct Engine {
  std::vector<dsr_triangle> mesh; /* current mesh (orc_mesh_scene) */
  dsr_settings s;
  dsr_calib calib;
  int W, H;          /* depth image size */
  int Wr, Hr;        /* rgb image size   */
  int noBuckets, noExcess, noTotalEntries, noBlocks;
  uint32_t hashMask;
  M4 calibInv;       /* trafo_rgb_to_depth.calib_inv */

  /* ITMVoxelBlockHash */
  std::vector<dsr_hash_entry> hashTable;
  std::vector<int32_t> excessAllocationList;
  int lastFreeExcessListId;
  /* ITMLocalVBA */
  std::vector<dsr_voxel> voxels;
  std::vector<int32_t> voxelAllocationList;
  int lastFreeBlockId;
  /* scratch of ITMSceneReconstructionEngine_CPU */
  std::vector<uint8_t> entriesAllocType;
  std::vector<int16_t> blockCoords; /* Vector4s per entry */

  RenderState live, freeview;
  bool freeviewValid = false;

  /* ITMView */
  bool hasView = false;
  std::vector<V4u> rgb;
  std::vector<float> depth;
  std::vector<float> depthTmp;
  /* ITMTrackingState */
  M4 M_d, invM_d;
  std::vector<V4f> pointsMap, normalsMap;

  int depthWeighting = 0;
  int stickyStatus = DSR_OK;
  int64_t decayedBlockCount = 0;
  int64_t framesProcessed = 0;
  std::deque<std::vector<int32_t>> decayFifo;

  /* ITMGlobalCache (host store) + ITMHashSwapState, only with use_swapping */
  std::vector<uint8_t> swapStates;
  std::vector<uint8_t> hasStored;
  std::unordered_map<int, std::vector<dsr_voxel>> storedBlocks;
  std::vector<uint8_t> ownsSlot; /* ITMGlobalCache keeps one fixed slot per entry: entries ever swapped out */
  int hostStoreSlots = 0;

  int threads = 1;
};

static inline uint32_t hashIndex(const Engine &e, int bx, int by, int bz) {
  /* ITMRepresentationAccess.h hashIndex */
  return (((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349669u) ^ ((uint32_t)bz * 83492791u)) & e.hashMask;
}

static void reset_scene(Engine &e) {
  /* ITMSceneReconstructionEngine_CPU<TVoxel,ITMVoxelBlockHash>::ResetScene
   * (reached from InfiniTamDriver.h:282-284). */
  dsr_hash_entry empty; memset(&empty, 0, sizeof empty); empty.ptr = -2;
  std::fill(e.hashTable.begin(), e.hashTable.end(), empty);
  for (int i = 0; i < e.noExcess; i++) e.excessAllocationList[i] = i;
  e.lastFreeExcessListId = e.noExcess - 1;
  std::fill(e.voxels.begin(), e.voxels.end(), default_voxel());
  for (int i = 0; i < e.noBlocks; i++) e.voxelAllocationList[i] = i;
  e.lastFreeBlockId = e.noBlocks - 1;
  std::fill(e.entriesAllocType.begin(), e.entriesAllocType.end(), 0);
  std::fill(e.blockCoords.begin(), e.blockCoords.end(), 0);
  for (RenderState *rs : {&e.live, &e.freeview}) {
    rs->noVisibleBlocks = 0;
    std::fill(rs->entriesVisibleType.begin(), rs->entriesVisibleType.end(), 0);
  }
  e.decayFifo.clear();
  e.decayedBlockCount = 0;
  std::fill(e.swapStates.begin(), e.swapStates.end(), 0);
  std::fill(e.hasStored.begin(), e.hasStored.end(), 0);
  e.storedBlocks.clear();
  std::fill(e.ownsSlot.begin(), e.ownsSlot.end(), 0);
  e.hostStoreSlots = 0;
  e.stickyStatus = DSR_OK;
}

/* ------------------------------------------------------------ view building */

/* ITMViewBuilder.h convertDepthAffineToFloat (via InfiniTamDriver.cpp:222). */
static void convert_depth(Engine &e, const int16_t *in) {
  const float a = e.calib.disparity_calib[0], b = e.calib.disparity_calib[1];
  for (int i = 0; i < e.W * e.H; i++) {
    int16_t d = in[i];
    e.depth[i] = (d <= 0 || d > 32000) ? -1.0f : (float)d * a + b;
  }
}

/* ITMViewBuilder.h filterDepth, applied 5x by ITMViewBuilder_CPU::DepthFiltering
 * when settings->useBilateralFilter. */
static void filter_depth_once(const Engine &e, const float *in, float *out) {
  const float MEAN_SIGMA_L = 1.2232f;
  for (int y = 2; y < e.H - 2; y++)
    for (int x = 2; x < e.W - 2; x++) {
      float z = in[x + y * e.W];
      if (z < 0.0f) { out[x + y * e.W] = -1.0f; continue; }
      float final_depth = 0.0f, w_sum = 0.0f;
      float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / sqrtf(z) * 0.25f);
      for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) {
          float tmpz = in[(x + j) + (y + i) * e.W];
          if (tmpz < 0.0f) continue;
          float dz = (tmpz - z); dz *= dz;
          float w = expf(-0.5f * ((abs(i) + abs(j)) * MEAN_SIGMA_L * MEAN_SIGMA_L + dz * sigma_z * sigma_z));
          w_sum += w;
          final_depth += w * tmpz;
        }
      final_depth /= w_sum;
      out[x + y * e.W] = final_depth;
    }
}
static void filter_depth(Engine &e) {
  e.depthTmp = e.depth;
  for (int k = 0; k < 5; k++) {
    if (k & 1) filter_depth_once(e, e.depthTmp.data(), e.depth.data());
    else filter_depth_once(e, e.depth.data(), e.depthTmp.data());
  }
  /* five passes: depth->tmp->depth->tmp->depth->tmp; result is in tmp */
  e.depth = e.depthTmp;
}

/* ------------------------------------------------------- allocation (A.2/3) */

/* ITMSceneReconstructionEngine.h checkPointVisibility / checkBlockVisibility. */
template <bool useSwapping>
static inline void checkPointVisibility(bool &isVisible, bool &isVisibleEnlarged, const V4f &pt_image,
                                        const M4 &M_d, const V4f &projParams_d, int W, int H) {
  V4f pt_buff = mul(M_d, pt_image);
  if (pt_buff.z < 1e-10f) return;
  pt_buff.x = projParams_d.x * pt_buff.x / pt_buff.z + projParams_d.z;
  pt_buff.y = projParams_d.y * pt_buff.y / pt_buff.z + projParams_d.w;
  if (pt_buff.x >= 0 && pt_buff.x < W && pt_buff.y >= 0 && pt_buff.y < H) {
    isVisible = true; isVisibleEnlarged = true;
  } else if (useSwapping) {
    int lx = -W / 8, ly = W + W / 8, lz = -H / 8, lw = H + H / 8;
    if (pt_buff.x >= lx && pt_buff.x < ly && pt_buff.y >= lz && pt_buff.y < lw) isVisibleEnlarged = true;
  }
}
template <bool useSwapping>
static inline void checkBlockVisibility(bool &isVisible, bool &isVisibleEnlarged, const int16_t pos[3],
                                        const M4 &M_d, const V4f &projParams_d, float voxelSize, int W, int H) {
  V4f p;
  float factor = (float)DSR_BLOCK_SIZE * voxelSize;
  isVisible = false; isVisibleEnlarged = false;
  p.x = (float)pos[0] * factor; p.y = (float)pos[1] * factor; p.z = (float)pos[2] * factor; p.w = 1.0f;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 0 0 0 */
  p.z += factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 0 0 1 */
  p.y += factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 0 1 1 */
  p.x += factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 1 1 1 */
  p.z -= factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 1 1 0 */
  p.y -= factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 1 0 0 */
  p.x -= factor; p.y += factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 0 1 0 */
  p.x += factor; p.y -= factor; p.z += factor;
  checkPointVisibility<useSwapping>(isVisible, isVisibleEnlarged, p, M_d, projParams_d, W, H); if (isVisible) return; /* 1 0 1 */
}

/* ITMSceneReconstructionEngine.h buildHashAllocAndVisibleTypePP.
 *
 * Delta for the fork's voxel GC (no CPU implementation exists in
 * the fork, InfiniTamDriver.h:198-206): decay leaves freed entries in place as
 * tombstones (ptr < -1, chain link kept).  The walk therefore (i) follows the
 * chain even through a free head and (ii) re-uses the FIRST free entry of the
 * chain ("in-place", type 1) before appending to the tail (type 2).  Without
 * decay no chain ever contains a free entry and a free head has offset 0, so this
 * is exactly upstream's behaviour. */
static inline void buildHashAllocAndVisibleTypePP(Engine &e, uint8_t *entriesVisibleType, int x, int y,
                                                  const M4 &invM_d, const V4f &invProj, float mu,
                                                  float oneOverVoxelSize) {
  const float *depth = e.depth.data();
  const dsr_hash_entry *hashTable = e.hashTable.data();
  float depth_measure = depth[x + y * e.W];
  if (depth_measure <= 0 || (depth_measure - mu) < 0 || (depth_measure - mu) < e.s.view_frustum_min ||
      (depth_measure + mu) > e.s.view_frustum_max)
    return;

  V4f pt_camera_f;
  pt_camera_f.z = depth_measure;
  pt_camera_f.x = pt_camera_f.z * (((float)x - invProj.z) * invProj.x);
  pt_camera_f.y = pt_camera_f.z * (((float)y - invProj.w) * invProj.y);
  float norm = sqrtf(pt_camera_f.x * pt_camera_f.x + pt_camera_f.y * pt_camera_f.y + pt_camera_f.z * pt_camera_f.z);

  V4f pt_buff, t;
  float f1 = 1.0f - mu / norm;
  pt_buff.x = pt_camera_f.x * f1; pt_buff.y = pt_camera_f.y * f1; pt_buff.z = pt_camera_f.z * f1; pt_buff.w = 1.0f;
  t = mul(invM_d, pt_buff);
  V3f point = {t.x * oneOverVoxelSize, t.y * oneOverVoxelSize, t.z * oneOverVoxelSize};
  float f2 = 1.0f + mu / norm;
  pt_buff.x = pt_camera_f.x * f2; pt_buff.y = pt_camera_f.y * f2; pt_buff.z = pt_camera_f.z * f2; pt_buff.w = 1.0f;
  t = mul(invM_d, pt_buff);
  V3f point_e = {t.x * oneOverVoxelSize, t.y * oneOverVoxelSize, t.z * oneOverVoxelSize};

  V3f direction = {point_e.x - point.x, point_e.y - point.y, point_e.z - point.z};
  norm = sqrtf(direction.x * direction.x + direction.y * direction.y + direction.z * direction.z);
  int noSteps = f2i(ceilf(2.0f * norm));
  float denom = (float)(noSteps - 1);
  direction.x /= denom; direction.y /= denom; direction.z /= denom;

  for (int i = 0; i < noSteps; i++) {
    int16_t bp[3] = {f2s(floorf(point.x)), f2s(floorf(point.y)), f2s(floorf(point.z))}; /* TO_SHORT_FLOOR3 */
    uint32_t hashIdx = hashIndex(e, bp[0], bp[1], bp[2]);
    bool isFound = false;
    int firstFree = -1;
    dsr_hash_entry he = hashTable[hashIdx];
    if (he.pos[0] == bp[0] && he.pos[1] == bp[1] && he.pos[2] == bp[2] && he.ptr >= -1) {
      entriesVisibleType[hashIdx] = (he.ptr == -1) ? (uint8_t)2 : (uint8_t)1;
      isFound = true;
    }
    if (!isFound) {
      if (he.ptr < -1) firstFree = (int)hashIdx;
      while (he.offset >= 1) {
        hashIdx = (uint32_t)(e.noBuckets + he.offset - 1);
        he = hashTable[hashIdx];
        if (he.pos[0] == bp[0] && he.pos[1] == bp[1] && he.pos[2] == bp[2] && he.ptr >= -1) {
          entriesVisibleType[hashIdx] = (he.ptr == -1) ? (uint8_t)2 : (uint8_t)1;
          isFound = true;
          break;
        }
        if (he.ptr < -1 && firstFree < 0) firstFree = (int)hashIdx;
      }
      if (!isFound) {
        bool isExcess = firstFree < 0;
        uint32_t target = isExcess ? hashIdx /* chain tail */ : (uint32_t)firstFree;
        e.entriesAllocType[target] = isExcess ? (uint8_t)2 : (uint8_t)1;
        if (!isExcess) entriesVisibleType[target] = 1;
        int16_t *bc = &e.blockCoords[4 * (size_t)target];
        bc[0] = bp[0]; bc[1] = bp[1]; bc[2] = bp[2]; bc[3] = 1;
      }
    }
    point.x += direction.x; point.y += direction.y; point.z += direction.z;
  }
}

/* ITMSceneReconstructionEngine_CPU<TVoxel,ITMVoxelBlockHash>::AllocateSceneFromDepth
 * (via ITMDenseMapper::ProcessFrame, InfiniTamDriver.h:140-145). */
static int allocate_scene_from_depth(Engine &e) {
  RenderState &rs = e.live;
  const float voxelSize = e.s.voxel_size;
  const float mu = e.s.mu;
  M4 M_d = e.M_d, invM_d;
  m4_inv(M_d, invM_d);
  V4f projParams_d = {e.calib.depth.fx, e.calib.depth.fy, e.calib.depth.cx, e.calib.depth.cy};
  V4f invProj = projParams_d;
  invProj.x = 1.0f / invProj.x; invProj.y = 1.0f / invProj.y;
  const float oneOverVoxelSize = 1.0f / (voxelSize * DSR_BLOCK_SIZE);
  const bool useSwapping = e.s.use_swapping != 0;
  uint8_t *evt = rs.entriesVisibleType.data();
  int status = DSR_OK;

  memset(e.entriesAllocType.data(), 0, (size_t)e.noTotalEntries);
  for (int i = 0; i < rs.noVisibleBlocks; i++) evt[rs.visibleEntryIDs[i]] = 3;

  /* build hashVisibility: raster order, last writer wins */
  for (int locId = 0; locId < e.W * e.H; locId++) {
    int y = locId / e.W, x = locId - y * e.W;
    buildHashAllocAndVisibleTypePP(e, evt, x, y, invM_d, invProj, mu, oneOverVoxelSize);
  }

  /* allocate, ascending entry index */
  int lastFreeVoxelBlockId = e.lastFreeBlockId, lastFreeExcessListId = e.lastFreeExcessListId;
  for (int targetIdx = 0; targetIdx < e.noTotalEntries; targetIdx++) {
    int vbaIdx, exlIdx;
    switch (e.entriesAllocType[targetIdx]) {
      case 1:
        vbaIdx = lastFreeVoxelBlockId; lastFreeVoxelBlockId--;
        if (vbaIdx >= 0) {
          const int16_t *bc = &e.blockCoords[4 * (size_t)targetIdx];
          dsr_hash_entry &he = e.hashTable[targetIdx];
          he.pos[0] = bc[0]; he.pos[1] = bc[1]; he.pos[2] = bc[2];
          he.ptr = e.voxelAllocationList[vbaIdx];
          /* upstream writes offset = 0; a free entry outside decay always has
           * offset 0 already, a tombstone keeps its link (see above). */
        } else status = DSR_E_OUT_OF_BLOCKS;
        break;
      case 2:
        vbaIdx = lastFreeVoxelBlockId; lastFreeVoxelBlockId--;
        exlIdx = lastFreeExcessListId; lastFreeExcessListId--;
        if (vbaIdx >= 0 && exlIdx >= 0) {
          const int16_t *bc = &e.blockCoords[4 * (size_t)targetIdx];
          dsr_hash_entry he; memset(&he, 0, sizeof he);
          he.pos[0] = bc[0]; he.pos[1] = bc[1]; he.pos[2] = bc[2];
          he.ptr = e.voxelAllocationList[vbaIdx];
          he.offset = 0;
          int exlOffset = e.excessAllocationList[exlIdx];
          e.hashTable[targetIdx].offset = exlOffset + 1;
          e.hashTable[e.noBuckets + exlOffset] = he;
          evt[e.noBuckets + exlOffset] = 1;
        } else status = DSR_E_OUT_OF_BLOCKS;
        break;
      default: break;
    }
  }

  /* build visible list, ascending */
  int noVisibleEntries = 0;
  for (int targetIdx = 0; targetIdx < e.noTotalEntries; targetIdx++) {
    uint8_t hashVisibleType = evt[targetIdx];
    const dsr_hash_entry &he = e.hashTable[targetIdx];
    if (hashVisibleType == 3) {
      bool isVisibleEnlarged, isVisible;
      if (useSwapping) {
        checkBlockVisibility<true>(isVisible, isVisibleEnlarged, he.pos, M_d, projParams_d, voxelSize, e.W, e.H);
        if (!isVisibleEnlarged) hashVisibleType = 0;
      } else {
        checkBlockVisibility<false>(isVisible, isVisibleEnlarged, he.pos, M_d, projParams_d, voxelSize, e.W, e.H);
        if (!isVisible) hashVisibleType = 0;
      }
      evt[targetIdx] = hashVisibleType;
    }
    if (useSwapping) {
      if (hashVisibleType > 0 && e.swapStates[targetIdx] != 2) e.swapStates[targetIdx] = 1;
    }
    if (hashVisibleType > 0) {
      if (noVisibleEntries < (int)rs.visibleEntryIDs.size()) rs.visibleEntryIDs[noVisibleEntries] = targetIdx;
      noVisibleEntries++;
    }
  }
  if (noVisibleEntries > (int)rs.visibleEntryIDs.size()) noVisibleEntries = (int)rs.visibleEntryIDs.size();

  /* reallocate deleted ones from previous swap operation */
  if (useSwapping) {
    for (int targetIdx = 0; targetIdx < e.noTotalEntries; targetIdx++) {
      if (evt[targetIdx] > 0 && e.hashTable[targetIdx].ptr == -1) {
        int vbaIdx = lastFreeVoxelBlockId; lastFreeVoxelBlockId--;
        if (vbaIdx >= 0) e.hashTable[targetIdx].ptr = e.voxelAllocationList[vbaIdx];
        else status = DSR_E_OUT_OF_BLOCKS;
      }
    }
  }

  /* heads are clamped at -1 (upstream lets them run negative; the fork throws,
   * InstanceReconstructor.cpp:662-671) */
  e.lastFreeBlockId = std::max(lastFreeVoxelBlockId, -1);
  e.lastFreeExcessListId = std::max(lastFreeExcessListId, -1);
  rs.noVisibleBlocks = noVisibleEntries;
  if (status != DSR_OK) e.stickyStatus = status;
  return status;
}

/* ---------------------------------------------------------- integration (A.4) */

/* ITMSceneReconstructionEngine.h interpolateBilinear<uchar4> (ITMPixelUtils.h). */
static inline V4f interpolateBilinear(const V4u *source, float px, float py, int W) {
  const int ix = f2i(floorf(px)), iy = f2i(floorf(py));
  const float dx = px - (float)ix, dy = py - (float)iy;
  V4u a = source[ix + iy * W];
  V4u b = {0, 0, 0, 0}, c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
  if (dx != 0) b = source[(ix + 1) + iy * W];
  if (dy != 0) c = source[ix + (iy + 1) * W];
  if (dx != 0 && dy != 0) d = source[(ix + 1) + (iy + 1) * W];
  V4f r;
  r.x = ((float)a.x * (1.0f - dx) * (1.0f - dy) + (float)b.x * dx * (1.0f - dy) + (float)c.x * (1.0f - dx) * dy + (float)d.x * dx * dy);
  r.y = ((float)a.y * (1.0f - dx) * (1.0f - dy) + (float)b.y * dx * (1.0f - dy) + (float)c.y * (1.0f - dx) * dy + (float)d.y * dx * dy);
  r.z = ((float)a.z * (1.0f - dx) * (1.0f - dy) + (float)b.z * dx * (1.0f - dy) + (float)c.z * (1.0f - dx) * dy + (float)d.z * dx * dy);
  r.w = ((float)a.w * (1.0f - dx) * (1.0f - dy) + (float)b.w * dx * (1.0f - dy) + (float)c.w * (1.0f - dx) * dy + (float)d.w * dx * dy);
  return r;
}

/* Fork: WeightParams.depthWeighting (InfiniTamDriver.h:100,138).  The fork's
 * formula is not recoverable from /root/reference; adopted definition:
 * newW = max(1, round(10 / z)) (closer measurements count more). */
static inline int depth_weight(float depth_measure) {
  int w = f2i(10.0f / depth_measure + 0.5f);
  return w < 1 ? 1 : w;
}

/* ITMSceneReconstructionEngine.h computeUpdatedVoxelDepthInfo */
static inline float computeUpdatedVoxelDepthInfo(dsr_voxel &voxel, const V4f &pt_model, const M4 &M_d,
                                                 const V4f &projParams_d, float mu, int maxW,
                                                 const float *depth, int W, int H, int depthWeighting) {
  V4f pt_camera = mul(M_d, pt_model);
  if (pt_camera.z <= 0) return -1;
  float px = projParams_d.x * pt_camera.x / pt_camera.z + projParams_d.z;
  float py = projParams_d.y * pt_camera.y / pt_camera.z + projParams_d.w;
  if ((px < 1) || (px > W - 2) || (py < 1) || (py > H - 2)) return -1;
  float depth_measure = depth[f2i(px + 0.5f) + f2i(py + 0.5f) * W];
  if (depth_measure <= 0.0f) return -1;
  float eta = depth_measure - pt_camera.z;
  if (eta < -mu) return eta;
  float oldF = sdf_to_float((float)voxel.sdf);
  int oldW = voxel.w_depth;
  const float q_ = eta / mu;
  float newF = (1.0f < q_) ? 1.0f : q_; /* MIN(1.0f, eta / mu) — ITMMath.h's macro ((a < b) ? a : b): a NaN quotient passes through (std::min would return 1) */
  int newW = depthWeighting ? depth_weight(depth_measure) : 1;
  newF = oldW * oldF + newW * newF;
  newW = oldW + newW;
  newF /= newW;
  newW = std::min(newW, maxW);
  voxel.sdf = sdf_from_float(newF);
  voxel.w_depth = (uint8_t)newW;
  return eta;
}

/* ITMSceneReconstructionEngine.h computeUpdatedVoxelColorInfo */
static inline void computeUpdatedVoxelColorInfo(dsr_voxel &voxel, const V4f &pt_model, const M4 &M_rgb,
                                                const V4f &projParams_rgb, int maxW, const V4u *rgb,
                                                int W, int H) {
  float oldW = (float)voxel.w_color;
  V3f oldC = {(float)voxel.clr[0] / 255.0f, (float)voxel.clr[1] / 255.0f, (float)voxel.clr[2] / 255.0f};
  V4f pt_camera = mul(M_rgb, pt_model);
  float px = projParams_rgb.x * pt_camera.x / pt_camera.z + projParams_rgb.z;
  float py = projParams_rgb.y * pt_camera.y / pt_camera.z + projParams_rgb.w;
  if ((px < 1) || (px > W - 2) || (py < 1) || (py > H - 2)) return;
  V4f m = interpolateBilinear(rgb, px, py, W);
  V3f rgb_measure = {m.x / 255.0f, m.y / 255.0f, m.z / 255.0f};
  float newW = 1;
  V3f newC = {oldC.x * oldW + rgb_measure.x * newW, oldC.y * oldW + rgb_measure.y * newW, oldC.z * oldW + rgb_measure.z * newW};
  newW = oldW + newW;
  newC.x /= newW; newC.y /= newW; newC.z /= newW;
  newW = std::min(newW, (float)maxW);
  voxel.clr[0] = (uint8_t)f2i(newC.x * 255.0f);
  voxel.clr[1] = (uint8_t)f2i(newC.y * 255.0f);
  voxel.clr[2] = (uint8_t)f2i(newC.z * 255.0f);
  voxel.w_color = (uint8_t)f2i(newW);
}

/* oracle-only diagnostics of the integration (orc_debug_integrate_stats): [0] blocks, [1] voxels whose depth was updated, [2] voxels
 * that passed the colour gate, [3] half blocks / [4] z-slices / [5] blocks with any update, [6] x columns with any update per
 * block, [7] x ROWS — the 64 (y, z) pairs of a block — with any updated voxel: the unit the HIP kernel stores (a lane owns a row:
 * 24 bytes written back per row with an update; tests/test_gpu_fullsize.py compares the kernel's own tallies with [7] and [2]) */
static long long g_int_stats[8];
static bool g_int_stats_on = false;

/* ITMSceneReconstructionEngine_CPU<TVoxel,ITMVoxelBlockHash>::IntegrateIntoScene */
static void integrate_into_scene(Engine &e) {
  const RenderState &rs = e.live;
  const float voxelSize = e.s.voxel_size, mu = e.s.mu;
  const int maxW = e.s.max_w;
  const M4 M_d = e.M_d;
  const M4 M_rgb = mul(e.calibInv, M_d);
  const V4f projParams_d = {e.calib.depth.fx, e.calib.depth.fy, e.calib.depth.cx, e.calib.depth.cy};
  const V4f projParams_rgb = {e.calib.rgb.fx, e.calib.rgb.fy, e.calib.rgb.cx, e.calib.rgb.cy};
  const bool stopAtMaxW = e.s.stop_integrating_at_max_w != 0;

#pragma omp parallel for schedule(dynamic, 64) num_threads(e.threads)
  for (int entryId = 0; entryId < rs.noVisibleBlocks; entryId++) {
    const dsr_hash_entry &he = e.hashTable[rs.visibleEntryIDs[entryId]];
    if (he.ptr < 0) continue;
    V3i globalPos = {he.pos[0] * DSR_BLOCK_SIZE, he.pos[1] * DSR_BLOCK_SIZE, he.pos[2] * DSR_BLOCK_SIZE};
    dsr_voxel *localVoxelBlock = &e.voxels[(size_t)he.ptr * DSR_BLOCK_SIZE3];
    long long nUpd = 0, nClr = 0; unsigned sliceMask = 0, xMask = 0; unsigned long long rowMask = 0;
    for (int z = 0; z < DSR_BLOCK_SIZE; z++)
      for (int y = 0; y < DSR_BLOCK_SIZE; y++)
        for (int x = 0; x < DSR_BLOCK_SIZE; x++) {
          int locId = x + y * DSR_BLOCK_SIZE + z * DSR_BLOCK_SIZE * DSR_BLOCK_SIZE;
          dsr_voxel &voxel = localVoxelBlock[locId];
          if (stopAtMaxW && voxel.w_depth == maxW) continue;
          V4f pt_model;
          pt_model.x = (float)(globalPos.x + x) * voxelSize;
          pt_model.y = (float)(globalPos.y + y) * voxelSize;
          pt_model.z = (float)(globalPos.z + z) * voxelSize;
          pt_model.w = 1.0f;
          /* ComputeUpdatedVoxelInfo<true,TVoxel>::compute */
          const dsr_voxel before = voxel;
          float eta = computeUpdatedVoxelDepthInfo(voxel, pt_model, M_d, projParams_d, mu, maxW, e.depth.data(),
                                                   e.W, e.H, e.depthWeighting);
          if (g_int_stats_on && (eta != -1 && !(eta < -mu))) { nUpd++; sliceMask |= 1u << z; xMask |= 1u << x; rowMask |= 1ull << (y + 8 * z); }
          (void)before;
          if ((eta > mu) || (fabsf(eta / mu) > 0.25f)) continue;
          computeUpdatedVoxelColorInfo(voxel, pt_model, M_rgb, projParams_rgb, maxW, e.rgb.data(), e.Wr, e.Hr);
          nClr++;
        }
    if (g_int_stats_on) {
#pragma omp critical
      {
        g_int_stats[0]++; g_int_stats[1] += nUpd; g_int_stats[2] += nClr;
        g_int_stats[3] += ((sliceMask & 0x0f) != 0) + ((sliceMask & 0xf0) != 0);
        g_int_stats[4] += __builtin_popcount(sliceMask);
        g_int_stats[5] += sliceMask != 0;
        g_int_stats[6] += __builtin_popcount(xMask); g_int_stats[7] += __builtin_popcountll(rowMask);
      }
    }
  }
}

/* ------------------------------------------------------------- raycast (A.5) */

struct IndexCache { /* ITMVoxelBlockHash::IndexCache */
  V3i blockPos = {0x7fffffff, 0x7fffffff, 0x7fffffff};
  long long blockPtr = -1; /* upstream: int (ptr * SDF_BLOCK_SIZE3 overflows it beyond 2^22 blocks; the fork's runtime sdfLocalBlockNum allows more) */
};

/* ITMRepresentationAccess.h pointToVoxelBlockPos */
static inline int pointToVoxelBlockPos(const V3i &p, V3i &b) {
  b.x = ((p.x < 0) ? p.x - DSR_BLOCK_SIZE + 1 : p.x) / DSR_BLOCK_SIZE;
  b.y = ((p.y < 0) ? p.y - DSR_BLOCK_SIZE + 1 : p.y) / DSR_BLOCK_SIZE;
  b.z = ((p.z < 0) ? p.z - DSR_BLOCK_SIZE + 1 : p.z) / DSR_BLOCK_SIZE;
  return p.x + (p.y - b.x) * DSR_BLOCK_SIZE + (p.z - b.y) * DSR_BLOCK_SIZE * DSR_BLOCK_SIZE -
         b.z * DSR_BLOCK_SIZE * DSR_BLOCK_SIZE * DSR_BLOCK_SIZE;
}

/* ITMRepresentationAccess.h readVoxel (cached and uncached forms) */
static inline dsr_voxel readVoxel(const Engine &e, const V3i &point, bool &isFound, IndexCache *cache) {
  V3i blockPos;
  int linearIdx = pointToVoxelBlockPos(point, blockPos);
  if (cache && blockPos.x == cache->blockPos.x && blockPos.y == cache->blockPos.y && blockPos.z == cache->blockPos.z) {
    isFound = true;
    return e.voxels[(size_t)cache->blockPtr + linearIdx];
  }
  int hashIdx = (int)hashIndex(e, blockPos.x, blockPos.y, blockPos.z);
  while (true) {
    const dsr_hash_entry &he = e.hashTable[hashIdx];
    if (he.pos[0] == blockPos.x && he.pos[1] == blockPos.y && he.pos[2] == blockPos.z && he.ptr >= 0) {
      isFound = true;
      if (cache) { cache->blockPos = blockPos; cache->blockPtr = (long long)he.ptr * DSR_BLOCK_SIZE3; }
      return e.voxels[(size_t)he.ptr * DSR_BLOCK_SIZE3 + linearIdx];
    }
    if (he.offset < 1) break;
    hashIdx = e.noBuckets + he.offset - 1;
  }
  isFound = false;
  return default_voxel();
}

static inline float ROUNDF(float x) { return (x < 0) ? (x - 0.5f) : (x + 0.5f); }

static inline float readFromSDF_float_uninterpolated(const Engine &e, const V3f &p, bool &isFound, IndexCache &cache) {
  V3i ip = {f2i(ROUNDF(p.x)), f2i(ROUNDF(p.y)), f2i(ROUNDF(p.z))};
  dsr_voxel res = readVoxel(e, ip, isFound, &cache);
  return sdf_to_float((float)res.sdf);
}

static inline float readFromSDF_float_interpolated(const Engine &e, const V3f &p, bool &isFound, IndexCache &cache) {
  float res1, res2, v1, v2;
  V3i pos = {f2i(floorf(p.x)), f2i(floorf(p.y)), f2i(floorf(p.z))}; /* TO_INT_FLOOR3 */
  V3f coeff = {p.x - (float)pos.x, p.y - (float)pos.y, p.z - (float)pos.z};
  auto rd = [&](int dx, int dy, int dz) -> float {
    V3i q = {pos.x + dx, pos.y + dy, pos.z + dz};
    return (float)readVoxel(e, q, isFound, &cache).sdf;
  };
  v1 = rd(0, 0, 0); v2 = rd(1, 0, 0);
  res1 = (1.0f - coeff.x) * v1 + coeff.x * v2;
  v1 = rd(0, 1, 0); v2 = rd(1, 1, 0);
  res1 = (1.0f - coeff.y) * res1 + coeff.y * ((1.0f - coeff.x) * v1 + coeff.x * v2);
  v1 = rd(0, 0, 1); v2 = rd(1, 0, 1);
  res2 = (1.0f - coeff.x) * v1 + coeff.x * v2;
  v1 = rd(0, 1, 1); v2 = rd(1, 1, 1);
  res2 = (1.0f - coeff.y) * res2 + coeff.y * ((1.0f - coeff.x) * v1 + coeff.x * v2);
  isFound = true;
  return sdf_to_float((1.0f - coeff.z) * res1 + coeff.z * res2);
}

/* oracle-only diagnostics of the ray march (orc_debug_raycast_stats) */
static long long g_rc_stats[8];
static bool g_rc_stats_on = false;
static int *g_rc_steps = nullptr;  /* per-pixel step counts of the last raycast (orc_debug_raycast_steps) */
static int g_rc_steps_w = 0;
/* orc_debug_raycast_stats2: what-if counters for two march accelerations (oracle-only study):
 * [0] found steps, [1] of those in blocks whose 512 sdf are all 32767 ("saturated block": the value is
 * known from the entry alone), [2] found steps with sdf == 1 in other blocks, [3] miss steps,
 * [4..6] miss steps inside a 2^3 / 4^3 / 8^3-block super-cell without ANY allocated block (the miss is
 * known without a table read), [7] rays */
static long long g_rc2[8];
static bool g_rc2_on = false;
static std::vector<uint8_t> g_rc2_blockSat;             /* per block index */
static std::unordered_map<unsigned long long, uint8_t> g_rc2_super[3]; /* occupied super-cells, 2/4/8 blocks wide */
static inline unsigned long long rc2_key(int bx, int by, int bz, int sh) {
  return ((unsigned long long)(unsigned short)(bx >> sh)) | ((unsigned long long)(unsigned short)(by >> sh) << 16) |
         ((unsigned long long)(unsigned short)(bz >> sh) << 32);
}

/* ITMVisualisationEngine.h castRay */
static inline bool castRay(const Engine &e, V4f &pt_out, int x, int y, const M4 &invM, const V4f &invProj,
                           float oneOverVoxelSize, float mu, const V2f &minmax) {
  V4f pt_camera_f, t;
  V3f pt_block_s, pt_block_e, rayDirection, pt_result;
  bool pt_found, hash_found;
  float sdfValue = 1.0f;
  float totalLength, stepLength, totalLengthMax, stepScale;

  stepScale = mu * oneOverVoxelSize;

  pt_camera_f.z = minmax.x;
  pt_camera_f.x = pt_camera_f.z * (((float)x - invProj.z) * invProj.x);
  pt_camera_f.y = pt_camera_f.z * (((float)y - invProj.w) * invProj.y);
  pt_camera_f.w = 1.0f;
  totalLength = sqrtf(pt_camera_f.x * pt_camera_f.x + pt_camera_f.y * pt_camera_f.y + pt_camera_f.z * pt_camera_f.z) * oneOverVoxelSize;
  t = mul(invM, pt_camera_f);
  pt_block_s = {t.x * oneOverVoxelSize, t.y * oneOverVoxelSize, t.z * oneOverVoxelSize};

  pt_camera_f.z = minmax.y;
  pt_camera_f.x = pt_camera_f.z * (((float)x - invProj.z) * invProj.x);
  pt_camera_f.y = pt_camera_f.z * (((float)y - invProj.w) * invProj.y);
  pt_camera_f.w = 1.0f;
  totalLengthMax = sqrtf(pt_camera_f.x * pt_camera_f.x + pt_camera_f.y * pt_camera_f.y + pt_camera_f.z * pt_camera_f.z) * oneOverVoxelSize;
  t = mul(invM, pt_camera_f);
  pt_block_e = {t.x * oneOverVoxelSize, t.y * oneOverVoxelSize, t.z * oneOverVoxelSize};

  rayDirection = {pt_block_e.x - pt_block_s.x, pt_block_e.y - pt_block_s.y, pt_block_e.z - pt_block_s.z};
  float direction_norm = 1.0f / sqrtf(rayDirection.x * rayDirection.x + rayDirection.y * rayDirection.y + rayDirection.z * rayDirection.z);
  rayDirection.x *= direction_norm; rayDirection.y *= direction_norm; rayDirection.z *= direction_norm;

  pt_result = pt_block_s;
  IndexCache cache;

  long long nMiss = 0, nFar = 0, nTri = 0, nSat = 0;
  long long c2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  while (totalLength < totalLengthMax) {
    sdfValue = readFromSDF_float_uninterpolated(e, pt_result, hash_found, cache);
    if (g_rc2_on) {
      V3i vp = {f2i(ROUNDF(pt_result.x)), f2i(ROUNDF(pt_result.y)), f2i(ROUNDF(pt_result.z))}, bp;
      pointToVoxelBlockPos(vp, bp);
      if (!hash_found) {
        c2[3]++;
        for (int k = 0; k < 3; k++) if (!g_rc2_super[k].count(rc2_key(bp.x, bp.y, bp.z, k + 1))) c2[4 + k]++;
      } else {
        c2[0]++;
        if (cache.blockPtr >= 0 && g_rc2_blockSat[(size_t)(cache.blockPtr / DSR_BLOCK_SIZE3)]) c2[1]++;
        else if (sdfValue >= 1.0f) c2[2]++;
      }
    }
    if (!hash_found) {
      stepLength = DSR_BLOCK_SIZE;
      nMiss++;
    } else {
      nFar++;
      if (sdfValue >= 1.0f) nSat++;
      if ((sdfValue <= 0.1f) && (sdfValue >= -0.5f)) { nTri++; sdfValue = readFromSDF_float_interpolated(e, pt_result, hash_found, cache); }
      if (sdfValue <= 0.0f) break;
      stepLength = std::max(sdfValue * stepScale, 1.0f);
    }
    pt_result.x += stepLength * rayDirection.x; pt_result.y += stepLength * rayDirection.y; pt_result.z += stepLength * rayDirection.z;
    totalLength += stepLength;
  }

  if (sdfValue <= 0.0f) {
    stepLength = sdfValue * stepScale;
    pt_result.x += stepLength * rayDirection.x; pt_result.y += stepLength * rayDirection.y; pt_result.z += stepLength * rayDirection.z;
    sdfValue = readFromSDF_float_interpolated(e, pt_result, hash_found, cache);
    stepLength = sdfValue * stepScale;
    pt_result.x += stepLength * rayDirection.x; pt_result.y += stepLength * rayDirection.y; pt_result.z += stepLength * rayDirection.z;
    pt_found = true;
  } else pt_found = false;

  if (g_rc2_on) {
#pragma omp critical
    { for (int k = 0; k < 7; k++) g_rc2[k] += c2[k]; g_rc2[7]++; }
  }
  if (g_rc_stats_on) {
#pragma omp critical
    {
      g_rc_stats[0]++; g_rc_stats[1] += nMiss; g_rc_stats[2] += nFar; g_rc_stats[3] += nTri;
      if (nMiss + nFar > g_rc_stats[4]) g_rc_stats[4] = nMiss + nFar;
      if (pt_found) g_rc_stats[5]++;
      if (nMiss + nFar > 200) g_rc_stats[6]++;
      if (nMiss + nFar > 50) g_rc_stats[7]++;
    }
  }
  if (g_rc_steps) g_rc_steps[x + y * g_rc_steps_w] = (int)(nMiss | (nSat << 10) | ((nFar - nSat) << 20));  /* packed 10-bit fields */
  pt_out.x = pt_result.x; pt_out.y = pt_result.y; pt_out.z = pt_result.z;
  pt_out.w = pt_found ? 1.0f : 0.0f;
  return pt_found;
}

/* ITMVisualisationEngine.h ProjectSingleBlock.  imgSize here is the compact
 * ceil(W/8) x ceil(H/8) range image (upstream keeps a full-size image and only
 * ever reads that corner: identical values for every cell castRay reads). */
static inline bool ProjectSingleBlock(const int16_t blockPos[3], const M4 &pose, const V4f &intrinsics, int imgW,
                                      int imgH, float voxelSize, V2i &upperLeft, V2i &lowerRight, V2f &zRange) {
  upperLeft = {imgW, imgH};
  lowerRight = {-1, -1};
  zRange = {FAR_AWAY, VERY_CLOSE};
  for (int corner = 0; corner < 8; ++corner) {
    int16_t tx = (int16_t)(blockPos[0] + ((corner & 1) ? 1 : 0));
    int16_t ty = (int16_t)(blockPos[1] + ((corner & 2) ? 1 : 0));
    int16_t tz = (int16_t)(blockPos[2] + ((corner & 4) ? 1 : 0));
    V4f pt3d = {(float)tx * (float)DSR_BLOCK_SIZE * voxelSize, (float)ty * (float)DSR_BLOCK_SIZE * voxelSize,
                (float)tz * (float)DSR_BLOCK_SIZE * voxelSize, 1.0f};
    pt3d = mul(pose, pt3d);
    if (pt3d.z < 1e-6f) continue;
    V2f pt2d;
    pt2d.x = (intrinsics.x * pt3d.x / pt3d.z + intrinsics.z) / (float)MINMAX_SUBSAMPLE;
    pt2d.y = (intrinsics.y * pt3d.y / pt3d.z + intrinsics.w) / (float)MINMAX_SUBSAMPLE;
    if ((float)upperLeft.x > floorf(pt2d.x)) upperLeft.x = f2i(floorf(pt2d.x));
    if ((float)lowerRight.x < ceilf(pt2d.x)) lowerRight.x = f2i(ceilf(pt2d.x));
    if ((float)upperLeft.y > floorf(pt2d.y)) upperLeft.y = f2i(floorf(pt2d.y));
    if ((float)lowerRight.y < ceilf(pt2d.y)) lowerRight.y = f2i(ceilf(pt2d.y));
    if (zRange.x > pt3d.z) zRange.x = pt3d.z;
    if (zRange.y < pt3d.z) zRange.y = pt3d.z;
  }
  if (upperLeft.x < 0) upperLeft.x = 0;
  if (upperLeft.y < 0) upperLeft.y = 0;
  if (lowerRight.x >= imgW) lowerRight.x = imgW - 1;
  if (lowerRight.y >= imgH) lowerRight.y = imgH - 1;
  if (upperLeft.x > lowerRight.x) return false;
  if (upperLeft.y > lowerRight.y) return false;
  if (zRange.x < VERY_CLOSE) zRange.x = VERY_CLOSE;
  if (zRange.y < VERY_CLOSE) return false;
  return true;
}

/* ITMVisualisationEngine_CPU<TVoxel,ITMVoxelBlockHash>::CreateExpectedDepths.
 * Upstream tiles each bounding box into 16x16 "rendering blocks" and stops
 * adding blocks after MAX_RENDERING_BLOCKS = 65536*4; the union of the tiles is
 * the bounding box, so the image is a min/max over covering boxes.  The cap is
 * NOT reproduced (it would silently drop blocks at the 5 mm preset, >262144
 * visible blocks) — DESIGN.md "deviations". */
static void create_expected_depths(Engine &e, RenderState &rs, const M4 &M, const V4f &intrinsics) {
  const int mw = (e.W + MINMAX_SUBSAMPLE - 1) / MINMAX_SUBSAMPLE, mh = (e.H + MINMAX_SUBSAMPLE - 1) / MINMAX_SUBSAMPLE;
  for (auto &px : rs.minmax) { px.x = FAR_AWAY; px.y = VERY_CLOSE; }
  for (int blockNo = 0; blockNo < rs.noVisibleBlocks; ++blockNo) {
    const dsr_hash_entry &bd = e.hashTable[rs.visibleEntryIDs[blockNo]];
    V2i ul, lr; V2f zr;
    bool valid = false;
    if (bd.ptr >= 0) valid = ProjectSingleBlock(bd.pos, M, intrinsics, mw, mh, e.s.voxel_size, ul, lr, zr);
    if (!valid) continue;
    for (int y = ul.y; y <= lr.y; ++y)
      for (int x = ul.x; x <= lr.x; ++x) {
        V2f &px = rs.minmax[x + y * mw];
        if (px.x > zr.x) px.x = zr.x;
        if (px.y < zr.y) px.y = zr.y;
      }
  }
}

/* ITMVisualisationEngine_CPU GenericRaycast */
static void generic_raycast(Engine &e, RenderState &rs, const M4 &invM, const V4f &projParams) {
  const int mw = (e.W + MINMAX_SUBSAMPLE - 1) / MINMAX_SUBSAMPLE;
  V4f invProj = projParams;
  invProj.x = 1.0f / invProj.x; invProj.y = 1.0f / invProj.y;
  const float mu = e.s.mu, oneOverVoxelSize = 1.0f / e.s.voxel_size;
#pragma omp parallel for schedule(dynamic, 256) num_threads(e.threads)
  for (int locId = 0; locId < e.W * e.H; ++locId) {
    int y = locId / e.W, x = locId - y * e.W;
    int locId2 = f2i(floorf((float)x / MINMAX_SUBSAMPLE)) + f2i(floorf((float)y / MINMAX_SUBSAMPLE)) * mw;
    castRay(e, rs.raycastResult[locId], x, y, invM, invProj, oneOverVoxelSize, mu, rs.minmax[locId2]);
  }
}

/* ITMVisualisationEngine.h computeNormalAndAngle<useSmoothing> (image space) */
template <bool useSmoothing>
static inline void computeNormalAndAngleImg(bool &foundPoint, int x, int y, const V4f *pointsRay, const V3f &lightSource,
                                            float voxelSize, int W, int H, V3f &outNormal, float &angle) {
  if (!foundPoint) return;
  V4f xp1_y, xm1_y, x_yp1, x_ym1;
  if (useSmoothing) {
    if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) { foundPoint = false; return; }
    xp1_y = pointsRay[(x + 2) + y * W]; x_yp1 = pointsRay[x + (y + 2) * W];
    xm1_y = pointsRay[(x - 2) + y * W]; x_ym1 = pointsRay[x + (y - 2) * W];
  } else {
    if (y <= 1 || y >= H - 2 || x <= 1 || x >= W - 2) { foundPoint = false; return; }
    xp1_y = pointsRay[(x + 1) + y * W]; x_yp1 = pointsRay[x + (y + 1) * W];
    xm1_y = pointsRay[(x - 1) + y * W]; x_ym1 = pointsRay[x + (y - 1) * W];
  }
  V4f diff_x = {0, 0, 0, 0}, diff_y = {0, 0, 0, 0};
  bool doPlus1 = false;
  if (xp1_y.w <= 0 || x_yp1.w <= 0 || xm1_y.w <= 0 || x_ym1.w <= 0) doPlus1 = true;
  else {
    diff_x = {xp1_y.x - xm1_y.x, xp1_y.y - xm1_y.y, xp1_y.z - xm1_y.z, xp1_y.w - xm1_y.w};
    diff_y = {x_yp1.x - x_ym1.x, x_yp1.y - x_ym1.y, x_yp1.z - x_ym1.z, x_yp1.w - x_ym1.w};
    float length_diff = std::max(diff_x.x * diff_x.x + diff_x.y * diff_x.y + diff_x.z * diff_x.z,
                                 diff_y.x * diff_y.x + diff_y.y * diff_y.y + diff_y.z * diff_y.z);
    if (length_diff * voxelSize * voxelSize > (0.15f * 0.15f)) doPlus1 = true;
  }
  if (doPlus1) {
    if (useSmoothing) {
      xp1_y = pointsRay[(x + 1) + y * W]; x_yp1 = pointsRay[x + (y + 1) * W];
      xm1_y = pointsRay[(x - 1) + y * W]; x_ym1 = pointsRay[x + (y - 1) * W];
      diff_x = {xp1_y.x - xm1_y.x, xp1_y.y - xm1_y.y, xp1_y.z - xm1_y.z, xp1_y.w - xm1_y.w};
      diff_y = {x_yp1.x - x_ym1.x, x_yp1.y - x_ym1.y, x_yp1.z - x_ym1.z, x_yp1.w - x_ym1.w};
    }
    if (xp1_y.w <= 0 || x_yp1.w <= 0 || xm1_y.w <= 0 || x_ym1.w <= 0) { foundPoint = false; return; }
  }
  outNormal.x = -(diff_x.y * diff_y.z - diff_x.z * diff_y.y);
  outNormal.y = -(diff_x.z * diff_y.x - diff_x.x * diff_y.z);
  outNormal.z = -(diff_x.x * diff_y.y - diff_x.y * diff_y.x);
  float normScale = 1.0f / sqrtf(outNormal.x * outNormal.x + outNormal.y * outNormal.y + outNormal.z * outNormal.z);
  outNormal.x *= normScale; outNormal.y *= normScale; outNormal.z *= normScale;
  angle = outNormal.x * lightSource.x + outNormal.y * lightSource.y + outNormal.z * lightSource.z;
  if (!(angle > 0.0f)) foundPoint = false;
}

static inline V4u grey(float angle) { /* drawPixelGrey */
  float outRes = (0.8f * angle + 0.2f) * 255.0f;
  uint8_t g = (uint8_t)f2i(outRes);
  return {g, g, g, g};
}

/* ITMVisualisationEngine_CPU CreateICPMaps_common + processPixelICP<true>
 * (via ITMTrackingController::Prepare, InfiniTamDriver.h:152). */
static void create_icp_maps(Engine &e) {
  RenderState &rs = e.live;
  M4 invM; m4_inv(e.M_d, invM);
  V4f proj = {e.calib.depth.fx, e.calib.depth.fy, e.calib.depth.cx, e.calib.depth.cy};
  generic_raycast(e, rs, invM, proj);
  V3f lightSource = {-invM.m[8], -invM.m[9], -invM.m[10]};
  const float voxelSize = e.s.voxel_size;
  const V4f *pointsRay = rs.raycastResult.data();
#pragma omp parallel for schedule(static) num_threads(e.threads)
  for (int y = 0; y < e.H; y++)
    for (int x = 0; x < e.W; x++) {
      int locId = x + y * e.W;
      V4f point = pointsRay[locId];
      bool foundPoint = point.w > 0.0f;
      V3f outNormal = {0, 0, 0}; float angle = 0;
      computeNormalAndAngleImg<true>(foundPoint, x, y, pointsRay, lightSource, voxelSize, e.W, e.H, outNormal, angle);
      if (foundPoint) {
        rs.raycastImage[locId] = grey(angle);
        e.pointsMap[locId] = {point.x * voxelSize, point.y * voxelSize, point.z * voxelSize, 1.0f};
        e.normalsMap[locId] = {outNormal.x, outNormal.y, outNormal.z, 0.0f};
      } else {
        V4f out4 = {0.0f, 0.0f, 0.0f, -1.0f};
        e.pointsMap[locId] = out4; e.normalsMap[locId] = out4;
        rs.raycastImage[locId] = {0, 0, 0, 0};
      }
    }
}

/* ITMRepresentationAccess.h computeSingleNormalFromSDF */
static inline V3f computeSingleNormalFromSDF(const Engine &e, const V3f &point) {
  bool isFound;
  V3f ret;
  V3i pos = {f2i(floorf(point.x)), f2i(floorf(point.y)), f2i(floorf(point.z))};
  V3f coeff = {point.x - (float)pos.x, point.y - (float)pos.y, point.z - (float)pos.z};
  V3f ncoeff = {1.0f - coeff.x, 1.0f - coeff.y, 1.0f - coeff.z};
  auto rd = [&](int dx, int dy, int dz) -> float {
    V3i q = {pos.x + dx, pos.y + dy, pos.z + dz};
    return (float)readVoxel(e, q, isFound, nullptr).sdf;
  };
  V4f front, back, tmp;
  front.x = rd(0, 0, 0); front.y = rd(1, 0, 0); front.z = rd(0, 1, 0); front.w = rd(1, 1, 0);
  back.x = rd(0, 0, 1); back.y = rd(1, 0, 1); back.z = rd(0, 1, 1); back.w = rd(1, 1, 1);
  float p1, p2, v1;
  /* gradient x */
  p1 = front.x * ncoeff.y * ncoeff.z + front.z * coeff.y * ncoeff.z + back.x * ncoeff.y * coeff.z + back.z * coeff.y * coeff.z;
  tmp.x = rd(-1, 0, 0); tmp.y = rd(-1, 1, 0); tmp.z = rd(-1, 0, 1); tmp.w = rd(-1, 1, 1);
  p2 = tmp.x * ncoeff.y * ncoeff.z + tmp.y * coeff.y * ncoeff.z + tmp.z * ncoeff.y * coeff.z + tmp.w * coeff.y * coeff.z;
  v1 = p1 * coeff.x + p2 * ncoeff.x;
  p1 = front.y * ncoeff.y * ncoeff.z + front.w * coeff.y * ncoeff.z + back.y * ncoeff.y * coeff.z + back.w * coeff.y * coeff.z;
  tmp.x = rd(2, 0, 0); tmp.y = rd(2, 1, 0); tmp.z = rd(2, 0, 1); tmp.w = rd(2, 1, 1);
  p2 = tmp.x * ncoeff.y * ncoeff.z + tmp.y * coeff.y * ncoeff.z + tmp.z * ncoeff.y * coeff.z + tmp.w * coeff.y * coeff.z;
  ret.x = sdf_to_float(p1 * ncoeff.x + p2 * coeff.x - v1);
  /* gradient y */
  p1 = front.x * ncoeff.x * ncoeff.z + front.y * coeff.x * ncoeff.z + back.x * ncoeff.x * coeff.z + back.y * coeff.x * coeff.z;
  tmp.x = rd(0, -1, 0); tmp.y = rd(1, -1, 0); tmp.z = rd(0, -1, 1); tmp.w = rd(1, -1, 1);
  p2 = tmp.x * ncoeff.x * ncoeff.z + tmp.y * coeff.x * ncoeff.z + tmp.z * ncoeff.x * coeff.z + tmp.w * coeff.x * coeff.z;
  v1 = p1 * coeff.y + p2 * ncoeff.y;
  p1 = front.z * ncoeff.x * ncoeff.z + front.w * coeff.x * ncoeff.z + back.z * ncoeff.x * coeff.z + back.w * coeff.x * coeff.z;
  tmp.x = rd(0, 2, 0); tmp.y = rd(1, 2, 0); tmp.z = rd(0, 2, 1); tmp.w = rd(1, 2, 1);
  p2 = tmp.x * ncoeff.x * ncoeff.z + tmp.y * coeff.x * ncoeff.z + tmp.z * ncoeff.x * coeff.z + tmp.w * coeff.x * coeff.z;
  ret.y = sdf_to_float(p1 * ncoeff.y + p2 * coeff.y - v1);
  /* gradient z */
  p1 = front.x * ncoeff.x * ncoeff.y + front.y * coeff.x * ncoeff.y + front.z * ncoeff.x * coeff.y + front.w * coeff.x * coeff.y;
  tmp.x = rd(0, 0, -1); tmp.y = rd(1, 0, -1); tmp.z = rd(0, 1, -1); tmp.w = rd(1, 1, -1);
  p2 = tmp.x * ncoeff.x * ncoeff.y + tmp.y * coeff.x * ncoeff.y + tmp.z * ncoeff.x * coeff.y + tmp.w * coeff.x * coeff.y;
  v1 = p1 * coeff.z + p2 * ncoeff.z;
  p1 = back.x * ncoeff.x * ncoeff.y + back.y * coeff.x * ncoeff.y + back.z * ncoeff.x * coeff.y + back.w * coeff.x * coeff.y;
  tmp.x = rd(0, 0, 2); tmp.y = rd(1, 0, 2); tmp.z = rd(0, 1, 2); tmp.w = rd(1, 1, 2);
  p2 = tmp.x * ncoeff.x * ncoeff.y + tmp.y * coeff.x * ncoeff.y + tmp.z * ncoeff.x * coeff.y + tmp.w * coeff.x * coeff.y;
  ret.z = sdf_to_float(p1 * ncoeff.z + p2 * coeff.z - v1);
  return ret;
}

/* ITMRepresentationAccess.h readFromSDF_color4u_interpolated */
static inline V4f readFromSDF_color4u_interpolated(const Engine &e, const V3f &point) {
  bool isFound;
  V3f ret = {0, 0, 0};
  V3i pos = {f2i(floorf(point.x)), f2i(floorf(point.y)), f2i(floorf(point.z))};
  V3f coeff = {point.x - (float)pos.x, point.y - (float)pos.y, point.z - (float)pos.z};
  auto acc = [&](int dx, int dy, int dz, float w) {
    V3i q = {pos.x + dx, pos.y + dy, pos.z + dz};
    dsr_voxel r = readVoxel(e, q, isFound, nullptr);
    ret.x += w * (float)r.clr[0]; ret.y += w * (float)r.clr[1]; ret.z += w * (float)r.clr[2];
  };
  acc(0, 0, 0, (1.0f - coeff.x) * (1.0f - coeff.y) * (1.0f - coeff.z));
  acc(1, 0, 0, (coeff.x) * (1.0f - coeff.y) * (1.0f - coeff.z));
  acc(0, 1, 0, (1.0f - coeff.x) * (coeff.y) * (1.0f - coeff.z));
  acc(1, 1, 0, (coeff.x) * (coeff.y) * (1.0f - coeff.z));
  acc(0, 0, 1, (1.0f - coeff.x) * (1.0f - coeff.y) * coeff.z);
  acc(1, 0, 1, (coeff.x) * (1.0f - coeff.y) * coeff.z);
  acc(0, 1, 1, (1.0f - coeff.x) * (coeff.y) * coeff.z);
  acc(1, 1, 1, (coeff.x) * (coeff.y) * coeff.z);
  return {ret.x / 255.0f, ret.y / 255.0f, ret.z / 255.0f, 255.0f / 255.0f};
}

/* ITMVisualisationEngine_CPU<TVoxel,ITMVoxelBlockHash>::FindVisibleBlocks */
static void find_visible_blocks(Engine &e, RenderState &rs, const M4 &M, const V4f &projParams) {
  int n = 0;
  for (int targetIdx = 0; targetIdx < e.noTotalEntries; targetIdx++) {
    const dsr_hash_entry &he = e.hashTable[targetIdx];
    uint8_t vis = 0;
    if (he.ptr >= 0) {
      bool isVisible, isVisibleEnlarged;
      checkBlockVisibility<false>(isVisible, isVisibleEnlarged, he.pos, M, projParams, e.s.voxel_size, e.W, e.H);
      vis = isVisible;
    }
    if (vis > 0 && n < (int)rs.visibleEntryIDs.size()) rs.visibleEntryIDs[n++] = targetIdx;
  }
  rs.noVisibleBlocks = n;
}

/* ITMVisualisationEngine_CPU RenderImage_common + the fork's two extra types
 * (InfiniTamDriver.cpp:16-34).  The fork's exact DEPTH / DEPTH_WEIGHT shaders are
 * not in /root/reference; adopted definitions:
 *   FREECAMERA_DEPTH: z of M * (p * voxelSize) in metres for every ray that hit,
 *     0 for a miss (consumers: InstanceReconstructor.cpp:861-867,896-897,
 *     EvaluationCallback.cpp:58-59);
 *   COLOUR_FROM_DEPTH_WEIGHT: t = w_depth(nearest voxel)/maxW clamped to [0,1],
 *     colour = (255*(1-t), 0, 255*t, 255): red = low weight, blue = high
 *     (README.md:30-34). */
static void render_image(Engine &e, RenderState &rs, const M4 &M, const V4f &proj, int type, V4u *outRgba, float *outDepth) {
  M4 invM; m4_inv(M, invM);
  generic_raycast(e, rs, invM, proj);
  V3f lightSource = {-invM.m[8], -invM.m[9], -invM.m[10]};
  const V4f *pointsRay = rs.raycastResult.data();
  const float voxelSize = e.s.voxel_size;
#pragma omp parallel for schedule(dynamic, 256) num_threads(e.threads)
  for (int locId = 0; locId < e.W * e.H; locId++) {
    V4f ptRay = pointsRay[locId];
    V3f point = {ptRay.x, ptRay.y, ptRay.z};
    bool foundPoint = ptRay.w > 0;
    V4u out = {0, 0, 0, 0};
    switch (type) {
      case DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME:
        if (foundPoint) { /* drawPixelColour */
          V4f clr = readFromSDF_color4u_interpolated(e, point);
          out.x = (uint8_t)f2i(clr.x * 255.0f); out.y = (uint8_t)f2i(clr.y * 255.0f); out.z = (uint8_t)f2i(clr.z * 255.0f); out.w = 255;
        }
        break;
      case DSR_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL:
      case DSR_IMAGE_FREECAMERA_SHADED: {
        V3f n = {0, 0, 0}; float angle = 0;
        if (foundPoint) { /* computeNormalAndAngle<TVoxel,TIndex> */
          n = computeSingleNormalFromSDF(e, point);
          float normScale = 1.0f / sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
          n.x *= normScale; n.y *= normScale; n.z *= normScale;
          angle = n.x * lightSource.x + n.y * lightSource.y + n.z * lightSource.z;
          if (!(angle > 0.0f)) foundPoint = false;
        }
        if (foundPoint) {
          if (type == DSR_IMAGE_FREECAMERA_SHADED) out = grey(angle);
          else { /* drawPixelNormal (w untouched -> 0 after Clear()) */
            out.x = (uint8_t)f2i((0.3f + (-n.x + 1.0f) * 0.35f) * 255.0f);
            out.y = (uint8_t)f2i((0.3f + (-n.y + 1.0f) * 0.35f) * 255.0f);
            out.z = (uint8_t)f2i((0.3f + (-n.z + 1.0f) * 0.35f) * 255.0f);
            out.w = 0;
          }
        }
      } break;
      case DSR_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT:
        if (foundPoint) {
          bool isFound;
          V3i ip = {f2i(ROUNDF(point.x)), f2i(ROUNDF(point.y)), f2i(ROUNDF(point.z))};
          dsr_voxel v = readVoxel(e, ip, isFound, nullptr);
          float t = (float)v.w_depth / (float)e.s.max_w;
          t = std::min(1.0f, std::max(0.0f, t));
          out.x = (uint8_t)f2i(255.0f * (1.0f - t)); out.y = 0; out.z = (uint8_t)f2i(255.0f * t); out.w = 255;
        }
        break;
      case DSR_IMAGE_FREECAMERA_DEPTH:
      default: break;
    }
    rs.raycastImage[locId] = out;
    if (outRgba) outRgba[locId] = out;
    if (outDepth) {
      float d = 0.0f;
      if (ptRay.w > 0) {
        V4f pm = {ptRay.x * voxelSize, ptRay.y * voxelSize, ptRay.z * voxelSize, 1.0f};
        d = mul(M, pm).z;
      }
      outDepth[locId] = d;
    }
  }
}

/* --------------------------------------------------------------- swapping */

/* ITMSwappingEngine.h combineVoxelDepthInformation / combineVoxelColorInformation
 * (CombineVoxelInformation<true>) */
static inline void combineVoxelInformation(const dsr_voxel &src, dsr_voxel &dst, int maxW) {
  {
    int newW = dst.w_depth, oldW = src.w_depth;
    float newF = sdf_to_float((float)dst.sdf), oldF = sdf_to_float((float)src.sdf);
    if (oldW != 0) {
      newF = oldW * oldF + newW * newF;
      newW = oldW + newW;
      newF /= newW;
      newW = std::min(newW, maxW);
      dst.w_depth = (uint8_t)newW;
      dst.sdf = sdf_from_float(newF);
    }
  }
  {
    int newW = dst.w_color, oldW = src.w_color;
    V3f newC = {(float)dst.clr[0] / 255.0f, (float)dst.clr[1] / 255.0f, (float)dst.clr[2] / 255.0f};
    V3f oldC = {(float)src.clr[0] / 255.0f, (float)src.clr[1] / 255.0f, (float)src.clr[2] / 255.0f};
    if (oldW != 0) {
      newC.x = oldC.x * (float)oldW + newC.x * (float)newW;
      newC.y = oldC.y * (float)oldW + newC.y * (float)newW;
      newC.z = oldC.z * (float)oldW + newC.z * (float)newW;
      newW = oldW + newW;
      newC.x /= (float)newW; newC.y /= (float)newW; newC.z /= (float)newW;
      newW = std::min(newW, maxW);
      dst.clr[0] = (uint8_t)f2i(newC.x * 255.0f); dst.clr[1] = (uint8_t)f2i(newC.y * 255.0f); dst.clr[2] = (uint8_t)f2i(newC.z * 255.0f);
      dst.w_color = (uint8_t)newW;
    }
  }
}

/* ITMSwappingEngine_CPU<TVoxel,ITMVoxelBlockHash>::IntegrateGlobalIntoLocal (+ LoadFromGlobalMemory):
 * the first <= SDF_TRANSFER_BLOCK_NUM entries (ascending) in state 1 are merged with their copy in
 * the global cache, if any, and become state 2.  Deviation: an entry whose stored copy could not be
 * given a block (voxel block array exhausted, ptr < 0) keeps state 1 instead of being merged
 * through an invalid pointer. */
static void swap_in(Engine &e) {
  std::vector<int> needed;
  for (int t = 0; t < e.noTotalEntries; t++) {
    if ((int)needed.size() >= DSR_TRANSFER_BLOCK_NUM) break;
    if (e.swapStates[t] == 1) needed.push_back(t);
  }
  const int maxW = e.s.max_w;
  for (int id : needed) {
    if (e.hasStored[id]) {
      const int ptr = e.hashTable[id].ptr;
      if (ptr < 0) continue;
      const std::vector<dsr_voxel> &src = e.storedBlocks[id];
      dsr_voxel *dst = &e.voxels[(size_t)ptr * DSR_BLOCK_SIZE3];
      for (int v = 0; v < DSR_BLOCK_SIZE3; v++) combineVoxelInformation(src[v], dst[v], maxW);
    }
    e.swapStates[id] = 2;
  }
}

/* ITMSwappingEngine_CPU<TVoxel,ITMVoxelBlockHash>::SaveToGlobalMemory: the first
 * <= SDF_TRANSFER_BLOCK_NUM entries (ascending) that are in state 2, resident and NOT visible are
 * copied to the global cache, their block is reset and pushed back on the free list, ptr = -1.
 * (Upstream bounds the free-list push with `vbaIdx < SDF_BUCKET_NUM - 1`; the block-array size
 * is the meaningful bound and the one used here — neither can trigger, since only allocated
 * blocks are returned.) */
static void swap_out(Engine &e) {
  const uint8_t *evt = e.live.entriesVisibleType.data();
  int count = 0;
  int noAllocatedVoxelEntries = e.lastFreeBlockId;
  for (int t = 0; t < e.noTotalEntries; t++) {
    if (count >= DSR_TRANSFER_BLOCK_NUM) break;
    const int localPtr = e.hashTable[t].ptr;
    if (e.swapStates[t] == 2 && localPtr >= 0 && evt[t] == 0) {
      dsr_voxel *blk = &e.voxels[(size_t)localPtr * DSR_BLOCK_SIZE3];
      e.storedBlocks[t].assign(blk, blk + DSR_BLOCK_SIZE3);
      e.hasStored[t] = 1;
      if (!e.ownsSlot[t]) { e.ownsSlot[t] = 1; e.hostStoreSlots++; }
      e.swapStates[t] = 0;
      int vbaIdx = noAllocatedVoxelEntries;
      if (vbaIdx < e.noBlocks - 1) {
        noAllocatedVoxelEntries++;
        e.voxelAllocationList[vbaIdx + 1] = localPtr;
        e.hashTable[t].ptr = -1;
        for (int i = 0; i < DSR_BLOCK_SIZE3; i++) blk[i] = default_voxel();
      }
      count++;
    }
  }
  e.lastFreeBlockId = noAllocatedVoxelEntries;
}

/* ------------------------------------------------------------------ decay */

/* Fork: ITMDenseMapper::Decay (InfiniTamDriver.h:201-235).  No CPU version
 * exists in the fork (InfiniTamDriver.h:198-200) — specification by inference,
 * SURVEY.md A.6 / DESIGN.md "voxel GC":
 *  - !forceAll: the current visible list is pushed on a FIFO; once more than
 *    minAge lists are queued the oldest is popped and its blocks are processed;
 *  - forceAll: every entry with ptr >= 0, ascending, is processed;
 *  - processing a block: voxels with w_depth <= maxWeight are reset to the
 *    default voxel; if afterwards all 512 voxels have w_depth == 0 the block is
 *    freed: its VBA slot is pushed back on the free list (in candidate order),
 *    the entry becomes a tombstone (ptr = -2, chain link kept), it leaves the
 *    live visible list, decayedBlockCount++. */
static void decay(Engine &e, int maxWeight, int minAge, bool forceAll) {
  RenderState &rs = e.live;
  std::vector<int32_t> cand;
  if (forceAll) {
    for (int t = 0; t < e.noTotalEntries; t++) if (e.hashTable[t].ptr >= 0) cand.push_back(t);
  } else {
    e.decayFifo.emplace_back(rs.visibleEntryIDs.begin(), rs.visibleEntryIDs.begin() + rs.noVisibleBlocks);
    if ((int)e.decayFifo.size() <= minAge) return;
    cand.swap(e.decayFifo.front());
    e.decayFifo.pop_front();
  }
  bool anyFreed = false;
  for (int32_t t : cand) {
    dsr_hash_entry &he = e.hashTable[t];
    if (he.ptr < 0) continue;
    dsr_voxel *blk = &e.voxels[(size_t)he.ptr * DSR_BLOCK_SIZE3];
    int empty = 0;
    for (int i = 0; i < DSR_BLOCK_SIZE3; i++) {
      if ((int)blk[i].w_depth <= maxWeight) blk[i] = default_voxel();
      if (blk[i].w_depth == 0) empty++;
    }
    if (empty == DSR_BLOCK_SIZE3) {
      e.lastFreeBlockId++;
      e.voxelAllocationList[e.lastFreeBlockId] = he.ptr;
      he.ptr = -2;
      rs.entriesVisibleType[t] = 0;
      if (!e.swapStates.empty()) { e.swapStates[t] = 0; if (e.hasStored[t]) { e.hasStored[t] = 0; e.storedBlocks.erase(t); } }
      e.decayedBlockCount++;
      anyFreed = true;
    }
  }
  if (anyFreed) {
    int n = 0;
    for (int i = 0; i < rs.noVisibleBlocks; i++) {
      int32_t id = rs.visibleEntryIDs[i];
      if (rs.entriesVisibleType[id] != 0) rs.visibleEntryIDs[n++] = id;
    }
    rs.noVisibleBlocks = n;
  }
}

} /* namespace */

/* =========================================================== exported C ABI */

struct dsr_engine { Engine e; };
#define E (h->e)

extern "C" {

int orc_abi_version(void) { return DSR_ABI_VERSION; }

void orc_default_settings(dsr_settings *s) {
  memset(s, 0, sizeof *s);
  /* ITMLibSettings.cpp (upstream v2): sceneParams(0.02f, 100, 0.005f, 0.2f, 3.0f, false) */
  s->voxel_size = 0.005f; s->mu = 0.02f; s->max_w = 100;
  s->view_frustum_min = 0.2f; s->view_frustum_max = 3.0f;
  s->stop_integrating_at_max_w = 0;
  s->sdf_local_block_num = DSR_DEFAULT_LOCAL_BLOCK_NUM;
  s->hash_bucket_num = DSR_DEFAULT_BUCKET_NUM;
  s->excess_list_size = DSR_DEFAULT_EXCESS_LIST_SIZE;
  s->use_swapping = 0; s->use_bilateral_filter = 0; s->device = -1; s->sync_status = 1;
}

const char *orc_last_error(void) { return g_err.c_str(); }

int orc_engine_create(const dsr_settings *settings, const dsr_calib *calib, dsr_engine **out) {
  if (!settings || !calib || !out) return fail(DSR_E_ARG, "null argument");
  const dsr_settings &s = *settings;
  if (s.hash_bucket_num <= 0 || (s.hash_bucket_num & (s.hash_bucket_num - 1))) return fail(DSR_E_ARG, "hash_bucket_num must be a power of two");
  if (s.excess_list_size <= 0 || s.sdf_local_block_num <= 0) return fail(DSR_E_ARG, "bad table sizes");
  if (!(s.voxel_size > 0) || !(s.mu > 0) || s.max_w < 1 || s.max_w > 255) return fail(DSR_E_ARG, "bad scene params");
  if (calib->depth.width <= 0 || calib->depth.height <= 0) return fail(DSR_E_ARG, "bad image size");
  dsr_engine *h = new (std::nothrow) dsr_engine();
  if (!h) return fail(DSR_E_NOMEM, "oom");
  Engine &e = h->e;
  e.s = s; e.calib = *calib;
  e.W = calib->depth.width; e.H = calib->depth.height;
  e.Wr = calib->rgb.width; e.Hr = calib->rgb.height;
  e.noBuckets = s.hash_bucket_num; e.noExcess = s.excess_list_size;
  e.noTotalEntries = e.noBuckets + e.noExcess; e.noBlocks = s.sdf_local_block_num;
  e.hashMask = (uint32_t)(e.noBuckets - 1);
  M4 trafo; memcpy(trafo.m, calib->trafo_rgb_to_depth, sizeof trafo.m);
  if (!m4_inv(trafo, e.calibInv)) { delete h; return fail(DSR_E_ARG, "singular trafo_rgb_to_depth"); }
  try {
    e.hashTable.resize(e.noTotalEntries);
    e.excessAllocationList.resize(e.noExcess);
    e.voxels.resize((size_t)e.noBlocks * DSR_BLOCK_SIZE3);
    e.voxelAllocationList.resize(e.noBlocks);
    e.entriesAllocType.resize(e.noTotalEntries);
    if (s.use_swapping) { e.swapStates.assign(e.noTotalEntries, 0); e.hasStored.assign(e.noTotalEntries, 0); e.ownsSlot.assign(e.noTotalEntries, 0); }
    e.blockCoords.resize(4 * (size_t)e.noTotalEntries);
    const int mw = (e.W + 7) / 8, mh = (e.H + 7) / 8;
    for (RenderState *rs : {&e.live, &e.freeview}) {
      rs->visibleEntryIDs.resize(e.noBlocks);
      rs->entriesVisibleType.resize(e.noTotalEntries);
      rs->minmax.resize((size_t)mw * mh);
      rs->raycastResult.resize((size_t)e.W * e.H);
      rs->raycastImage.resize((size_t)e.W * e.H);
    }
    e.rgb.resize((size_t)e.Wr * e.Hr);
    e.depth.resize((size_t)e.W * e.H);
    e.pointsMap.resize((size_t)e.W * e.H);
    e.normalsMap.resize((size_t)e.W * e.H);
  } catch (const std::bad_alloc &) { delete h; return fail(DSR_E_NOMEM, "oom"); }
  e.M_d = m4_identity(); e.invM_d = m4_identity();
  reset_scene(e);
  *out = h;
  return DSR_OK;
}

void orc_engine_destroy(dsr_engine *h) { delete h; }

int orc_reset_scene(dsr_engine *h) { if (!h) return fail(DSR_E_ARG, "null"); reset_scene(E); return DSR_OK; }
int orc_sync(dsr_engine *h) { return h ? DSR_OK : fail(DSR_E_ARG, "null"); }
int orc_device_synchronize(void) { return DSR_OK; }  // the CPU restatement runs synchronously
int orc_device_mem_info(int, uint64_t *free_bytes, uint64_t *total_bytes) {  /* the "device" is the host: its RAM */
  if (!free_bytes || !total_bytes) return fail(DSR_E_ARG, "null argument");
  const long page = sysconf(_SC_PAGESIZE);
  *total_bytes = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)page;
  *free_bytes = (uint64_t)sysconf(_SC_AVPHYS_PAGES) * (uint64_t)page;
  return DSR_OK;
}

int orc_update_view(dsr_engine *h, const uint8_t *rgba, const int16_t *depth_mm) {
  if (!h || !rgba || !depth_mm) return fail(DSR_E_ARG, "null");
  memcpy(E.rgb.data(), rgba, (size_t)E.Wr * E.Hr * 4);
  convert_depth(E, depth_mm);
  if (E.s.use_bilateral_filter) filter_depth(E);
  E.hasView = true;
  return DSR_OK;
}
int orc_update_view_dev(dsr_engine *h, const void *rgba, const void *depth_mm) {
  return orc_update_view(h, (const uint8_t *)rgba, (const int16_t *)depth_mm);
}
/* InfiniTamDriver::UpdateView as a whole (InfiniTamDriver.cpp:211-224): CvToItm(rgb_image) (:81-100, restated in
 * orc_bgr_to_rgba) followed by viewBuilder->UpdateView. */
int orc_bgr_to_rgba(const uint8_t *bgr, uint8_t *rgba_out, int n);
int orc_update_view_bgr(dsr_engine *h, const uint8_t *bgr, const int16_t *depth_mm) {
  if (!h || !bgr || !depth_mm) return fail(DSR_E_ARG, "null");
  std::vector<uint8_t> rgba((size_t)E.Wr * E.Hr * 4);
  int st = orc_bgr_to_rgba(bgr, rgba.data(), E.Wr * E.Hr);
  if (st) return st;
  return orc_update_view(h, rgba.data(), depth_mm);
}
int orc_set_view_float(dsr_engine *h, const uint8_t *rgba, const float *depth_m) {
  if (!h || !rgba || !depth_m) return fail(DSR_E_ARG, "null");
  memcpy(E.rgb.data(), rgba, (size_t)E.Wr * E.Hr * 4);
  memcpy(E.depth.data(), depth_m, (size_t)E.W * E.H * sizeof(float));
  E.hasView = true;
  return DSR_OK;
}
int orc_set_view_float_dev(dsr_engine *h, const void *rgba, const void *depth_m) {
  return orc_set_view_float(h, (const uint8_t *)rgba, (const float *)depth_m);
}
int orc_pin_host_buffer(void *ptr, size_t bytes) { return (ptr && bytes) ? DSR_OK : fail(DSR_E_ARG, "null"); }  /* nothing to pin on the host */
int orc_unpin_host_buffer(void *ptr) { return ptr ? DSR_OK : fail(DSR_E_ARG, "null"); }
int orc_get_view(dsr_engine *h, uint8_t *rgba_out, float *depth_m_out) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (rgba_out) memcpy(rgba_out, E.rgb.data(), (size_t)E.Wr * E.Hr * 4);
  if (depth_m_out) memcpy(depth_m_out, E.depth.data(), (size_t)E.W * E.H * sizeof(float));
  return DSR_OK;
}

/* ITMPose::SetInvM: invM.inv(M); GetInvM(): M.inv(ret). */
int orc_set_pose_inv_m(dsr_engine *h, const float inv_m[16]) {
  if (!h || !inv_m) return fail(DSR_E_ARG, "null");
  M4 im; memcpy(im.m, inv_m, sizeof im.m);
  if (!m4_inv(im, E.M_d)) return fail(DSR_E_ARG, "singular pose");
  m4_inv(E.M_d, E.invM_d);
  return DSR_OK;
}
int orc_set_pose_m(dsr_engine *h, const float m[16]) {
  if (!h || !m) return fail(DSR_E_ARG, "null");
  memcpy(E.M_d.m, m, sizeof E.M_d.m);
  if (!m4_inv(E.M_d, E.invM_d)) return fail(DSR_E_ARG, "singular pose");
  return DSR_OK;
}
int orc_get_pose(dsr_engine *h, float m_out[16], float inv_m_out[16]) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (m_out) memcpy(m_out, E.M_d.m, sizeof E.M_d.m);
  if (inv_m_out) memcpy(inv_m_out, E.invM_d.m, sizeof E.invM_d.m);
  return DSR_OK;
}

int orc_set_fusion_weight_params(dsr_engine *h, int depth_weighting) {
  if (!h) return fail(DSR_E_ARG, "null");
  E.depthWeighting = depth_weighting ? 1 : 0;
  return DSR_OK;
}

int orc_allocate_scene_from_depth(dsr_engine *h) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  int st = allocate_scene_from_depth(E);
  if (st != DSR_OK) return fail(st, "out of voxel blocks / excess list entries");
  return DSR_OK;
}
int orc_integrate_into_scene(dsr_engine *h) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  integrate_into_scene(E);
  return DSR_OK;
}
/* ITMDenseMapper::ProcessFrame */
int orc_process_frame(dsr_engine *h) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  int st = allocate_scene_from_depth(E);
  integrate_into_scene(E);
  if (E.s.use_swapping) { swap_in(E); swap_out(E); }  /* ITMDenseMapper::ProcessFrame */
  E.framesProcessed++;
  if (st != DSR_OK) return fail(st, "out of voxel blocks / excess list entries");
  return DSR_OK;
}
/* ITMTrackingController::Prepare */
int orc_prepare(dsr_engine *h) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (E.live.noVisibleBlocks <= 0) return DSR_OK; /* InfiniTamDriver.h:150 */
  V4f proj = {E.calib.depth.fx, E.calib.depth.fy, E.calib.depth.cx, E.calib.depth.cy};
  create_expected_depths(E, E.live, E.M_d, proj);
  create_icp_maps(E);
  return DSR_OK;
}

int orc_decay(dsr_engine *h, int max_weight, int min_age, int force_all_voxels) {
  if (!h) return fail(DSR_E_ARG, "null");
  decay(E, max_weight, min_age, force_all_voxels != 0);
  return DSR_OK;
}

/* ITMMainEngine::GetImage */
int orc_get_image(dsr_engine *h, int type, const float pose_m[16], const float intrinsics[4], uint8_t *rgba_out,
                  float *depth_out) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  const size_t P = (size_t)E.W * E.H;
  switch (type) {
    case DSR_IMAGE_ORIGINAL_RGB:
      if (rgba_out) memcpy(rgba_out, E.rgb.data(), P * 4);
      return DSR_OK;
    case DSR_IMAGE_SCENERAYCAST:
      if (rgba_out) memcpy(rgba_out, E.live.raycastImage.data(), P * 4);
      return DSR_OK;
    case DSR_IMAGE_FREECAMERA_SHADED:
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME:
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL:
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT:
    case DSR_IMAGE_FREECAMERA_DEPTH: {
      M4 M = E.M_d;
      if (pose_m) memcpy(M.m, pose_m, sizeof M.m);
      V4f proj = {E.calib.depth.fx, E.calib.depth.fy, E.calib.depth.cx, E.calib.depth.cy};
      if (intrinsics) proj = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
      find_visible_blocks(E, E.freeview, M, proj);
      create_expected_depths(E, E.freeview, M, proj);
      render_image(E, E.freeview, M, proj, type, (V4u *)rgba_out, depth_out);
      E.freeviewValid = true;
      return DSR_OK;
    }
    default: return fail(DSR_E_ARG, "unsupported image type");
  }
}
int orc_get_image_dev(dsr_engine *h, int type, const float pose_m[16], const float intrinsics[4], void *rgba_out,
                      void *depth_out) {
  return orc_get_image(h, type, pose_m, intrinsics, (uint8_t *)rgba_out, (float *)depth_out);
}

/* DepthProvider::DepthFromDisparityMap (src/DynSLAM/DepthProvider.h:94-137), restated. */
int orc_depth_from_disparity(const float *disparity, int16_t *depth_mm_out, int n, float baseline_m, float focal_px,
                             float scale, float min_depth_m, float max_depth_m) {
  if (!disparity || !depth_mm_out || n <= 0) return fail(DSR_E_ARG, "bad disparity arguments");
  const float kMetersToMillimeters = 1000.0f;
  int32_t min_depth_mm = static_cast<int32_t>(min_depth_m * kMetersToMillimeters);
  int32_t max_depth_mm = static_cast<int32_t>(max_depth_m * kMetersToMillimeters);
  if (max_depth_mm >= 32767) return fail(DSR_E_ARG, "maximum depth does not fit an int16 millimetre map");
  for (int i = 0; i < n; i++) {
    float disp = disparity[i];
    /* static_cast<int32_t>(float) is undefined out of range in C++; the adopted definition is
     * the saturating one (f2i), which yields the same final value: both extremes are clipped to 0 */
    int32_t depth_mm = f2i(kMetersToMillimeters * scale * ((baseline_m * focal_px) / disp));
    if (std::abs(disp) < 1e-5) depth_mm = 0;
    if (depth_mm > max_depth_mm || depth_mm < min_depth_mm) depth_mm = 0;
    depth_mm_out[i] = static_cast<int16_t>(depth_mm);
  }
  return DSR_OK;
}
int orc_depth_from_disparity_dev(int, void *, const void *d, void *o, int n, float b, float f, float s, float mn, float mx) {
  return orc_depth_from_disparity((const float *)d, (int16_t *)o, n, b, f, s, mn, mx);
}

/* PrecomputedDepthProvider::ReadPrecomputed (PrecomputedDepthProvider.cpp:22-75), restated with iostreams:
 * the OpenCV FileStorage XML dump of a CV_16SC1 matrix (node "depth-frame": rows, cols, dt = "s", data) */
static int read_depth_xml_impl(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return fail(DSR_E_ARG, "bad arguments");
  std::ifstream in(path, std::ios::binary);
  if (!in) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  std::stringstream ss; ss << in.rdbuf();
  const std::string doc = ss.str();
  auto element = [](const std::string &d, const std::string &tag, std::string &body) -> bool {
    size_t p0 = 0;
    while ((p0 = d.find("<" + tag, p0)) != std::string::npos) {
      const char c = d[p0 + tag.size() + 1];
      if (c == '>' || isspace((unsigned char)c)) break;
      p0++;
    }
    if (p0 == std::string::npos) return false;
    const size_t gt = d.find('>', p0), close = d.find("</" + tag + ">", p0);
    if (gt == std::string::npos || close == std::string::npos || close < gt) return false;
    body = d.substr(gt + 1, close - gt - 1);
    return true;
  };
  std::string node, t;
  if (!element(doc, "depth-frame", node)) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  int rows = 0, cols = 0;
  if (!element(node, "rows", t)) return fail(DSR_E_IO, "no rows");
  rows = atoi(t.c_str());  /* no exceptions across the C ABI: garbage reads as 0 -> "empty matrix" below */
  if (!element(node, "cols", t)) return fail(DSR_E_IO, "no cols");
  cols = atoi(t.c_str());
  if (!element(node, "dt", t)) return fail(DSR_E_IO, "no dt");
  std::string dt; for (char c : t) if (!isspace((unsigned char)c)) dt += c;
  if (dt != "s") return fail(DSR_E_IO, "Precomputed depth map had the wrong format."); /* out.type() != CV_16SC1 */
  if ((long long)rows * cols > (1ll << 28) || rows > (1 << 20) || cols > (1 << 20)) return fail(DSR_E_IO, "depth-frame: implausible rows x cols");
  *width = cols; *height = rows;
  if (rows <= 0 || cols <= 0) return fail(DSR_E_IO, "Could not read precomputed depth map");
  if (!depth_mm_out || (long long)rows * cols > capacity) return fail(DSR_E_ARG, "depth map larger than the buffer");
  if (!element(node, "data", t)) return fail(DSR_E_IO, "no data");
  std::istringstream data(t);
  long long i = 0, n = (long long)rows * cols;
  long v;
  while (i < n && (data >> v)) depth_mm_out[i++] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));  /* cv::saturate_cast<short> */
  if (i != n) return fail(DSR_E_IO, "depth-frame <data> holds fewer values than rows x cols");
  return DSR_OK;
}
/* single-channel PFM: "Pf", width height, scale (< 0: little endian), raster bottom row first; returned
 * top row first (what ReadFilePFM of the reference's pfmLib submodule — absent — hands to OpenCV) */
static int read_pfm_impl(const char *path, float *out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return fail(DSR_E_ARG, "bad arguments");
  std::ifstream in(path, std::ios::binary);
  if (!in) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  std::string magic; int w = 0, h = 0; double scale = 0;
  in >> magic >> w >> h >> scale;
  if (!in || magic != "Pf") return fail(DSR_E_IO, "not a single-channel PFM file");
  in.get();
  if ((long long)w * h > (1ll << 28) || w > (1 << 20) || h > (1 << 20)) return fail(DSR_E_IO, "PFM: implausible width x height");
  *width = w; *height = h;
  if (w <= 0 || h <= 0) return fail(DSR_E_IO, "Could not read precomputed depth map");
  if (!out || (long long)w * h > capacity) return fail(DSR_E_ARG, "PFM image larger than the buffer");
  std::vector<unsigned char> row((size_t)w * 4);
  for (int r = 0; r < h; r++) {
    in.read(reinterpret_cast<char *>(row.data()), (std::streamsize)row.size());
    if (!in) return fail(DSR_E_IO, "PFM raster shorter than width x height");
    for (int c = 0; c < w; c++) {
      const unsigned char *b = &row[(size_t)c * 4];
      uint32_t bits = scale < 0 ? ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24))
                                : ((uint32_t)b[3] | ((uint32_t)b[2] << 8) | ((uint32_t)b[1] << 16) | ((uint32_t)b[0] << 24));
      float f; memcpy(&f, &bits, 4);
      out[(size_t)(h - 1 - r) * w + c] = f;
    }
  }
  return DSR_OK;
}
/* size reported with DSR_OK and DSR_E_ARG (buffer too small: the size query), 0 x 0 after DSR_E_IO */
int orc_read_depth_xml(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height) {
  const int st = read_depth_xml_impl(path, depth_mm_out, capacity, width, height);
  if (st == DSR_E_IO && width && height) *width = *height = 0;
  return st;
}
int orc_read_pfm(const char *path, float *out, int capacity, int *width, int *height) {
  const int st = read_pfm_impl(path, out, capacity, width, height);
  if (st == DSR_E_IO && width && height) *width = *height = 0;
  return st;
}
/* the input_is_depth_ clamp, int16 branch (PrecomputedDepthProvider.cpp:55-74) */
int orc_clip_depth_mm(int16_t *depth_mm, int n, float max_depth_m) {
  if (!depth_mm || n <= 0) return fail(DSR_E_ARG, "bad clip arguments");
  const float kMetersToMillimeters = 1000.0f;
  float max_depth_mm_f = max_depth_m * kMetersToMillimeters;
  float r = roundf(max_depth_mm_f);
  int16_t max_depth_mm_s = (int16_t)(r >= 32767.0f ? 32767 : (r <= -32768.0f ? -32768 : (int)r)); /* saturating, like f2i */
  for (int i = 0; i < n; i++) {
    int16_t depth = depth_mm[i];
    if (depth > max_depth_mm_s) depth_mm[i] = 0;
  }
  return DSR_OK;
}
int orc_clip_depth_mm_dev(int, void *, void *d, int n, float m) { return orc_clip_depth_mm((int16_t *)d, n, m); }

/* InfiniTamDriver.cpp:81-100 CvToItm(cv::Mat3b) */
int orc_bgr_to_rgba(const uint8_t *bgr, uint8_t *rgba_out, int n) {
  if (!bgr || !rgba_out || n <= 0) return fail(DSR_E_ARG, "bad conversion arguments");
  for (int idx = 0; idx < n; ++idx) {
    const uint8_t *col = bgr + 3 * (size_t)idx;
    uint8_t *d = rgba_out + 4 * (size_t)idx; /* Vector4u {r, g, b, a} */
    d[2] = col[0]; /* .b */
    d[1] = col[1]; /* .g */
    d[0] = col[2]; /* .r */
    d[3] = 255u;   /* .a */
  }
  return DSR_OK;
}
int orc_bgr_to_rgba_dev(int, void *, const void *i, void *o, int n) { return orc_bgr_to_rgba((const uint8_t *)i, (uint8_t *)o, n); }

/* InfiniTamDriver.cpp:108-120 ItmToCv(ITMUChar4Image) */
int orc_rgba_to_bgr(const uint8_t *rgba, uint8_t *bgr_out, int n) {
  if (!rgba || !bgr_out || n <= 0) return fail(DSR_E_ARG, "bad conversion arguments");
  for (int idx = 0; idx < n; ++idx) {
    const uint8_t *s_ = rgba + 4 * (size_t)idx;
    uint8_t *d = bgr_out + 3 * (size_t)idx;
    d[0] = s_[2]; d[1] = s_[1]; d[2] = s_[0]; /* cv::Vec3b(.b, .g, .r) */
  }
  return DSR_OK;
}
int orc_rgba_to_bgr_dev(int, void *, const void *i, void *o, int n) { return orc_rgba_to_bgr((const uint8_t *)i, (uint8_t *)o, n); }

/* InfiniTamDriver.cpp:128-139 FloatDepthmapToShort; out-of-range conversions as f2i (saturating) */
int orc_depth_m_to_mm(const float *depth_m, int16_t *depth_mm_out, int n) {
  if (!depth_m || !depth_mm_out || n <= 0) return fail(DSR_E_ARG, "bad conversion arguments");
  const int kMetersToMillimeters = 1000;
  for (int i = 0; i < n; ++i) depth_mm_out[i] = (int16_t)f2i(depth_m[i] * kMetersToMillimeters);
  return DSR_OK;
}
int orc_depth_m_to_mm_dev(int, void *, const void *i, void *o, int n) { return orc_depth_m_to_mm((const float *)i, (int16_t *)o, n); }

/* InfiniTamDriver.h:154-156: ItmToCv(*view->rgb) + ItmDepthToCv(*view->depth) of the current view */
int orc_get_view_previews(dsr_engine *h, uint8_t *bgr_out, int16_t *depth_mm_out) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (E.W != E.Wr || E.H != E.Hr) return fail(DSR_E_ARG, "rgb and depth sizes differ");
  int st = DSR_OK;
  if (bgr_out) st = orc_rgba_to_bgr(reinterpret_cast<const uint8_t *>(E.rgb.data()), bgr_out, E.W * E.H);
  if (st == DSR_OK && depth_mm_out) st = orc_depth_m_to_mm(E.depth.data(), depth_mm_out, E.W * E.H);
  return st;
}
int orc_get_no_visible_blocks(dsr_engine *h, int32_t *out) {
  if (!h || !out) return fail(DSR_E_ARG, "null");
  *out = E.live.noVisibleBlocks;
  return DSR_OK;
}

/* ProcessSilhouette_CPU (InstanceReconstructor.cpp:59-133), restated on the engines' views. */
int orc_view_extract_silhouette(dsr_engine *m, dsr_engine *inst, const uint8_t *mask, int x0, int y0, int box_w, int box_h) {
  if (!m || !inst || !mask || box_w <= 0 || box_h <= 0) return fail(DSR_E_ARG, "bad silhouette arguments");
  Engine &src = m->e, &dst = inst->e;
  if (!src.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (src.W != dst.W || src.H != dst.H || src.Wr != dst.Wr || src.Hr != dst.Hr || src.W != src.Wr) return fail(DSR_E_ARG, "size mismatch");
  const int frame_width = src.W, frame_height = src.H;
  memset(dst.rgb.data(), 255, (size_t)frame_width * frame_height * sizeof(V4u));
  memset(dst.depth.data(), 0, (size_t)frame_width * frame_height * sizeof(float));
  for (int row = 0; row < box_h; ++row)
    for (int col = 0; col < box_w; ++col) {
      int copy_row = row + y0, copy_col = col + x0;
      if (copy_row < 0 || copy_row >= frame_height || copy_col < 0 || copy_col >= frame_width) continue;
      int copy_idx = copy_row * frame_width + copy_col;
      if (mask[row * box_w + col] == 1) {
        dst.rgb[copy_idx] = src.rgb[copy_idx];
        dst.depth[copy_idx] = src.depth[copy_idx];
      } else {
        dst.rgb[copy_idx].x = 255; dst.rgb[copy_idx].y = 255; dst.rgb[copy_idx].z = 255;
      }
    }
  dst.hasView = true;
  return DSR_OK;
}
/* RemoveSilhouette_CPU (InstanceReconstructor.cpp:135-170), restated. */
int orc_view_remove_silhouette(dsr_engine *h, const uint8_t *mask, int x0, int y0, int box_w, int box_h) {
  if (!h || !mask || box_w <= 0 || box_h <= 0) return fail(DSR_E_ARG, "bad silhouette arguments");
  if (!E.hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  for (int row = 0; row < box_h; ++row)
    for (int col = 0; col < box_w; ++col) {
      int frame_row = row + y0, frame_col = col + x0;
      if (frame_row < 0 || frame_row >= E.H || frame_col < 0 || frame_col >= E.W) continue;
      int frame_idx = frame_row * E.W + frame_col;
      if (mask[row * box_w + col] == 1) {
        E.rgb[frame_idx] = {0, 0, 0, 0};
        E.depth[frame_idx] = 0.0f;
      }
    }
  return DSR_OK;
}

/* InstanceReconstructor::CompositeInstances / CompositeColor / CompositeDepth
 * (InstanceReconstructor.cpp:851-990), restated on raw buffers. */
int orc_composite_instances(uint8_t *target_rgba, float *target_depth, const uint8_t *layers_rgba,
                            const float *layers_depth, const int32_t *track_ids, int n_layers, int n_pixels,
                            float tint_strength, int dim_background) {
  static const int pal[10][3] = {{0x1f, 0x77, 0xb4}, {0xff, 0x7f, 0x0e}, {0x2c, 0xa0, 0x2c}, {0xd6, 0x27, 0x28},
                                 {0x94, 0x67, 0xbd}, {0x8c, 0x56, 0x4b}, {0xe3, 0x77, 0xc2}, {0x71, 0x71, 0x71},
                                 {0xbc, 0xbd, 0x22}, {0x17, 0xbe, 0xcf}};
  if (!target_depth || n_layers < 0 || n_pixels <= 0) return fail(DSR_E_ARG, "bad composite arguments");
  if (target_rgba && dim_background) {
    float dim_factor = 0.10f;
    for (int i = 0; i < n_pixels; i++)
      for (int ch = 0; ch < 3; ch++) target_rgba[4 * i + ch] = static_cast<uint8_t>(target_rgba[4 * i + ch] * (1.0 - dim_factor));
  }
  const float kColorBoost = 0.50f;
  for (int l = 0; l < n_layers; l++) {
    const float *s_depth = layers_depth + (size_t)l * n_pixels;
    const uint8_t *s_col = layers_rgba ? layers_rgba + (size_t)l * n_pixels * 4 : nullptr;
    const int *tint = pal[((track_ids[l] % 10) + 10) % 10];
    for (int idx = 0; idx < n_pixels; idx++) {
      bool instance_on_top = (s_depth[idx] != 0 && (target_depth[idx] == 0 || target_depth[idx] > s_depth[idx]));
      if (!instance_on_top) continue;
      target_depth[idx] = s_depth[idx];
      if (target_rgba) {
        double col_strength = 1.0 + kColorBoost - tint_strength;
        for (int ch = 0; ch < 3; ch++)
          target_rgba[4 * idx + ch] = static_cast<uint8_t>(std::min(255.0, s_col[4 * idx + ch] * col_strength + tint[ch] * tint_strength));
      }
    }
  }
  return DSR_OK;
}
/* test infrastructure: "device" pointers are host pointers here */
int orc_view_extract_silhouette_dev(dsr_engine *m, dsr_engine *inst, const void *mask, int x0, int y0, int box_w, int box_h) {
  return orc_view_extract_silhouette(m, inst, (const uint8_t *)mask, x0, y0, box_w, box_h);
}
int orc_view_remove_silhouette_dev(dsr_engine *h, const void *mask, int x0, int y0, int box_w, int box_h) {
  return orc_view_remove_silhouette(h, (const uint8_t *)mask, x0, y0, box_w, box_h);
}
// dsr_view_split_silhouette: the two host loops in the reference's order (InstanceReconstructor.cpp:238-263)
int orc_view_split_silhouette(dsr_engine *m, dsr_engine *inst, const uint8_t *copy_mask, int x0, int y0, int box_w, int box_h,
                              const uint8_t *delete_mask, int dx0, int dy0, int dbox_w, int dbox_h) {
  int st = orc_view_extract_silhouette(m, inst, copy_mask, x0, y0, box_w, box_h);
  if (st) return st;
  return orc_view_remove_silhouette(m, delete_mask, dx0, dy0, dbox_w, dbox_h);
}
int orc_view_split_silhouette_dev(dsr_engine *m, dsr_engine *inst, const void *copy_mask, int x0, int y0, int box_w, int box_h,
                                  const void *delete_mask, int dx0, int dy0, int dbox_w, int dbox_h) {
  return orc_view_split_silhouette(m, inst, (const uint8_t *)copy_mask, x0, y0, box_w, box_h, (const uint8_t *)delete_mask, dx0, dy0, dbox_w, dbox_h);
}
int orc_composite_layer_ptrs_dev(int, void *, void *target_rgba, void *target_depth, const void *const *layer_rgba_ptrs,
                                 const void *const *layer_depth_ptrs, const int32_t *track_ids, int n_layers, int n_pixels,
                                 float tint_strength, int dim_background) {
  if (n_layers < 0 || n_pixels <= 0 || (n_layers > 0 && (!layer_depth_ptrs || (target_rgba && !layer_rgba_ptrs)))) return fail(DSR_E_ARG, "bad composite arguments");
  std::vector<uint8_t> lr(target_rgba ? (size_t)n_layers * n_pixels * 4 : 0);
  std::vector<float> ld((size_t)n_layers * n_pixels);
  for (int l = 0; l < n_layers; ++l) {  /* gather the layers, then the restated loops below */
    memcpy(ld.data() + (size_t)l * n_pixels, layer_depth_ptrs[l], (size_t)n_pixels * 4);
    if (target_rgba) memcpy(lr.data() + (size_t)l * n_pixels * 4, layer_rgba_ptrs[l], (size_t)n_pixels * 4);
  }
  return orc_composite_instances((uint8_t *)target_rgba, (float *)target_depth, target_rgba ? lr.data() : nullptr, ld.data(), track_ids,
                                 n_layers, n_pixels, tint_strength, dim_background);
}
int orc_composite_instances_dev(int, void *, void *target_rgba, void *target_depth, const void *layers_rgba,
                                const void *layers_depth, const int32_t *track_ids, int n_layers, int n_pixels,
                                float tint_strength, int dim_background) {
  return orc_composite_instances((uint8_t *)target_rgba, (float *)target_depth, (const uint8_t *)layers_rgba,
                                 (const float *)layers_depth, track_ids, n_layers, n_pixels, tint_strength, dim_background);
}

/* ---- the multi-GPU exchange (include/dsr.h) as the CPU checker sees it: "devices" are host memory, every rank lives in this
 * process, the layers of all ranks lie in one buffer — the all-gather has nothing to move.  Rank-per-process mode exists only
 * for a world of one (the checker has no transport; the multi-process CPU tests exchange the layers over gloo). */
struct dsr_exchange {
  int nRanks = 0, slots = 0, P = 0;
  std::vector<uint8_t> all;                  /* nRanks x slots layers: float depth plane, then RGBA plane */
  std::vector<uint8_t> targetRgba;
  std::vector<float> targetDepth;
};
static int orc_exchange_make(int n_ranks, int slots, int P, dsr_exchange **out) {
  if (n_ranks <= 0 || slots <= 0 || P <= 0 || !out) return fail(DSR_E_ARG, "bad exchange arguments");
  dsr_exchange *x = new dsr_exchange();
  x->nRanks = n_ranks; x->slots = slots; x->P = P;
  x->all.assign((size_t)n_ranks * slots * P * 8, 0);
  x->targetRgba.assign((size_t)P * 4, 0); x->targetDepth.assign((size_t)P, 0.0f);
  *out = x;
  return DSR_OK;
}
int orc_exchange_create(const int32_t *devices, int n_ranks, int slots_per_rank, int n_pixels, dsr_exchange **out) {
  if (!devices) return fail(DSR_E_ARG, "bad exchange arguments");
  return orc_exchange_make(n_ranks, slots_per_rank, n_pixels, out);
}
int orc_exchange_unique_id(uint8_t id_out[128]) { if (!id_out) return fail(DSR_E_ARG, "null"); memset(id_out, 0, 128); return DSR_OK; }
int orc_exchange_create_rank(const uint8_t unique_id[128], int world_size, int rank, int, int slots_per_rank, int n_pixels, dsr_exchange **out) {
  if (!unique_id || world_size != 1 || rank != 0) return fail(DSR_E_ARG, "the CPU checker has no transport between processes");
  return orc_exchange_make(1, slots_per_rank, n_pixels, out);
}
void orc_exchange_destroy(dsr_exchange *x) { delete x; }
void *orc_exchange_stream(dsr_exchange *, int) { return nullptr; }
int orc_exchange_layer_ptrs(dsr_exchange *x, int on_rank, int rank, int slot, void **rgba, void **depth) {
  if (!x || on_rank < 0 || on_rank >= x->nRanks || rank < 0 || rank >= x->nRanks || slot < 0 || slot >= x->slots) return fail(DSR_E_ARG, "bad exchange layer");
  uint8_t *base = x->all.data() + ((size_t)rank * x->slots + slot) * x->P * 8;
  if (depth) *depth = base;
  if (rgba) *rgba = base + (size_t)x->P * 4;
  return DSR_OK;
}
int orc_exchange_slot_ptrs(dsr_exchange *x, int rank, int slot, void **rgba, void **depth) { return orc_exchange_layer_ptrs(x, rank, rank, slot, rgba, depth); }
int orc_exchange_render_slot(dsr_exchange *x, int rank, int slot, dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4]) {
  void *rgba = nullptr, *depth = nullptr;
  int st = orc_exchange_slot_ptrs(x, rank, slot, &rgba, &depth);
  if (st) return st;
  if (!e) { memset(depth, 0, (size_t)x->P * 4); return DSR_OK; }
  if (e->e.W * e->e.H != x->P) return fail(DSR_E_ARG, "image size differs from the exchange's");
  return orc_get_image_dev(e, type, pose_m, intrinsics, rgba, depth);
}
int orc_exchange_gather(dsr_exchange *x) { return x ? DSR_OK : fail(DSR_E_ARG, "null exchange"); }
int orc_exchange_set_collective(dsr_exchange *x, int, int) { return x ? DSR_OK : fail(DSR_E_ARG, "null exchange"); }  // one address space: nothing to move
int orc_exchange_timing(dsr_exchange *x, int, double *gather_ms, double *composite_ms, int32_t *n_gathers, int32_t *n_composites) {
  if (!x) return fail(DSR_E_ARG, "null exchange");
  if (gather_ms) *gather_ms = 0.0;
  if (composite_ms) *composite_ms = 0.0;
  if (n_gathers) *n_gathers = 0;
  if (n_composites) *n_composites = 0;
  return DSR_OK;
}
int orc_exchange_target_ptrs(dsr_exchange *x, int rank, void **rgba, void **depth) {
  if (!x || rank < 0 || rank >= x->nRanks) return fail(DSR_E_ARG, "bad exchange rank");
  if (rgba) *rgba = x->targetRgba.data();
  if (depth) *depth = x->targetDepth.data();
  return DSR_OK;
}
int orc_exchange_clear_target(dsr_exchange *x, int rank) {
  if (!x || rank < 0 || rank >= x->nRanks) return fail(DSR_E_ARG, "bad exchange rank");
  std::fill(x->targetRgba.begin(), x->targetRgba.end(), 0); std::fill(x->targetDepth.begin(), x->targetDepth.end(), 0.0f);
  return DSR_OK;
}
int orc_composite_layer_ptrs_dev(int, void *, void *target_rgba, void *target_depth, const void *const *layer_rgba_ptrs,
                                 const void *const *layer_depth_ptrs, const int32_t *track_ids, int n_layers, int n_pixels,
                                 float tint_strength, int dim_background);
int orc_exchange_composite(dsr_exchange *x, int root_rank, dsr_engine *, void *target_rgba, void *target_depth, const int32_t *ranks,
                           const int32_t *slots, const int32_t *track_ids, int n_layers, float tint_strength, int dim_background) {
  if (!x || root_rank < 0 || root_rank >= x->nRanks || n_layers < 0 || (n_layers > 0 && (!ranks || !slots || !track_ids))) return fail(DSR_E_ARG, "bad composite arguments");
  if (!target_depth) { target_rgba = x->targetRgba.data(); target_depth = x->targetDepth.data(); }
  std::vector<const void *> rp((size_t)std::max(n_layers, 1)), dp((size_t)std::max(n_layers, 1));
  for (int l = 0; l < n_layers; ++l) {
    void *r = nullptr, *d = nullptr;
    int st = orc_exchange_layer_ptrs(x, root_rank, ranks[l], slots[l], &r, &d);
    if (st) return st;
    rp[l] = r; dp[l] = d;
  }
  if (n_layers == 0) return DSR_OK;
  return orc_composite_layer_ptrs_dev(-1, nullptr, target_rgba, target_depth, target_rgba ? rp.data() : nullptr, dp.data(), track_ids, n_layers,
                                      x->P, tint_strength, dim_background);
}
int orc_exchange_gather_and_composite(dsr_exchange *x, int root_rank, dsr_engine *te, void *target_rgba, void *target_depth, const int32_t *ranks,
                                      const int32_t *slots, const int32_t *track_ids, int n_layers, float tint_strength, int dim_background) {
  if (!x) return fail(DSR_E_ARG, "null exchange");
  if (root_rank < 0 || root_rank >= x->nRanks) return DSR_OK;
  return orc_exchange_composite(x, root_rank, te, target_rgba, target_depth, ranks, slots, track_ids, n_layers, tint_strength, dim_background);
}
int orc_exchange_read_target(dsr_exchange *x, int rank, uint8_t *rgba_out, float *depth_out) {
  if (!x || rank < 0 || rank >= x->nRanks) return fail(DSR_E_ARG, "bad exchange rank");
  if (rgba_out) memcpy(rgba_out, x->targetRgba.data(), (size_t)x->P * 4);
  if (depth_out) memcpy(depth_out, x->targetDepth.data(), (size_t)x->P * 4);
  return DSR_OK;
}
int orc_exchange_sync(dsr_exchange *x) { return x ? DSR_OK : fail(DSR_E_ARG, "null exchange"); }

int orc_get_stats(dsr_engine *h, dsr_stats *out) {
  if (!h || !out) return fail(DSR_E_ARG, "null");
  memset(out, 0, sizeof *out);
  out->num_allocated_voxel_blocks = E.noBlocks;
  out->last_free_block_id = E.lastFreeBlockId;
  out->last_free_excess_list_id = E.lastFreeExcessListId;
  out->no_visible_blocks = E.live.noVisibleBlocks;
  out->no_total_entries = E.noTotalEntries;
  out->voxel_bytes = (int)sizeof(dsr_voxel);
  out->block_voxels = DSR_BLOCK_SIZE3;
  out->sticky_status = E.stickyStatus;
  out->decayed_block_count = E.decayedBlockCount;
  out->frames_processed = E.framesProcessed;
  out->no_visible_blocks_freeview = E.freeview.noVisibleBlocks;
  out->host_store_slots = E.hostStoreSlots;
  out->host_store_capacity_slots = E.hostStoreSlots;
  return DSR_OK;
}

int orc_dump_hash_table(dsr_engine *h, dsr_hash_entry *out) {
  if (!h || !out) return fail(DSR_E_ARG, "null");
  memcpy(out, E.hashTable.data(), (size_t)E.noTotalEntries * sizeof(dsr_hash_entry));
  return DSR_OK;
}
int orc_dump_visible_list(dsr_engine *h, int freeview, int32_t *ids_out, int32_t *n) {
  if (!h || !n) return fail(DSR_E_ARG, "null");
  RenderState &rs = freeview ? E.freeview : E.live;
  *n = rs.noVisibleBlocks;
  if (ids_out) memcpy(ids_out, rs.visibleEntryIDs.data(), (size_t)rs.noVisibleBlocks * sizeof(int32_t));
  return DSR_OK;
}
int orc_dump_visible_types(dsr_engine *h, uint8_t *out) {
  if (!h || !out) return fail(DSR_E_ARG, "null");
  memcpy(out, E.live.entriesVisibleType.data(), (size_t)E.noTotalEntries);
  return DSR_OK;
}
int orc_dump_voxel_blocks(dsr_engine *h, int first_block, int n_blocks, dsr_voxel *out) {
  if (!h || !out || first_block < 0 || n_blocks < 0 || first_block + n_blocks > E.noBlocks) return fail(DSR_E_ARG, "bad range");
  memcpy(out, &E.voxels[(size_t)first_block * DSR_BLOCK_SIZE3], (size_t)n_blocks * DSR_BLOCK_SIZE3 * sizeof(dsr_voxel));
  return DSR_OK;
}
int orc_dump_allocation_lists(dsr_engine *h, int32_t *voxel_alloc_list, int32_t *excess_alloc_list) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (voxel_alloc_list) memcpy(voxel_alloc_list, E.voxelAllocationList.data(), (size_t)E.noBlocks * 4);
  if (excess_alloc_list) memcpy(excess_alloc_list, E.excessAllocationList.data(), (size_t)E.noExcess * 4);
  return DSR_OK;
}
int orc_dump_render_state(dsr_engine *h, int which, float *minmax, float *raycast_result, float *points, float *normals,
                          uint8_t *raycast_image) {
  if (!h) return fail(DSR_E_ARG, "null");
  RenderState &rs = which ? E.freeview : E.live;
  const size_t P = (size_t)E.W * E.H;
  if (minmax) memcpy(minmax, rs.minmax.data(), rs.minmax.size() * sizeof(V2f));
  if (raycast_result) memcpy(raycast_result, rs.raycastResult.data(), P * sizeof(V4f));
  if (points) memcpy(points, E.pointsMap.data(), P * sizeof(V4f));
  if (normals) memcpy(normals, E.normalsMap.data(), P * sizeof(V4f));
  if (raycast_image) memcpy(raycast_image, rs.raycastImage.data(), P * 4);
  return DSR_OK;
}

int orc_dump_swap_state(dsr_engine *h, uint8_t *states, uint8_t *has_stored) {
  if (!h) return fail(DSR_E_ARG, "null");
  if (E.swapStates.empty()) return fail(DSR_E_ARG, "swapping is not enabled");
  if (states) memcpy(states, E.swapStates.data(), (size_t)E.noTotalEntries);
  if (has_stored) memcpy(has_stored, E.hasStored.data(), (size_t)E.noTotalEntries);
  return DSR_OK;
}
int orc_dump_stored_block(dsr_engine *h, int entry, dsr_voxel *out, int *present) {
  if (!h || !present || entry < 0 || entry >= E.noTotalEntries) return fail(DSR_E_ARG, "bad entry");
  *present = 0;
  if (E.swapStates.empty() || !E.hasStored[entry]) return DSR_OK;
  *present = 1;
  if (out) memcpy(out, E.storedBlocks[entry].data(), DSR_BLOCK_SIZE3 * sizeof(dsr_voxel));
  return DSR_OK;
}

/* the oracle divides with the C `/` operator everywhere: nothing to self-test */
int orc_selftest_division(int, uint64_t, uint64_t, uint64_t *mismatches) { if (mismatches) *mismatches = 0; return DSR_OK; }

/* stream ordering / bandwidth probe: nothing to order or measure on the CPU */
int orc_wait_for_stream(dsr_engine *h, void *) { return h ? DSR_OK : DSR_E_ARG; }
// dsr_batch_*: the checker's form is the reference's own loop — one instance after the other (InstanceReconstructor.cpp:315-361)
struct dsr_batch { dsr_engine *source; std::vector<dsr_engine *> vols; };
int orc_batch_create(dsr_engine *source, dsr_engine *const *volumes, int n_volumes, dsr_batch **out) {
  if (!source || !volumes || n_volumes <= 0 || n_volumes > 8 || !out) return fail(DSR_E_ARG, "a batch holds 1..8 volumes");
  dsr_batch *b = new dsr_batch();
  b->source = source;
  for (int k = 0; k < n_volumes; ++k) { if (!volumes[k] || volumes[k] == source) { delete b; return fail(DSR_E_ARG, "bad volume"); } b->vols.push_back(volumes[k]); }
  *out = b;
  return DSR_OK;
}
void orc_batch_destroy(dsr_batch *b) { delete b; }
int orc_batch_fuse(dsr_batch *b, const dsr_batch_item *items, int n_items, int32_t *status_out) {
  if (!b || !items || n_items <= 0) return fail(DSR_E_ARG, "bad batch arguments");
  for (int i = 0; i < n_items; ++i) {
    const dsr_batch_item &it = items[i];
    if (status_out) status_out[i] = DSR_OK;
    if (it.volume >= (int)b->vols.size() || it.volume < -1) return fail(DSR_E_ARG, "bad batch volume index");
    dsr_engine *e = it.volume >= 0 ? b->vols[it.volume] : nullptr;
    int st = DSR_OK;
    if (e && (st = orc_view_extract_silhouette(b->source, e, (const uint8_t *)it.copy_mask_dev, it.x0, it.y0, it.box_w, it.box_h))) return st;
    if (it.delete_mask_dev && (st = orc_view_remove_silhouette(b->source, (const uint8_t *)it.delete_mask_dev, it.dx0, it.dy0, it.dbox_w, it.dbox_h))) return st;
    if (!e) continue;
    if ((st = orc_set_pose_inv_m(e, it.inv_m))) return st;
    st = orc_process_frame(e);
    if (st == DSR_E_OUT_OF_BLOCKS) { if (status_out) status_out[i] = st; }  // the fork's exception, caught by the host (:662-671)
    else if (st) return st;
    if ((st = orc_prepare(e))) return st;
  }
  return DSR_OK;
}
int orc_batch_render(dsr_batch *b, int type, const dsr_batch_render_item *items, int n_items) {
  if (!b || !items || n_items <= 0) return fail(DSR_E_ARG, "bad batch arguments");
  for (int i = 0; i < n_items; ++i) {
    const dsr_batch_render_item &it = items[i];
    if (it.volume < 0 || it.volume >= (int)b->vols.size()) return fail(DSR_E_ARG, "bad batch volume index");
    int st = orc_get_image_dev(b->vols[it.volume], type, it.pose_m, nullptr, it.rgba_out_dev, it.depth_out_dev);
    if (st) return st;
  }
  return DSR_OK;
}
int orc_pin_host_thread(int) { return DSR_OK; }  // no GPU to be near to
int orc_engine_share_stream(dsr_engine *h, dsr_engine *owner) { return (h && owner && h != owner) ? DSR_OK : DSR_E_ARG; }  // no streams here
int orc_stream_wait_for_engine(dsr_engine *h, void *) { return h ? DSR_OK : DSR_E_ARG; }
int orc_measure_copy_bandwidth(int, uint64_t, int, double *gbps_out) { if (gbps_out) *gbps_out = 0.0; return DSR_OK; }
int orc_measure_copy_bandwidth_spread(int, uint64_t, int, double out[3]) { if (out) out[0] = out[1] = out[2] = 0.0; return DSR_OK; }

int orc_profile_enable(dsr_engine *h, int) { return h ? DSR_OK : DSR_E_ARG; }
int orc_profile_reset(dsr_engine *h) { return h ? DSR_OK : DSR_E_ARG; }
int orc_profile_get(dsr_engine *, dsr_kernel_time *, int) { return 0; }

/* ------------------------------------------------------------- meshing (8f row 4) */

#include "mc_tables.h"

/* ITMMeshingEngine.h findPointNeighbors: the 8 corners of the cell at blockLocation, in the cube
 * numbering of the tables; false when a corner is missing or still at the initial sdf (1.0). */
static inline bool findPointNeighbors(const Engine &e, V3f *p, float *sdf, const V3i &blockLocation) {
  static const int off[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
  for (int k = 0; k < 8; ++k) {
    V3i q = {blockLocation.x + off[k][0], blockLocation.y + off[k][1], blockLocation.z + off[k][2]};
    bool found;
    dsr_voxel v = readVoxel(e, q, found, nullptr);
    sdf[k] = sdf_to_float((float)v.sdf);
    if (!found || sdf[k] == 1.0f) return false;
    p[k] = {(float)q.x, (float)q.y, (float)q.z};
  }
  return true;
}

/* ITMMeshingEngine.h sdfInterp */
static inline V3f sdfInterp(const V3f &p1, const V3f &p2, float valp1, float valp2) {
  if (fabsf(0.0f - valp1) < 0.00001f) return p1;
  if (fabsf(0.0f - valp2) < 0.00001f) return p2;
  if (fabsf(valp1 - valp2) < 0.00001f) return p1;
  const float t = (0.0f - valp1) / (valp2 - valp1);
  return {p1.x + t * (p2.x - p1.x), p1.y + t * (p2.y - p1.y), p1.z + t * (p2.z - p1.z)};
}

/* ITMMeshingEngine.h buildVertList */
static inline int buildVertList(const Engine &e, V3f *vertList, const V3i &globalPos, const V3i &localPos) {
  V3f points[8];
  float sdfVals[8];
  V3i loc = {globalPos.x + localPos.x, globalPos.y + localPos.y, globalPos.z + localPos.z};
  if (!findPointNeighbors(e, points, sdfVals, loc)) return -1;
  int cubeIndex = 0;
  for (int k = 0; k < 8; ++k) if (sdfVals[k] < 0) cubeIndex |= 1 << k;
  if (kMcEdgeTable[cubeIndex] == 0) return -1;
  static const int ends[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  for (int k = 0; k < 12; ++k)
    if (kMcEdgeTable[cubeIndex] & (1 << k))
      vertList[k] = sdfInterp(points[ends[k][0]], points[ends[k][1]], sdfVals[ends[k][0]], sdfVals[ends[k][1]]);
  return cubeIndex;
}

/* ITMMeshingEngine_CPU<TVoxel,ITMVoxelBlockHash>::MeshScene */
int orc_mesh_scene(dsr_engine *h, uint64_t *n_triangles) {
  if (!h) return fail(DSR_E_ARG, "null engine");
  Engine &e = E;
  e.mesh.clear();
  const uint64_t noMaxTriangles = (uint64_t)e.s.sdf_local_block_num * 32u;
  const float factor = e.s.voxel_size;
  uint64_t noTriangles = 0;
  for (int entryId = 0; entryId < e.noTotalEntries; entryId++) {
    const dsr_hash_entry &he = e.hashTable[entryId];
    if (he.ptr < 0) continue;
    V3i globalPos = {he.pos[0] * DSR_BLOCK_SIZE, he.pos[1] * DSR_BLOCK_SIZE, he.pos[2] * DSR_BLOCK_SIZE};
    for (int z = 0; z < DSR_BLOCK_SIZE; z++)
      for (int y = 0; y < DSR_BLOCK_SIZE; y++)
        for (int x = 0; x < DSR_BLOCK_SIZE; x++) {
          V3f vertList[12];
          int cubeIndex = buildVertList(e, vertList, globalPos, V3i{x, y, z});
          if (cubeIndex < 0) continue;
          for (int i = 0; kMcTriTable[cubeIndex][i] != -1; i += 3) {
            const V3f &a = vertList[kMcTriTable[cubeIndex][i]], &b = vertList[kMcTriTable[cubeIndex][i + 1]],
                      &c = vertList[kMcTriTable[cubeIndex][i + 2]];
            dsr_triangle t = {{a.x * factor, a.y * factor, a.z * factor},
                              {b.x * factor, b.y * factor, b.z * factor},
                              {c.x * factor, c.y * factor, c.z * factor}};
            /* triangles[noTriangles] = t; if (noTriangles < noMaxTriangles - 1) noTriangles++;
             * => the first noMaxTriangles - 1 triangles survive */
            if (noTriangles < noMaxTriangles - 1) { e.mesh.push_back(t); noTriangles++; }
          }
        }
  }
  if (n_triangles) *n_triangles = noTriangles;
  return DSR_OK;
}

int orc_mesh_get(dsr_engine *h, dsr_triangle *out, uint64_t first, uint64_t count) {
  if (!h || (!out && count)) return fail(DSR_E_ARG, "null");
  if (first + count > E.mesh.size()) return fail(DSR_E_ARG, "triangle range outside the mesh");
  if (count) memcpy(out, E.mesh.data() + first, (size_t)count * sizeof(dsr_triangle));
  return DSR_OK;
}

/* ITMMesh::WriteOBJ */
int orc_mesh_write_obj(dsr_engine *h, const char *path) {
  if (!h || !path) return fail(DSR_E_ARG, "null");
  FILE *f = fopen(path, "w+");
  if (!f) return fail(DSR_E_ARG, "cannot open the OBJ file for writing");
  for (const dsr_triangle &t : E.mesh) {
    fprintf(f, "v %f %f %f\n", t.p0[0], t.p0[1], t.p0[2]);
    fprintf(f, "v %f %f %f\n", t.p1[0], t.p1[1], t.p1[2]);
    fprintf(f, "v %f %f %f\n", t.p2[0], t.p2[1], t.p2[2]);
  }
  for (uint64_t i = 0; i < E.mesh.size(); i++)
    fprintf(f, "f %llu %llu %llu\n", (unsigned long long)(i * 3 + 2 + 1), (unsigned long long)(i * 3 + 1 + 1),
            (unsigned long long)(i * 3 + 0 + 1));
  fclose(f);
  return DSR_OK;
}

int orc_mesh_free(dsr_engine *h) {
  if (!h) return fail(DSR_E_ARG, "null engine");
  E.mesh.clear(); E.mesh.shrink_to_fit();
  return DSR_OK;
}

int orc_save_scene_to_mesh(dsr_engine *h, const char *path) {
  int st = orc_mesh_scene(h, nullptr);
  if (st == DSR_OK) st = orc_mesh_write_obj(h, path);
  if (h) orc_mesh_free(h);
  return st;
}

/* oracle-only: ray-march statistics since the last call with reset != 0:
 * rays, miss steps, found steps, trilinear samples, max steps of a ray, hits, rays > 200 steps, rays > 50 steps */
int orc_debug_raycast_stats(int enable, int reset, long long out[8]) {
  g_rc_stats_on = enable != 0;
  if (out) memcpy(out, g_rc_stats, sizeof g_rc_stats);
  if (reset) memset(g_rc_stats, 0, sizeof g_rc_stats);
  return DSR_OK;
}

/* oracle-only: record the number of march steps of every ray of the following raycasts into
 * out[W*H] (NULL stops recording) */
int orc_debug_raycast_stats2(void *engine, int enable, long long out[8]) {
  g_rc2_on = enable != 0;
  if (out) memcpy(out, g_rc2, sizeof g_rc2);
  memset(g_rc2, 0, sizeof g_rc2);
  if (g_rc2_on && engine) { /* snapshot of the scene the NEXT raycast will march through */
    Engine &e = reinterpret_cast<dsr_engine *>(engine)->e;
    g_rc2_blockSat.assign((size_t)e.noBlocks, 0);
    for (int k = 0; k < 3; k++) g_rc2_super[k].clear();
    for (int t = 0; t < e.noTotalEntries; t++) {
      const dsr_hash_entry &he = e.hashTable[t];
      if (he.ptr < 0) continue;
      const dsr_voxel *blk = &e.voxels[(size_t)he.ptr * DSR_BLOCK_SIZE3];
      bool sat = true;
      for (int v = 0; v < DSR_BLOCK_SIZE3 && sat; v++) sat = blk[v].sdf == 32767;
      g_rc2_blockSat[(size_t)he.ptr] = sat;
      for (int k = 0; k < 3; k++) g_rc2_super[k][rc2_key(he.pos[0], he.pos[1], he.pos[2], k + 1)] = 1;
    }
  }
  return 0;
}

int orc_debug_raycast_steps(int *out, int width) {
  g_rc_steps = out; g_rc_steps_w = width;
  return DSR_OK;
}

/* oracle-only: integration statistics since the last reset: blocks, depth-updated voxels,
 * colour-updated voxels, half blocks / z-slices / blocks with at least one depth update */
int orc_debug_integrate_stats(int enable, int reset, long long out[8]) {
  g_int_stats_on = enable != 0;
  if (out) memcpy(out, g_int_stats, sizeof g_int_stats);
  if (reset) memset(g_int_stats, 0, sizeof g_int_stats);
  return DSR_OK;
}

/* oracle-only: number of OpenMP threads for the data-parallel loops (integrate,
 * raycast, shading).  Results do not depend on it. */
int orc_set_threads(dsr_engine *h, int n) {
  if (!h || n < 1) return fail(DSR_E_ARG, "bad thread count");
  E.threads = n;
  return DSR_OK;
}

} /* extern "C" */
