// k_batch.h — the instance volumes of ONE GPU as a batch: every kernel of an instance frame launched once for all of them.
//
// An instance volume's frame is eight launches (k_small.h), most of them far too small for the chip: two run as ONE workgroup,
// the raycasts are a few hundred live rays in a frame of 465 k pixels.  N volumes driven one after the other — the reference's
// loop (InstanceReconstructor.cpp:315-361), and round 4's eight streams — pay N chains of dependent launches; kernels of
// different streams overlap little (round 4: 8 x 21 launches in 1.27 ms, ~7 us per launch whichever queue it sits in).  Here the
// volume is a grid dimension: blockIdx.z (or .x for the one-workgroup kernels) selects the volume, its parameters come from two
// small tables (BatchVolP: pointers, constant, in HBM; BatchFrameP: this call's cameras — since round 6 the whole table travels
// as a KERNEL ARGUMENT of every launch, 2.8 KB, instead of through a device copy that two k_batch_set launches per call had to
// write first), and the kernel BODIES are the per-volume functions of the other headers — alloc_mark_pixel,
// small_alloc_visible_body, integrate_body, cast_ray, icp_pixel, small_freeview_body, render_pixel — so every volume's
// arithmetic, and every digest, is what the per-volume launches give (tests/test_gpu_batch.py).
#pragma once
#include "k_integrate.h"
#include "k_small.h"

namespace dsr {

constexpr int kBatchMax = 8;       // volumes per launch (a larger scene runs several batches)

struct BatchVolP {  // one volume: what does not change from call to call
  SceneP s;
  float *depth;
  uchar4 *rgb;  // its view
  uint8_t *visType;
  int4 *workList;
  int32_t *visibleIDs;
  int4 *visBlocks;
  int2 *minmax;
  float4 *raycastResult;
  uchar4 *raycastImage;  // live render state
  float4 *pointsMap, *normalsMap;
  uint2 *integrateStats;
  int32_t *fvVisibleIDs;
  int4 *fvVisBlocks;
  int2 *fvMinmax;
  float4 *fvRaycastResult;
  uchar4 *fvRaycastImage;
  int32_t *allocList;  // free-view render state + the staging list of k_small_freeview
  int32_t *statusDev;
  int32_t *rayBox, *fvRayBox;  // the range images' box records (k_raycast.h RB_*)
  int numTiles, noBlocks, gridIntegrate, lists;  // lists: the sorted list of allocated entries is kept (k_small.h)
};

struct BatchFrameP {  // one volume, this call
  FrameP p;           // the fusion / tracking camera, or the free camera of a render
  int active;         // 0: the volume takes no part in this call
  int tileX0, tileY0, tilesX, tilesY;  // the allocation mark's grid: the 16x16 tiles of the silhouette's box
  int publishSeq;
  int type;           // render: image type
  int pad;
  uchar4 *outRgba;
  float *outDepth;    // render: the caller's HBM buffers (an exchange slot)
};

struct BatchFrames { BatchFrameP f[kBatchMax]; };
static_assert(sizeof(BatchFrames) <= 3584, "the per-call table must fit the kernel argument segment (4 KB) next to the other arguments");

// ---- the view split of up to kBatchMax instances in one pass over the frame: item after item in the host's order, each cut-out
// sees the blanking of the items before it — exactly what the calls one after the other produce (masks may overlap)
struct BatchSplitItem {
  const uint8_t *mask, *rmask;  // copy mask (null: this instance is only blanked here, its volume lives elsewhere), delete mask
  uchar4 *dstRgb;
  float *dstDepth;
  int x0, y0, bw, bh, rx0, ry0, rbw, rbh;
  int wx0, wy0, wx1, wy1;  // the pixels of the cut-out that have to be written (k_edges.h k_split_silhouette `wr`)
};
struct BatchSplit { BatchSplitItem it[kBatchMax]; };
__global__ __launch_bounds__(256) void k_batch_split(uchar4 *srcRgb, float *srcDepth, int W, int H, BatchSplit b, int n) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const int idx = x + y * W;
  uchar4 c = srcRgb[idx];
  float d = srcDepth[idx];
  bool blanked = false;
  for (int k = 0; k < n; ++k) {
    const BatchSplitItem &it = b.it[k];
    if (it.mask && x >= it.wx0 && x < it.wx1 && y >= it.wy0 && y < it.wy1) {
      const int col = x - it.x0, row = y - it.y0;
      if (col >= 0 && col < it.bw && row >= 0 && row < it.bh && it.mask[row * it.bw + col] == 1) {
        it.dstRgb[idx] = c;
        it.dstDepth[idx] = d;
      } else {
        it.dstRgb[idx] = make_uchar4(255, 255, 255, 255);
        it.dstDepth[idx] = 0.0f;
      }
    }
    if (it.rmask) {
      const int rcol = x - it.rx0, rrow = y - it.ry0;
      if (rcol >= 0 && rcol < it.rbw && rrow >= 0 && rrow < it.rbh && it.rmask[rrow * it.rbw + rcol] == 1) {
        c = make_uchar4(0, 0, 0, 0);
        d = 0.0f;
        blanked = true;
      }
    }
  }
  if (blanked) { srcRgb[idx] = c; srcDepth[idx] = d; }
}

// ---- fusion + tracking render of every active volume
__global__ __launch_bounds__(256) void k_batch_alloc_mark(const BatchFrames frames, const BatchVolP *__restrict__ vols) {
  const BatchFrameP &f = frames.f[blockIdx.z];
  if (!f.active || (int)blockIdx.x >= f.tilesX || (int)blockIdx.y >= f.tilesY) return;
  const BatchVolP &v = vols[blockIdx.z];
  const int x = ((int)blockIdx.x + f.tileX0) * 16 + (threadIdx.x & 15), y = ((int)blockIdx.y + f.tileY0) * 16 + (threadIdx.x >> 4);
  alloc_mark_pixel<true>(f.p, v.s, v.depth, v.visType, x, y, x < f.p.W && y < f.p.H);
}

__global__ __launch_bounds__(kSmallThreads) void k_batch_small_alloc_visible(const BatchFrames frames,
                                                                             const BatchVolP *__restrict__ vols) {
  const BatchFrameP &f = frames.f[blockIdx.x];
  if (!f.active) return;
  const BatchVolP &v = vols[blockIdx.x];
  small_alloc_visible_body(f.p, v.s, v.depth, v.visType, v.numTiles, v.workList, v.visibleIDs, v.visBlocks, v.noBlocks, v.statusDev,
                           f.publishSeq, v.minmax, v.rayBox, v.lists);
}

template <bool RGB_SAME, bool PLAIN>
__global__ __launch_bounds__(64 * kIntegrateWaves, 7) void k_batch_integrate(const BatchFrames frames,
                                                                             const BatchVolP *__restrict__ vols) {
  const BatchFrameP &f = frames.f[blockIdx.y];
  const BatchVolP &v = vols[blockIdx.y];
  if (!f.active || (int)blockIdx.x >= v.gridIntegrate) return;
  integrate_body<RGB_SAME, PLAIN, 8, true>(f.p, v.s, v.depth, v.rgb, v.visBlocks, v.integrateStats, (int)blockIdx.x, v.gridIntegrate);
}

__global__ __launch_bounds__(256, 8) void k_batch_raycast(const BatchFrames frames, const BatchVolP *__restrict__ vols) {
  const BatchFrameP &f = frames.f[blockIdx.z];
  if (!f.active) return;
  const BatchVolP &v = vols[blockIdx.z];
  if (v.s.ctr[CTR_NO_VISIBLE_LIVE] <= 0) return;  // Prepare() is skipped without visible blocks
  if (v.rayBox) {  // (null only under DSR_RAY_BOX=0: the A/B against full-frame kernels)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) raybox_mark_ran(v.rayBox, f.p);
    if (!raybox_tile(v.rayBox, RB_DIRTY, blockIdx.x, blockIdx.y)) return;  // the tile keeps its miss (k_raycast.h RB_*)
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (x >= f.p.W || y >= f.p.H) return;
  const int mw = (f.p.W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const float2 mm = reinterpret_cast<const float2 *>(v.minmax)[(x >> 3) + (y >> 3) * mw];
  RC_STAT(RcStats st;)
  v.raycastResult[x + y * f.p.W] = cast_ray<DeviceOps>(f.p, v.s, x, y, mm RC_STAT(, st));
}

__global__ __launch_bounds__(256) void k_batch_icp_maps(const BatchFrames frames, const BatchVolP *__restrict__ vols) {
  const BatchFrameP &f = frames.f[blockIdx.z];
  if (!f.active) return;
  const BatchVolP &v = vols[blockIdx.z];
  if (v.s.ctr[CTR_NO_VISIBLE_LIVE] <= 0 || (v.rayBox && !raybox_tile(v.rayBox, RB_DIRTY, blockIdx.x, blockIdx.y))) return;
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= f.p.W || y >= f.p.H) return;
  float4 point, normal;
  uchar4 grey;
  icp_pixel<DeviceOps>(f.p, v.raycastResult, x, y, point, normal, grey);
  const int locId = x + y * f.p.W;
  v.raycastImage[locId] = grey;
  v.pointsMap[locId] = point;
  v.normalsMap[locId] = normal;
}

// ---- the preview: free-view list + range image, then the raycast that shades its own pixels, of every active volume
// frameTable (may be null): the volume's record of THIS call is also left in HBM for the paired render that follows — two
// BatchFrames do not fit one launch's kernel arguments (k_batch_raycast_pair)
__global__ __launch_bounds__(kSmallThreads) void k_batch_small_freeview(const BatchFrames frames,
                                                                        const BatchVolP *__restrict__ vols,
                                                                        BatchFrameP *__restrict__ frameTable) {
  const BatchFrameP &f = frames.f[blockIdx.x];
  if (frameTable) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&frames.f[blockIdx.x]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(frameTable + blockIdx.x);
    if (threadIdx.x < sizeof(BatchFrameP) / 4) dst[threadIdx.x] = src[threadIdx.x];
  }
  if (!f.active) return;
  const BatchVolP &v = vols[blockIdx.x];
  small_freeview_body(f.p, v.s, v.allocList, v.fvVisibleIDs, v.fvVisBlocks, v.noBlocks, v.fvMinmax, v.fvRayBox, v.lists);
}

__global__ __launch_bounds__(256) void k_batch_raycast_render(const BatchFrames frames, const BatchVolP *__restrict__ vols) {
  __shared__ int s_blocks[256][9];
  const BatchFrameP &f = frames.f[blockIdx.z];
  if (!f.active) return;
  const BatchVolP &v = vols[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (v.fvRayBox && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) raybox_mark_ran(v.fvRayBox, f.p);
  if (v.fvRayBox && !raybox_tile(v.fvRayBox, RB_DIRTY, blockIdx.x, blockIdx.y)) {  // the caller's buffers get the miss, ours keep it
    if (x >= f.p.W || y >= f.p.H) return;
    if (f.outRgba) f.outRgba[x + y * f.p.W] = make_uchar4(0, 0, 0, 0);
    if (f.outDepth) f.outDepth[x + y * f.p.W] = 0.0f;
    return;
  }
  if (x >= f.p.W || y >= f.p.H) return;
  const int mw = (f.p.W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  const float2 mm = reinterpret_cast<const float2 *>(v.fvMinmax)[(x >> 3) + (y >> 3) * mw];
  RC_STAT(RcStats st;)
  const float4 pt = cast_ray<DeviceOps>(f.p, v.s, x, y, mm RC_STAT(, st));
  const int locId = x + y * f.p.W;
  v.fvRaycastResult[locId] = pt;
  const uchar4 out = render_pixel<DeviceOps>(f.p, v.s, f.type, pt, s_blocks[threadIdx.x]);
  v.fvRaycastImage[locId] = out;
  if (f.outRgba) f.outRgba[locId] = out;
  if (f.outDepth) f.outDepth[locId] = render_depth(f.p, pt);
}

// k_raycast_pair for a batch: blockIdx.z < nv is k_batch_raycast of volume z (this frame's fusion cameras, kernel arguments),
// blockIdx.z >= nv is k_batch_raycast_render of volume z - nv (the preview cameras, from the table k_batch_small_freeview left)
__global__ __launch_bounds__(256) void k_batch_raycast_pair(const BatchFrames live, const BatchFrameP *__restrict__ freeTable,
                                                            const BatchVolP *__restrict__ vols, int nv) {
  __shared__ int s_blocks[256][9];
  if ((int)blockIdx.z < nv) {
    const BatchFrameP &f = live.f[blockIdx.z];
    if (!f.active) return;
    const BatchVolP &v = vols[blockIdx.z];
    raycast_box_tile(f.p, v.s, CTR_NO_VISIBLE_LIVE, reinterpret_cast<const float2 *>(v.minmax), v.raycastResult, v.rayBox, blockIdx.x, blockIdx.y);
  } else {
    const BatchFrameP &f = freeTable[blockIdx.z - nv];
    if (!f.active) return;
    const BatchVolP &v = vols[blockIdx.z - nv];
    raycast_render_tile(f.p, v.s, reinterpret_cast<const float2 *>(v.fvMinmax), v.fvRaycastResult, f.type, v.fvRaycastImage, f.outDepth,
                        f.outRgba, v.fvRayBox, blockIdx.x, blockIdx.y, s_blocks[threadIdx.x]);
  }
}

}  // namespace dsr
