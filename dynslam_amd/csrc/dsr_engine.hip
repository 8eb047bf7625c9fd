// dsr_engine.hip — host side of the HIP engine: implements the C ABI of include/dsr.h.
//
// One dsr_engine = one ITMMainEngine (InfiniTamDriver.h:79): scene (hash table, excess list,
// voxel block array), two render states, a view and the tracking pose — all resident in HBM.
// Every call enqueues kernels on the engine's own HIP stream; the hot path
// (update_view_dev / process_frame / prepare / decay) never synchronises: list lengths and
// free-list heads stay in device memory and kernels use fixed grids that read them there.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see __graft_entry__.build()).
#include <sched.h>

#include "dsr_internal.h"
using namespace dsr_internal;
#include "k_alloc.h"
#include "k_composite.h"
#include "k_decay.h"
#include "k_integrate.h"
#include "k_raycast.h"
#include "k_swap.h"
#include "k_mesh.h"
#include "k_small.h"
#include "k_batch.h"

// the volume batch's deferred work (paired render; defined with dsr_batch below, inside its extern "C" block)
extern "C" {
int dsri_batch_flush(dsr_batch *b);
bool dsri_batch_is_live(dsr_batch *b);
void dsri_batch_drop_deferred(dsr_batch *b);
}

namespace dsr_internal {
std::atomic<int> g_enginesOnDevice[64];  // live engines per device (range-image overlap policy, allocate_scene)
}

namespace {

std::atomic<unsigned long long> g_devMask{0};  // devices engines were created on (dsr_device_synchronize)
std::mutex g_engineMutex;
std::vector<dsr_engine *> g_liveEngines;         // every engine of the process: dsr_device_synchronize queues their deferred work first
// DSR_PIPELINED_VIEW=2: per GPU ONE view stream for all engines and ONE fusion stream for all instance-sized volumes (a host drives
// its instance volumes one after the other anyway): a map + N instances are then 4-5 streams instead of 2N + 4, and the map's
// fusion stream need not share a hardware queue with anybody
hipStream_t g_sharedViewStream[64] = {}, g_sharedSmallStream[64] = {};

// ----------------------------------------------------------------- host matrices
// ORUtils::Matrix4f helpers, float, same operation order as the oracle (host code is
// compiled with -ffp-contract=off as well).

Mat4 m4_identity() { Mat4 r; memset(&r, 0, sizeof r); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

Mat4 m4_mul(const Mat4 &l, const Mat4 &r) {
  Mat4 o;
  for (int x = 0; x < 4; x++)
    for (int y = 0; y < 4; y++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += l.m[k * 4 + y] * r.m[x * 4 + k];
      o.m[x * 4 + y] = s;
    }
  return o;
}

// ORUtils Matrix4::inv (cofactor expansion on the transposed source)
bool m4_inv(const Mat4 &in, Mat4 &out) {
  float t[12], s[16], det;
  float *d = out.m;
  for (int i = 0; i < 4; i++) { s[i] = in.m[i * 4]; s[i + 4] = in.m[i * 4 + 1]; s[i + 8] = in.m[i * 4 + 2]; s[i + 12] = in.m[i * 4 + 3]; }
  t[0] = s[10] * s[15]; t[1] = s[11] * s[14]; t[2] = s[9] * s[15]; t[3] = s[11] * s[13];
  t[4] = s[9] * s[14]; t[5] = s[10] * s[13]; t[6] = s[8] * s[15]; t[7] = s[11] * s[12];
  t[8] = s[8] * s[14]; t[9] = s[10] * s[12]; t[10] = s[8] * s[13]; t[11] = s[9] * s[12];
  d[0] = (t[0] * s[5] + t[3] * s[6] + t[4] * s[7]) - (t[1] * s[5] + t[2] * s[6] + t[5] * s[7]);
  d[1] = (t[1] * s[4] + t[6] * s[6] + t[9] * s[7]) - (t[0] * s[4] + t[7] * s[6] + t[8] * s[7]);
  d[2] = (t[2] * s[4] + t[7] * s[5] + t[10] * s[7]) - (t[3] * s[4] + t[6] * s[5] + t[11] * s[7]);
  d[3] = (t[5] * s[4] + t[8] * s[5] + t[11] * s[6]) - (t[4] * s[4] + t[9] * s[5] + t[10] * s[6]);
  d[4] = (t[1] * s[1] + t[2] * s[2] + t[5] * s[3]) - (t[0] * s[1] + t[3] * s[2] + t[4] * s[3]);
  d[5] = (t[0] * s[0] + t[7] * s[2] + t[8] * s[3]) - (t[1] * s[0] + t[6] * s[2] + t[9] * s[3]);
  d[6] = (t[3] * s[0] + t[6] * s[1] + t[11] * s[3]) - (t[2] * s[0] + t[7] * s[1] + t[10] * s[3]);
  d[7] = (t[4] * s[0] + t[9] * s[1] + t[10] * s[2]) - (t[5] * s[0] + t[8] * s[1] + t[11] * s[2]);
  t[0] = s[2] * s[7]; t[1] = s[3] * s[6]; t[2] = s[1] * s[7]; t[3] = s[3] * s[5];
  t[4] = s[1] * s[6]; t[5] = s[2] * s[5]; t[6] = s[0] * s[7]; t[7] = s[3] * s[4];
  t[8] = s[0] * s[6]; t[9] = s[2] * s[4]; t[10] = s[0] * s[5]; t[11] = s[1] * s[4];
  d[8] = (t[0] * s[13] + t[3] * s[14] + t[4] * s[15]) - (t[1] * s[13] + t[2] * s[14] + t[5] * s[15]);
  d[9] = (t[1] * s[12] + t[6] * s[14] + t[9] * s[15]) - (t[0] * s[12] + t[7] * s[14] + t[8] * s[15]);
  d[10] = (t[2] * s[12] + t[7] * s[13] + t[10] * s[15]) - (t[3] * s[12] + t[6] * s[13] + t[11] * s[15]);
  d[11] = (t[5] * s[12] + t[8] * s[13] + t[11] * s[14]) - (t[4] * s[12] + t[9] * s[13] + t[10] * s[14]);
  d[12] = (t[2] * s[10] + t[5] * s[11] + t[1] * s[9]) - (t[4] * s[11] + t[0] * s[9] + t[3] * s[10]);
  d[13] = (t[8] * s[11] + t[0] * s[8] + t[7] * s[10]) - (t[6] * s[10] + t[9] * s[11] + t[1] * s[8]);
  d[14] = (t[6] * s[9] + t[11] * s[11] + t[3] * s[8]) - (t[10] * s[11] + t[2] * s[8] + t[7] * s[9]);
  d[15] = (t[10] * s[10] + t[4] * s[8] + t[9] * s[9]) - (t[8] * s[9] + t[11] * s[10] + t[5] * s[8]);
  det = s[0] * d[0] + s[1] * d[1] + s[2] * d[2] + s[3] * d[3];
  if (det == 0.0f) return false;
  float inv = 1.0f / det;
  for (int i = 0; i < 16; i++) d[i] *= inv;
  return true;
}

}  // namespace


namespace {

void prof_resolve(dsr_engine *e) {
  if (e->profPending.empty()) return;
  if (e->viewStream) (void)hipStreamSynchronize(e->viewStream);
  (void)hipStreamSynchronize(e->stream);
  if (e->sideStream) (void)hipStreamSynchronize(e->sideStream);
  if (e->device >= 0 && e->device < 64 && g_ioStream[e->device]) (void)hipStreamSynchronize(g_ioStream[e->device]);
  for (auto &p : e->profPending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { e->profRecs[p.rec].ms += ms; e->profRecs[p.rec].launches++; }
    e->eventPool.push_back(p.a); e->eventPool.push_back(p.b);
  }
  e->profPending.clear();
}

FrameP make_frame_params(const dsr_engine *e, const Mat4 &M, const Mat4 &invM, const float proj[4]) {
  FrameP p;
  memset(&p, 0, sizeof p);
  p.M = M; p.invM = invM;
  p.M_rgb = m4_mul(e->calibInv, M);
  p.proj = make_float4(proj[0], proj[1], proj[2], proj[3]);
  p.proj_rgb = make_float4(e->calib.rgb.fx, e->calib.rgb.fy, e->calib.rgb.cx, e->calib.rgb.cy);
  p.mu = e->s.mu; p.voxelSize = e->s.voxel_size;
  p.vfMin = e->s.view_frustum_min; p.vfMax = e->s.view_frustum_max;
  p.W = e->W; p.H = e->H; p.Wr = e->Wr; p.Hr = e->Hr;
  p.maxW = e->s.max_w; p.stopAtMaxW = e->s.stop_integrating_at_max_w; p.depthWeighting = e->depthWeighting;
  p.rgbSame = (memcmp(&p.M_rgb, &p.M, sizeof(Mat4)) == 0 && memcmp(&p.proj, &p.proj_rgb, sizeof(float4)) == 0 &&
               e->W == e->Wr && e->H == e->Hr) ? 1 : 0;
  p.noBuckets = e->noBuckets; p.noExcess = e->noExcess; p.noTotalEntries = e->E; p.noBlocks = e->noBlocks;
  p.hashMask = (uint32_t)(e->noBuckets - 1);
  p.maxSteps = e->maxSteps;
  p.useSwapping = e->s.use_swapping;
  return p;
}

void depth_proj(const dsr_engine *e, float proj[4]) {
  proj[0] = e->calib.depth.fx; proj[1] = e->calib.depth.fy; proj[2] = e->calib.depth.cx; proj[3] = e->calib.depth.cy;
}

int reset_scene(dsr_engine *e) {
  e->sceneVersion++;
  e->noVisibleValid = false;
  e->listVersion++;
  LAUNCH(e, "reset", k_reset_table, dim3(div_up(e->E, 256)), dim3(256), e->scene.table, e->E, e->scene.allocKey);
  LAUNCH(e, "reset", k_iota, dim3(div_up(e->noExcess, 256)), dim3(256), e->scene.excessAllocList, e->noExcess);
  LAUNCH(e, "reset", k_iota, dim3(div_up(e->noBlocks, 256)), dim3(256), e->scene.voxelAllocList, e->noBlocks);
  LAUNCH(e, "reset", k_reset_vba, dim3(4096), dim3(256), reinterpret_cast<uint4 *>(e->scene.vba),
         (size_t)e->noBlocks * (kBlockBytes / 16));
  LAUNCH(e, "reset", k_reset_counters, dim3(1), dim3(64), e->scene.ctr, e->scene.work, e->noBlocks, e->noExcess);
  if (e->scene.swapState) {
    HIP_TRY(hipMemsetAsync(e->scene.swapState, 0, (size_t)e->E, e->stream));
    HIP_TRY(hipMemsetAsync(e->scene.swapStored, 0, (size_t)e->E, e->stream));
    HIP_TRY(hipMemsetAsync(e->scene.swapSlot, 0xff, (size_t)e->E * 4, e->stream));  // -1: the entry owns no host slot yet
    HIP_TRY(hipStreamSynchronize(e->stream));
    // the slabs are kept; the slot counter restarts with the counters
    e->hostUsedUpper = 0; e->hostUsedPending = false; e->hostUsedCallsSince = 0;
  }
  HIP_TRY(hipMemsetAsync(e->live.visType, 0, (size_t)e->E, e->stream));
  HIP_TRY(hipMemsetAsync(e->freeview.visType, 0, (size_t)e->E, e->stream));
  HIP_TRY(hipMemsetAsync(e->scene.allocGrp, 0, (size_t)e->numTilesE * (kTile / 32) * 4, e->stream));
  HIP_TRY(hipMemsetAsync(e->scene.allocTile, 0, ((size_t)e->numTilesE + 1) * 8, e->stream));
  if (e->scene.visBits) {
    HIP_TRY(hipMemsetAsync(e->scene.visGrp, 0, (size_t)kSmallBitWords * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->scene.visBits, 0, (size_t)kSmallBitWords * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->scene.allocBits, 0, (size_t)kSmallBitWords * 4, e->stream));
  }
  e->fifoHead = 0; e->fifoLen = 0;
  // (the render buffers keep whatever the previous scene left in them: the next raycast of each render state is a full-frame one)
  for (RenderStateDev *rs : {&e->live, &e->freeview})
    if (rs->rayBox) {
      hipLaunchKernelGGL(k_raybox_reset, dim3(1), dim3(RB_WORDS), 0, e->stream, rs->rayBox, (e->W + 7) / 8, (e->H + 7) / 8, e->rayBoxLive ? 1 : 0);
    }
  e->rayBoxLive = true;
  HIP_TRY(hipGetLastError());
  if (e->statusHost) {
    // the published words describe a scene that no longer exists; the next allocation publishes number statusSeq + 1
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->statusHost[0] = 0; e->statusHost[1] = DSR_OK;
  }
  return DSR_OK;
}

void free_all(dsr_engine *e) {
  auto F = [](void *p) { if (p) (void)hipFree(p); };
  F(e->scene.table); F(e->scene.vba); F(e->scene.voxelAllocList); F(e->scene.excessAllocList);
  F(e->scene.ctr); F(e->scene.work); F(e->scene.allocKey); F(e->scene.allocGrp); F(e->scene.allocTile);
  F(e->scene.visGrp); F(e->scene.visBits); F(e->scene.allocBits); F(e->scene.allocIds);
  for (RenderStateDev *rs : {&e->live, &e->freeview}) {
    F(rs->visibleIDs); F(rs->visibleIDsAlt); F(rs->visBlocks); F(rs->visBlocksAlt); F(rs->visType); F(rs->minmax); F(rs->raycastResult); F(rs->raycastImage); F(rs->rayBox);
  }
  F(e->tileSums); F(e->integrateStats); F(e->allocList); F(e->allocWork); F(e->meshTris); F(e->rgb); F(e->depth); F(e->depthTmp); F(e->rawDepth); F(e->pointsMap); F(e->normalsMap);
  F(e->freeDepth); F(e->aosScratch);
  F(e->fifoPlanes); F(e->decayCand); F(e->decayFlags);
  if (e->maskHost) (void)hipHostFree(e->maskHost);
  for (auto ev : e->maskEvent) if (ev) (void)hipEventDestroy(ev);
  F(e->scene.swapState); F(e->scene.swapStored); F(e->swapStagingDev); F(e->swapIdsDev); F(e->swapFlagsDev);
  F(e->scene.swapSlot); F(e->scene.hostSlabs);
  if (e->hostUsedSeen) (void)hipHostFree(e->hostUsedSeen);
  if (e->hostUsedEvent) (void)hipEventDestroy(e->hostUsedEvent);
  for (auto p : e->hostSlabs) (void)hipHostFree(p);
  for (int k = 0; k < 2; ++k) {
    if (e->upPin[k]) (void)hipHostFree(e->upPin[k]);
    if (e->upSlotFree[k]) (void)hipEventDestroy(e->upSlotFree[k]);
  }
  F(e->upDev); F(e->pvDev); F(e->xferRgb); F(e->xferDepth);
  F(e->rgbAlt); F(e->depthAlt);
  for (hipEvent_t ev : {e->evAltFree, e->evFusionRead}) if (ev) (void)hipEventDestroy(ev);
  if (e->viewStream && e->ownsViewStream) (void)hipStreamDestroy(e->viewStream);
  if (e->pvPin) (void)hipHostFree(e->pvPin);
  if (e->statusHost) (void)hipHostFree(e->statusHost);
  for (hipEvent_t ev : {e->evUploaded, e->evIngested, e->evView, e->evViewRead}) if (ev) (void)hipEventDestroy(ev);
  if (e->xEvent) (void)hipEventDestroy(e->xEvent);
  if (e->xEvent2) (void)hipEventDestroy(e->xEvent2);
  if (e->orderEvent) (void)hipEventDestroy(e->orderEvent);
  for (auto &p : e->profPending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto ev : e->eventPool) (void)hipEventDestroy(ev);
  if (e->evList) (void)hipEventDestroy(e->evList);
  if (e->evExpected) (void)hipEventDestroy(e->evExpected);
  if (e->sideStream) (void)hipStreamDestroy(e->sideStream);
  if (e->stream && e->ownsStream) (void)hipStreamDestroy(e->stream);
}

// AllocateSceneFromDepth: mark -> ordered commit -> ordered visible list
int expected_depths(dsr_engine *e, RenderStateDev &rs, const FrameP &p);

int allocate_scene(dsr_engine *e) {
  e->sceneVersion++;
  e->noVisibleValid = false;
  e->listVersion++;
  float proj[4]; depth_proj(e, proj);
  FrameP p = make_frame_params(e, e->M_d, e->invM_d, proj);
  RenderStateDev &rs = e->live;
  { int st = before_fusion(e); if (st) return st; }
  // a range image of the PREVIOUS list may still be running on the side stream (back-to-back fusion calls without a Prepare in
  // between): it reads the list and the count this call rewrites
  if (e->sidePending) { HIP_TRY(hipStreamWaitEvent(e->stream, e->evExpected, 0)); e->sidePending = false; }
  if (e->smallPath) {
    // an instance-sized volume: the mark over the box the view is not empty in, then ONE workgroup for everything that needs the
    // whole table in order — commit, visible list, range image (k_small.h)
    const int tx0 = e->viewBox[0] / 16, ty0 = e->viewBox[1] / 16;
    const int tx1 = div_up(std::min(e->viewBox[2], e->W), 16), ty1 = div_up(std::min(e->viewBox[3], e->H), 16);
    if (tx1 > tx0 && ty1 > ty0)
      LAUNCH(e, "alloc_mark", k_alloc_mark<true>, dim3(tx1 - tx0, ty1 - ty0), dim3(256), p, e->scene, (const float *)e->depth,
             rs.visType, tx0, ty0);
    if (e->statusDev) e->statusSeq++;
    {
      ProfScope _ps(e, "small_alloc_visible");
      hipLaunchKernelGGL(k_small_alloc_visible, dim3(1), dim3(kSmallThreads), e->smallLdsBytes, e->stream, p, e->scene,
                         (const float *)e->depth, rs.visType, e->numTilesE, e->allocWork, rs.visibleIDs, rs.visBlocks, e->noBlocks,
                         e->statusDev, e->statusSeq, reinterpret_cast<int2 *>(rs.minmax), rs.rayBox, e->smallLists ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
    { int st = after_fusion(e); if (st) return st; }
    e->liveExp.valid = true; e->liveExp.onSide = false; e->liveExp.version = e->listVersion; e->liveExp.M = e->M_d;
    memcpy(e->liveExp.proj, proj, sizeof proj);
    return DSR_OK;
  }
  LAUNCH(e, "retest_prev_visible", k_retest_previous_visible, dim3(1024), dim3(256), p, e->scene,
         (const int4 *)rs.visBlocks, rs.visType);
  int2 *allocTile = reinterpret_cast<int2 *>(e->scene.allocTile);
  LAUNCH(e, "alloc_mark", k_alloc_mark<false>, dim3(div_up(e->W, 16), div_up(e->H, 16)), dim3(256), p, e->scene,
         (const float *)e->depth, rs.visType, 0, 0);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), allocTile, e->numTilesE, e->scene, (int)SCAN_ALLOC, 0);
  LAUNCH(e, "alloc_commit", k_alloc_commit, dim3(e->numTilesE), dim3(kTileThreads), p, e->scene, allocTile, e->allocWork);
  LAUNCH(e, "alloc_apply", k_alloc_apply, dim3(256), dim3(256), p, e->scene, (const float *)e->depth,
         (const int4 *)e->allocWork, rs.visType);
  LAUNCH(e, "visible_count", k_visible_count<false>, dim3(e->numTilesE), dim3(kTileThreads), p, e->scene, rs.visType, e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene,
         (int)SCAN_VISIBLE_LIVE, e->noBlocks);
  if (e->statusDev) e->statusSeq++;
  LAUNCH(e, "visible_write", k_visible_write, dim3(e->numTilesE), dim3(kTileThreads), e->E, (const uint8_t *)rs.visType,
         (const int2 *)e->tileSums, rs.visibleIDs, e->noBlocks, e->scene, (int)e->s.use_swapping, rs.visBlocks, e->statusDev,
         e->statusSeq);
  HIP_TRY(hipGetLastError());
  { int st = after_fusion(e); if (st) return st; }
  e->liveExp.valid = false;
  // ... when this volume has the GPU to itself: with other volumes' engines on the device (a map + its instance volumes)
  // their streams fill the phases the side stream would, and the extra stream only costs (configs[2]: 733 vs 686 frames/s)
  if (e->overlapExpected && (e->device >= 64 || g_enginesOnDevice[e->device].load(std::memory_order_relaxed) <= 1)) {
    // the list is final: start the live view's range image on the side stream, under the integration that follows
    HIP_TRY(hipEventRecord(e->evList, e->stream));
    HIP_TRY(hipStreamWaitEvent(e->sideStream, e->evList, 0));
    {
      StreamSwap sw(e, e->sideStream);
      int st = expected_depths(e, rs, p);
      if (st) return st;
    }
    HIP_TRY(hipEventRecord(e->evExpected, e->sideStream));
    e->sidePending = true;
    e->liveExp.valid = true; e->liveExp.onSide = true; e->liveExp.version = e->listVersion; e->liveExp.M = e->M_d;
    memcpy(e->liveExp.proj, proj, sizeof proj);
  }
  return DSR_OK;
}

int integrate_scene(dsr_engine *e) {
  e->sceneVersion++;
  float proj[4]; depth_proj(e, proj);
  FrameP p = make_frame_params(e, e->M_d, e->invM_d, proj);
  const bool plain = !p.depthWeighting && !p.stopAtMaxW && e->shortDivMuExact;
  { int st = before_fusion(e); if (st) return st; }
  // (XLDS: the wave-uniform x terms of the camera transform through LDS, k_integrate.h — 562 -> 540-545 us on the bench workload,
  //  bit-identical, profiles/r04b_integrate_xlds_ab.log; the round-3 form is no longer instantiated)
#define LAUNCH_INTEGRATE(A, B, VOX, OCC)                                                                     \
  LAUNCH(e, "integrate", (k_integrate<A, B, VOX, OCC, true>), dim3(e->gridIntegrate), dim3(256), p, e->scene, \
         (const float *)e->depth, (const uchar4 *)e->rgb, (const int4 *)e->live.visBlocks, e->integrateStats)
#define LAUNCH_INTEGRATE_V(VOX, OCC)                                                                         \
  do {                                                                                                       \
    if (p.rgbSame) { if (plain) LAUNCH_INTEGRATE(true, true, VOX, OCC); else LAUNCH_INTEGRATE(true, false, VOX, OCC); } \
    else { if (plain) LAUNCH_INTEGRATE(false, true, VOX, OCC); else LAUNCH_INTEGRATE(false, false, VOX, OCC); }         \
  } while (0)
  // whole block per wave, 8 voxels per lane, register allocation for 7 waves per SIMD (k_integrate.h).  (The XLDS form needs 57
  // VGPRs, so 8 waves are resident anyway; compiled FOR 8 the scalar-register budget shrinks: 40 instead of 23 spill writes.)
  LAUNCH_INTEGRATE_V(8, 7);
#undef LAUNCH_INTEGRATE_V
#undef LAUNCH_INTEGRATE
  HIP_TRY(hipGetLastError());
  return after_fusion(e);
}

int expected_depths(dsr_engine *e, RenderStateDev &rs, const FrameP &p) {
  const int mw = (e->W + 7) / 8, mh = (e->H + 7) / 8;
  if (e->smallVolume && (size_t)mw * mh * sizeof(int2) <= 64 * 1024) {
    // an instance-sized volume: one workgroup, one launch (k_raycast.h k_expected_depth_one)
    ProfScope _ps(e, "expected_depth");
    hipLaunchKernelGGL(k_expected_depth_one, dim3(1), dim3(1024), (size_t)mw * mh * sizeof(int2), e->stream, p, e->scene,
                       (const int4 *)rs.visBlocks, rs.ctrIdx, reinterpret_cast<int2 *>(rs.minmax),
                       rs.ctrIdx == CTR_NO_VISIBLE_LIVE ? 1 : 0, rs.rayBox);
    return DSR_OK;
  }
  LAUNCH(e, "minmax_init", k_minmax_init, dim3(div_up(mw * mh, 256)), dim3(256), rs.minmax, mw * mh,
         (const int32_t *)e->scene.ctr, rs.ctrIdx == CTR_NO_VISIBLE_LIVE ? (int)CTR_NO_VISIBLE_LIVE : -1);
  const size_t ldsBytes = (size_t)mw * mh * sizeof(int2);
  if (ldsBytes <= 64 * 1024) {
    // range image privatised in LDS by a few large workgroups (k_raycast.h)
    ProfScope _ps(e, "expected_depth");
    hipLaunchKernelGGL(k_expected_depth_lds, dim3(e->gridExpected), dim3(1024), ldsBytes, e->stream, p, e->scene,
                       (const int4 *)rs.visBlocks, rs.ctrIdx, reinterpret_cast<int2 *>(rs.minmax));
  } else {
    LAUNCH(e, "expected_depth", k_expected_depth, dim3(1024), dim3(256), p, e->scene, (const int4 *)rs.visBlocks,
           rs.ctrIdx, reinterpret_cast<int2 *>(rs.minmax));
  }
  return DSR_OK;
}

int launch_raycast(dsr_engine *e, const char *name, const FrameP &p, RenderStateDev &rs) {
  dim3 g(div_up(e->W, 16), div_up(e->H, 16));
  if (rs.rayBox) LAUNCH(e, name, k_raycast_box, g, dim3(256), p, e->scene, rs.ctrIdx, (const float2 *)rs.minmax, rs.raycastResult, rs.rayBox);
  else LAUNCH(e, name, k_raycast, g, dim3(256), p, e->scene, rs.ctrIdx, (const float2 *)rs.minmax, rs.raycastResult);
  return DSR_OK;
}

// ---- the paired render (dsr_engine::pairRender): the deferred tracking render of one engine
bool pair_render_default() {
  static const bool on = !(getenv("DSR_PAIR_RENDER") && atoi(getenv("DSR_PAIR_RENDER")) == 0);
  return on;
}
int launch_icp_maps(dsr_engine *e, const FrameP &p) {
  RenderStateDev &rs = e->live;
  dim3 g(div_up(e->W, 16), div_up(e->H, 16));
  if (rs.rayBox) LAUNCH(e, "icp_maps", k_icp_maps_box, g, dim3(256), p, e->scene, (const float4 *)rs.raycastResult, e->pointsMap,
                        e->normalsMap, rs.raycastImage, (const int32_t *)rs.rayBox);
  else LAUNCH(e, "icp_maps", k_icp_maps, g, dim3(256), p, e->scene, (const float4 *)rs.raycastResult, e->pointsMap,
              e->normalsMap, rs.raycastImage);
  return DSR_OK;
}
int flush_track_render(dsr_engine *e) {
  if (!e->trackRender.pending) return DSR_OK;
  e->trackRender.pending = false;
  int st = launch_raycast(e, "raycast", e->trackRender.p, e->live);
  if (st) return st;
  if ((st = launch_icp_maps(e, e->trackRender.p))) return st;
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

const char *status_text(int status) {
  return status == DSR_E_OUT_OF_BLOCKS ? "out of voxel blocks / excess list entries"
                                       : "allocation ray longer than the order key allows: the pose is not rigid (k_alloc.h)";
}

// the status word, and with it (same copy, same synchronisation) the live view's noVisibleBlocks: the host reads that
// count right after fusion (InfiniTamDriver.h:150) and should not pay a second synchronisation for it
int sticky_status(dsr_engine *e, int *status) {
  static_assert(CTR_NO_VISIBLE_LIVE + 2 == CTR_STATUS, "the three words are fetched with one copy");
  int32_t w[3] = {0, 0, 0};
  HIP_TRY(hipMemcpyAsync(w, e->scene.ctr + CTR_NO_VISIBLE_LIVE, sizeof w, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  *status = w[2];
  e->noVisibleSeen = w[0]; e->noVisibleValid = true;
  return DSR_OK;
}

// The same two words as k_visible_write PUBLISHED them for allocation number e->statusSeq (pinned, device-mapped memory): the
// host polls instead of draining the stream, so it returns as soon as the allocation kernels of this frame have run — the
// integration and everything after it are still in flight.  Falls back to the copy above when the stream ran dry without the
// word arriving (a failed launch) or the engine has no published word.
int published_status(dsr_engine *e, int *status) {
  if (!e->statusHost) return sticky_status(e, status);
  const int seq = e->statusSeq;
  for (unsigned spins = 1;; ++spins) {
    // (>= in wrap-safe arithmetic: a second allocation queued before this wait has published a later number already)
    if ((int)((unsigned)__atomic_load_n(e->statusHost + 2, __ATOMIC_ACQUIRE) - (unsigned)seq) >= 0) break;
    if ((spins & 0xfff) == 0) {
      const hipError_t q = hipStreamQuery(e->stream);
      if (q == hipSuccess) {
        if ((int)((unsigned)__atomic_load_n(e->statusHost + 2, __ATOMIC_ACQUIRE) - (unsigned)seq) >= 0) break;
        return sticky_status(e, status);
      }
      if (q != hipErrorNotReady) return fail(DSR_E_DEVICE, std::string("engine stream: ") + hipGetErrorString(q));
      sched_yield();
    }
  }
  *status = e->statusHost[1];
  e->noVisibleSeen = e->statusHost[0]; e->noVisibleValid = true;
  return DSR_OK;
}

int ensure_fifo(dsr_engine *e, int slotsNeeded) {
  if (slotsNeeded <= e->fifoCap) return DSR_OK;
  // grow the ring (min_age went up), keeping queue order; (min_age + 1) x E / 8 bytes in total
  e->fifoPlaneWords = ((size_t)e->E + 31) / 32;
  uint32_t *np = nullptr;
  int st = dmalloc(&np, (size_t)slotsNeeded * e->fifoPlaneWords);
  if (st) return st;
  for (int i = 0; i < e->fifoLen; ++i) {
    const int old = (e->fifoHead + i) % e->fifoCap;
    HIP_TRY(hipMemcpyAsync(np + (size_t)i * e->fifoPlaneWords, e->fifoPlanes + (size_t)old * e->fifoPlaneWords,
                           e->fifoPlaneWords * 4, hipMemcpyDeviceToDevice, e->stream));
  }
  if (e->fifoPlanes) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(e->fifoPlanes);
  }
  e->fifoPlanes = np;
  e->fifoCap = slotsNeeded;
  e->fifoHead = 0;
  return DSR_OK;
}

// Host store (ITMGlobalCache): a pool of pinned slabs that the swap kernels address directly
// (k_swap.h).  The host's only job is to keep the pool ahead of the device's slot counter: it
// tracks an upper bound of the counter (every swap-out batch takes at most kTransferBlocks
// slots), tightened by an asynchronous read-back that is never waited for, and adds a slab when
// the bound comes within one batch of the capacity.
uint8_t *host_slot_ptr(dsr_engine *e, long long slot) {
  return e->hostSlabs[(size_t)(slot / e->scene.slabBlocks)] + (size_t)(slot % e->scene.slabBlocks) * kBlockBytes;
}

int add_host_slab(dsr_engine *e) {
  if ((int)e->hostSlabs.size() >= dsr_engine::kMaxHostSlabs) return fail(DSR_E_NOMEM, "host store is full");
  uint8_t *slab = nullptr;
  if (hipHostMalloc(reinterpret_cast<void **>(&slab), (size_t)e->scene.slabBlocks * kBlockBytes, hipHostMallocDefault) != hipSuccess)
    return fail(DSR_E_NOMEM, "host store slab allocation failed");
  e->hostSlabs.push_back(slab);
  // publish the pointer to the kernels (ordered on the stream before the next swap kernels)
  HIP_TRY(hipMemcpyAsync(e->scene.hostSlabs + (e->hostSlabs.size() - 1), &e->hostSlabs.back(), sizeof(uint8_t *),
                         hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));  // the source of that copy is this vector's storage
  return DSR_OK;
}

// ITMSwappingEngine::IntegrateGlobalIntoLocal: host store -> staging -> combine into the local blocks
int swap_in(dsr_engine *e) {
  LAUNCH(e, "swap_list", (k_swap_count<false>), dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
         (const uint8_t *)e->live.visType, e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene, (int)SCAN_SWAP_IN,
         (int)kTransferBlocks);
  LAUNCH(e, "swap_list", (k_swap_write<false>), dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
         (const uint8_t *)e->live.visType, (const int2 *)e->tileSums, e->swapIdsDev, e->swapFlagsDev);
  LAUNCH(e, "swapin_fetch", k_swapin_fetch, dim3(1024), dim3(256), e->scene, (const int32_t *)e->swapIdsDev,
         (const uint8_t *)e->swapFlagsDev, e->swapStagingDev);
  LAUNCH(e, "swapin_combine", k_swapin_combine, dim3(1024), dim3(256), e->scene, (int)e->s.max_w,
         (const int32_t *)e->swapIdsDev, (const uint8_t *)e->swapFlagsDev, (const uint8_t *)e->swapStagingDev);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

// ITMSwappingEngine::SaveToGlobalMemory: invisible resident blocks -> host store
int swap_out(dsr_engine *e) {
  // tighten the bound with the last read-back, if it has arrived
  if (e->hostUsedPending && hipEventQuery(e->hostUsedEvent) == hipSuccess) {
    e->hostUsedUpper = std::min(e->hostUsedUpper, (long long)*e->hostUsedSeen + e->hostUsedCallsSince * kTransferBlocks);
    e->hostUsedPending = false;
  }
  while ((long long)e->hostSlabs.size() * e->scene.slabBlocks < e->hostUsedUpper + kTransferBlocks) {
    int st = add_host_slab(e);
    if (st) return st;
  }
  LAUNCH(e, "swap_list", (k_swap_count<true>), dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
         (const uint8_t *)e->live.visType, e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene, (int)SCAN_SWAP_OUT,
         (int)kTransferBlocks);
  LAUNCH(e, "swap_list", (k_swap_write<true>), dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
         (const uint8_t *)e->live.visType, (const int2 *)e->tileSums, e->swapIdsDev, e->swapFlagsDev);
  LAUNCH(e, "swapout_move", k_swapout_move, dim3(1024), dim3(256), e->scene, (const int32_t *)e->swapIdsDev);
  e->hostUsedUpper += kTransferBlocks;
  e->hostUsedCallsSince++;
  if (!e->hostUsedPending) {  // ask for the counter; the answer is picked up by a later frame
    HIP_TRY(hipMemcpyAsync(e->hostUsedSeen, e->scene.ctr + CTR_HOST_USED, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipEventRecord(e->hostUsedEvent, e->stream));
    e->hostUsedPending = true;
    e->hostUsedCallsSince = 0;
  }
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

}  // namespace

std::string &dsr_internal::last_error() {
  thread_local std::string message;
  return message;
}
int dsr_internal::engine_set_device(dsr_engine *e) { return set_device(e); }
void dsr_internal::prof_resolve_pending(dsr_engine *e) { prof_resolve(e); }
int dsr_internal::engine_flush_deferred(dsr_engine *e) {
  int st = flush_track_render(e);
  if (st) return st;
  if (e->ownerBatch) {
    if (dsri_batch_is_live(e->ownerBatch)) return dsri_batch_flush(e->ownerBatch);
    e->ownerBatch = nullptr;  // destroyed since
  }
  return DSR_OK;
}
void dsr_internal::engine_prof_resolve(dsr_engine *e) { prof_resolve(e); }

extern "C" {

int dsr_abi_version(void) { return DSR_ABI_VERSION; }

void dsr_default_settings(dsr_settings *s) {
  memset(s, 0, sizeof *s);
  s->voxel_size = 0.005f; s->mu = 0.02f; s->max_w = 100;
  s->view_frustum_min = 0.2f; s->view_frustum_max = 3.0f;
  s->stop_integrating_at_max_w = 0;
  s->sdf_local_block_num = DSR_DEFAULT_LOCAL_BLOCK_NUM;
  s->hash_bucket_num = DSR_DEFAULT_BUCKET_NUM;
  s->excess_list_size = DSR_DEFAULT_EXCESS_LIST_SIZE;
  s->use_swapping = 0; s->use_bilateral_filter = 0; s->device = -1; s->sync_status = 1;
}

const char *dsr_last_error(void) { return dsr_internal::last_error().c_str(); }

int dsr_engine_create(const dsr_settings *settings, const dsr_calib *calib, dsr_engine **out) {
  if (!settings || !calib || !out) return fail(DSR_E_ARG, "null argument");
  const dsr_settings &s = *settings;
  if (s.hash_bucket_num <= 0 || (s.hash_bucket_num & (s.hash_bucket_num - 1))) return fail(DSR_E_ARG, "hash_bucket_num must be a power of two");
  if (s.excess_list_size <= 0 || s.sdf_local_block_num <= 0) return fail(DSR_E_ARG, "bad table sizes");
  if (!(s.voxel_size > 0) || !(s.mu > 0) || s.max_w < 1 || s.max_w > 255) return fail(DSR_E_ARG, "bad scene params");
  if (calib->depth.width <= 0 || calib->depth.height <= 0) return fail(DSR_E_ARG, "bad image size");
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
    return fail(DSR_E_DEVICE, "no HIP device: the HIP engine has no CPU fallback");
  dsr_engine *e = new (std::nothrow) dsr_engine();
  if (!e) return fail(DSR_E_NOMEM, "oom");
  e->s = s; e->calib = *calib;
  if (s.device >= 0) e->device = s.device;
  else if (hipGetDevice(&e->device) != hipSuccess) e->device = 0;
  if (e->device >= nDev) { delete e; return fail(DSR_E_ARG, "device ordinal out of range"); }
  if (e->device < 64) g_devMask.fetch_or(1ull << e->device);
  const int countedDevice = e->device;
  e->W = calib->depth.width; e->H = calib->depth.height; e->Wr = calib->rgb.width; e->Hr = calib->rgb.height;
  e->P = e->W * e->H;
  e->noBuckets = s.hash_bucket_num; e->noExcess = s.excess_list_size; e->E = e->noBuckets + e->noExcess;
  e->noBlocks = s.sdf_local_block_num;
  e->numTilesE = div_up(e->E, kTile); e->numTilesB = div_up(e->noBlocks, kTile);
  e->numTilesMax = std::max(e->numTilesE, e->numTilesB);
  {
    // bound on noSteps = ceil(2*|dir|), |dir| ~ 2*mu / (8*voxelSize) block units
    double len = 2.0 * (double)s.mu / (8.0 * (double)s.voxel_size);
    double S = std::ceil(2.0 * len * 1.05) + 3.0;
    if (S * (double)e->P >= 4294967295.0) { delete e; return fail(DSR_E_ARG, "mu/voxel_size ratio too large for the 32-bit allocation key"); }
    e->maxSteps = (uint32_t)S;
  }
  if (const char *ge = getenv("DSR_GRID_EXPECTED")) e->gridExpected = std::max(1, atoi(ge));
  e->smallVolume = s.sdf_local_block_num <= 16384;
  if (const char *sv = getenv("DSR_SMALL_VOLUME")) e->smallVolume = atoi(sv) != 0;  // tests: both paths on any volume
  e->gridDecay = std::min(32768, std::max(256, s.sdf_local_block_num / 16));
  if (const char *gd = getenv("DSR_GRID_DECAY")) e->gridDecay = std::max(1, atoi(gd));
  e->gridIntegrate = std::min(16384, std::max(256, s.sdf_local_block_num / 4));
  if (const char *gi = getenv("DSR_GRID_INTEGRATE")) e->gridIntegrate = std::max(1, atoi(gi));
  Mat4 trafo; memcpy(trafo.m, calib->trafo_rgb_to_depth, sizeof trafo.m);
  if (!m4_inv(trafo, e->calibInv)) { delete e; return fail(DSR_E_ARG, "singular trafo_rgb_to_depth"); }
  e->M_d = m4_identity(); e->invM_d = m4_identity();

  int st = set_device(e);
  if (st) { delete e; return st; }
#define ALLOC(expr) if ((st = (expr)) != DSR_OK) { free_all(e); delete e; return st; }
  // The view pipeline (see dsr_engine::pipelinedView).  Engines a HOST waits on (sync_status: DynSLAM's call pattern, the shim) get
  // form 2 by default since round 5 — the view double buffered, ONE view stream per GPU for all engines and one fusion stream for
  // all instance-sized volumes: configs[2] through the C++ host 410 -> 457 frames/s (profiles/r04d_*, r05d), the whole GPU suite
  // green under it (profiles/r05e_gpu_suite_pv2.log).  Engines driven without status waits (the bench, the sharded scene, a volume
  // batch) keep one stream each and no view stream.  env DSR_PIPELINED_VIEW=0 / 1 / 2 overrides (1: streams per engine).
  // dsr_settings.view_pipeline (ABI 5) names a form explicitly — a host that wants its status-reporting volumes in a batch or on a
  // shared stream asks for _OFF instead of switching the whole process with the environment variable.
  if (s.view_pipeline < DSR_VIEW_PIPELINE_AUTO || s.view_pipeline > DSR_VIEW_PIPELINE_SHARED) { delete e; return fail(DSR_E_ARG, "bad view_pipeline"); }
  const int pvMode = s.view_pipeline == DSR_VIEW_PIPELINE_OFF ? 0 : s.view_pipeline == DSR_VIEW_PIPELINE_PER_ENGINE ? 1
                     : s.view_pipeline == DSR_VIEW_PIPELINE_SHARED ? 3
                     : getenv("DSR_PIPELINED_VIEW") ? atoi(getenv("DSR_PIPELINED_VIEW")) : (s.sync_status ? 3 : 0);
  auto shared_stream = [&](hipStream_t *table) -> hipStream_t {
    if (e->device < 0 || e->device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_ioMutex);
    if (!table[e->device] && create_stream(&table[e->device]) != hipSuccess) table[e->device] = nullptr;
    return table[e->device];
  };
  // Form 3 (round 6, the default of engines a host waits on): as form 2, but the instance-sized volumes fuse on the shared VIEW
  // stream itself — an instance's cut-out, mark, lists, integration and renders are then ordered by ONE queue, where form 2 paid a
  // cross-queue dependency (13-20 us on this part) between the view stream and the small volumes' fusion stream per instance and
  // frame: configs[2] through the C++ host 696-702 -> 748-756 frames/s at 20 steps, 774-776 -> 808-814 at 45, identical results
  // (profiles/r06q_through_shim_pv3.log).  env DSR_PIPELINED_VIEW=2: form 2.
  if (pvMode == 3 && s.sdf_local_block_num <= 16384 && (e->stream = shared_stream(g_sharedViewStream))) e->ownsStream = false;
  else if (pvMode >= 2 && s.sdf_local_block_num <= 16384 && (e->stream = shared_stream(g_sharedSmallStream))) e->ownsStream = false;
  else if (create_stream(&e->stream) != hipSuccess) { delete e; return fail(DSR_E_DEVICE, "hipStreamCreate failed"); }
  // The side stream exists only for volumes whose integration is long enough to hide something under (not for instance-sized
  // ones, not for a map at the reference's 5 cm / 2^18 blocks, whose whole frame is 0.24 ms), and at DEFAULT priority: every stream of a process competes for the same few hardware queues, and a scene of one
  // map + N instance volumes is N + 1 engines — with a (high-priority) side stream per engine `bench.py --instance-volumes 8`
  // fell from 4900 to 1400 volume-frames/s, with plain ones to 4570 (profiles/r03n_*; GPU_MAX_HW_QUEUES tells the same story).
  e->overlapExpected = s.sdf_local_block_num >= (1 << 20);  // >= 4 GiB of voxels: fine voxels, integrations of hundreds of us
  if (const char *ov = getenv("DSR_OVERLAP_EXPECTED")) e->overlapExpected = atoi(ov) != 0;
  if (e->overlapExpected &&
      (hipStreamCreateWithFlags(&e->sideStream, hipStreamNonBlocking) != hipSuccess ||
       hipEventCreateWithFlags(&e->evList, order_event_flags()) != hipSuccess ||
       hipEventCreateWithFlags(&e->evExpected, order_event_flags()) != hipSuccess)) { free_all(e); delete e; return fail(DSR_E_DEVICE, "side stream creation failed"); }
  ALLOC(dmalloc(&e->scene.table, (size_t)e->E));
  ALLOC(dmalloc(&e->scene.vba, (size_t)e->noBlocks * kBlockBytes));
  ALLOC(dmalloc(&e->scene.voxelAllocList, (size_t)e->noBlocks));
  ALLOC(dmalloc(&e->scene.excessAllocList, (size_t)e->noExcess));
  ALLOC(dmalloc(&e->scene.ctr, (size_t)CTR_COUNT));
  ALLOC(dmalloc(&e->scene.work, (size_t)WORK_COUNT));
  ALLOC(dmalloc(&e->scene.allocKey, (size_t)e->E));
  ALLOC(dmalloc(&e->scene.allocGrp, (size_t)e->numTilesE * (kTile / 32)));
  ALLOC(dmalloc(&e->scene.allocTile, (size_t)e->numTilesE + 1));
  {
    const int cells = ((e->W + 7) / 8) * ((e->H + 7) / 8);
    e->smallPath = e->smallVolume && !s.use_swapping && e->E <= kSmallMaxEntries && e->E % 8 == 0 && e->numTilesE <= kSmallMaxTiles &&
                   small_lds_bytes(cells) <= 64 * 1024;
    if (e->smallPath) {
      ALLOC(dmalloc(&e->scene.visGrp, (size_t)kSmallBitWords * 4));
      ALLOC(dmalloc(&e->scene.visBits, (size_t)kSmallBitWords));
      ALLOC(dmalloc(&e->scene.allocBits, (size_t)kSmallBitWords));
      ALLOC(dmalloc(&e->scene.allocIds, (size_t)e->noBlocks));
      e->smallLdsBytes = small_lds_bytes(cells);
      const size_t withLists = std::max(e->smallLdsBytes, sizeof(SmallShared) + small_lists_lds_bytes(e->noBlocks));
      e->smallLists = withLists <= 64 * 1024 && !(getenv("DSR_SMALL_LISTS") && atoi(getenv("DSR_SMALL_LISTS")) == 0);
      if (e->smallLists) e->smallLdsBytes = withLists;
      e->pairRender = pair_render_default();  // (takes effect with a range-image box: dsr_prepare)
    }
  }
  ALLOC(dmalloc(&e->allocWork, (size_t)std::min((double)e->noBlocks, (double)e->P * e->maxSteps)));
  const int mw = (e->W + 7) / 8, mh = (e->H + 7) / 8;
  for (RenderStateDev *rs : {&e->live, &e->freeview}) {
    ALLOC(dmalloc(&rs->visibleIDs, (size_t)e->noBlocks));
    ALLOC(dmalloc(&rs->visBlocks, (size_t)e->noBlocks));
    ALLOC(dmalloc(&rs->visType, (size_t)e->E));
    ALLOC(dmalloc(&rs->minmax, (size_t)mw * mh));
    ALLOC(dmalloc(&rs->raycastResult, (size_t)e->P));
    ALLOC(dmalloc(&rs->raycastImage, (size_t)e->P));
    // a range image that ONE workgroup builds (expected_depths below) carries its box (env DSR_RAY_BOX=0: full-frame kernels, A/B)
    if (e->smallVolume && (size_t)mw * mh * sizeof(int2) <= 64 * 1024 && !(getenv("DSR_RAY_BOX") && atoi(getenv("DSR_RAY_BOX")) == 0))
      ALLOC(dmalloc(&rs->rayBox, (size_t)RB_WORDS));
  }
  ALLOC(dmalloc(&e->live.visibleIDsAlt, (size_t)e->noBlocks));
  ALLOC(dmalloc(&e->live.visBlocksAlt, (size_t)e->noBlocks));
  e->live.ctrIdx = CTR_NO_VISIBLE_LIVE; e->freeview.ctrIdx = CTR_NO_VISIBLE_FREE;
  ALLOC(dmalloc(&e->tileSums, (size_t)e->numTilesMax + 1));
  e->integrateStatsCount = (size_t)e->gridIntegrate * kIntegrateWaves;
  ALLOC(dmalloc(&e->integrateStats, e->integrateStatsCount));
  (void)hipMemsetAsync(e->integrateStats, 0, e->integrateStatsCount * sizeof(uint2), e->stream);
  ALLOC(dmalloc(&e->rgb, (size_t)e->Wr * e->Hr));
  ALLOC(dmalloc(&e->depth, (size_t)e->P));
  ALLOC(dmalloc(&e->depthTmp, (size_t)e->P));
  ALLOC(dmalloc(&e->rawDepth, (size_t)e->P + 4));
  ALLOC(dmalloc(&e->pointsMap, (size_t)e->P));
  ALLOC(dmalloc(&e->normalsMap, (size_t)e->P));
  ALLOC(dmalloc(&e->freeDepth, (size_t)e->P));
  ALLOC(dmalloc(&e->decayFlags, (size_t)e->noBlocks));
  ALLOC(dmalloc(&e->decayCand, (size_t)e->noBlocks));
  ALLOC(dmalloc(&e->allocList, (size_t)e->noBlocks));
  if (s.use_swapping) {
    ALLOC(dmalloc(&e->scene.swapState, (size_t)e->E));
    ALLOC(dmalloc(&e->scene.swapStored, (size_t)e->E));
    ALLOC(dmalloc(&e->swapStagingDev, (size_t)kTransferBlocks * kBlockBytes));
    ALLOC(dmalloc(&e->swapIdsDev, (size_t)kTransferBlocks));
    ALLOC(dmalloc(&e->swapFlagsDev, (size_t)kTransferBlocks));
    ALLOC(dmalloc(&e->scene.swapSlot, (size_t)e->E));
    ALLOC(dmalloc(&e->scene.hostSlabs, (size_t)dsr_engine::kMaxHostSlabs));
    if (hipHostMalloc(reinterpret_cast<void **>(&e->hostUsedSeen), 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&e->hostUsedEvent, hipEventDisableTiming) != hipSuccess) {
      free_all(e); delete e; return fail(DSR_E_NOMEM, "host store bookkeeping allocation failed");
    }
    e->scene.slabBlocks = kSlabBlocksDefault;
    if (const char *sb = getenv("DSR_SLAB_BLOCKS")) e->scene.slabBlocks = std::max(1, atoi(sb));  // tests: force slab growth
    ALLOC(add_host_slab(e));  // the first slab, so that the first frames never wait for one
  }
  // (form 1 — a view stream and a fusion stream PER ENGINE — measured in round 4: 388 frames/s at the runtime's default of 4 hardware
  //  queues against 453 without a view pipeline, 488-498 only with GPU_MAX_HW_QUEUES=16: twice the streams share the queues)
  e->pipelinedView = pvMode != 0;
  if (e->pipelinedView && pvMode >= 2 && (e->viewStream = shared_stream(g_sharedViewStream))) e->ownsViewStream = false;
  else if (e->pipelinedView && create_stream(&e->viewStream) != hipSuccess) {
    free_all(e); delete e; return fail(DSR_E_DEVICE, "view stream creation failed");
  }
  if (s.sync_status) {
    // the published status word (k_visible_write): pinned, device-mapped, coherent — a handful of bytes
    if (hipHostMalloc(reinterpret_cast<void **>(&e->statusHost), 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&e->statusDev), e->statusHost, 0) != hipSuccess) {
      if (e->statusHost) (void)hipHostFree(e->statusHost);
      e->statusHost = e->statusDev = nullptr;  // no mapped host memory here: the status is fetched with a copy as before
    } else memset(e->statusHost, 0, 64);
    if (getenv("DSR_NO_PUBLISHED_STATUS") && e->statusHost) { (void)hipHostFree(e->statusHost); e->statusHost = e->statusDev = nullptr; }
  }
  ALLOC(short_division_exact(e->stream, s.mu, &e->shortDivMuExact));
  // clear image-sized buffers once so that dumps before the first frame are defined
  for (RenderStateDev *rs : {&e->live, &e->freeview}) {
    (void)hipMemsetAsync(rs->raycastResult, 0, (size_t)e->P * 16, e->stream);
    (void)hipMemsetAsync(rs->raycastImage, 0, (size_t)e->P * 4, e->stream);
    (void)hipMemsetAsync(rs->minmax, 0, (size_t)mw * mh * 8, e->stream);
    (void)hipMemsetAsync(rs->visibleIDs, 0, (size_t)e->noBlocks * 4, e->stream);
  }
  (void)hipMemsetAsync(e->pointsMap, 0, (size_t)e->P * 16, e->stream);
  (void)hipMemsetAsync(e->normalsMap, 0, (size_t)e->P * 16, e->stream);
  ALLOC(reset_scene(e));
  if (hipStreamSynchronize(e->stream) != hipSuccess) { free_all(e); delete e; return fail(DSR_E_DEVICE, "engine initialisation failed"); }
#undef ALLOC
  if (countedDevice < 64) g_enginesOnDevice[countedDevice].fetch_add(1);
  { std::lock_guard<std::mutex> lock(g_engineMutex); g_liveEngines.push_back(e); }
  *out = e;
  return DSR_OK;
}

void dsr_engine_destroy(dsr_engine *e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> lock(g_engineMutex);
    g_liveEngines.erase(std::remove(g_liveEngines.begin(), g_liveEngines.end(), e), g_liveEngines.end());
  }
  (void)hipSetDevice(e->device);
  if (e->ownerBatch && dsri_batch_is_live(e->ownerBatch)) dsri_batch_drop_deferred(e->ownerBatch);  // (its deferred work dies with an engine of the batch)
  if (e->viewStream) (void)hipStreamSynchronize(e->viewStream);
  if (e->borrowedStream) (void)hipDeviceSynchronize();  // (its owner may have been destroyed already: the handle is not touched)
  else if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->sideStream) (void)hipStreamSynchronize(e->sideStream);
  if (e->device >= 0 && e->device < 64 && g_ioStream[e->device]) (void)hipStreamSynchronize(g_ioStream[e->device]);
  if (e->device < 64) g_enginesOnDevice[e->device].fetch_sub(1);
  free_all(e);
  delete e;
}

int dsr_reset_scene(dsr_engine *e) {
  CHECK_E(e);
  return reset_scene(e);
}

int dsr_sync(dsr_engine *e) {
  CHECK_E(e);
  if (e->viewStream) HIP_TRY(hipStreamSynchronize(e->viewStream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->sideStream) HIP_TRY(hipStreamSynchronize(e->sideStream));
  return DSR_OK;
}

int dsr_device_synchronize(void) {
  int prev = 0;
  const bool havePrev = hipGetDevice(&prev) == hipSuccess;
  const unsigned long long mask = g_devMask.load();
  int rc = DSR_OK;
  {  // the host's "everything is done" point: deferred tracking renders (paired render) are queued first — like every call that
     // names an engine, this one is made from the thread that drives the engines (dsr.h threading contract)
    std::vector<dsr_engine *> engines;
    { std::lock_guard<std::mutex> lock(g_engineMutex); engines = g_liveEngines; }
    for (dsr_engine *e : engines)
      if (e->trackRender.pending || e->ownerBatch) {
        if (hipSetDevice(e->device) != hipSuccess) continue;
        const int st = dsr_internal::engine_flush_deferred(e);
        if (st) rc = st;
      }
  }
  for (int d = 0; d < 64; d++) {
    if (!((mask >> d) & 1ull)) continue;
    hipError_t err = hipSetDevice(d);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) rc = fail(DSR_E_DEVICE, std::string("device ") + std::to_string(d) + ": " + hipGetErrorString(err));
  }
  if (havePrev) (void)hipSetDevice(prev);
  return rc;
}

int dsr_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (!free_bytes || !total_bytes) return fail(DSR_E_ARG, "null argument");
  int prev = 0;
  const bool havePrev = hipGetDevice(&prev) == hipSuccess;
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  size_t f = 0, t = 0;
  const hipError_t err = hipMemGetInfo(&f, &t);
  if (device >= 0 && havePrev) (void)hipSetDevice(prev);
  if (err != hipSuccess) return fail(DSR_E_DEVICE, std::string("hipMemGetInfo: ") + hipGetErrorString(err));
  *free_bytes = f; *total_bytes = t;
  return DSR_OK;
}

// ---- stream ordering without host synchronisation (dsr.h)

int dsr_wait_for_stream(dsr_engine *e, void *hip_stream) {
  CHECK_E_NOFLUSH(e);  // (deferred work is queued later, i.e. behind this wait as well)
  if (!e->orderEvent) HIP_TRY(hipEventCreateWithFlags(&e->orderEvent, hipEventDisableTiming));  // system scope: the other side is not ours
  HIP_TRY(hipEventRecord(e->orderEvent, (hipStream_t)hip_stream));
  HIP_TRY(hipStreamWaitEvent(e->stream, e->orderEvent, 0));
  if (e->pipelinedView) HIP_TRY(hipStreamWaitEvent(e->viewStream, e->orderEvent, 0));  // "_dev" view inputs are read there
  return DSR_OK;
}

int dsr_stream_wait_for_engine(dsr_engine *e, void *hip_stream) {
  CHECK_E(e);
  if (!e->orderEvent) HIP_TRY(hipEventCreateWithFlags(&e->orderEvent, hipEventDisableTiming));  // system scope: the other side is not ours
  HIP_TRY(hipEventRecord(e->orderEvent, e->stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, e->orderEvent, 0));
  if (e->pipelinedView && e->viewEventValid) HIP_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, e->evView, 0));
  return DSR_OK;
}

// ---- pose

int dsr_set_pose_inv_m(dsr_engine *e, const float inv_m[16]) {
  if (!e || !inv_m) return fail(DSR_E_ARG, "null");
  Mat4 im; memcpy(im.m, inv_m, sizeof im.m);
  Mat4 M;
  if (!m4_inv(im, M)) return fail(DSR_E_ARG, "singular pose");
  e->M_d = M;
  m4_inv(e->M_d, e->invM_d);
  return DSR_OK;
}
int dsr_set_pose_m(dsr_engine *e, const float m[16]) {
  if (!e || !m) return fail(DSR_E_ARG, "null");
  Mat4 M; memcpy(M.m, m, sizeof M.m);
  Mat4 inv;
  if (!m4_inv(M, inv)) return fail(DSR_E_ARG, "singular pose");
  e->M_d = M; e->invM_d = inv;
  return DSR_OK;
}
int dsr_get_pose(dsr_engine *e, float m_out[16], float inv_m_out[16]) {
  if (!e) return fail(DSR_E_ARG, "null");
  if (m_out) memcpy(m_out, e->M_d.m, sizeof e->M_d.m);
  if (inv_m_out) memcpy(inv_m_out, e->invM_d.m, sizeof e->invM_d.m);
  return DSR_OK;
}

// ---- fusion

int dsr_set_fusion_weight_params(dsr_engine *e, int depth_weighting) {
  if (!e) return fail(DSR_E_ARG, "null");
  e->depthWeighting = depth_weighting ? 1 : 0;
  return DSR_OK;
}

int dsr_allocate_scene_from_depth(dsr_engine *e) {
  CHECK_E(e);
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  int st = allocate_scene(e);
  if (st) return st;
  if (e->s.sync_status) {
    int status = DSR_OK;
    st = published_status(e, &status);
    if (st) return st;
    if (status != DSR_OK) return fail(status, status_text(status));
  }
  return DSR_OK;
}

int dsr_integrate_into_scene(dsr_engine *e) {
  CHECK_E(e);
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  return integrate_scene(e);
}

int dsr_process_frame(dsr_engine *e) {
  CHECK_E(e);
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  int st = allocate_scene(e);
  if (st) return st;
  st = integrate_scene(e);
  if (st) return st;
  if (e->s.use_swapping) {  // ITMDenseMapper::ProcessFrame: CPU -> GPU, then GPU -> CPU
    if ((st = swap_in(e))) return st;
    if ((st = swap_out(e))) return st;
  }
  e->framesProcessed++;
  if (e->s.sync_status) {
    int status = DSR_OK;
    st = published_status(e, &status);  // final after the allocation kernels: the integration is still running
    if (st) return st;
    if (status != DSR_OK) {
      // the fork throws per failing frame: clear the sticky word after reporting it
      (void)hipMemsetAsync(e->scene.ctr + CTR_STATUS, 0, 4, e->stream);
      if (e->statusHost) e->statusHost[1] = DSR_OK;  // the published copy of the word just cleared
      return fail(status, status_text(status));
    }
  }
  return DSR_OK;
}

int dsr_prepare(dsr_engine *e) {
  CHECK_E(e);
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  float proj[4]; depth_proj(e, proj);
  FrameP p = make_frame_params(e, e->M_d, e->invM_d, proj);
  RenderStateDev &rs = e->live;
  if (e->liveExp.valid && e->liveExp.version == e->listVersion && memcmp(e->liveExp.M.m, e->M_d.m, sizeof e->M_d.m) == 0 &&
      memcmp(e->liveExp.proj, proj, sizeof proj) == 0) {
    if (e->liveExp.onSide) {
      HIP_TRY(hipStreamWaitEvent(e->stream, e->evExpected, 0));  // computed under the integration (allocate_scene)
      e->sidePending = false;
    }  // (else: k_small_alloc_visible wrote it on this stream)
  } else {
    // a stale one may still be writing the image — whether or not it is still marked valid
    if (e->sidePending) { HIP_TRY(hipStreamWaitEvent(e->stream, e->evExpected, 0)); e->sidePending = false; }
    e->liveExp.valid = false;
    int st = expected_depths(e, rs, p);
    if (st) return st;
  }
  // (Round 4 measured the raycast + ICP maps on the side stream with the next frame's read-only prefix under them: the kernels
  //  overlap and the raycast pays for it, 422 -> 444 us.  Archived: profiles/r05_pruned_variants.diff, r04c_overlap_prepare_ab.log.)
  if (e->pairRender && rs.rayBox) {
    // the tracking render waits for the next call: with the preview render it goes out as one launch (dsr_engine::pairRender)
    e->trackRender.pending = true;
    e->trackRender.p = p;
    HIP_TRY(hipGetLastError());
    return DSR_OK;
  }
  {
    int st = launch_raycast(e, "raycast", p, rs);
    if (st) return st;
    if ((st = launch_icp_maps(e, p))) return st;
  }
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

int dsr_decay(dsr_engine *e, int max_weight, int min_age, int force_all_voxels) {
  CHECK_E(e);
  if (min_age < 0) return fail(DSR_E_ARG, "negative min_age");
  e->sceneVersion++;
  e->noVisibleValid = false;
  e->listVersion++;
  if (e->sidePending) { HIP_TRY(hipStreamWaitEvent(e->stream, e->evExpected, 0)); e->sidePending = false; }
  RenderStateDev &rs = e->live;
  const int32_t *cand = nullptr;
  const int32_t *nCandPtr = nullptr;
  if (force_all_voxels) {
    LAUNCH(e, "decay_candidates", k_allocated_count, dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E, e->tileSums);
    LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene, (int)SCAN_NCAND,
           e->noBlocks);
    LAUNCH(e, "decay_candidates", k_allocated_write, dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
           (const int2 *)e->tileSums, e->decayCand, e->noBlocks);
    cand = e->decayCand;
    nCandPtr = e->scene.ctr + CTR_DECAY_NCAND;
  } else {
    int st = ensure_fifo(e, std::max(min_age + 1, e->fifoLen + 1));
    if (st) return st;
    const int slot = (e->fifoHead + e->fifoLen) % e->fifoCap;
    uint32_t *plane = e->fifoPlanes + (size_t)slot * e->fifoPlaneWords;
    HIP_TRY(hipMemsetAsync(plane, 0, e->fifoPlaneWords * 4, e->stream));
    LAUNCH(e, "decay_fifo_push", k_fifo_push_bits, dim3(512), dim3(256), (const int32_t *)rs.visibleIDs,
           (const int32_t *)e->scene.ctr, plane);
    e->fifoLen++;
    if (e->fifoLen <= min_age) { HIP_TRY(hipGetLastError()); return DSR_OK; }
    // pop the oldest plane: ordered compaction of its set bits = the visible list that was pushed
    const uint8_t *oldest = reinterpret_cast<const uint8_t *>(e->fifoPlanes + (size_t)e->fifoHead * e->fifoPlaneWords);
    LAUNCH(e, "decay_candidates", k_bits_count, dim3(e->numTilesE), dim3(kTileThreads), oldest, e->E, e->tileSums);
    LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene, (int)SCAN_NCAND,
           e->noBlocks);
    LAUNCH(e, "decay_candidates", k_bits_write, dim3(e->numTilesE), dim3(kTileThreads), oldest, e->E, (const int2 *)e->tileSums,
           e->decayCand, e->noBlocks);
    cand = e->decayCand;
    nCandPtr = e->scene.ctr + CTR_DECAY_NCAND;
    e->fifoHead = (e->fifoHead + 1) % e->fifoCap;
    e->fifoLen--;
  }
  // a short dependent chain per block (entry -> weights -> the other planes) and no pipelining in the
  // kernel: many waves, few blocks each (env DSR_GRID_DECAY)
  const float muv = e->s.mu;  // the kernels' rejectedPassGate (k_integrate.h), negated
  const int zeroIsReset = (((-1.0f > muv) || (fabsf(-1.0f / muv) > 0.25f)) && e->s.max_w >= 1) ? 1 : 0;
  LAUNCH(e, "decay_blocks", k_decay_blocks, dim3(e->gridDecay), dim3(256), e->scene, cand, nCandPtr, max_weight,
         e->decayFlags, zeroIsReset);
  LAUNCH(e, "decay_count", k_flag_count, dim3(e->numTilesB), dim3(kTileThreads), (const uint8_t *)e->decayFlags, nCandPtr,
         e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesB, e->scene, (int)SCAN_DECAY, 0);
  LAUNCH(e, "decay_commit", k_decay_commit, dim3(e->numTilesB), dim3(kTileThreads), e->scene, cand, nCandPtr,
         (const uint8_t *)e->decayFlags, (const int2 *)e->tileSums, rs.visType);
  // drop freed entries from the live visible list (ordered compaction into the alternate buffer)
  LAUNCH(e, "decay_compact", k_live_keep_count, dim3(e->numTilesB), dim3(kTileThreads), (const int32_t *)rs.visibleIDs,
         (const int32_t *)e->scene.ctr, (const uint8_t *)rs.visType, e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesB, e->scene,
         (int)SCAN_COMPACT_LIVE, e->noBlocks);
  LAUNCH(e, "decay_compact", k_live_keep_write, dim3(e->numTilesB), dim3(kTileThreads), (const int32_t *)rs.visibleIDs,
         (const int32_t *)e->scene.ctr, (const uint8_t *)rs.visType, (const int2 *)e->tileSums, rs.visibleIDsAlt,
         (const int4 *)rs.visBlocks, rs.visBlocksAlt);
  std::swap(rs.visibleIDs, rs.visibleIDsAlt);
  std::swap(rs.visBlocks, rs.visBlocksAlt);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

// ---- rendering

static int render_common(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], void *rgba_out,
                         void *depth_out, bool outIsDevice) {
  // a deferred tracking render (dsr_prepare) goes out WITH a free-camera render into device buffers; before anything else
  const bool freeCamera = type >= DSR_IMAGE_FREECAMERA_SHADED && type <= DSR_IMAGE_FREECAMERA_DEPTH;
  if (!(e->trackRender.pending && freeCamera && outIsDevice && e->smallPath) || !e->hasView) {
    int st = dsr_internal::engine_flush_deferred(e);
    if (st) return st;
  } else if (e->ownerBatch && dsri_batch_is_live(e->ownerBatch)) {
    int st = dsri_batch_flush(e->ownerBatch);
    if (st) return st;
  }
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  const hipMemcpyKind kind = outIsDevice ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const size_t P = (size_t)e->P;
  switch (type) {
    case DSR_IMAGE_ORIGINAL_RGB:
      if (rgba_out) {
        int st = before_fusion(e);
        if (st) return st;
        HIP_TRY(hipMemcpyAsync(rgba_out, e->rgb, P * 4, kind, e->stream));
        if ((st = after_fusion(e))) return st;
      }
      break;
    case DSR_IMAGE_SCENERAYCAST:
      if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, e->live.raycastImage, P * 4, kind, e->stream));
      break;
    case DSR_IMAGE_FREECAMERA_SHADED:
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME:
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL:
    case DSR_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT:
    case DSR_IMAGE_FREECAMERA_DEPTH: {
      Mat4 M = e->M_d, invM;
      if (pose_m) memcpy(M.m, pose_m, sizeof M.m);
      if (!m4_inv(M, invM)) return fail(DSR_E_ARG, "singular free-camera pose");
      float proj[4]; depth_proj(e, proj);
      if (intrinsics) memcpy(proj, intrinsics, sizeof proj);
      FrameP p = make_frame_params(e, M, invM, proj);
      RenderStateDev &rs = e->freeview;
      dim3 g(div_up(e->W, 16), div_up(e->H, 16));
      const bool cached = e->fvValid && e->fvVersion == e->sceneVersion && memcmp(e->fvM.m, M.m, sizeof M.m) == 0 &&
                          memcmp(e->fvProj, proj, sizeof proj) == 0;
      if (e->trackRender.pending && cached) { int st = flush_track_render(e); if (st) return st; }
      if (!cached && e->smallPath && outIsDevice && e->trackRender.pending) {
        // PAIRED: the free-view list, then the tracking raycast and the preview raycast as ONE launch, then the ICP maps
        const int cells = ((e->W + 7) / 8) * ((e->H + 7) / 8);
        {
          ProfScope _ps(e, "small_freeview");
          hipLaunchKernelGGL(k_small_freeview, dim3(1), dim3(kSmallThreads), small_lds_bytes(cells), e->stream, p, e->scene, e->allocList,
                             rs.visibleIDs, rs.visBlocks, e->noBlocks, reinterpret_cast<int2 *>(rs.minmax), rs.rayBox, e->smallLists ? 1 : 0);
        }
        RenderHalfP fv;
        fv.minmax = (const float2 *)rs.minmax; fv.raycastResult = rs.raycastResult; fv.outRgba = rs.raycastImage;
        fv.outDepth = (float *)depth_out; fv.outRgba2 = (uchar4 *)rgba_out; fv.rb = rs.rayBox; fv.type = type;
        e->trackRender.pending = false;
        LAUNCH(e, "raycast_pair", k_raycast_pair, dim3(g.x, g.y, 2), dim3(256), e->trackRender.p, p, e->scene,
               (const float2 *)e->live.minmax, e->live.raycastResult, e->live.rayBox, fv);
        int st = launch_icp_maps(e, e->trackRender.p);
        if (st) return st;
        HIP_TRY(hipGetLastError());
        e->fvValid = true; e->fvVersion = e->sceneVersion; e->fvM = M; memcpy(e->fvProj, proj, sizeof proj);
        break;
      }
      if (!cached && e->smallPath) {
        // FindVisibleBlocks + CreateExpectedDepths in ONE workgroup (k_small.h), then the raycast that shades its own pixels
        const int cells = ((e->W + 7) / 8) * ((e->H + 7) / 8);
        {
          ProfScope _ps(e, "small_freeview");
          hipLaunchKernelGGL(k_small_freeview, dim3(1), dim3(kSmallThreads), small_lds_bytes(cells), e->stream, p, e->scene, e->allocList,
                             rs.visibleIDs, rs.visBlocks, e->noBlocks, reinterpret_cast<int2 *>(rs.minmax), rs.rayBox, e->smallLists ? 1 : 0);
        }
        e->fvValid = true; e->fvVersion = e->sceneVersion; e->fvM = M; memcpy(e->fvProj, proj, sizeof proj);
        if (outIsDevice) {
          LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
                 rs.raycastResult, type, rs.raycastImage, (float *)depth_out, (uchar4 *)rgba_out, rs.rayBox);
          HIP_TRY(hipGetLastError());
          break;
        }
        LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
               rs.raycastResult, type, rs.raycastImage, depth_out ? e->freeDepth : (float *)nullptr, (uchar4 *)nullptr, rs.rayBox);
        HIP_TRY(hipGetLastError());
        if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rs.raycastImage, P * 4, kind, e->stream));
        if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, e->freeDepth, P * 4, kind, e->stream));
        break;
      } else if (!cached && e->smallVolume) {
        // FindVisibleBlocks by ONE sweep over the table (frustum test inside) + ordered compaction: 3 launches where the
        // cached list of allocated entries below takes 7 — that list pays when a large, unchanged map is rendered from
        // several cameras; an instance volume changes every frame and its table sweep is a few microseconds
        LAUNCH(e, "freeview_visible", k_visible_count<true>, dim3(e->numTilesE), dim3(kTileThreads), p, e->scene, rs.visType, e->tileSums);
        LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene,
               (int)SCAN_VISIBLE_FREE, e->noBlocks);
        LAUNCH(e, "freeview_visible", k_visible_write, dim3(e->numTilesE), dim3(kTileThreads), e->E, (const uint8_t *)rs.visType,
               (const int2 *)e->tileSums, rs.visibleIDs, e->noBlocks, e->scene, 0, rs.visBlocks, (int32_t *)nullptr, 0);
        int st = expected_depths(e, rs, p);
        if (st) return st;
        e->fvValid = true; e->fvVersion = e->sceneVersion; e->fvM = M; memcpy(e->fvProj, proj, sizeof proj);
        // the raycast shades its own pixels (k_raycast_render): one launch less in an instance volume's frame
        if (outIsDevice) {
          LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
                 rs.raycastResult, type, rs.raycastImage, (float *)depth_out, (uchar4 *)rgba_out, rs.rayBox);
          HIP_TRY(hipGetLastError());
          break;
        }
        LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
               rs.raycastResult, type, rs.raycastImage, depth_out ? e->freeDepth : (float *)nullptr, (uchar4 *)nullptr, rs.rayBox);
        HIP_TRY(hipGetLastError());
        if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rs.raycastImage, P * 4, kind, e->stream));
        if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, e->freeDepth, P * 4, kind, e->stream));
        break;
      } else if (!cached) {
      // FindVisibleBlocks: the allocated entries (ascending list, rebuilt when the scene has changed)
      // are tested densely against the free camera's frustum, the visible ones compacted in order
      if (e->allocListVersion != e->sceneVersion) {
        LAUNCH(e, "allocated_list", k_allocated_count, dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E, e->tileSums);
        LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene,
               (int)SCAN_ALLOCATED, e->noBlocks);
        LAUNCH(e, "allocated_list", k_allocated_write, dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
               (const int2 *)e->tileSums, e->allocList, e->noBlocks);
        e->allocListVersion = e->sceneVersion;
      }
      const int32_t *nAlloc = e->scene.ctr + CTR_NO_ALLOCATED;
      LAUNCH(e, "freeview_visible", k_freeview_test, dim3(4096), dim3(256), p, e->scene, (const int32_t *)e->allocList, nAlloc,
             e->decayFlags);
      LAUNCH(e, "freeview_visible", k_flag_count, dim3(e->numTilesB), dim3(kTileThreads), (const uint8_t *)e->decayFlags, nAlloc,
             e->tileSums);
      LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesB, e->scene,
             (int)SCAN_VISIBLE_FREE, e->noBlocks);
      LAUNCH(e, "freeview_visible", k_flag_write, dim3(e->numTilesB), dim3(kTileThreads), (const int32_t *)e->allocList,
             (const uint8_t *)e->decayFlags, nAlloc, (const int2 *)e->tileSums, rs.visibleIDs, e->noBlocks,
             (const dsr_hash_entry *)e->scene.table, rs.visBlocks);
      int st = expected_depths(e, rs, p);
      if (st) return st;
      launch_raycast(e, "raycast_freeview", p, rs);
      e->fvValid = true; e->fvVersion = e->sceneVersion; e->fvM = M; memcpy(e->fvProj, proj, sizeof proj);
      }
      if (outIsDevice) {  // the shading writes the caller's HBM buffers itself: no copy launches behind it
        LAUNCH(e, "render", k_render, g, dim3(256), p, e->scene, type, (const float4 *)rs.raycastResult, rs.raycastImage,
               (float *)depth_out, (uchar4 *)rgba_out);
        HIP_TRY(hipGetLastError());
        break;
      }
      LAUNCH(e, "render", k_render, g, dim3(256), p, e->scene, type, (const float4 *)rs.raycastResult, rs.raycastImage,
             depth_out ? e->freeDepth : (float *)nullptr, (uchar4 *)nullptr);
      HIP_TRY(hipGetLastError());
      if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, rs.raycastImage, P * 4, kind, e->stream));
      if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, e->freeDepth, P * 4, kind, e->stream));
      break;
    }
    default: return fail(DSR_E_ARG, "unsupported image type");
  }
  if (!outIsDevice) HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

}  // extern "C"
int dsr_internal::engine_render(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], void *rgba_out,
                                void *depth_out, bool outIsDevice) {
  return render_common(e, type, pose_m, intrinsics, rgba_out, depth_out, outIsDevice);
}
extern "C" {

int dsr_get_image(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], uint8_t *rgba_out,
                  float *depth_out) {
  CHECK_E_NOFLUSH(e);  // (render_common queues or consumes the deferred tracking render itself)
  return render_common(e, type, pose_m, intrinsics, rgba_out, depth_out, false);
}

int dsr_get_image_dev(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], void *rgba_out_dev,
                      void *depth_out_dev) {
  CHECK_E_NOFLUSH(e);
  return render_common(e, type, pose_m, intrinsics, rgba_out_dev, depth_out_dev, true);
}

// One volume per GPU next to the engine that holds the full frame: `e` gives up its own stream and queues its work on `owner`'s —
// the view split, the fusion and the renders of the pair are then ordered by ONE queue, with no cross-stream event in the frame
// (an unsatisfied cross-queue dependency costs tens of microseconds each time the host runs ahead: DESIGN.md 6.5).  Both engines
// must live on one GPU and be driven from one thread; `e` must be idle.
int dsr_engine_share_stream(dsr_engine *e, dsr_engine *owner) {
  CHECK_E(e);
  if (!owner || owner == e) return fail(DSR_E_ARG, "bad stream owner");
  if (owner->device != e->device) return fail(DSR_E_ARG, "engines on different GPUs cannot share a stream");
  if (e->pipelinedView || owner->pipelinedView) return fail(DSR_E_ARG, "not with pipelined views (they have streams of their own)");
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->sideStream) HIP_TRY(hipStreamSynchronize(e->sideStream));
  if (e->ownsStream && e->stream) (void)hipStreamDestroy(e->stream);
  e->stream = owner->stream;
  e->ownsStream = false;
  e->borrowedStream = true;
  e->overlapExpected = false;  // (the side stream's events assume a stream of the engine's own)
  e->liveExp.valid = false;
  return DSR_OK;
}

// ---- the instance volumes of one GPU as a batch (include/dsr.h "volume batch", k_batch.h) ----------------------------------

struct dsr_batch {
  dsr_engine *source = nullptr;
  int device = 0;  // (kept apart from `source`: the engines may be gone by the time the batch is destroyed)
  std::vector<dsr_engine *> vols;
  std::vector<BatchVolP> volsHost;
  BatchVolP *volsDev = nullptr;  // (the per-call records travel as kernel arguments: k_batch.h BatchFrames)
  // the paired render of the batch (dsr_engine::pairRender): dsr_batch_fuse defers the tracking render of its volumes; the next
  // dsr_batch_render sends it out with the preview raycasts as one launch, any other call on an engine of the batch queues it
  bool pair = false, pendingTrack = false;
  BatchFrames liveFrames;
  BatchFrameP *freeTableDev = nullptr;
};
namespace {
std::mutex g_batchMutex;
std::vector<dsr_batch *> g_liveBatches;  // (an engine's ownerBatch may outlive the batch)
}
bool dsri_batch_is_live(dsr_batch *b) {
  std::lock_guard<std::mutex> lock(g_batchMutex);
  return std::find(g_liveBatches.begin(), g_liveBatches.end(), b) != g_liveBatches.end();
}
void dsri_batch_drop_deferred(dsr_batch *b) { b->pendingTrack = false; }
int dsri_batch_flush(dsr_batch *b) {
  if (!b->pendingTrack) return DSR_OK;
  b->pendingTrack = false;
  dsr_engine *src = b->source;
  const int nv = (int)b->vols.size();
  const dim3 img(div_up(src->W, 16), div_up(src->H, 16), nv);
  LAUNCH(src, "batch_raycast", k_batch_raycast, img, dim3(256), b->liveFrames, (const BatchVolP *)b->volsDev);
  LAUNCH(src, "batch_icp_maps", k_batch_icp_maps, img, dim3(256), b->liveFrames, (const BatchVolP *)b->volsDev);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

static BatchVolP batch_vol_record(dsr_engine *e) {
  BatchVolP v;
  memset(&v, 0, sizeof v);
  v.s = e->scene; v.depth = e->depth; v.rgb = e->rgb; v.visType = e->live.visType; v.workList = e->allocWork;
  v.visibleIDs = e->live.visibleIDs; v.visBlocks = e->live.visBlocks; v.minmax = reinterpret_cast<int2 *>(e->live.minmax);
  v.raycastResult = e->live.raycastResult; v.raycastImage = e->live.raycastImage; v.pointsMap = e->pointsMap; v.normalsMap = e->normalsMap;
  v.integrateStats = e->integrateStats;
  v.fvVisibleIDs = e->freeview.visibleIDs; v.fvVisBlocks = e->freeview.visBlocks; v.fvMinmax = reinterpret_cast<int2 *>(e->freeview.minmax);
  v.fvRaycastResult = e->freeview.raycastResult; v.fvRaycastImage = e->freeview.raycastImage; v.allocList = e->allocList;
  v.statusDev = e->statusDev;
  v.rayBox = e->live.rayBox; v.fvRayBox = e->freeview.rayBox;
  v.numTiles = e->numTilesE; v.noBlocks = e->noBlocks; v.gridIntegrate = e->gridIntegrate; v.lists = e->smallLists ? 1 : 0;
  return v;
}
// the voxel GC swaps a volume's list buffers (dsr_decay): bring the device records up to date before they are used
static int batch_refresh(dsr_batch *b) {
  for (size_t k = 0; k < b->vols.size(); ++k) {
    const BatchVolP v = batch_vol_record(b->vols[k]);
    if (memcmp(&v, &b->volsHost[k], sizeof v) == 0) continue;
    b->volsHost[k] = v;
    HIP_TRY(hipMemcpyAsync(b->volsDev + k, &b->volsHost[k], sizeof v, hipMemcpyHostToDevice, b->source->stream));
    HIP_TRY(hipStreamSynchronize(b->source->stream));  // (rare; the source of the copy is pageable)
  }
  return DSR_OK;
}
int dsr_batch_create(dsr_engine *source, dsr_engine *const *volumes, int n_volumes, dsr_batch **out) {
  CHECK_E(source);
  if (!volumes || n_volumes <= 0 || n_volumes > kBatchMax || !out) return fail(DSR_E_ARG, "a batch holds 1..8 volumes");
  if (source->pipelinedView) return fail(DSR_E_ARG, "not with pipelined views");
  for (int k = 0; k < n_volumes; ++k) {
    dsr_engine *e = volumes[k];
    if (!e || e == source) return fail(DSR_E_ARG, "bad volume");
    for (int j = 0; j < k; ++j) if (volumes[j] == e) return fail(DSR_E_ARG, "a volume is listed twice");
    if (!e->smallPath) return fail(DSR_E_ARG, "a batch takes instance-sized volumes (k_small.h path) only");
    if (e->device != source->device || e->pipelinedView) return fail(DSR_E_ARG, "the volumes of a batch live on the source engine's GPU, without pipelined views");
    if (e->W != source->W || e->H != source->H || e->Wr != source->Wr || e->Hr != source->Hr || e->W != e->Wr || e->H != e->Hr)
      return fail(DSR_E_ARG, "main and instance engines must share the image size");
    if (e->s.stop_integrating_at_max_w != volumes[0]->s.stop_integrating_at_max_w || e->shortDivMuExact != volumes[0]->shortDivMuExact)
      return fail(DSR_E_ARG, "the volumes of a batch must share their fusion parameters");
  }
  dsr_batch *b = new (std::nothrow) dsr_batch();
  if (!b) return fail(DSR_E_NOMEM, "oom");
  b->source = source;
  b->device = source->device;
  for (int k = 0; k < n_volumes; ++k) {
    dsr_engine *e = volumes[k];
    if (e->stream != source->stream) {  // one queue for the whole batch
      int st = dsr_engine_share_stream(e, source);
      if (st) { delete b; return st; }
    }
    b->vols.push_back(e);
    b->volsHost.push_back(batch_vol_record(e));
  }
  b->pair = pair_render_default();
  for (int k = 0; k < n_volumes; ++k) b->pair = b->pair && volumes[k]->live.rayBox && volumes[k]->freeview.rayBox;
  if (hipMalloc(reinterpret_cast<void **>(&b->volsDev), sizeof(BatchVolP) * kBatchMax) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&b->freeTableDev), sizeof(BatchFrameP) * kBatchMax) != hipSuccess ||
      hipMemcpy(b->volsDev, b->volsHost.data(), sizeof(BatchVolP) * b->vols.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(b->freeTableDev, 0, sizeof(BatchFrameP) * kBatchMax) != hipSuccess) {
    if (b->volsDev) (void)hipFree(b->volsDev);
    if (b->freeTableDev) (void)hipFree(b->freeTableDev);
    delete b;
    return fail(DSR_E_NOMEM, "batch tables");
  }
  // calls on any engine of the batch queue the batch's deferred work first (CHECK_E)
  source->ownerBatch = b;
  for (dsr_engine *e : b->vols) { e->ownerBatch = b; e->pairRender = false; }  // (rendered through the batch: its own deferral)
  { std::lock_guard<std::mutex> lock(g_batchMutex); g_liveBatches.push_back(b); }
  *out = b;
  return DSR_OK;
}

void dsr_batch_destroy(dsr_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  (void)hipDeviceSynchronize();
  {  // (deferred work is dropped: the engines may be gone, and nobody can ask for its results through this batch any more)
    std::lock_guard<std::mutex> lock(g_batchMutex);
    g_liveBatches.erase(std::remove(g_liveBatches.begin(), g_liveBatches.end(), b), g_liveBatches.end());
  }
  (void)hipFree(b->volsDev);
  (void)hipFree(b->freeTableDev);
  delete b;
}

// InstanceReconstructor::ProcessFrame for the instances of this GPU (InstanceReconstructor.cpp:238-263,569-700): per item, in the
// host's order, ProcessSilhouette + RemoveSilhouette, SetPose, Integrate, PrepareNextStep — as 2 + 6 launches for ALL of them.
int dsr_batch_fuse(dsr_batch *b, const dsr_batch_item *items, int n_items, int32_t *status_out) {
  if (!b || !items || n_items <= 0) return fail(DSR_E_ARG, "bad batch arguments");
  dsr_engine *src = b->source;
  CHECK_E(src);
  if (!src->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  const int nv = (int)b->vols.size();
  std::vector<int> itemOf(nv, -1);
  bool anyBlank = false;
  for (int i = 0; i < n_items; ++i) {
    const dsr_batch_item &it = items[i];
    if (it.volume >= nv || it.volume < -1) return fail(DSR_E_ARG, "bad batch volume index");
    if (it.volume >= 0) {
      if (itemOf[it.volume] >= 0) return fail(DSR_E_ARG, "a volume appears twice in one frame");
      if (!it.copy_mask_dev || it.box_w <= 0 || it.box_h <= 0) return fail(DSR_E_ARG, "bad silhouette arguments");
      itemOf[it.volume] = i;
    }
    if (it.delete_mask_dev) { if (it.dbox_w <= 0 || it.dbox_h <= 0) return fail(DSR_E_ARG, "bad silhouette arguments"); anyBlank = true; }
    if (it.volume >= 0) {  // every pose is checked BEFORE anything is queued or any engine's bookkeeping changes
      Mat4 invM, M;
      memcpy(invM.m, it.inv_m, sizeof invM.m);
      if (!m4_inv(invM, M)) return fail(DSR_E_ARG, "singular pose");
    }
  }
  int st = batch_refresh(b);
  if (st) return st;
  hipStream_t S = src->stream;
  // ---- the view split: every cut-out and every blanking in ceil(n / 8) passes over the frame
  if (anyBlank && (st = begin_view_modify(src))) return st;
  for (int v = 0; v < nv; ++v)
    if (itemOf[v] >= 0 && (st = before_view_write(b->vols[v], S))) return st;
  for (int first = 0; first < n_items; first += kBatchMax) {
    BatchSplit sp;
    memset(&sp, 0, sizeof sp);
    const int n = std::min(kBatchMax, n_items - first);
    for (int k = 0; k < n; ++k) {
      const dsr_batch_item &it = items[first + k];
      BatchSplitItem &o = sp.it[k];
      o.rmask = (const uint8_t *)it.delete_mask_dev; o.rx0 = it.dx0; o.ry0 = it.dy0; o.rbw = it.dbox_w; o.rbh = it.dbox_h;
      if (it.volume >= 0) {
        dsr_engine *e = b->vols[it.volume];
        o.mask = (const uint8_t *)it.copy_mask_dev; o.x0 = it.x0; o.y0 = it.y0; o.bw = it.box_w; o.bh = it.box_h;
        o.dstRgb = e->rgb; o.dstDepth = e->depth;
        int wr[4];
        cutout_write_region(e, true, it.x0, it.y0, it.box_w, it.box_h, wr);  // (batch volumes: one view buffer each, written here)
        o.wx0 = wr[0]; o.wy0 = wr[1]; o.wx1 = wr[2]; o.wy1 = wr[3];
        cutout_written(e, true, it.x0, it.y0, it.box_w, it.box_h);
      }
    }
    LAUNCH(src, "batch_split", k_batch_split, dim3(div_up(src->W, 16), div_up(src->H, 16)), dim3(256), src->rgb, src->depth, src->W, src->H, sp, n);
  }
  HIP_TRY(hipGetLastError());
  if (anyBlank && (st = view_written(src, S))) return st;
  // ---- per volume: pose, this frame's parameters, the bookkeeping of allocate_scene / integrate_scene / dsr_prepare
  BatchFrames frames;
  memset(&frames, 0, sizeof frames);
  BatchFrameP *fr = frames.f;
  int maxTilesX = 0, maxTilesY = 0, maxGrid = 0, rgbSame = -1, plain = -1;
  for (int v = 0; v < nv; ++v) {
    if (itemOf[v] < 0) continue;
    dsr_engine *e = b->vols[v];
    const dsr_batch_item &it = items[itemOf[v]];
    if ((st = view_written(e, S))) return st;
    e->viewBox[0] = std::max(0, it.x0); e->viewBox[1] = std::max(0, it.y0);
    e->viewBox[2] = std::min(e->W, it.x0 + it.box_w); e->viewBox[3] = std::min(e->H, it.y0 + it.box_h);
    if ((st = dsr_set_pose_inv_m(e, it.inv_m))) return st;
    float proj[4]; depth_proj(e, proj);
    BatchFrameP &f = fr[v];
    f.p = make_frame_params(e, e->M_d, e->invM_d, proj);
    f.active = 1;
    f.tileX0 = e->viewBox[0] / 16; f.tileY0 = e->viewBox[1] / 16;
    f.tilesX = std::max(0, div_up(e->viewBox[2], 16) - f.tileX0); f.tilesY = std::max(0, div_up(e->viewBox[3], 16) - f.tileY0);
    maxTilesX = std::max(maxTilesX, f.tilesX); maxTilesY = std::max(maxTilesY, f.tilesY); maxGrid = std::max(maxGrid, e->gridIntegrate);
    // k_integrate's specialisations (colour camera == depth camera; no weighting options + exact short division): the batch takes
    // the one EVERY active volume qualifies for — the general forms compute the same values (a pose with a -0 entry already makes
    // calib_inv * M differ from M in a sign of zero, and a single volume then runs the general kernel too)
    const int rs = f.p.rgbSame ? 1 : 0, pl = (!f.p.depthWeighting && !f.p.stopAtMaxW && e->shortDivMuExact) ? 1 : 0;
    rgbSame = rgbSame < 0 ? rs : (rgbSame & rs); plain = plain < 0 ? pl : (plain & pl);
    if (e->statusDev) e->statusSeq++;
    f.publishSeq = e->statusSeq;
    e->sceneVersion += 2; e->noVisibleValid = false; e->listVersion++; e->framesProcessed++;
    e->liveExp.valid = true; e->liveExp.onSide = false; e->liveExp.version = e->listVersion; e->liveExp.M = e->M_d;
    memcpy(e->liveExp.proj, proj, sizeof proj);
  }
  if (rgbSame < 0) {  // only blanking in this frame
    if (status_out) for (int i = 0; i < n_items; ++i) status_out[i] = DSR_OK;
    return DSR_OK;
  }
  const dim3 img(div_up(src->W, 16), div_up(src->H, 16), nv);
  if (maxTilesX > 0 && maxTilesY > 0)
    LAUNCH(src, "batch_alloc_mark", k_batch_alloc_mark, dim3(maxTilesX, maxTilesY, nv), dim3(256), frames, (const BatchVolP *)b->volsDev);
  const int cells = ((src->W + 7) / 8) * ((src->H + 7) / 8);
  {
    ProfScope _ps(src, "batch_small_alloc_visible");
    size_t lds = small_lds_bytes(cells);
    for (dsr_engine *e : b->vols) lds = std::max(lds, e->smallLdsBytes);
    hipLaunchKernelGGL(k_batch_small_alloc_visible, dim3(nv), dim3(kSmallThreads), lds, S, frames,
                       (const BatchVolP *)b->volsDev);
  }
#define BATCH_INTEGRATE(A, B) LAUNCH(src, "batch_integrate", (k_batch_integrate<A, B>), dim3(maxGrid, nv), dim3(256), frames, (const BatchVolP *)b->volsDev)
  if (rgbSame) { if (plain) BATCH_INTEGRATE(true, true); else BATCH_INTEGRATE(true, false); }
  else { if (plain) BATCH_INTEGRATE(false, true); else BATCH_INTEGRATE(false, false); }
#undef BATCH_INTEGRATE
  if (b->pair) {  // the tracking render goes out with the preview raycasts of the next dsr_batch_render, or with the next other call
    b->pendingTrack = true;
    b->liveFrames = frames;
  } else {
    LAUNCH(src, "batch_raycast", k_batch_raycast, img, dim3(256), frames, (const BatchVolP *)b->volsDev);
    LAUNCH(src, "batch_icp_maps", k_batch_icp_maps, img, dim3(256), frames, (const BatchVolP *)b->volsDev);
  }
  HIP_TRY(hipGetLastError());
  if (status_out) {
    for (int i = 0; i < n_items; ++i) status_out[i] = DSR_OK;
    for (int v = 0; v < nv; ++v) {
      if (itemOf[v] < 0 || !b->vols[v]->s.sync_status) continue;
      int status = DSR_OK;
      if ((st = published_status(b->vols[v], &status))) return st;
      status_out[itemOf[v]] = status;
      if (status != DSR_OK) {  // the fork throws per failing frame: clear the sticky word after reporting it (dsr_process_frame)
        (void)hipMemsetAsync(b->vols[v]->scene.ctr + CTR_STATUS, 0, 4, S);
        if (b->vols[v]->statusHost) b->vols[v]->statusHost[1] = DSR_OK;
      }
    }
  }
  return DSR_OK;
}

// GetImage + GetFloatImage of every listed volume from its own free camera (CompositeInstances, InstanceReconstructor.cpp:956-986):
// two launches for all of them, straight into the caller's HBM buffers (exchange slots)
int dsr_batch_render(dsr_batch *b, int type, const dsr_batch_render_item *items, int n_items) {
  if (!b || !items || n_items <= 0) return fail(DSR_E_ARG, "bad batch arguments");
  dsr_engine *src = b->source;
  CHECK_E_NOFLUSH(src);  // (the batch's deferred tracking render is consumed below)
  { int st0 = flush_track_render(src); if (st0) return st0; }
  if (type < DSR_IMAGE_FREECAMERA_SHADED || type > DSR_IMAGE_FREECAMERA_DEPTH) return fail(DSR_E_ARG, "unsupported image type");
  const int nv = (int)b->vols.size();
  int st = batch_refresh(b);
  if (st) return st;
  BatchFrames frames;
  memset(&frames, 0, sizeof frames);
  BatchFrameP *fr = frames.f;
  bool any = false;
  for (int i = 0; i < n_items; ++i) {
    const dsr_batch_render_item &it = items[i];
    if (it.volume < 0 || it.volume >= nv || fr[it.volume].active) return fail(DSR_E_ARG, "bad batch volume index");
    dsr_engine *e = b->vols[it.volume];
    if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
    Mat4 M, invM;
    memcpy(M.m, it.pose_m, sizeof M.m);
    if (!m4_inv(M, invM)) return fail(DSR_E_ARG, "singular free-camera pose");
    float proj[4]; depth_proj(e, proj);
    BatchFrameP &f = fr[it.volume];
    f.p = make_frame_params(e, M, invM, proj);
    f.active = 1; f.type = type;
    f.outRgba = (uchar4 *)it.rgba_out_dev; f.outDepth = (float *)it.depth_out_dev;
    e->fvValid = true; e->fvVersion = e->sceneVersion; e->fvM = M; memcpy(e->fvProj, proj, sizeof proj);
    any = true;
  }
  if (!any) return DSR_OK;
  const int cells = ((src->W + 7) / 8) * ((src->H + 7) / 8);
  if (b->pendingTrack) {
    // PAIRED: the free-view lists, then the tracking raycasts and the preview raycasts of every volume as ONE launch, then the
    // ICP maps (k_batch.h k_batch_raycast_pair)
    {
      ProfScope _ps(src, "batch_small_freeview");
      hipLaunchKernelGGL(k_batch_small_freeview, dim3(nv), dim3(kSmallThreads), small_lds_bytes(cells), src->stream, frames,
                         (const BatchVolP *)b->volsDev, b->freeTableDev);
    }
    b->pendingTrack = false;
    LAUNCH(src, "batch_raycast_pair", k_batch_raycast_pair, dim3(div_up(src->W, 16), div_up(src->H, 16), 2 * nv), dim3(256),
           b->liveFrames, (const BatchFrameP *)b->freeTableDev, (const BatchVolP *)b->volsDev, nv);
    LAUNCH(src, "batch_icp_maps", k_batch_icp_maps, dim3(div_up(src->W, 16), div_up(src->H, 16), nv), dim3(256), b->liveFrames,
           (const BatchVolP *)b->volsDev);
    HIP_TRY(hipGetLastError());
    return DSR_OK;
  }
  {
    ProfScope _ps(src, "batch_small_freeview");
    hipLaunchKernelGGL(k_batch_small_freeview, dim3(nv), dim3(kSmallThreads), small_lds_bytes(cells), src->stream, frames,
                       (const BatchVolP *)b->volsDev, (BatchFrameP *)nullptr);
  }
  LAUNCH(src, "batch_raycast_render", k_batch_raycast_render, dim3(div_up(src->W, 16), div_up(src->H, 16), nv), dim3(256),
         frames, (const BatchVolP *)b->volsDev);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

int dsr_dump_swap_state(dsr_engine *e, uint8_t *states, uint8_t *has_stored) {
  CHECK_E(e);
  if (!e->scene.swapState) return fail(DSR_E_ARG, "swapping is not enabled");
  if (states) HIP_TRY(hipMemcpyAsync(states, e->scene.swapState, (size_t)e->E, hipMemcpyDeviceToHost, e->stream));
  if (has_stored) HIP_TRY(hipMemcpyAsync(has_stored, e->scene.swapStored, (size_t)e->E, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

int dsr_dump_stored_block(dsr_engine *e, int entry, dsr_voxel *out, int *present) {
  CHECK_E(e);
  if (!present || entry < 0 || entry >= e->E) return fail(DSR_E_ARG, "bad entry");
  *present = 0;
  if (!e->scene.swapState) return DSR_OK;
  uint8_t flag = 0;
  HIP_TRY(hipMemcpy(&flag, e->scene.swapStored + entry, 1, hipMemcpyDeviceToHost));
  if (!flag) return DSR_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));  // the swap-out kernels write the host store asynchronously
  int32_t slot = -1;
  HIP_TRY(hipMemcpy(&slot, e->scene.swapSlot + entry, 4, hipMemcpyDeviceToHost));
  if (slot < 0 || slot >= (long long)e->hostSlabs.size() * e->scene.slabBlocks) return fail(DSR_E_DEVICE, "host store inconsistent");
  *present = 1;
  if (out) {
    const uint8_t *b = host_slot_ptr(e, slot);
    for (int v = 0; v < kBlockSize3; ++v) {
      dsr_voxel o; memset(&o, 0, sizeof o);
      memcpy(&o.sdf, b + kOffSdf + v * 2, 2);
      o.w_depth = b[kOffWDepth + v]; o.w_color = b[kOffClr + v * 4 + 3];
      o.clr[0] = b[kOffClr + v * 4]; o.clr[1] = b[kOffClr + v * 4 + 1]; o.clr[2] = b[kOffClr + v * 4 + 2];
      out[v] = o;
    }
  }
  return DSR_OK;
}

// ---- meshing

int dsr_mesh_free(dsr_engine *e) {
  CHECK_E(e);
  if (e->meshTris) { HIP_TRY(hipStreamSynchronize(e->stream)); (void)hipFree(e->meshTris); e->meshTris = nullptr; }
  e->meshCount = 0;
  return DSR_OK;
}

// ITMMeshingEngine::MeshScene (an offline dump: host synchronisation is fine here)
int dsr_mesh_scene(dsr_engine *e, uint64_t *n_triangles) {
  CHECK_E(e);
  int st = dsr_mesh_free(e);
  if (st) return st;
  // ascending list of the allocated entries (shared with Decay(forceAllVoxels))
  LAUNCH(e, "mesh_candidates", k_allocated_count, dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E, e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, e->numTilesE, e->scene, (int)SCAN_NCAND, e->noBlocks);
  LAUNCH(e, "mesh_candidates", k_allocated_write, dim3(e->numTilesE), dim3(kTileThreads), e->scene, e->E,
         (const int2 *)e->tileSums, e->decayCand, e->noBlocks);
  const int32_t *nPtr = e->scene.ctr + CTR_DECAY_NCAND;
  int n = 0;
  HIP_TRY(hipMemcpyAsync(&n, nPtr, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (n_triangles) *n_triangles = 0;
  if (n <= 0) return DSR_OK;
  uint32_t *blockCount = nullptr, *blockOffset = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&blockCount), (size_t)n * 4));
  if (hipMalloc(reinterpret_cast<void **>(&blockOffset), (size_t)n * 4) != hipSuccess) { (void)hipFree(blockCount); return fail(DSR_E_NOMEM, "mesh scratch allocation failed"); }
  MeshP mp; mp.voxelSize = e->s.voxel_size; mp.hashMask = (uint32_t)(e->noBuckets - 1); mp.noBuckets = e->noBuckets;
  const int grid = std::min(8192, div_up(n, kMeshWaves));
  const int tiles = div_up(n, kTile);
  LAUNCH(e, "mesh_count", (k_mesh_blocks<false>), dim3(grid), dim3(64 * kMeshWaves), e->scene, mp, (const int32_t *)e->decayCand, nPtr,
         blockCount, (const uint32_t *)nullptr, (dsr_triangle *)nullptr, 0ull);
  LAUNCH(e, "mesh_scan", k_u32_tile_sums, dim3(tiles), dim3(kTileThreads), (const uint32_t *)blockCount, nPtr, e->tileSums);
  LAUNCH(e, "scan_tiles", k_scan_tile_sums, dim3(1), dim3(1024), e->tileSums, tiles, e->scene, (int)SCAN_MESH, 0);
  LAUNCH(e, "mesh_scan", k_u32_tile_offsets, dim3(tiles), dim3(kTileThreads), (const uint32_t *)blockCount, nPtr,
         (const int2 *)e->tileSums, blockOffset);
  int total = 0;
  hipError_t err = hipMemcpyAsync(&total, e->scene.ctr + CTR_MESH_TOTAL, 4, hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  st = DSR_OK;
  if (err != hipSuccess) st = fail(DSR_E_DEVICE, hipGetErrorString(err));
  else if (total < 0) st = fail(DSR_E_ARG, "mesh has more than 2^31 triangles");
  // ITMMesh: noMaxTriangles = maxBlocks * 32; the append keeps the first noMaxTriangles - 1
  const unsigned long long cap = (unsigned long long)e->noBlocks * 32ull - 1ull;
  const unsigned long long keep = std::min((unsigned long long)std::max(total, 0), cap);
  if (st == DSR_OK && keep > 0) {
    if (hipMalloc(reinterpret_cast<void **>(&e->meshTris), (size_t)keep * sizeof(dsr_triangle)) != hipSuccess) {
      e->meshTris = nullptr;
      st = fail(DSR_E_NOMEM, "mesh triangle buffer allocation failed");
    } else {
      LAUNCH(e, "mesh_write", (k_mesh_blocks<true>), dim3(grid), dim3(64 * kMeshWaves), e->scene, mp, (const int32_t *)e->decayCand,
             nPtr, blockCount, (const uint32_t *)blockOffset, e->meshTris, keep);
      err = hipStreamSynchronize(e->stream);
      if (err != hipSuccess) st = fail(DSR_E_DEVICE, hipGetErrorString(err));
      else e->meshCount = keep;
    }
  }
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(blockCount); (void)hipFree(blockOffset);
  if (st == DSR_OK && n_triangles) *n_triangles = e->meshCount;
  return st;
}

int dsr_mesh_get(dsr_engine *e, dsr_triangle *out, uint64_t first, uint64_t count) {
  CHECK_E(e);
  if (!out && count) return fail(DSR_E_ARG, "null");
  if (first + count > e->meshCount) return fail(DSR_E_ARG, "triangle range outside the mesh");
  if (count) {
    HIP_TRY(hipMemcpyAsync(out, e->meshTris + first, (size_t)count * sizeof(dsr_triangle), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return DSR_OK;
}

// ITMMesh::WriteOBJ
int dsr_mesh_write_obj(dsr_engine *e, const char *path) {
  CHECK_E(e);
  if (!path) return fail(DSR_E_ARG, "null path");
  FILE *f = fopen(path, "w+");
  if (!f) return fail(DSR_E_ARG, "cannot open the OBJ file for writing");
  const uint64_t chunk = 1u << 20;
  std::vector<dsr_triangle> buf((size_t)std::min<uint64_t>(chunk, e->meshCount));
  int st = DSR_OK;
  for (uint64_t first = 0; first < e->meshCount && st == DSR_OK; first += chunk) {
    const uint64_t cnt = std::min<uint64_t>(chunk, e->meshCount - first);
    st = dsr_mesh_get(e, buf.data(), first, cnt);
    for (uint64_t i = 0; i < cnt && st == DSR_OK; ++i) {
      const dsr_triangle &t = buf[(size_t)i];
      fprintf(f, "v %f %f %f\n", t.p0[0], t.p0[1], t.p0[2]);
      fprintf(f, "v %f %f %f\n", t.p1[0], t.p1[1], t.p1[2]);
      fprintf(f, "v %f %f %f\n", t.p2[0], t.p2[1], t.p2[2]);
    }
  }
  for (uint64_t i = 0; i < e->meshCount && st == DSR_OK; i++)
    fprintf(f, "f %llu %llu %llu\n", (unsigned long long)(i * 3 + 2 + 1), (unsigned long long)(i * 3 + 1 + 1),
            (unsigned long long)(i * 3 + 0 + 1));
  fclose(f);
  return st;
}

// ITMMainEngine::SaveSceneToMesh
int dsr_save_scene_to_mesh(dsr_engine *e, const char *path) {
  int st = dsr_mesh_scene(e, nullptr);
  if (st == DSR_OK) st = dsr_mesh_write_obj(e, path);
  if (e) (void)dsr_mesh_free(e);
  return st;
}

#ifdef DSR_RAYCAST_STATS
// measurement builds only (tools/raycast_wave_stats.py): where k_raycast writes its 12 words per wave
int dsr_debug_raycast_stats(void *dev_buf) {
  unsigned int *p = (unsigned int *)dev_buf;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rcStats), &p, sizeof(p)));
  return DSR_OK;
}
#endif

#ifdef DSR_SMALL_CLOCKS
// measurement builds only (tools/small_kernel_clocks.py): where the one-workgroup kernels stamp their phase clocks
int dsr_debug_small_clocks(void *dev_buf) {
  unsigned long long *p = (unsigned long long *)dev_buf;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_smallClk), &p, sizeof(p)));
  return DSR_OK;
}
#endif

// ---- statistics / dumps

int dsr_get_stats(dsr_engine *e, dsr_stats *out) {
  CHECK_E(e);
  if (!out) return fail(DSR_E_ARG, "null");
  int32_t ctr[CTR_COUNT];
  unsigned long long work[WORK_COUNT];
  HIP_TRY(hipMemcpyAsync(ctr, e->scene.ctr, sizeof ctr, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(work, e->scene.work, sizeof work, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  memset(out, 0, sizeof *out);
  out->num_allocated_voxel_blocks = e->noBlocks;
  out->last_free_block_id = ctr[CTR_LAST_FREE_BLOCK];
  out->last_free_excess_list_id = ctr[CTR_LAST_FREE_EXCESS];
  out->no_visible_blocks = ctr[CTR_NO_VISIBLE_LIVE];
  e->noVisibleSeen = ctr[CTR_NO_VISIBLE_LIVE]; e->noVisibleValid = true;
  out->no_total_entries = e->E;
  out->voxel_bytes = (int)sizeof(dsr_voxel);
  out->block_voxels = kBlockSize3;
  out->sticky_status = ctr[CTR_STATUS];
  out->decayed_block_count = (int64_t)work[WORK_DECAYED_BLOCKS];
  out->frames_processed = e->framesProcessed;
  out->no_visible_blocks_freeview = ctr[CTR_NO_VISIBLE_FREE];
  out->host_store_slots = e->s.use_swapping ? ctr[CTR_HOST_USED] : 0;
  out->host_store_capacity_slots = (int32_t)std::min<long long>((long long)e->hostSlabs.size() * e->scene.slabBlocks, 0x7fffffff);
  return DSR_OK;
}

int dsr_get_no_visible_blocks(dsr_engine *e, int32_t *out) {
  CHECK_E(e);
  if (!out) return fail(DSR_E_ARG, "null");
  if (!e->noVisibleValid) {
    int status = 0;
    int st = sticky_status(e, &status);  // one 12-byte copy + synchronisation; the status word stays sticky
    if (st) return st;
  }
  *out = e->noVisibleSeen;
  return DSR_OK;
}

// InfiniTamDriver::PrepareNextStep's "Keep the OpenCV previews up to date" (InfiniTamDriver.h:154-156):
// ItmToCv(*view->rgb) + ItmDepthToCv(*view->depth) from the engine's device-resident view — two conversion kernels, two
// D2H copies, ONE synchronisation (the host-buffer conversions dsr_rgba_to_bgr / dsr_depth_m_to_mm cost an upload, a
// download and a synchronisation EACH)
int dsr_dump_hash_table(dsr_engine *e, dsr_hash_entry *out) {
  CHECK_E(e);
  if (!out) return fail(DSR_E_ARG, "null");
  HIP_TRY(hipMemcpyAsync(out, e->scene.table, (size_t)e->E * sizeof(dsr_hash_entry), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

int dsr_dump_visible_list(dsr_engine *e, int freeview, int32_t *ids_out, int32_t *n) {
  CHECK_E(e);
  if (!n) return fail(DSR_E_ARG, "null");
  RenderStateDev &rs = freeview ? e->freeview : e->live;
  int32_t cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, e->scene.ctr + rs.ctrIdx, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  *n = cnt;
  if (ids_out && cnt > 0) {
    HIP_TRY(hipMemcpyAsync(ids_out, rs.visibleIDs, (size_t)cnt * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return DSR_OK;
}

int dsr_dump_visible_types(dsr_engine *e, uint8_t *out) {
  CHECK_E(e);
  if (!out) return fail(DSR_E_ARG, "null");
  HIP_TRY(hipMemcpyAsync(out, e->live.visType, (size_t)e->E, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

int dsr_dump_voxel_blocks(dsr_engine *e, int first_block, int n_blocks, dsr_voxel *out) {
  CHECK_E(e);
  if (!out || first_block < 0 || n_blocks < 0 || (long long)first_block + n_blocks > e->noBlocks) return fail(DSR_E_ARG, "bad block range");
  const int chunk = 16384;  // 64 MiB of AoS voxels per pass
  if (e->aosScratchBlocks < std::min(chunk, n_blocks)) {
    if (e->aosScratch) (void)hipFree(e->aosScratch);
    e->aosScratch = nullptr;
    e->aosScratchBlocks = std::min(chunk, std::max(n_blocks, 1));
    int st = dmalloc(&e->aosScratch, (size_t)e->aosScratchBlocks * kBlockSize3);
    if (st) { e->aosScratchBlocks = 0; return st; }
  }
  for (int done = 0; done < n_blocks; done += e->aosScratchBlocks) {
    const int nb = std::min(e->aosScratchBlocks, n_blocks - done);
    LAUNCH(e, "blocks_to_aos", k_blocks_to_aos, dim3(std::min(2048, div_up(nb, 4))), dim3(256),
           (const uint8_t *)e->scene.vba, first_block + done, nb, e->aosScratch);
    HIP_TRY(hipMemcpyAsync(out + (size_t)done * kBlockSize3, e->aosScratch, (size_t)nb * kBlockSize3 * sizeof(dsr_voxel),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return DSR_OK;
}

int dsr_dump_allocation_lists(dsr_engine *e, int32_t *voxel_alloc_list, int32_t *excess_alloc_list) {
  CHECK_E(e);
  if (voxel_alloc_list) HIP_TRY(hipMemcpyAsync(voxel_alloc_list, e->scene.voxelAllocList, (size_t)e->noBlocks * 4, hipMemcpyDeviceToHost, e->stream));
  if (excess_alloc_list) HIP_TRY(hipMemcpyAsync(excess_alloc_list, e->scene.excessAllocList, (size_t)e->noExcess * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

int dsr_dump_render_state(dsr_engine *e, int which, float *minmax, float *raycast_result, float *points, float *normals,
                          uint8_t *raycast_image) {
  CHECK_E(e);
  RenderStateDev &rs = which ? e->freeview : e->live;
  const size_t P = (size_t)e->P;
  const int mw = (e->W + 7) / 8, mh = (e->H + 7) / 8;
  if (!which && e->sidePending) { HIP_TRY(hipStreamWaitEvent(e->stream, e->evExpected, 0)); e->sidePending = false; }  // a range image still in flight on the side stream
  if (minmax) HIP_TRY(hipMemcpyAsync(minmax, rs.minmax, (size_t)mw * mh * 8, hipMemcpyDeviceToHost, e->stream));
  // the far-plane start points of the misses outside the last raycast's box: never read by the path, completed for the dump
  if (raycast_result && rs.rayBox)
    hipLaunchKernelGGL(k_raycast_fill_outside, dim3(div_up(e->W, 16), div_up(e->H, 16)), dim3(256), 0, e->stream, e->scene,
                       (const int32_t *)rs.rayBox, rs.raycastResult);
  if (raycast_result) HIP_TRY(hipMemcpyAsync(raycast_result, rs.raycastResult, P * 16, hipMemcpyDeviceToHost, e->stream));
  if (points) HIP_TRY(hipMemcpyAsync(points, e->pointsMap, P * 16, hipMemcpyDeviceToHost, e->stream));
  if (normals) HIP_TRY(hipMemcpyAsync(normals, e->normalsMap, P * 16, hipMemcpyDeviceToHost, e->stream));
  if (raycast_image) HIP_TRY(hipMemcpyAsync(raycast_image, rs.raycastImage, P * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

}  // extern "C"
