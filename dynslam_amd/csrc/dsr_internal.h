// dsr_internal.h — what the translation units of libdsr_hip.so share (NOT part of the C ABI: include/dsr.h is):
//   dsr_engine.hip    the engine — creation, allocation, integration, GC, swapping, raycast / render, the volume batch, meshing, dumps
//   dsr_view.hip      the view and the edges of the path — frames in, the view pipeline, the instance view split, layout conversions,
//                     depth ingest, the two previews
//   dsr_exchange.hip  the multi-GPU layer exchange (RCCL, loaded on first use) and the compositing entry points
//   dsr_hostio.hip    host-side I/O of the boundary: precomputed depth / disparity files, page-locking of the host's buffers
//   dsr_profile.hip   HIP-event profile read-out, the division self-tests, the HBM copy probe
// Every kernel header (k_*.h) is included by exactly ONE of them: kernels have external linkage.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "dsr_device.h"

using namespace dsr;

namespace dsr_internal {
std::string &last_error();  // thread-local message behind dsr_last_error() (dsr_engine.hip)
inline int fail(int code, const std::string &msg) { last_error() = msg; return code; }
}  // namespace dsr_internal
using dsr_internal::fail;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      char _b[512];                                                                                \
      snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return fail(DSR_E_DEVICE, _b);                                                               \
    }                                                                                              \
  } while (0)

struct RenderStateDev {  // ITMRenderState_VH
  int32_t *visibleIDs = nullptr;
  int32_t *visibleIDsAlt = nullptr;  // ping-pong target of the post-decay compaction (live only)
  int4 *visBlocks = nullptr;         // the visible-block stream: one 16-byte record per list entry (dsr_device.h)
  int4 *visBlocksAlt = nullptr;
  uint8_t *visType = nullptr;
  float2 *minmax = nullptr;
  float4 *raycastResult = nullptr;
  uchar4 *raycastImage = nullptr;
  // instance-sized volumes: the box record of the range image (k_raycast.h RB_*) — the pixel kernels behind the range image skip
  // the tiles outside it; null for a map-sized volume (full-frame kernels)
  int32_t *rayBox = nullptr;
  int ctrIdx = CTR_NO_VISIBLE_LIVE;
};

struct ProfRec { std::string name; double ms = 0; long long launches = 0; };

struct dsr_engine {
  dsr_settings s;
  dsr_calib calib;
  int device = 0;
  hipStream_t stream = nullptr;
  // The range image of the live view (K6) needs the visible list and the pose — not a single voxel — so it is computed on
  // a SIDE stream while k_integrate runs (the host's Integrate(); PrepareNextStep(); pair, InfiniTamDriver.h:137-158):
  // its LDS / atomic / latency phases hide under the VALU-bound integration.  dsr_prepare takes the result when list and
  // camera are still the ones it was computed for, else it recomputes on the main stream.  env DSR_OVERLAP_EXPECTED=0: off.
  // Measured (profiles/r03j_range_image_overlap_ab.json): 1.176 vs 1.192 ms per frame.  K6's 1024-thread, 58 KB-LDS
  // workgroups only find room as integration workgroups retire, so under a profiler its SPAN is the integration's (~510 us
  // for ~40 us of work): a span, not a cost.  Tried on top: raised wave priority (s_setprio 3: no change — the waves are not
  // resident, not slow) and the global-atomics kernel, whose 256-thread workgroups do co-reside (182 us) but whose atomics
  // slow the integration to 692 us (profiles/r03m_*).
  hipStream_t sideStream = nullptr;
  hipEvent_t evList = nullptr, evExpected = nullptr;
  bool overlapExpected = true;
  unsigned long long listVersion = 0;  // bumped by every call that rewrites the live visible list
  struct { bool valid = false; bool onSide = false; unsigned long long version = 0; Mat4 M; float proj[4] = {0, 0, 0, 0}; } liveExp;
  int W = 0, H = 0, Wr = 0, Hr = 0, P = 0;
  int noBuckets = 0, noExcess = 0, E = 0, noBlocks = 0;
  int numTilesE = 0, numTilesB = 0, numTilesMax = 0;
  uint32_t maxSteps = 0;
  int gridPersistent = 2048;
  int gridDecay = 2048;
  // k_integrate grid: more, finer strided shares balance the tail (5 mm bench: 1280 workgroups
  // (= resident) 918 us, 4096 872 us, 8192 840 us, 16384 835 us, whole-block variant); scaled down
  // for small volumes.
  // env DSR_GRID_INTEGRATE overrides.
  int gridIntegrate = 8192;
  // a volume of instance size (7142 blocks in the reference, InstanceReconstructor.cpp:379): its frames are bound by the number
  // of launches, not by bandwidth, so the paths with fewer, simpler launches are taken (expected depths in one workgroup,
  // free-view visible list by one sweep instead of through the cached list of allocated entries); results are identical
  bool smallVolume = false;
  // ... and, when the table is no larger than upstream's (1 179 648 entries) and nothing is swapped: the one-workgroup kernels of
  // k_small.h — commit + visible list + range image as ONE launch, the free-view list + range image as one (21 -> 9 launches
  // per instance frame); results identical, both paths under test
  bool smallPath = false;
  // ... with the allocated entries also kept as a sorted list (SceneP::allocIds, k_small.h round 6): the visible lists are dense
  // passes over it instead of sweeps of the bit planes.  env DSR_SMALL_LISTS=0: the sweeps only (A/B; both under test)
  bool smallLists = false;
  size_t smallLdsBytes = 0;  // dynamic LDS of the one-workgroup kernels: the range image, or the merge's scratch aliased with it
  // the box (pixels, end exclusive) outside which the current view's depth is known to be 0: set by the silhouette cut-out
  // that produced an instance's view, the whole image after any other writer.  The allocation's per-pixel mark runs over it.
  int viewBox[4] = {0, 0, 0, 0};
  // ... and (round 6) the box outside which the view BUFFER is known to hold the blank constants of a cut-out (rgb 255, depth 0):
  // the next cut-out writes only its own box and this one.  Invalid after any other writer of the view.
  int blankBox[4] = {0, 0, 0, 0};
  bool blankValid = false;
  int gridExpected = 128;  // workgroups of k_expected_depth_lds (env DSR_GRID_EXPECTED; 64: 60 us, 128: 44, 256: 84)
  Mat4 calibInv, M_d, invM_d;

  SceneP scene{};
  RenderStateDev live, freeview;
  int2 *tileSums = nullptr;
  uint2 *integrateStats = nullptr;  // per wave of k_integrate: {lanes that stored depth planes, colour voxels}
  size_t integrateStatsCount = 0;   // its length (gridIntegrate * kIntegrateWaves)
  int4 *allocWork = nullptr;  // ordered work list of the frame's allocations
  // free-view cache: DynSLAM renders several image types from ONE pose per redraw (GetImage colour +
  // GetFloatImage depth, InfiniTamDriver.cpp:165-209); while neither the scene nor the camera has
  // changed, FindVisibleBlocks + CreateExpectedDepths + the raycast are reused and only the
  // shading runs again
  unsigned long long sceneVersion = 0;
  int32_t *allocList = nullptr;              // ascending list of the allocated entries, valid for allocListVersion
  unsigned long long allocListVersion = ~0ull;
  bool fvValid = false;
  unsigned long long fvVersion = 0;
  Mat4 fvM;
  float fvProj[4] = {0, 0, 0, 0};
  dsr_triangle *meshTris = nullptr;  // current mesh (dsr_mesh_scene), device
  uint64_t meshCount = 0;

  // view
  bool hasView = false;
  uchar4 *rgb = nullptr;
  float *depth = nullptr, *depthTmp = nullptr;
  short *rawDepth = nullptr;
  // tracking state point cloud
  float4 *pointsMap = nullptr, *normalsMap = nullptr;
  // scratch
  float *freeDepth = nullptr;
  dsr_voxel *aosScratch = nullptr;
  int aosScratchBlocks = 0;

  int depthWeighting = 0;
  bool shortDivMuExact = false;  // div_short(x, mu) == x / mu for every x (checked at creation)
  long long framesProcessed = 0;

  // voxel GC FIFO of visible lists
  uint32_t *fifoPlanes = nullptr;    // ring storage (device): fifoCap planes of fifoPlaneWords words, a bit per entry (k_decay.h)
  size_t fifoPlaneWords = 0;
  int fifoCap = 0, fifoHead = 0, fifoLen = 0;
  int32_t *decayCand = nullptr;      // forceAll candidate list
  // host swapping (use_swapping): ITMGlobalCache = host store of plane-wise 4 KiB blocks
  uint8_t *swapStagingDev = nullptr;             // 16 MiB: fetched host copies of a swap-in batch
  int32_t *swapIdsDev = nullptr;
  uint8_t *swapFlagsDev = nullptr;
  // host store (ITMGlobalCache): pinned slabs the GPU reads and writes directly (k_swap.h)
  std::vector<uint8_t *> hostSlabs;              // e->scene.slabBlocks blocks each; also listed in scene.hostSlabs
  static constexpr int kMaxHostSlabs = 4096;     // 256 GiB of host store
  long long hostUsedUpper = 0;                   // upper bound of CTR_HOST_USED after the enqueued frames
  int32_t *hostUsedSeen = nullptr;               // pinned: asynchronous read-back of CTR_HOST_USED
  hipEvent_t hostUsedEvent = nullptr;
  bool hostUsedPending = false;
  long long hostUsedCallsSince = 0;              // swap-out batches enqueued since that read-back was issued
  // silhouette masks handed over as HOST buffers (instance view split): a ring of pinned, device-mapped staging slots.
  // The host copies the mask into a slot and the silhouette kernel reads it from there over the host link (10-20 KB,
  // once): no copy command, no synchronisation — the caller's buffer is free when the call returns and a slot is reused
  // only once the kernel that read it has run.  (A hipMemcpyAsync from the pinned slot into a device twin was measured
  // first: the copy engine's hand-over to the compute queue costs ~40 us per mask, configs[2] 623 -> 505 frames/s.)
  static constexpr int kMaskSlots = 32;  // two masks per instance and frame: a scene of up to 16 instances never waits on a slot
  uint8_t *maskHost = nullptr, *maskHostDev = nullptr;  // the ring and its device-side address
  size_t maskSlotBytes = 0;
  hipEvent_t maskEvent[kMaskSlots] = {};
  bool maskEventUsed[kMaskSlots] = {};
  int maskNext = 0;
  // noVisibleBlocks of the live view as the host last saw it (read together with the status word: dsr_process_frame with
  // sync_status, dsr_get_stats); valid until the next call that changes the list
  int32_t noVisibleSeen = 0;
  bool noVisibleValid = false;
  // ---- host buffers in and out WITHOUT draining the engine's stream (DESIGN.md "through the host").  DynSLAM's host hands
  // every frame over as pageable host buffers and wants two previews and a status word back per frame and per driver
  // (InfiniTamDriver.cpp:211-224, InfiniTamDriver.h:137-158); waiting for the engine's stream at each of these calls exposes the
  // integration and the raycast to the host serially.  Instead: frames are copied into a pinned slot (two, alternating) and
  // uploaded on the GPU's I/O stream (one per device, shared by the engines of the process) into a landing buffer the ingest
  // kernel reads; the status words are PUBLISHED by k_visible_write into a pinned, device-mapped word the host polls; previews
  // and view read-backs run on the I/O stream after the last kernel that wrote the view (evView) — none of them waits for
  // k_integrate or k_raycast.
  // PIPELINED VIEW (opt-in: env DSR_PIPELINED_VIEW=1, see dsr_engine_create for the measurements): everything that writes or
  // modifies the view — ingest, SetView, the silhouette kernels — runs on the engine's VIEW stream, and the view is double
  // buffered: a frame's view is built in the buffer fusion is not reading, so the next frame's view split (and with it the
  // instance volumes' whole frames) proceeds while this volume's integration and raycast are still running.  Without it the
  // view kernels of frame i + 1 queue behind the raycast of frame i on the one stream, and a host that waits for an instance's
  // allocation status waits for the map's whole previous frame (configs[2] through the reference's call pattern).
  bool pipelinedView = false;
  bool ownsStream = true, ownsViewStream = true;  // false: the per-GPU shared streams (DSR_PIPELINED_VIEW=2)
  bool borrowedStream = false;                    // dsr_engine_share_stream: the stream is another engine's (which may be gone by now)
  hipStream_t viewStream = nullptr;
  uchar4 *rgbAlt = nullptr;
  float *depthAlt = nullptr;
  hipEvent_t evAltFree = nullptr;      // recorded on the fusion stream when the buffers were swapped: readers of the old view are behind it
  bool altFreeValid = false;
  hipEvent_t evFusionRead = nullptr;   // the last fusion work that read the view ...
  const float *fusionReadDepth = nullptr;  // ... and which buffer it read
  uint8_t *upPin[2] = {nullptr, nullptr};
  size_t upBytes = 0, upDepthOff = 0;
  hipEvent_t upSlotFree[2] = {nullptr, nullptr};
  bool upSlotUsed[2] = {false, false};
  int upNext = 0;
  uint8_t *upDev = nullptr;                    // landing buffer of the upload: colour, then depth
  hipEvent_t evUploaded = nullptr, evIngested = nullptr;
  bool ingestPending = false;
  hipEvent_t evView = nullptr;                 // recorded after the last kernel that wrote this engine's view
  bool viewEventValid = false;
  hipEvent_t evViewRead = nullptr;             // the I/O stream's last read of the view (previews, dsr_get_view)
  bool viewReadEver = false;
  uint8_t *pvPin = nullptr, *pvDev = nullptr;  // previews: packed BGR (3 B / pixel), then int16 millimetres — pinned pair (the conversion kernel of a small volume stores into it directly) and the HBM scratch of a map-sized volume
  size_t pvMmOff = 0;
  int32_t *statusHost = nullptr, *statusDev = nullptr;  // {noVisibleBlocks, status, sequence number}, pinned + mapped
  int statusSeq = 0;
  // cross-GPU view split (main engine on one GPU, the instance volume on another): the cut-out is produced here, then peer-copied
  uchar4 *xferRgb = nullptr;
  float *xferDepth = nullptr;
  bool sidePending = false;          // evExpected has been recorded and not been waited for by the main stream since
  hipEvent_t xEvent = nullptr;       // as instance: orders the main stream after this engine's queued work
  hipEvent_t xEvent2 = nullptr;      // as main engine: orders the instance stream after a view split
  bool xEvent2System = false;        // ... created with a system-scope release (an instance on another GPU has waited for it)
  hipEvent_t orderEvent = nullptr;   // dsr_wait_for_stream / dsr_stream_wait_for_engine
  uint8_t *decayFlags = nullptr;

  // PAIRED RENDER (round 6; instance-sized volumes).  dsr_prepare's tracking render (k_raycast_box + k_icp_maps_box) is not queued
  // at once: it is DEFERRED until the next call on this engine.  When that call is the preview render (dsr_get_image_dev /
  // dsr_exchange_render_slot) — the reference's order per instance: Integrate, PrepareNextStep, later GetImage from the GUI's
  // camera — both raycasts go out as ONE launch (k_raycast.h k_raycast_pair) and overlap on one queue; any other call
  // (CHECK_E below) queues the deferred work first, so no caller ever sees a stream without it.  What the pending launch needs is
  // kept here: the camera of the prepare call.  env DSR_PAIR_RENDER=0: off.
  bool pairRender = false;
  bool rayBoxLive = false;   // the ray-box records have been initialised once (a later reset keeps their LAST / POSE part)
  struct { bool pending = false; FrameP p; } trackRender;
  struct dsr_batch *ownerBatch = nullptr;  // the volume batch this engine is a source / volume of (it may hold deferred work too)

  // profiling
  int profiling = 0;  // 0 off, 1 every kernel, 2 the two dominant kernels only
  std::vector<ProfRec> profRecs;
  std::map<std::string, int> profIndex;
  struct Pending { int rec; hipEvent_t a, b; };
  std::vector<Pending> profPending;
  std::vector<hipEvent_t> eventPool;
};

namespace dsr_internal {
// ---- small helpers every translation unit of the engine uses
inline int set_device(dsr_engine *e) {
  HIP_TRY(hipSetDevice(e->device));
  return DSR_OK;
}
inline hipEvent_t get_event(dsr_engine *e) {
  if (!e->eventPool.empty()) { hipEvent_t ev = e->eventPool.back(); e->eventPool.pop_back(); return ev; }
  hipEvent_t ev = nullptr;
  (void)hipEventCreate(&ev);
  return ev;
}
void prof_resolve_pending(dsr_engine *e);  // dsr_engine.hip: resolve the HIP-event pairs recorded so far
struct ProfScope {  // HIP events around the launches of a scope, on e->stream (dsr_profile_*)
  dsr_engine *e; int rec = -1; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(dsr_engine *e_, const char *name) : e(e_) {
    if (!e->profiling) return;
    if (e->profiling == 2 && strcmp(name, "integrate") != 0 && strcmp(name, "raycast") != 0 && strcmp(name, "raycast_tail") != 0) return;
    auto it = e->profIndex.find(name);
    if (it == e->profIndex.end()) {
      rec = (int)e->profRecs.size();
      e->profIndex[name] = rec;
      ProfRec r; r.name = name; e->profRecs.push_back(r);
    } else rec = it->second;
    if (e->profPending.size() > 8192) prof_resolve_pending(e);
    a = get_event(e); b = get_event(e);
    (void)hipEventRecord(a, e->stream);
  }
  ~ProfScope() {
    if (rec < 0) return;
    (void)hipEventRecord(b, e->stream);
    e->profPending.push_back({rec, a, b});
  }
};
// kernels enqueued inside the scope go to `s` (LAUNCH and ProfScope read e->stream)
struct StreamSwap {
  dsr_engine *e; hipStream_t saved;
  StreamSwap(dsr_engine *e_, hipStream_t s) : e(e_), saved(e_->stream) { e->stream = s; }
  ~StreamSwap() { e->stream = saved; }
};
inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
template <class T>
int dmalloc(T **p, size_t n) {
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
  return DSR_OK;
}
// ---- dsr_view.hip: the I/O stream of a GPU, event flavours, the ordering of the view's writers and readers
extern std::mutex g_ioMutex;
extern hipStream_t g_ioStream[64];
hipError_t create_stream(hipStream_t *out);
int io_stream(dsr_engine *e, hipStream_t *out);
unsigned order_event_flags();
int make_event(hipEvent_t *ev, bool hostWaits = false);
int before_view_write(dsr_engine *e, hipStream_t stream);
int view_written(dsr_engine *e, hipStream_t stream);
int begin_view_modify(dsr_engine *e);
int before_fusion(dsr_engine *e);
int after_fusion(dsr_engine *e);
hipStream_t vstream(dsr_engine *e);
// the pixels a cut-out into `instance`'s view has to write for a mask box (x0, y0, w, h), and the bookkeeping behind it
void cutout_write_region(const dsr_engine *instance, bool direct, int x0, int y0, int w, int h, int wr[4]);
void cutout_written(dsr_engine *instance, bool direct, int x0, int y0, int w, int h);
// dsr_engine.hip
extern std::atomic<int> g_enginesOnDevice[64];  // live engines per device (range-image overlap policy, preview stores)
int engine_set_device(dsr_engine *e);
int engine_flush_deferred(dsr_engine *e);  // queue what dsr_prepare / dsr_batch_fuse deferred (paired render)
void engine_prof_resolve(dsr_engine *e);
int engine_render(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], void *rgba_out, void *depth_out,
                  bool outIsDevice);
// dsr_hostio.hip: is [p, p + bytes) inside a range the caller page-locked through dsr_pin_host_buffer?
bool host_range_pinned(const void *p, size_t bytes);
// dsr_profile.hip: div_short(x, b) == x / b for every x? (k_integrate.h; checked once per engine for its mu)
int short_division_exact(hipStream_t stream, float b, bool *exact);
template <class T>
int device_alloc(T **p, size_t n) {
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
  return DSR_OK;
}
}  // namespace dsr_internal
using dsr_internal::host_range_pinned;
using dsr_internal::short_division_exact;

// entry of (nearly) every engine call: the engine's GPU becomes current and work this engine — or the batch it belongs to — has
// deferred (paired render) is queued; CHECK_E_NOFLUSH: the few calls that consume the deferred work or cannot be affected by it
#define LAUNCH(e, name, kernel, grid, block, ...)                                \
  do {                                                                           \
    dsr_internal::ProfScope _ps((e), (name));                                    \
    hipLaunchKernelGGL(kernel, grid, block, 0, (e)->stream, __VA_ARGS__);        \
  } while (0)

#define CHECK_E_NOFLUSH(e)                                  \
  if (!(e)) return fail(DSR_E_ARG, "null engine");          \
  { int _st = dsr_internal::engine_set_device(e); if (_st) return _st; }
#define CHECK_E(e)                                          \
  CHECK_E_NOFLUSH(e)                                        \
  if ((e)->trackRender.pending || (e)->ownerBatch) { int _st = dsr_internal::engine_flush_deferred(e); if (_st) return _st; }
