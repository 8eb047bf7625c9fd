// dsr_engine.hip — host side of the HIP engine: implements the C ABI of include/dsr.h.
//
// One dsr_engine = one ITMMainEngine (InfiniTamDriver.h:79): scene (hash table, excess list,
// voxel block array), two render states, a view and the tracking pose — all resident in HBM.
// Every call enqueues kernels on the engine's own HIP stream; the hot path
// (update_view_dev / process_frame / prepare / decay) never synchronises: list lengths and
// free-list heads stay in device memory and kernels use fixed grids that read them there.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: librccl is loaded on first use (exchange_* below), the library does not link it

#include <dlfcn.h>
#include <sched.h>
#include <unistd.h>

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
#include "k_alloc.h"
#include "k_composite.h"
#include "k_decay.h"
#include "k_edges.h"
#include "k_integrate.h"
#include "k_raycast.h"
#include "k_swap.h"
#include "k_mesh.h"
#include "k_small.h"

using namespace dsr;

namespace {

thread_local std::string g_err;
std::atomic<unsigned long long> g_devMask{0};  // devices engines were created on (dsr_device_synchronize)
std::atomic<int> g_enginesOnDevice[64];         // live engines per device (range-image overlap policy, allocate_scene)
int fail(int code, const std::string &msg) { g_err = msg; return code; }
std::mutex g_pinMutex;
std::map<uintptr_t, size_t> g_pinned;           // host ranges the caller page-locked through dsr_pin_host_buffer
bool host_range_pinned(const void *p, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_pinMutex);
  if (g_pinned.empty()) return false;
  auto it = g_pinned.upper_bound((uintptr_t)p);
  if (it == g_pinned.begin()) return false;
  --it;
  return (uintptr_t)p + bytes <= it->first + it->second;
}
std::mutex g_ioMutex;
hipStream_t g_ioStream[64] = {};                // per GPU: uploads, previews and view read-backs of every engine on it
// DSR_PIPELINED_VIEW=2: per GPU ONE view stream for all engines and ONE fusion stream for all instance-sized volumes (a host drives
// its instance volumes one after the other anyway): a map + N instances are then 4-5 streams instead of 2N + 4, and the map's
// fusion stream need not share a hardware queue with anybody
hipStream_t g_sharedViewStream[64] = {}, g_sharedSmallStream[64] = {};

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      char _b[512];                                                                                \
      snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return fail(DSR_E_DEVICE, _b);                                                               \
    }                                                                                              \
  } while (0)

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

struct RenderStateDev {  // ITMRenderState_VH
  int32_t *visibleIDs = nullptr;
  int32_t *visibleIDsAlt = nullptr;  // ping-pong target of the post-decay compaction (live only)
  int4 *visBlocks = nullptr;         // the visible-block stream: one 16-byte record per list entry (dsr_device.h)
  int4 *visBlocksAlt = nullptr;
  uint8_t *visType = nullptr;
  float2 *minmax = nullptr;
  float4 *raycastResult = nullptr;
  uchar4 *raycastImage = nullptr;
  int ctrIdx = CTR_NO_VISIBLE_LIVE;
};

struct ProfRec { std::string name; double ms = 0; long long launches = 0; };

}  // namespace

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
  // the box (pixels, end exclusive) outside which the current view's depth is known to be 0: set by the silhouette cut-out
  // that produced an instance's view, the whole image after any other writer.  The allocation's per-pixel mark runs over it.
  int viewBox[4] = {0, 0, 0, 0};
  int gridExpected = 128;  // workgroups of k_expected_depth_lds (env DSR_GRID_EXPECTED; 64: 60 us, 128: 44, 256: 84)
  Mat4 calibInv, M_d, invM_d;

  SceneP scene{};
  RenderStateDev live, freeview;
  int2 *tileSums = nullptr;
  uint2 *integrateStats = nullptr;  // per wave of k_integrate: {lanes that stored depth planes, colour voxels}
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
  uint8_t *pvPin = nullptr, *pvDev = nullptr;  // previews: packed BGR (3 B / pixel), then int16 millimetres
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

  // profiling
  int profiling = 0;  // 0 off, 1 every kernel, 2 the two dominant kernels only
  std::vector<ProfRec> profRecs;
  std::map<std::string, int> profIndex;
  struct Pending { int rec; hipEvent_t a, b; };
  std::vector<Pending> profPending;
  std::vector<hipEvent_t> eventPool;
};

namespace {

int set_device(dsr_engine *e) {
  HIP_TRY(hipSetDevice(e->device));
  return DSR_OK;
}

hipEvent_t get_event(dsr_engine *e) {
  if (!e->eventPool.empty()) { hipEvent_t ev = e->eventPool.back(); e->eventPool.pop_back(); return ev; }
  hipEvent_t ev = nullptr;
  (void)hipEventCreate(&ev);
  return ev;
}

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

struct ProfScope {
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
    if (e->profPending.size() > 8192) prof_resolve(e);
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

#define LAUNCH(e, name, kernel, grid, block, ...)                                \
  do {                                                                           \
    ProfScope _ps((e), (name));                                                  \
    hipLaunchKernelGGL(kernel, grid, block, 0, (e)->stream, __VA_ARGS__);        \
  } while (0)

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

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
    HIP_TRY(hipMemsetAsync(e->scene.visBits, 0, (size_t)kSmallBitWords * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->scene.allocBits, 0, (size_t)kSmallBitWords * 4, e->stream));
  }
  e->fifoHead = 0; e->fifoLen = 0;
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
  F(e->scene.visBits); F(e->scene.allocBits);
  for (RenderStateDev *rs : {&e->live, &e->freeview}) {
    F(rs->visibleIDs); F(rs->visibleIDsAlt); F(rs->visBlocks); F(rs->visBlocksAlt); F(rs->visType); F(rs->minmax); F(rs->raycastResult); F(rs->raycastImage);
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

// div_short(a, b, RN(1/b)) against a / b for every numerator mantissa (a in [1, 2): division is
// scale invariant while nothing under- or overflows, and symmetric in the signs)
__global__ __launch_bounds__(256) void k_check_short_division(float b, unsigned long long *mismatches) {
  const float y = 1.0f / b;
  unsigned long long bad = 0;
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
    const float a = __uint_as_float(0x3f800000u | m);
    if (__float_as_uint(div_short(a, b, y)) != __float_as_uint(a / b)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

// true when the one-correction division is exact for this divisor (k_integrate.h div_short)
int short_division_exact(hipStream_t stream, float b, bool *exact) {
  unsigned long long *d = nullptr, h = 1;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 8));
  (void)hipMemsetAsync(d, 0, 8, stream);
  hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, stream, b, d);
  hipError_t err = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, stream);
  if (err == hipSuccess) err = hipStreamSynchronize(stream);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(DSR_E_DEVICE, "short-division check failed to run");
  *exact = (h == 0);
  return DSR_OK;
}

template <class T>
int dmalloc(T **p, size_t n) {
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
  return DSR_OK;
}

// ---- host buffers in and out without draining the engine's stream (see dsr_engine) ---------------------------------

// (Round 4 measured the small streams — instance volumes, view operations, I/O — at the device's highest priority: no gain at the
//  runtime's default number of hardware queues, profiles/r04d_through_shim_queues.log.)
hipError_t create_stream(hipStream_t *out) { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }

int io_stream(dsr_engine *e, hipStream_t *out) {
  if (e->device < 0 || e->device >= 64) return fail(DSR_E_ARG, "device ordinal beyond the I/O stream table");
  std::lock_guard<std::mutex> lock(g_ioMutex);
  if (!g_ioStream[e->device]) HIP_TRY(create_stream(&g_ioStream[e->device]));
  *out = g_ioStream[e->device];
  return DSR_OK;
}

// Events that only order one stream of a GPU after another stream of the SAME GPU need a device-scope release; HIP's default is a
// system-scope one (the XCD L2s written back and invalidated for the host's sake) at every record — there are 6-8 such records in
// an instance volume's frame.  Events the HOST waits on before reading pinned memory (preview read-backs, the host store's
// counter), events that stand between kernels and COPY-ENGINE transfers (the view event the I/O stream's read-backs wait for, the
// upload events), events handed to streams that are not ours (dsr_wait_for_stream / dsr_stream_wait_for_engine) and events waited
// for from another GPU keep the default.  (Free-running instance frame 348 -> 295 us, profiles/r04g_instance_frame_sysscope.json.)
unsigned order_event_flags() { return hipEventDisableTiming | hipEventReleaseToDevice; }
int make_event(hipEvent_t *ev, bool hostWaits = false) {
  if (!*ev) HIP_TRY(hipEventCreateWithFlags(ev, hostWaits ? hipEventDisableTiming : order_event_flags()));
  return DSR_OK;
}

// call before enqueuing a kernel that WRITES e's view on `stream`: readers on the I/O stream (previews, read-backs) first
int before_view_write(dsr_engine *e, hipStream_t stream) {
  if (e->viewReadEver) HIP_TRY(hipStreamWaitEvent(stream, e->evViewRead, 0));
  return DSR_OK;
}
// ... and after it
int view_written(dsr_engine *e, hipStream_t stream) {
  e->hasView = true;
  if (!e->s.sync_status && !e->pipelinedView) {
    // an engine driven without status waits (bench, the sharded scene): nobody reads its view back as a rule, and a record per
    // view operation is a packet in the frame's dependent chain — the event is recorded when a reader turns up (io_reads_view)
    e->viewEventValid = false;
    return DSR_OK;
  }
  int st = make_event(&e->evView, true);  // system scope: what waits for it on the I/O stream are copy-engine reads of the view
  if (st) return st;
  HIP_TRY(hipEventRecord(e->evView, stream));
  e->viewEventValid = true;
  return DSR_OK;
}
// the I/O stream becomes a reader of e's view as it is after everything queued so far that writes it
int io_reads_view(dsr_engine *e, hipStream_t io) {
  if (!e->viewEventValid) {  // no record at write time (see view_written): after everything queued on the engine's streams so far
    int st = make_event(&e->evView, true);
    if (st) return st;
    HIP_TRY(hipEventRecord(e->evView, e->stream));
    e->viewEventValid = true;
  }
  HIP_TRY(hipStreamWaitEvent(io, e->evView, 0));
  return DSR_OK;
}
int io_read_done(dsr_engine *e, hipStream_t io) {
  int st = make_event(&e->evViewRead, true);  // the host waits on it and then reads pinned memory
  if (st) return st;
  HIP_TRY(hipEventRecord(e->evViewRead, io));
  e->viewReadEver = true;
  return DSR_OK;
}

// ---- the pipelined view (see dsr_engine): which stream a view operation of `e` runs on, and the hand-over of buffers
hipStream_t vstream(dsr_engine *e) { return e->pipelinedView ? e->viewStream : e->stream; }

struct ViewTarget { uchar4 *rgb; float *depth; };

// `ws` is about to REPLACE e's whole view (ingest, SetView, a cut-out from another engine's view): -> the buffers to write
int begin_view_replace(dsr_engine *e, hipStream_t ws, ViewTarget *t) {
  int st = before_view_write(e, ws);  // readers on the I/O stream
  if (st) return st;
  e->viewBox[0] = 0; e->viewBox[1] = 0; e->viewBox[2] = e->W; e->viewBox[3] = e->H;  // (a cut-out narrows it afterwards)
  if (!e->pipelinedView) { t->rgb = e->rgb; t->depth = e->depth; return DSR_OK; }
  if (!e->rgbAlt) {
    if ((st = dmalloc(&e->rgbAlt, (size_t)e->Wr * e->Hr)) || (st = dmalloc(&e->depthAlt, (size_t)e->P))) return st;
    if ((st = make_event(&e->evAltFree)) || (st = make_event(&e->evFusionRead))) return st;
  }
  if (e->altFreeValid) HIP_TRY(hipStreamWaitEvent(ws, e->evAltFree, 0));  // fusion work that read this buffer when it was current
  if (ws != e->viewStream && e->viewEventValid) HIP_TRY(hipStreamWaitEvent(ws, e->evView, 0));  // a writer on another stream before us
  t->rgb = e->rgbAlt; t->depth = e->depthAlt;
  return DSR_OK;
}
// ... has queued its writes: the new view becomes current
int end_view_replace(dsr_engine *e, hipStream_t ws, bool recordView = true) {
  if (e->pipelinedView) {
    std::swap(e->rgb, e->rgbAlt);
    std::swap(e->depth, e->depthAlt);
    // whatever reads the old view (now the spare buffer) has been queued on the fusion stream by now
    HIP_TRY(hipEventRecord(e->evAltFree, e->stream));
    e->altFreeValid = true;
  }
  return recordView ? view_written(e, ws) : DSR_OK;
}
// an in-place modification of the CURRENT view on e's view stream (blanking a silhouette): after the fusion that read this buffer
int begin_view_modify(dsr_engine *e) {
  hipStream_t ws = vstream(e);
  int st = before_view_write(e, ws);
  if (st) return st;
  if (e->pipelinedView && e->fusionReadDepth == e->depth) HIP_TRY(hipStreamWaitEvent(ws, e->evFusionRead, 0));
  return DSR_OK;
}
// fusion (allocation, integration, anything on the engine's stream that READS the view) starts / has been queued
int before_fusion(dsr_engine *e) {
  if (e->pipelinedView && e->viewEventValid) HIP_TRY(hipStreamWaitEvent(e->stream, e->evView, 0));
  return DSR_OK;
}
int after_fusion(dsr_engine *e) {
  if (e->pipelinedView && e->evFusionRead) {
    HIP_TRY(hipEventRecord(e->evFusionRead, e->stream));
    e->fusionReadDepth = e->depth;
  }
  return DSR_OK;
}

// A frame handed over as host buffers: copied into a pinned slot (the caller's buffers are free on return), uploaded on the
// I/O stream into the landing buffer; the engine's stream waits for the upload, not the host.  -> device addresses of the two
// parts.  The caller enqueues its ingest kernel on e->stream and then calls upload_consumed().
int upload_frame(dsr_engine *e, hipStream_t consumer, const void *colour, size_t cBytes, const void *depth, size_t dBytes,
                 const uint8_t **cDev, const uint8_t **dDev) {
  hipStream_t io = nullptr;
  int st = io_stream(e, &io);
  if (st) return st;
  if (!e->upDev) {
    e->upDepthOff = (((size_t)e->Wr * e->Hr * 4) + 255) / 256 * 256;
    e->upBytes = e->upDepthOff + (size_t)e->P * 4;
    for (int k = 0; k < 2; ++k) {
      if (hipHostMalloc(reinterpret_cast<void **>(&e->upPin[k]), e->upBytes, hipHostMallocDefault) != hipSuccess)
        return fail(DSR_E_NOMEM, "pinned frame staging allocation failed");
      HIP_TRY(hipEventCreateWithFlags(&e->upSlotFree[k], hipEventDisableTiming));
    }
    if ((st = dmalloc(&e->upDev, e->upBytes))) return st;
    // (system scope: the two events stand between copy-engine transfers and kernels)
    if ((st = make_event(&e->evUploaded, true)) || (st = make_event(&e->evIngested, true))) return st;
  }
  if (cBytes > e->upDepthOff || e->upDepthOff + dBytes > e->upBytes) return fail(DSR_E_ARG, "frame larger than the staging slot");
  const int s = e->upNext;
  e->upNext ^= 1;
  if (e->upSlotUsed[s]) HIP_TRY(hipEventSynchronize(e->upSlotFree[s]));  // the upload of two frames ago: long done
  memcpy(e->upPin[s], colour, cBytes);
  memcpy(e->upPin[s] + e->upDepthOff, depth, dBytes);
  if (e->ingestPending) HIP_TRY(hipStreamWaitEvent(io, e->evIngested, 0));  // the previous ingest kernel reads the landing buffer
  // (the frame is staged even when the caller's buffers are page-locked: "free on return" is part of the contract, and a copy
  //  straight out of the caller's buffer would still be reading it after the call)
  HIP_TRY(hipMemcpyAsync(e->upDev, e->upPin[s], cBytes, hipMemcpyHostToDevice, io));
  HIP_TRY(hipMemcpyAsync(e->upDev + e->upDepthOff, e->upPin[s] + e->upDepthOff, dBytes, hipMemcpyHostToDevice, io));
  HIP_TRY(hipEventRecord(e->upSlotFree[s], io));
  e->upSlotUsed[s] = true;
  HIP_TRY(hipEventRecord(e->evUploaded, io));
  HIP_TRY(hipStreamWaitEvent(consumer, e->evUploaded, 0));
  *cDev = e->upDev;
  *dDev = e->upDev + e->upDepthOff;
  return DSR_OK;
}
int upload_consumed(dsr_engine *e, hipStream_t consumer) {
  HIP_TRY(hipEventRecord(e->evIngested, consumer));
  e->ingestPending = true;
  return DSR_OK;
}

// ITMViewBuilder::UpdateView's optional bilateral passes on a view whose float depth is already in `depth` (on e->stream as the
// caller has set it)
int filter_view(dsr_engine *e, float *depth) {
  if (!e->s.use_bilateral_filter) return DSR_OK;
  HIP_TRY(hipMemcpyAsync(e->depthTmp, depth, (size_t)e->P * 4, hipMemcpyDeviceToDevice, e->stream));
  dim3 g(div_up(e->W, 16), div_up(e->H, 16));
  for (int k = 0; k < 5; ++k) {
    if (k & 1) LAUNCH(e, "filter_depth", k_filter_depth, g, dim3(256), (const float *)e->depthTmp, depth, e->W, e->H);
    else LAUNCH(e, "filter_depth", k_filter_depth, g, dim3(256), (const float *)depth, e->depthTmp, e->W, e->H);
  }
  HIP_TRY(hipMemcpyAsync(depth, e->depthTmp, (size_t)e->P * 4, hipMemcpyDeviceToDevice, e->stream));
  return DSR_OK;
}

// UpdateView from device-resident RGBA + int16 mm (the caller's HBM buffers, or the landing buffer of an upload): one fused
// ingest kernel when both are 16-byte aligned.  Runs on the view stream; `uploaded`: the inputs are the landing buffer.
int convert_view(dsr_engine *e, const void *rgbDev, const void *depthDev, bool uploaded = false) {
  const float a = e->calib.disparity_calib[0], b = e->calib.disparity_calib[1];
  const size_t rgbBytes = (size_t)e->Wr * e->Hr * 4;
  hipStream_t ws = vstream(e);
  ViewTarget t;
  int st = begin_view_replace(e, ws, &t);
  if (st) return st;
  {
    StreamSwap sw(e, ws);
    if (((uintptr_t)rgbDev & 15) == 0 && ((uintptr_t)depthDev & 15) == 0) {
      const int nRgbVec = (int)(rgbBytes / 16), nQuads = div_up(e->P, 4);
      LAUNCH(e, "view_ingest", k_view_ingest, dim3(div_up(std::max(nRgbVec, nQuads), 256)), dim3(256), (const uint4 *)rgbDev,
             reinterpret_cast<uint4 *>(t.rgb), nRgbVec, e->Wr * e->Hr, (const short *)depthDev, t.depth, e->P, a, b);
    } else {
      HIP_TRY(hipMemcpyAsync(t.rgb, rgbDev, rgbBytes, hipMemcpyDeviceToDevice, ws));
      HIP_TRY(hipMemcpyAsync(e->rawDepth, depthDev, (size_t)e->P * 2, hipMemcpyDeviceToDevice, ws));
      LAUNCH(e, "depth_to_float", k_depth_to_float, dim3(div_up(div_up(e->P, 4), 256)), dim3(256), e->rawDepth, t.depth,
             e->P, a, b);
    }
    HIP_TRY(hipGetLastError());
    if (uploaded && (st = upload_consumed(e, ws))) return st;
    // ITMViewBuilder::UpdateView: five ping-pong passes, result copied back into view->depth
    if ((st = filter_view(e, t.depth))) return st;
  }
  return end_view_replace(e, ws);
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
    const int cells = ((e->W + 7) / 8) * ((e->H + 7) / 8);
    {
      ProfScope _ps(e, "small_alloc_visible");
      hipLaunchKernelGGL(k_small_alloc_visible, dim3(1), dim3(kSmallThreads), small_lds_bytes(cells), e->stream, p, e->scene,
                         (const float *)e->depth, rs.visType, e->numTilesE, e->allocWork, rs.visibleIDs, rs.visBlocks, e->noBlocks,
                         e->statusDev, e->statusSeq, reinterpret_cast<int2 *>(rs.minmax));
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
                       rs.ctrIdx == CTR_NO_VISIBLE_LIVE ? 1 : 0);
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
  LAUNCH(e, name, k_raycast, g, dim3(256), p, e->scene, rs.ctrIdx, (const float2 *)rs.minmax, rs.raycastResult);
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

// ---- RCCL, loaded on first use: the library itself does not link librccl (a host without multi-GPU needs never pays for it,
// and a process that already holds a copy — PyTorch ships its own — keeps using that one)
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
RcclApi *rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)  // a copy the process has loaded already (torch's) wins
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : paths)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.lib) {
      const char *why = dlerror();  // (dlerror() clears the message: one call)
      api.error = std::string("librccl not found: ") + (why ? why : "");
      return;
    }
#define RCCL_SYM(field, name)                                                           \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));            \
    if (!api.field && api.error.empty()) api.error = std::string("librccl lacks ") + name;
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId") RCCL_SYM(CommInitRank, "ncclCommInitRank") RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy") RCCL_SYM(AllGather, "ncclAllGather") RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd") RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
  });
  return &api;
}
// RCCL prints a version banner on STDOUT when a process's first communicator comes up; a host that reports on stdout (a bench
// line, DynSLAM's own logs piped to a tool) must not find it there: fd 1 points at stderr while the communicator is created.
std::mutex g_stdoutSwapMutex;  // the descriptor swap is process-wide: one communicator creation at a time
struct StdoutToStderr {
  std::lock_guard<std::mutex> lock{g_stdoutSwapMutex};
  int saved = -1;
  StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
  ~StdoutToStderr() { if (saved >= 0) { fflush(stdout); dup2(saved, 1); close(saved); } }
};
#define RCCL_TRY(api, expr)                                                                              \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) return fail(DSR_E_DEVICE, std::string(#expr) + ": " + (api)->GetErrorString(_r)); \
  } while (0)

}  // namespace

// One exchange = the layer buffers of the fused preview on every GPU this process drives, the communicator(s) between them and
// one stream per GPU (include/dsr.h "multi-GPU").  GROUP = one GPU's worth of ranks: the unit the collective sees.  The gathered
// buffer holds groups x perGroup x slots layers of 8 bytes per pixel (float depth plane, then RGBA plane); a rank's own slots
// lie INSIDE the gathered buffer of its GPU (in-place all-gather), so ranks that share a GPU exchange nothing at all.
struct dsr_exchange {
  int nRanks = 0, slots = 0, P = 0;
  size_t layerBytes = 0, chunkBytes = 0;  // chunk = one group's share of the gathered buffer
  int groups = 0, perGroup = 0;
  std::vector<int> groupOfRank, indexInGroup;
  bool rankMode = false;
  struct Dev {
    int device = 0, group = 0;
    hipStream_t stream = nullptr;
    uint8_t *all = nullptr;                 // gathered layers
    uchar4 *targetRgba = nullptr;           // the exchange's own composite target (lazily)
    float *targetDepth = nullptr;
    ncclComm_t comm = nullptr;
  };
  std::vector<Dev> devs;                    // local GPUs
  std::vector<int> devOfRank;               // index into devs, -1: a rank of another process
  bool useRccl = false;
};

namespace {

size_t layer_index(const dsr_exchange *x, int rank, int slot) {
  return ((size_t)x->groupOfRank[rank] * x->perGroup + x->indexInGroup[rank]) * x->slots + slot;
}
dsr_exchange::Dev *local_dev(dsr_exchange *x, int rank) {
  if (!x || rank < 0 || rank >= x->nRanks || x->devOfRank[rank] < 0) return nullptr;
  return &x->devs[x->devOfRank[rank]];
}
void exchange_free(dsr_exchange *x) {
  if (!x) return;
  RcclApi *api = x->useRccl ? rccl_api() : nullptr;
  for (auto &d : x->devs) {
    (void)hipSetDevice(d.device);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.comm && api && api->CommDestroy) (void)api->CommDestroy(d.comm);
    if (d.all) (void)hipFree(d.all);
    if (d.targetRgba) (void)hipFree(d.targetRgba);
    if (d.targetDepth) (void)hipFree(d.targetDepth);
    if (d.stream) (void)hipStreamDestroy(d.stream);
  }
  delete x;
}
int exchange_alloc(dsr_exchange *x) {
  x->layerBytes = (size_t)x->P * 8;
  x->chunkBytes = x->layerBytes * x->perGroup * x->slots;
  for (auto &d : x->devs) {
    HIP_TRY(hipSetDevice(d.device));
    HIP_TRY(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.all), x->chunkBytes * x->groups));
    HIP_TRY(hipMemsetAsync(d.all, 0, x->chunkBytes * x->groups, d.stream));  // empty layers: depth 0 never wins a pixel
    HIP_TRY(hipStreamSynchronize(d.stream));
  }
  return DSR_OK;
}
int exchange_target(dsr_exchange *x, dsr_exchange::Dev *d) {
  if (d->targetRgba) return DSR_OK;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->targetRgba), (size_t)x->P * 4));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d->targetDepth), (size_t)x->P * 4));
  HIP_TRY(hipMemsetAsync(d->targetRgba, 0, (size_t)x->P * 4, d->stream));
  HIP_TRY(hipMemsetAsync(d->targetDepth, 0, (size_t)x->P * 4, d->stream));
  return DSR_OK;
}

}  // namespace

// ---- HBM ceiling probe kernel (dsr_measure_copy_bandwidth)
typedef float copy_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy16(const copy_v4f *__restrict__ in, copy_v4f *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);  // streaming: no reuse to keep in L2
    else out[i] = in[i];
  }
}

// per-pixel conversion kernels of the boundary (k_edges.h): device-resident and host-buffer drivers
template <class K, class TI, class TO>
int convert_dev(K kernel, int device, void *hip_stream, const void *in, void *out, int n) {
  if (!in || !out || n <= 0) return fail(DSR_E_ARG, "bad conversion arguments");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, (const TI *)in, (TO *)out, n);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}
// Host-buffer form of the conversions (what InfiniTamDriver.cpp:81-144 calls every frame): the device
// scratch is kept per thread and per GPU and only ever grows, so a frame costs two copies and one
// launch — no hipMalloc / hipFree (both synchronise the device) on the per-frame path.
struct ConvScratch {
  int device = -1;
  uint8_t *in = nullptr, *out = nullptr;
  size_t inCap = 0, outCap = 0;
  // never freed at thread / process exit: the HIP runtime may already be gone by then
};
static int conv_reserve(uint8_t **buf, size_t *cap, size_t bytes) {
  if (*cap >= bytes) return DSR_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr; *cap = 0;
  const size_t want = bytes + bytes / 4;  // head room: images of a sequence differ little in size
  int st = dmalloc(buf, want);
  if (st) return st;
  *cap = want;
  return DSR_OK;
}
template <class K, class TI, class TO>
int convert_host(K kernel, const void *in, size_t inBytes, void *out, size_t outBytes, int n) {
  if (!in || !out || n <= 0) return fail(DSR_E_ARG, "bad conversion arguments");
  static thread_local ConvScratch sc;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (sc.device != dev) {  // the scratch belongs to the GPU it was allocated on
    if (sc.in) (void)hipFree(sc.in);
    if (sc.out) (void)hipFree(sc.out);
    sc.in = sc.out = nullptr; sc.inCap = sc.outCap = 0;
    sc.device = dev;
  }
  int st = conv_reserve(&sc.in, &sc.inCap, inBytes);
  if (st) return st;
  if ((st = conv_reserve(&sc.out, &sc.outCap, outBytes))) return st;
  HIP_TRY(hipMemcpy(sc.in, in, inBytes, hipMemcpyHostToDevice));
  st = convert_dev<K, TI, TO>(kernel, -1, nullptr, sc.in, sc.out, n);
  if (st) return st;
  if (hipMemcpy(out, sc.out, outBytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(DSR_E_DEVICE, "conversion copy failed");
  return DSR_OK;
}

#define CHECK_E(e)                                          \
  if (!(e)) return fail(DSR_E_ARG, "null engine");          \
  { int _st = set_device(e); if (_st) return _st; }

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

const char *dsr_last_error(void) { return g_err.c_str(); }

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
  const int pvMode = getenv("DSR_PIPELINED_VIEW") ? atoi(getenv("DSR_PIPELINED_VIEW")) : 0;
  auto shared_stream = [&](hipStream_t *table) -> hipStream_t {
    if (e->device < 0 || e->device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_ioMutex);
    if (!table[e->device] && create_stream(&table[e->device]) != hipSuccess) table[e->device] = nullptr;
    return table[e->device];
  };
  if (pvMode == 2 && s.sdf_local_block_num <= 16384 && (e->stream = shared_stream(g_sharedSmallStream))) e->ownsStream = false;
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
    e->smallPath = e->smallVolume && !s.use_swapping && e->E <= kSmallMaxEntries && e->numTilesE <= kSmallMaxTiles &&
                   small_lds_bytes(cells) <= 64 * 1024;
    if (e->smallPath) {
      ALLOC(dmalloc(&e->scene.visBits, (size_t)kSmallBitWords));
      ALLOC(dmalloc(&e->scene.allocBits, (size_t)kSmallBitWords));
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
  }
  ALLOC(dmalloc(&e->live.visibleIDsAlt, (size_t)e->noBlocks));
  ALLOC(dmalloc(&e->live.visBlocksAlt, (size_t)e->noBlocks));
  e->live.ctrIdx = CTR_NO_VISIBLE_LIVE; e->freeview.ctrIdx = CTR_NO_VISIBLE_FREE;
  ALLOC(dmalloc(&e->tileSums, (size_t)e->numTilesMax + 1));
  ALLOC(dmalloc(&e->integrateStats, (size_t)e->gridIntegrate * kIntegrateWaves));
  (void)hipMemsetAsync(e->integrateStats, 0, (size_t)e->gridIntegrate * kIntegrateWaves * sizeof(uint2), e->stream);
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
  // OFF by default.  Measured through the C++ host at configs[2] (5 mm map + 4 instance volumes, profiles/r04d_through_shim_queues.log):
  // 453 frames/s with one stream per engine, 388 with the view streams at the runtime's default of 4 hardware queues (twice the
  // streams share them: a stream's packets wait behind another stream's dependent chain in the same queue), 450 with the small
  // streams at high priority, 488-498 with GPU_MAX_HW_QUEUES=16 (+ priority) — a gain only with a process-wide runtime setting
  // the library cannot make.  env DSR_PIPELINED_VIEW=1 enables it; the parity suite runs both forms.
  e->pipelinedView = false;
  if (const char *pv = getenv("DSR_PIPELINED_VIEW")) e->pipelinedView = atoi(pv) != 0;
  if (e->pipelinedView && pvMode == 2 && (e->viewStream = shared_stream(g_sharedViewStream))) e->ownsViewStream = false;
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
  *out = e;
  return DSR_OK;
}

void dsr_engine_destroy(dsr_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
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
  CHECK_E(e);
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

// ---- view

int dsr_update_view(dsr_engine *e, const uint8_t *rgba, const int16_t *depth_mm) {
  CHECK_E(e);
  if (!rgba || !depth_mm) return fail(DSR_E_ARG, "null image");
  const uint8_t *cDev = nullptr, *dDev = nullptr;
  int st = upload_frame(e, vstream(e), rgba, (size_t)e->Wr * e->Hr * 4, depth_mm, (size_t)e->P * 2, &cDev, &dDev);
  if (st) return st;
  return convert_view(e, cDev, dDev, true);  // the landing buffer's two parts are 256-byte aligned
}

int dsr_update_view_bgr(dsr_engine *e, const uint8_t *bgr, const int16_t *depth_mm) {
  CHECK_E(e);
  if (!bgr || !depth_mm) return fail(DSR_E_ARG, "null image");
  const uint8_t *cDev = nullptr, *dDev = nullptr;
  hipStream_t ws = vstream(e);
  int st = upload_frame(e, ws, bgr, (size_t)e->Wr * e->Hr * 3, depth_mm, (size_t)e->P * 2, &cDev, &dDev);
  if (st) return st;
  ViewTarget t;
  if ((st = begin_view_replace(e, ws, &t))) return st;
  const float a = e->calib.disparity_calib[0], b = e->calib.disparity_calib[1];
  {
    StreamSwap sw(e, ws);
    if (e->Wr * e->Hr == e->P) {
      LAUNCH(e, "view_ingest", k_view_ingest_bgr, dim3(div_up(div_up(e->P, 4), 256)), dim3(256), (const uint32_t *)cDev,
             reinterpret_cast<uint4 *>(t.rgb), e->P, (const short *)dDev, t.depth, a, b);
    } else {
      LAUNCH(e, "view_ingest", k_bgr_to_rgba, dim3(div_up(e->Wr * e->Hr, 256)), dim3(256), cDev, t.rgb, e->Wr * e->Hr);
      LAUNCH(e, "depth_to_float", k_depth_to_float, dim3(div_up(div_up(e->P, 4), 256)), dim3(256), (const short *)dDev, t.depth, e->P, a, b);
    }
    HIP_TRY(hipGetLastError());
    if ((st = upload_consumed(e, ws)) || (st = filter_view(e, t.depth))) return st;
  }
  return end_view_replace(e, ws);
}

int dsr_update_view_dev(dsr_engine *e, const void *rgba_dev, const void *depth_mm_dev) {
  CHECK_E(e);
  if (!rgba_dev || !depth_mm_dev) return fail(DSR_E_ARG, "null image");
  return convert_view(e, rgba_dev, depth_mm_dev);
}

int dsr_set_view_float(dsr_engine *e, const uint8_t *rgba, const float *depth_m) {
  CHECK_E(e);
  if (!rgba || !depth_m) return fail(DSR_E_ARG, "null image");
  const uint8_t *cDev = nullptr, *dDev = nullptr;
  hipStream_t ws = vstream(e);
  int st = upload_frame(e, ws, rgba, (size_t)e->Wr * e->Hr * 4, depth_m, (size_t)e->P * 4, &cDev, &dDev);
  if (st) return st;
  ViewTarget t;
  if ((st = begin_view_replace(e, ws, &t))) return st;
  {
    StreamSwap sw(e, ws);
    LAUNCH(e, "set_view", k_set_view_ingest, dim3(div_up(std::max(e->Wr * e->Hr, e->P), 256)), dim3(256), (const uchar4 *)cDev, t.rgb,
           e->Wr * e->Hr, (const float *)dDev, t.depth, e->P);
    HIP_TRY(hipGetLastError());
    if ((st = upload_consumed(e, ws))) return st;
  }
  return end_view_replace(e, ws);
}

int dsr_set_view_float_dev(dsr_engine *e, const void *rgba_dev, const void *depth_m_dev) {
  CHECK_E(e);
  if (!rgba_dev || !depth_m_dev) return fail(DSR_E_ARG, "null image");
  hipStream_t ws = vstream(e);
  ViewTarget t;
  int st = begin_view_replace(e, ws, &t);
  if (st) return st;
  {
    StreamSwap sw(e, ws);
    HIP_TRY(hipMemcpyAsync(t.rgb, rgba_dev, (size_t)e->Wr * e->Hr * 4, hipMemcpyDeviceToDevice, ws));
    LAUNCH(e, "set_view", k_copy_depth_finite, dim3(div_up(e->P, 256)), dim3(256), (const float *)depth_m_dev, t.depth, e->P);
    HIP_TRY(hipGetLastError());
  }
  return end_view_replace(e, ws);
}

// view->rgb / view->depth ->UpdateHostFromDevice(): on the I/O stream, after the last kernel that wrote the view — not after
// the fusion and the raycast that may be queued behind it on the engine's stream
int dsr_pin_host_buffer(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(DSR_E_ARG, "null buffer");
  if (host_range_pinned(ptr, bytes)) return DSR_OK;
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  std::lock_guard<std::mutex> lock(g_pinMutex);
  g_pinned[(uintptr_t)ptr] = bytes;
  return DSR_OK;
}
int dsr_unpin_host_buffer(void *ptr) {
  if (!ptr) return fail(DSR_E_ARG, "null buffer");
  {
    std::lock_guard<std::mutex> lock(g_pinMutex);
    auto it = g_pinned.find((uintptr_t)ptr);
    if (it == g_pinned.end()) return fail(DSR_E_ARG, "not a buffer pinned through dsr_pin_host_buffer");
    g_pinned.erase(it);
  }
  HIP_TRY(hipHostUnregister(ptr));
  return DSR_OK;
}

int dsr_get_view(dsr_engine *e, uint8_t *rgba_out, float *depth_m_out) {
  CHECK_E(e);
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  hipStream_t io = nullptr;
  int st = io_stream(e, &io);
  if (st || (st = io_reads_view(e, io))) return st;
  if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, e->rgb, (size_t)e->Wr * e->Hr * 4, hipMemcpyDeviceToHost, io));
  if (depth_m_out) HIP_TRY(hipMemcpyAsync(depth_m_out, e->depth, (size_t)e->P * 4, hipMemcpyDeviceToHost, io));
  if ((st = io_read_done(e, io))) return st;
  HIP_TRY(hipEventSynchronize(e->evViewRead));
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
  dim3 g(div_up(e->W, 16), div_up(e->H, 16));
  // (Round 4 measured the raycast + ICP maps on the side stream with the next frame's read-only prefix under them: the kernels
  //  overlap and the raycast pays for it, 422 -> 444 us.  Archived: profiles/r05_pruned_variants.diff, r04c_overlap_prepare_ab.log.)
  {
    int st = launch_raycast(e, "raycast", p, rs);
    if (st) return st;
    LAUNCH(e, "icp_maps", k_icp_maps, g, dim3(256), p, e->scene, (const float4 *)rs.raycastResult, e->pointsMap,
           e->normalsMap, rs.raycastImage);
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
      if (!cached && e->smallPath) {
        // FindVisibleBlocks + CreateExpectedDepths in ONE workgroup (k_small.h), then the raycast that shades its own pixels
        const int cells = ((e->W + 7) / 8) * ((e->H + 7) / 8);
        {
          ProfScope _ps(e, "small_freeview");
          hipLaunchKernelGGL(k_small_freeview, dim3(1), dim3(kSmallThreads), small_lds_bytes(cells), e->stream, p, e->scene, e->allocList,
                             rs.visibleIDs, rs.visBlocks, e->noBlocks, reinterpret_cast<int2 *>(rs.minmax));
        }
        e->fvValid = true; e->fvVersion = e->sceneVersion; e->fvM = M; memcpy(e->fvProj, proj, sizeof proj);
        if (outIsDevice) {
          LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
                 rs.raycastResult, type, rs.raycastImage, (float *)depth_out, (uchar4 *)rgba_out);
          HIP_TRY(hipGetLastError());
          break;
        }
        LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
               rs.raycastResult, type, rs.raycastImage, depth_out ? e->freeDepth : (float *)nullptr, (uchar4 *)nullptr);
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
                 rs.raycastResult, type, rs.raycastImage, (float *)depth_out, (uchar4 *)rgba_out);
          HIP_TRY(hipGetLastError());
          break;
        }
        LAUNCH(e, "raycast_freeview", k_raycast_render, g, dim3(256), p, e->scene, (const float2 *)rs.minmax,
               rs.raycastResult, type, rs.raycastImage, depth_out ? e->freeDepth : (float *)nullptr, (uchar4 *)nullptr);
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

int dsr_get_image(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], uint8_t *rgba_out,
                  float *depth_out) {
  CHECK_E(e);
  return render_common(e, type, pose_m, intrinsics, rgba_out, depth_out, false);
}

int dsr_get_image_dev(dsr_engine *e, int type, const float pose_m[16], const float intrinsics[4], void *rgba_out_dev,
                      void *depth_out_dev) {
  CHECK_E(e);
  return render_common(e, type, pose_m, intrinsics, rgba_out_dev, depth_out_dev, true);
}

// ---- edges of the path: depth ingest, instance view split

int dsr_depth_from_disparity_dev(int device, void *hip_stream, const void *disparity_dev, void *depth_mm_out_dev, int n,
                                 float baseline_m, float focal_px, float scale, float min_depth_m, float max_depth_m) {
  if (!disparity_dev || !depth_mm_out_dev || n <= 0) return fail(DSR_E_ARG, "bad disparity arguments");
  const int minMm = (int)(min_depth_m * 1000.0f), maxMm = (int)(max_depth_m * 1000.0f);
  if (maxMm >= 32767) return fail(DSR_E_ARG, "maximum depth does not fit an int16 millimetre map (DepthProvider.h:110-116)");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_depth_from_disparity, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream,
                     (const float *)disparity_dev, (short *)depth_mm_out_dev, n, baseline_m, focal_px, scale, minMm, maxMm);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

int dsr_depth_from_disparity(const float *disparity, int16_t *depth_mm_out, int n, float baseline_m, float focal_px,
                             float scale, float min_depth_m, float max_depth_m) {
  if (!disparity || !depth_mm_out || n <= 0) return fail(DSR_E_ARG, "bad disparity arguments");
  float *d = nullptr; short *o = nullptr;
  int st = dmalloc(&d, (size_t)n);
  if (st) return st;
  if ((st = dmalloc(&o, (size_t)n))) { (void)hipFree(d); return st; }
  hipError_t err = hipMemcpy(d, disparity, (size_t)n * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    st = dsr_depth_from_disparity_dev(-1, nullptr, d, o, n, baseline_m, focal_px, scale, min_depth_m, max_depth_m);
    if (st == DSR_OK) err = hipMemcpy(depth_mm_out, o, (size_t)n * 2, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d); (void)hipFree(o);
  if (st) return st;
  if (err != hipSuccess) return fail(DSR_E_DEVICE, "disparity conversion copy failed");
  return DSR_OK;
}

// ---- layout shims of the host at the boundary (InfiniTamDriver.cpp:81-144)

int dsr_bgr_to_rgba_dev(int device, void *hip_stream, const void *bgr_dev, void *rgba_out_dev, int n) {
  return convert_dev<decltype(&k_bgr_to_rgba), uint8_t, uchar4>(k_bgr_to_rgba, device, hip_stream, bgr_dev, rgba_out_dev, n);
}
int dsr_bgr_to_rgba(const uint8_t *bgr, uint8_t *rgba_out, int n) {
  return convert_host<decltype(&k_bgr_to_rgba), uint8_t, uchar4>(k_bgr_to_rgba, bgr, (size_t)n * 3, rgba_out, (size_t)n * 4, n);
}
int dsr_rgba_to_bgr_dev(int device, void *hip_stream, const void *rgba_dev, void *bgr_out_dev, int n) {
  return convert_dev<decltype(&k_rgba_to_bgr), uchar4, uint8_t>(k_rgba_to_bgr, device, hip_stream, rgba_dev, bgr_out_dev, n);
}
int dsr_rgba_to_bgr(const uint8_t *rgba, uint8_t *bgr_out, int n) {
  return convert_host<decltype(&k_rgba_to_bgr), uchar4, uint8_t>(k_rgba_to_bgr, rgba, (size_t)n * 4, bgr_out, (size_t)n * 3, n);
}
int dsr_depth_m_to_mm_dev(int device, void *hip_stream, const void *depth_m_dev, void *depth_mm_out_dev, int n) {
  return convert_dev<decltype(&k_depth_m_to_mm), float, short>(k_depth_m_to_mm, device, hip_stream, depth_m_dev, depth_mm_out_dev, n);
}
int dsr_depth_m_to_mm(const float *depth_m, int16_t *depth_mm_out, int n) {
  return convert_host<decltype(&k_depth_m_to_mm), float, short>(k_depth_m_to_mm, depth_m, (size_t)n * 4, depth_mm_out, (size_t)n * 2, n);
}

// ---- precomputed depth / disparity maps on disk (PrecomputedDepthProvider.cpp:22-75) -------------------
// Host-side parsing (disk I/O is not GPU work); the clamp and the disparity -> depth step that follow run on
// the GPU (k_clip_depth_mm, k_depth_from_disparity).
static short clip_limit_mm(float max_depth_m) {
  // static_cast<int16_t>(round(GetMaxDepthMeters() * kMetersToMillimeters)) (:57-58)
  const float f = roundf(max_depth_m * 1000.0f);
  return (short)(f >= 32767.0f ? 32767 : (f <= -32768.0f ? -32768 : (int)f));
}
int dsr_clip_depth_mm_dev(int device, void *hip_stream, void *depth_mm_dev, int n, float max_depth_m) {
  if (!depth_mm_dev || n <= 0) return fail(DSR_E_ARG, "bad clip arguments");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_clip_depth_mm, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, (short *)depth_mm_dev, n,
                     clip_limit_mm(max_depth_m));
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}
int dsr_clip_depth_mm(int16_t *depth_mm, int n, float max_depth_m) {
  if (!depth_mm || n <= 0) return fail(DSR_E_ARG, "bad clip arguments");
  short *d = nullptr;
  int st = dmalloc(&d, (size_t)n);
  if (st) return st;
  hipError_t err = hipMemcpy(d, depth_mm, (size_t)n * 2, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    st = dsr_clip_depth_mm_dev(-1, nullptr, d, n, max_depth_m);
    if (st == DSR_OK) err = hipMemcpy(depth_mm, d, (size_t)n * 2, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (st) return st;
  if (err != hipSuccess) return fail(DSR_E_DEVICE, "clip copy failed");
  return DSR_OK;
}

// the text between <tag ...> and </tag> of the first such element at or after `from` (FileStorage XML is
// flat enough for this: the node "depth-frame" holds <rows>, <cols>, <dt>, <data>)
static bool xml_element(const std::string &doc, const char *tag, size_t from, size_t *begin, size_t *end) {
  const std::string open = std::string("<") + tag;
  size_t p0 = doc.find(open, from);
  while (p0 != std::string::npos) {
    const char c = p0 + open.size() < doc.size() ? doc[p0 + open.size()] : 0;
    if (c == '>' || c == ' ' || c == '\t' || c == '\n' || c == '\r') break;
    p0 = doc.find(open, p0 + 1);
  }
  if (p0 == std::string::npos) return false;
  const size_t gt = doc.find('>', p0);
  if (gt == std::string::npos) return false;
  const size_t close = doc.find(std::string("</") + tag + ">", gt);
  if (close == std::string::npos) return false;
  *begin = gt + 1; *end = close;
  return true;
}
static bool read_whole_file(const char *path, std::string *out) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}

static int read_depth_xml_impl(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return fail(DSR_E_ARG, "bad arguments");
  std::string doc;
  if (!read_whole_file(path, &doc)) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  size_t nb, ne, b, e2;
  if (!xml_element(doc, "depth-frame", 0, &nb, &ne)) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  const std::string node = doc.substr(nb, ne - nb);
  int rows = 0, cols = 0;
  if (!xml_element(node, "rows", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <rows>");
  rows = atoi(node.substr(b, e2 - b).c_str());
  if (!xml_element(node, "cols", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <cols>");
  cols = atoi(node.substr(b, e2 - b).c_str());
  if (!xml_element(node, "dt", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <dt>");
  std::string dt = node.substr(b, e2 - b);
  dt.erase(std::remove_if(dt.begin(), dt.end(), [](char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }), dt.end());
  if (dt != "s") return fail(DSR_E_IO, "Precomputed depth map had the wrong format.");  // :42-44: CV_16SC1 only
  // a size no camera produces is a malformed file, not something to allocate for (the size query hands it to the caller)
  if ((long long)rows * cols > (1ll << 28) || rows > (1 << 20) || cols > (1 << 20)) return fail(DSR_E_IO, "depth-frame: implausible rows x cols");
  *width = cols; *height = rows;
  if (rows <= 0 || cols <= 0) return fail(DSR_E_IO, "Could not read precomputed depth map: empty matrix");
  if (!depth_mm_out || (long long)rows * cols > capacity) return fail(DSR_E_ARG, "depth map larger than the buffer");
  if (!xml_element(node, "data", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <data>");
  const char *p = node.c_str() + b, *end = node.c_str() + e2;
  const long long n = (long long)rows * cols;
  long long i = 0;
  while (i < n) {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
    if (p >= end) break;
    char *next = nullptr;
    const long v = strtol(p, &next, 10);
    if (next == p) return fail(DSR_E_IO, "malformed <data> in depth-frame");
    depth_mm_out[i++] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));  // cv::saturate_cast<short>
    p = next;
  }
  if (i != n) return fail(DSR_E_IO, "depth-frame <data> holds fewer values than rows x cols");
  return DSR_OK;
}

static int read_pfm_impl(const char *path, float *out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return fail(DSR_E_ARG, "bad arguments");
  FILE *f = fopen(path, "rb");
  if (!f) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  char magic[3] = {0, 0, 0};
  int w = 0, h = 0;
  float scale = 0.0f;
  // "Pf" <ws> width <ws> height <ws> scale <single whitespace byte> raster
  if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "Pf") != 0 || fscanf(f, "%d %d %f", &w, &h, &scale) != 3) {
    fclose(f);
    return fail(DSR_E_IO, "not a single-channel PFM (\"Pf\") file");
  }
  (void)fgetc(f);
  if ((long long)w * h > (1ll << 28) || w > (1 << 20) || h > (1 << 20)) { fclose(f); return fail(DSR_E_IO, "PFM: implausible width x height"); }
  *width = w; *height = h;
  if (w <= 0 || h <= 0) { fclose(f); return fail(DSR_E_IO, "Could not read precomputed depth map: empty image"); }
  if (!out || (long long)w * h > capacity) { fclose(f); return fail(DSR_E_ARG, "PFM image larger than the buffer"); }
  const bool fileLittle = scale < 0.0f;
  const uint16_t probe = 1;
  const bool hostLittle = *reinterpret_cast<const uint8_t *>(&probe) == 1;
  for (int r = h - 1; r >= 0; --r) {  // the file's first row is the image's bottom row
    float *row = out + (size_t)r * w;
    if (fread(row, 4, (size_t)w, f) != (size_t)w) { fclose(f); return fail(DSR_E_IO, "PFM raster shorter than width x height"); }
    if (fileLittle != hostLittle)
      for (int c = 0; c < w; ++c) {
        uint32_t v; memcpy(&v, row + c, 4);
        v = (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
        memcpy(row + c, &v, 4);
      }
  }
  fclose(f);
  return DSR_OK;
}

// The size is reported whenever the header could be read (DSR_OK, and DSR_E_ARG for a buffer that is too small: the
// size query of a caller that allocates afterwards); after DSR_E_IO it is 0 x 0, never a half-parsed value.
int dsr_read_depth_xml(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height) {
  const int st = read_depth_xml_impl(path, depth_mm_out, capacity, width, height);
  if (st == DSR_E_IO && width && height) *width = *height = 0;
  return st;
}
int dsr_read_pfm(const char *path, float *out, int capacity, int *width, int *height) {
  const int st = read_pfm_impl(path, out, capacity, width, height);
  if (st == DSR_E_IO && width && height) *width = *height = 0;
  return st;
}

// -> device-side address of the staged mask (see the ring's description in dsr_engine); `mask_slot_used` must be called
// after the kernel that reads it has been enqueued
static int upload_mask(dsr_engine *e, const uint8_t *mask, int box_w, int box_h, const uint8_t **devOut, int *slotOut) {
  const size_t n = (size_t)box_w * box_h;
  if (e->maskSlotBytes < n) {  // grow: rare (a mask larger than any before) — drain, then reallocate the ring
    HIP_TRY(hipStreamSynchronize(vstream(e)));
    if (e->maskHost) (void)hipHostFree(e->maskHost);
    e->maskHost = e->maskHostDev = nullptr; e->maskSlotBytes = 0;
    const size_t slot = ((n + n / 2 + 4095) / 4096) * 4096;
    if (hipHostMalloc(reinterpret_cast<void **>(&e->maskHost), slot * dsr_engine::kMaskSlots, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
      return fail(DSR_E_NOMEM, "mask staging allocation failed");
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&e->maskHostDev), e->maskHost, 0) != hipSuccess)
      return fail(DSR_E_DEVICE, "mask staging is not device-visible");
    e->maskSlotBytes = slot;
    for (bool &u : e->maskEventUsed) u = false;
  }
  const int s = e->maskNext;
  e->maskNext = (s + 1) % dsr_engine::kMaskSlots;
  if (!e->maskEvent[s]) HIP_TRY(hipEventCreateWithFlags(&e->maskEvent[s], hipEventDisableTiming));
  if (e->maskEventUsed[s]) HIP_TRY(hipEventSynchronize(e->maskEvent[s]));  // the kernel that last read this slot (kMaskSlots masks ago)
  memcpy(e->maskHost + (size_t)s * e->maskSlotBytes, mask, n);  // the caller's (pageable) buffer is free after this line
  *devOut = e->maskHostDev + (size_t)s * e->maskSlotBytes;
  *slotOut = s;
  return DSR_OK;
}
static int mask_slot_used(dsr_engine *e, int slot) {
  HIP_TRY(hipEventRecord(e->maskEvent[slot], vstream(e)));
  e->maskEventUsed[slot] = true;
  return DSR_OK;
}

// One volume per GPU: the main engine (the full frame) and the instance volume may live on different devices.  The cut-out is
// produced on main's GPU into a transfer pair, sent with one peer copy per plane (xGMI; through the host where the GPUs have no
// peer access) on main's stream, and the instance's stream waits for the event behind it.
static int enable_peer_access(int from, int to) {
  static std::mutex m;
  static unsigned long long done[64] = {};
  if (from == to || from >= 64 || to >= 64) return DSR_OK;
  std::lock_guard<std::mutex> lock(m);
  if (done[from] & (1ull << to)) return DSR_OK;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(from) == hipSuccess) {
      const hipError_t err = hipDeviceEnablePeerAccess(to, 0);
      if (err != hipSuccess && err != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // the copy falls back to staging
    }
    (void)hipSetDevice(prev);
  }
  done[from] |= 1ull << to;
  return DSR_OK;
}

// maskDev == nullptr: `mask` is a host buffer, staged through the engine's pinned ring (no synchronisation).
// rbw > 0: the same launch also blanks the silhouette `rmask` in the main view (dsr_view_split_silhouette) — the cut-out reads
// the pixel first, as the two host loops would (InstanceReconstructor.cpp:238-263).
static int extract_silhouette(dsr_engine *main_engine, dsr_engine *instance, const uint8_t *mask, const uint8_t *maskDev,
                              int x0, int y0, int box_w, int box_h, const uint8_t *rmask = nullptr, const uint8_t *rmaskDev = nullptr,
                              int rx0 = 0, int ry0 = 0, int rbw = 0, int rbh = 0) {
  CHECK_E(main_engine);
  if (!instance || (!mask && !maskDev) || box_w <= 0 || box_h <= 0) return fail(DSR_E_ARG, "bad silhouette arguments");
  const bool blank = rbw > 0 || rbh > 0 || rmask || rmaskDev;
  if (blank && ((!rmask && !rmaskDev) || rbw <= 0 || rbh <= 0)) return fail(DSR_E_ARG, "bad silhouette arguments");
  if (!main_engine->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (instance->W != main_engine->W || instance->H != main_engine->H ||
      instance->Wr != main_engine->Wr || instance->Hr != main_engine->Hr || main_engine->W != main_engine->Wr)
    return fail(DSR_E_ARG, "main and instance engines must share the image size");
  int maskSlot = -1, rmaskSlot = -1;
  if (!maskDev) {
    int st = upload_mask(main_engine, mask, box_w, box_h, &maskDev, &maskSlot);
    if (st) return st;
  }
  if (blank && !rmaskDev) {
    if (rmask == mask && rbw == box_w && rbh == box_h) rmaskDev = maskDev;  // one mask for both (the sharded scene): staged once
    else { int st = upload_mask(main_engine, rmask, rbw, rbh, &rmaskDev, &rmaskSlot); if (st) return st; }
  }
  dsr_engine *e = main_engine;
  const bool forcePeerPath = getenv("DSR_FORCE_PEER_PATH") != nullptr;  // tests: the cross-GPU path on one GPU
  const bool peer = instance->device != e->device || forcePeerPath;
  // Runs on the MAIN engine's view stream: ordered after the producer of its view and before any later blanking.  The kernel
  // REPLACES the instance's view: a pipelined instance takes it in its spare buffer (begin_view_replace: only the fusion that
  // last read that buffer is waited for); otherwise work queued on the instance's stream (the previous frame's integration) may
  // still be reading the one buffer, and the main side first waits for all of it.  An instance that SHARES the main engine's
  // stream (dsr_engine_share_stream: one volume per GPU next to its view engine) is ordered by that stream alone: no event.
  hipStream_t ws = vstream(e);
  const bool sameStream = !instance->pipelinedView && instance->stream == ws;
  if (!instance->pipelinedView && !sameStream) {
    // (an event is created and recorded with its own stream's device current; WAITING for it works from any device)
    if (peer) HIP_TRY(hipSetDevice(instance->device));
    if (!instance->xEvent) HIP_TRY(hipEventCreateWithFlags(&instance->xEvent, instance->device != e->device ? hipEventDisableTiming : order_event_flags()));
    HIP_TRY(hipEventRecord(instance->xEvent, instance->stream));
    if (peer) HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamWaitEvent(ws, instance->xEvent, 0));
  }
  if (blank) { int st = begin_view_modify(e); if (st) return st; }
  ViewTarget t;
  {
    // (a pipelined instance allocates its spare view buffers and their events on first use: on ITS GPU, not on main's)
    if (peer) HIP_TRY(hipSetDevice(instance->device));
    if (instance->pipelinedView && !instance->rgbAlt) {
      int st = dmalloc(&instance->rgbAlt, (size_t)instance->Wr * instance->Hr);
      if (st || (st = dmalloc(&instance->depthAlt, (size_t)instance->P)) || (st = make_event(&instance->evAltFree, instance->device != e->device)) ||
          (st = make_event(&instance->evFusionRead, instance->device != e->device))) { if (peer) (void)hipSetDevice(e->device); return st; }
    }
    if (peer) HIP_TRY(hipSetDevice(e->device));
    int st = begin_view_replace(instance, ws, &t);
    if (st) return st;
  }
  uchar4 *dstRgb = t.rgb;
  float *dstDepth = t.depth;
  if (peer) {
    if (!e->xferRgb) {
      int st = dmalloc(&e->xferRgb, (size_t)e->P);
      if (st || (st = dmalloc(&e->xferDepth, (size_t)e->P))) return st;
    }
    enable_peer_access(e->device, instance->device);
    enable_peer_access(instance->device, e->device);
    dstRgb = e->xferRgb; dstDepth = e->xferDepth;
  }
  {
    StreamSwap sw(e, ws);
    if (blank)
      LAUNCH(e, "split_silhouette", k_split_silhouette, dim3(div_up(e->W, 16), div_up(e->H, 16)), dim3(256), e->rgb, e->depth, dstRgb,
             dstDepth, e->W, e->H, maskDev, x0, y0, box_w, box_h, rmaskDev, rx0, ry0, rbw, rbh);
    else
      LAUNCH(e, "extract_silhouette", k_extract_silhouette, dim3(div_up(e->W, 16), div_up(e->H, 16)), dim3(256),
             (const uchar4 *)e->rgb, (const float *)e->depth, dstRgb, dstDepth, e->W, e->H,
             maskDev, x0, y0, box_w, box_h);
  }
  HIP_TRY(hipGetLastError());
  if (maskSlot >= 0) { int st = mask_slot_used(e, maskSlot); if (st) return st; }
  if (rmaskSlot >= 0) { int st = mask_slot_used(e, rmaskSlot); if (st) return st; }
  if (blank) { int st = view_written(e, ws); if (st) return st; }
  if (peer) {
    HIP_TRY(hipMemcpyPeerAsync(t.rgb, instance->device, e->xferRgb, e->device, (size_t)e->P * 4, ws));
    HIP_TRY(hipMemcpyPeerAsync(t.depth, instance->device, e->xferDepth, e->device, (size_t)e->P * 4, ws));
  }
  int stv = DSR_OK;
  if (sameStream) {
    stv = end_view_replace(instance, ws);
  } else {
    // the instance's side: its "view written" event is recorded on a stream of ITS device (its view stream / its only stream),
    // behind a wait for the main side — so every event is only ever recorded with its own device's streams
    if (!e->xEvent2 || (instance->device != e->device && !e->xEvent2System)) {  // waited for from another GPU: system scope
      if (e->xEvent2) (void)hipEventDestroy(e->xEvent2);
      e->xEvent2 = nullptr;
      e->xEvent2System = instance->device != e->device;
      HIP_TRY(hipEventCreateWithFlags(&e->xEvent2, e->xEvent2System ? hipEventDisableTiming : order_event_flags()));
    }
    HIP_TRY(hipEventRecord(e->xEvent2, ws));
    if (peer) HIP_TRY(hipSetDevice(instance->device));
    {
      hipStream_t is = vstream(instance);
      const hipError_t werr = hipStreamWaitEvent(is, e->xEvent2, 0);
      if (werr != hipSuccess) stv = fail(DSR_E_DEVICE, std::string("hipStreamWaitEvent: ") + hipGetErrorString(werr));
      else stv = end_view_replace(instance, is);  // buffers swapped; the instance's view is final once `is` has passed this point
    }
    if (peer) HIP_TRY(hipSetDevice(e->device));
  }
  // outside the mask's box the cut-out is empty (depth 0): the instance's allocation mark need not look there
  instance->viewBox[0] = std::max(0, x0); instance->viewBox[1] = std::max(0, y0);
  instance->viewBox[2] = std::min(e->W, x0 + box_w); instance->viewBox[3] = std::min(e->H, y0 + box_h);
  return stv;
}

int dsr_view_extract_silhouette(dsr_engine *main_engine, dsr_engine *instance, const uint8_t *mask, int x0, int y0,
                                int box_w, int box_h) {
  return extract_silhouette(main_engine, instance, mask, nullptr, x0, y0, box_w, box_h);
}
int dsr_view_extract_silhouette_dev(dsr_engine *main_engine, dsr_engine *instance, const void *mask_dev, int x0, int y0,
                                    int box_w, int box_h) {
  if (!mask_dev) return fail(DSR_E_ARG, "bad silhouette arguments");
  return extract_silhouette(main_engine, instance, nullptr, (const uint8_t *)mask_dev, x0, y0, box_w, box_h);
}
int dsr_view_split_silhouette(dsr_engine *main_engine, dsr_engine *instance, const uint8_t *copy_mask, int x0, int y0, int box_w,
                              int box_h, const uint8_t *delete_mask, int dx0, int dy0, int dbox_w, int dbox_h) {
  if (!delete_mask) return fail(DSR_E_ARG, "bad silhouette arguments");
  return extract_silhouette(main_engine, instance, copy_mask, nullptr, x0, y0, box_w, box_h, delete_mask, nullptr, dx0, dy0, dbox_w, dbox_h);
}
int dsr_view_split_silhouette_dev(dsr_engine *main_engine, dsr_engine *instance, const void *copy_mask_dev, int x0, int y0, int box_w,
                                  int box_h, const void *delete_mask_dev, int dx0, int dy0, int dbox_w, int dbox_h) {
  if (!copy_mask_dev || !delete_mask_dev) return fail(DSR_E_ARG, "bad silhouette arguments");
  return extract_silhouette(main_engine, instance, nullptr, (const uint8_t *)copy_mask_dev, x0, y0, box_w, box_h, nullptr,
                            (const uint8_t *)delete_mask_dev, dx0, dy0, dbox_w, dbox_h);
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

static int remove_silhouette(dsr_engine *e, const uint8_t *mask, const uint8_t *maskDev, int x0, int y0, int box_w, int box_h) {
  CHECK_E(e);
  if ((!mask && !maskDev) || box_w <= 0 || box_h <= 0) return fail(DSR_E_ARG, "bad silhouette arguments");
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (e->W != e->Wr || e->H != e->Hr) return fail(DSR_E_ARG, "rgb and depth sizes differ");
  int maskSlot = -1;
  if (!maskDev) {
    int st = upload_mask(e, mask, box_w, box_h, &maskDev, &maskSlot);
    if (st) return st;
  }
  { int st = begin_view_modify(e); if (st) return st; }
  {
    StreamSwap sw(e, vstream(e));
    LAUNCH(e, "remove_silhouette", k_remove_silhouette, dim3(div_up(box_w, 16), div_up(box_h, 16)), dim3(256), e->rgb,
           e->depth, e->W, e->H, maskDev, x0, y0, box_w, box_h);
  }
  HIP_TRY(hipGetLastError());
  if (maskSlot >= 0) { int st = mask_slot_used(e, maskSlot); if (st) return st; }
  return view_written(e, vstream(e));
}
int dsr_view_remove_silhouette(dsr_engine *e, const uint8_t *mask, int x0, int y0, int box_w, int box_h) {
  return remove_silhouette(e, mask, nullptr, x0, y0, box_w, box_h);
}
int dsr_view_remove_silhouette_dev(dsr_engine *e, const void *mask_dev, int x0, int y0, int box_w, int box_h) {
  if (!mask_dev) return fail(DSR_E_ARG, "bad silhouette arguments");
  return remove_silhouette(e, nullptr, (const uint8_t *)mask_dev, x0, y0, box_w, box_h);
}

// ---- instance compositing

int dsr_composite_instances_dev(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                const void *layers_rgba_dev, const void *layers_depth_dev, const int32_t *track_ids,
                                int n_layers, int n_pixels, float tint_strength, int dim_background) {
  if (!target_depth_dev || n_layers < 0 || n_pixels <= 0) return fail(DSR_E_ARG, "bad composite arguments");
  if (n_layers > 0 && (!layers_depth_dev || !track_ids || (target_rgba_dev && !layers_rgba_dev)))
    return fail(DSR_E_ARG, "null layer buffers");
  if (n_layers > kMaxCompositeLayers) return fail(DSR_E_ARG, "too many layers (max 64)");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  const CompositeP c = composite_params(track_ids, n_layers, n_pixels, tint_strength, dim_background);
  CompositeLayers none;
  memset(&none, 0, sizeof none);
  hipLaunchKernelGGL(k_composite<false>, dim3((n_pixels + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, c,
                     (uchar4 *)target_rgba_dev, (float *)target_depth_dev, (const uchar4 *)layers_rgba_dev,
                     (const float *)layers_depth_dev, none);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

int dsr_composite_layer_ptrs_dev(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                 const void *const *layer_rgba_ptrs, const void *const *layer_depth_ptrs,
                                 const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength,
                                 int dim_background) {
  if (!target_depth_dev || n_layers < 0 || n_pixels <= 0) return fail(DSR_E_ARG, "bad composite arguments");
  if (n_layers > 0 && (!layer_depth_ptrs || !track_ids || (target_rgba_dev && !layer_rgba_ptrs)))
    return fail(DSR_E_ARG, "null layer buffers");
  if (n_layers > kMaxCompositeLayers) return fail(DSR_E_ARG, "too many layers (max 64)");
  CompositeLayers lp;
  memset(&lp, 0, sizeof lp);
  for (int l = 0; l < n_layers; ++l) {
    lp.depth[l] = (const float *)layer_depth_ptrs[l];
    lp.rgba[l] = target_rgba_dev ? (const uchar4 *)layer_rgba_ptrs[l] : nullptr;
    if (!lp.depth[l] || (target_rgba_dev && !lp.rgba[l])) return fail(DSR_E_ARG, "null layer buffers");
  }
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  const CompositeP c = composite_params(track_ids, n_layers, n_pixels, tint_strength, dim_background);
  hipLaunchKernelGGL(k_composite<true>, dim3((n_pixels + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, c,
                     (uchar4 *)target_rgba_dev, (float *)target_depth_dev, (const uchar4 *)nullptr, (const float *)nullptr, lp);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

int dsr_composite_instances(uint8_t *target_rgba, float *target_depth, const uint8_t *layers_rgba,
                            const float *layers_depth, const int32_t *track_ids, int n_layers, int n_pixels,
                            float tint_strength, int dim_background) {
  if (!target_depth || n_layers < 0 || n_pixels <= 0) return fail(DSR_E_ARG, "bad composite arguments");
  const size_t P = (size_t)n_pixels, L = (size_t)n_layers;
  uchar4 *tR = nullptr, *lR = nullptr;
  float *tD = nullptr, *lD = nullptr;
  int st = DSR_OK;
  auto cleanup = [&]() { if (tR) (void)hipFree(tR); if (lR) (void)hipFree(lR); if (tD) (void)hipFree(tD); if (lD) (void)hipFree(lD); };
  if ((st = dmalloc(&tD, P))) { cleanup(); return st; }
  if (L && (st = dmalloc(&lD, P * L))) { cleanup(); return st; }
  if (target_rgba && (st = dmalloc(&tR, P))) { cleanup(); return st; }
  if (target_rgba && L && (st = dmalloc(&lR, P * L))) { cleanup(); return st; }
#define CP(expr) if ((expr) != hipSuccess) { cleanup(); return fail(DSR_E_DEVICE, "composite copy failed"); }
  CP(hipMemcpy(tD, target_depth, P * 4, hipMemcpyHostToDevice));
  if (L) CP(hipMemcpy(lD, layers_depth, P * L * 4, hipMemcpyHostToDevice));
  if (tR) CP(hipMemcpy(tR, target_rgba, P * 4, hipMemcpyHostToDevice));
  if (lR) CP(hipMemcpy(lR, layers_rgba, P * L * 4, hipMemcpyHostToDevice));
  st = dsr_composite_instances_dev(-1, nullptr, tR, tD, lR, lD, track_ids, n_layers, n_pixels, tint_strength, dim_background);
  if (st) { cleanup(); return st; }
  CP(hipDeviceSynchronize());
  CP(hipMemcpy(target_depth, tD, P * 4, hipMemcpyDeviceToHost));
  if (tR) CP(hipMemcpy(target_rgba, tR, P * 4, hipMemcpyDeviceToHost));
#undef CP
  cleanup();
  return DSR_OK;
}

// ---- multi-GPU exchange (include/dsr.h): layers of the fused preview, RCCL all-gather, composite

static int exchange_create_common(dsr_exchange *x, int slots_per_rank, int n_pixels) {
  if (slots_per_rank <= 0 || n_pixels <= 0) return fail(DSR_E_ARG, "bad exchange arguments");
  x->slots = slots_per_rank; x->P = n_pixels;
  return exchange_alloc(x);
}

int dsr_exchange_create(const int32_t *devices, int n_ranks, int slots_per_rank, int n_pixels, dsr_exchange **out) {
  if (!devices || n_ranks <= 0 || !out) return fail(DSR_E_ARG, "bad exchange arguments");
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) return fail(DSR_E_DEVICE, "no HIP device: the exchange has no CPU fallback");
  int prev = 0;
  (void)hipGetDevice(&prev);
  dsr_exchange *x = new (std::nothrow) dsr_exchange();
  if (!x) return fail(DSR_E_NOMEM, "oom");
  x->nRanks = n_ranks;
  x->groupOfRank.assign(n_ranks, 0); x->indexInGroup.assign(n_ranks, 0); x->devOfRank.assign(n_ranks, -1);
  std::vector<int> count;
  for (int r = 0; r < n_ranks; ++r) {
    const int dv = devices[r] < 0 ? prev : devices[r];
    if (dv >= nDev) { delete x; return fail(DSR_E_ARG, "device ordinal out of range"); }
    int g = -1;
    for (size_t k = 0; k < x->devs.size(); ++k) if (x->devs[k].device == dv) g = (int)k;
    if (g < 0) { dsr_exchange::Dev d; d.device = dv; d.group = (int)x->devs.size(); x->devs.push_back(d); count.push_back(0); g = d.group; }
    x->groupOfRank[r] = g; x->indexInGroup[r] = count[g]++; x->devOfRank[r] = g;
  }
  x->groups = (int)x->devs.size();
  x->perGroup = *std::max_element(count.begin(), count.end());
  int st = exchange_create_common(x, slots_per_rank, n_pixels);
  // one communicator rank per GPU; a single GPU has nothing to exchange (DSR_EXCHANGE_FORCE_RCCL: a 1-rank communicator anyway,
  // so that the RCCL path runs on a one-GPU box)
  if (st == DSR_OK && (x->groups > 1 || getenv("DSR_EXCHANGE_FORCE_RCCL"))) {
    RcclApi *api = rccl_api();
    if (!api->error.empty()) st = fail(DSR_E_DEVICE, api->error);
    else {
      std::vector<int> devlist; std::vector<ncclComm_t> comms(x->devs.size());
      for (auto &d : x->devs) devlist.push_back(d.device);
      ncclResult_t r;
      { StdoutToStderr quiet; r = api->CommInitAll(comms.data(), (int)devlist.size(), devlist.data()); }
      if (r != ncclSuccess) st = fail(DSR_E_DEVICE, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
      else { for (size_t k = 0; k < comms.size(); ++k) x->devs[k].comm = comms[k]; x->useRccl = true; }
    }
  }
  (void)hipSetDevice(prev);
  if (st) { exchange_free(x); return st; }
  *out = x;
  return DSR_OK;
}

int dsr_exchange_unique_id(uint8_t id_out[128]) {
  if (!id_out) return fail(DSR_E_ARG, "null");
  static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
  RcclApi *api = rccl_api();
  if (!api->error.empty()) return fail(DSR_E_DEVICE, api->error);
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return DSR_OK;
}

int dsr_exchange_create_rank(const uint8_t unique_id[128], int world_size, int rank, int device, int slots_per_rank, int n_pixels,
                             dsr_exchange **out) {
  if (!unique_id || world_size <= 0 || rank < 0 || rank >= world_size || !out) return fail(DSR_E_ARG, "bad exchange arguments");
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) return fail(DSR_E_DEVICE, "no HIP device: the exchange has no CPU fallback");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (device < 0) device = prev;
  if (device >= nDev) return fail(DSR_E_ARG, "device ordinal out of range");
  RcclApi *api = rccl_api();
  if (!api->error.empty()) return fail(DSR_E_DEVICE, api->error);
  dsr_exchange *x = new (std::nothrow) dsr_exchange();
  if (!x) return fail(DSR_E_NOMEM, "oom");
  x->rankMode = true; x->nRanks = world_size; x->groups = world_size; x->perGroup = 1;
  x->groupOfRank.resize(world_size); x->indexInGroup.assign(world_size, 0); x->devOfRank.assign(world_size, -1);
  for (int r = 0; r < world_size; ++r) x->groupOfRank[r] = r;
  dsr_exchange::Dev d; d.device = device; d.group = rank;
  x->devs.push_back(d);
  x->devOfRank[rank] = 0;
  int st = exchange_create_common(x, slots_per_rank, n_pixels);
  if (st == DSR_OK) {
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r;
    { StdoutToStderr quiet; r = (hipSetDevice(device) == hipSuccess) ? api->CommInitRank(&x->devs[0].comm, world_size, id, rank) : ncclUnhandledCudaError; }
    if (r != ncclSuccess) st = fail(DSR_E_DEVICE, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
    else x->useRccl = true;
  }
  (void)hipSetDevice(prev);
  if (st) { exchange_free(x); return st; }
  *out = x;
  return DSR_OK;
}

void dsr_exchange_destroy(dsr_exchange *x) {
  int prev = 0;
  const bool havePrev = hipGetDevice(&prev) == hipSuccess;
  exchange_free(x);
  if (havePrev) (void)hipSetDevice(prev);
}

void *dsr_exchange_stream(dsr_exchange *x, int rank) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  return d ? (void *)d->stream : nullptr;
}

int dsr_exchange_layer_ptrs(dsr_exchange *x, int on_rank, int rank, int slot, void **rgba_dev, void **depth_dev) {
  dsr_exchange::Dev *d = local_dev(x, on_rank);
  if (!d || rank < 0 || rank >= x->nRanks || slot < 0 || slot >= x->slots) return fail(DSR_E_ARG, "bad exchange layer");
  uint8_t *base = d->all + layer_index(x, rank, slot) * x->layerBytes;
  if (depth_dev) *depth_dev = base;                       // float depth plane first,
  if (rgba_dev) *rgba_dev = base + (size_t)x->P * 4;      // then the RGBA plane
  return DSR_OK;
}

int dsr_exchange_slot_ptrs(dsr_exchange *x, int rank, int slot, void **rgba_dev, void **depth_dev) {
  return dsr_exchange_layer_ptrs(x, rank, rank, slot, rgba_dev, depth_dev);
}

int dsr_exchange_render_slot(dsr_exchange *x, int rank, int slot, dsr_engine *e, int type, const float pose_m[16],
                             const float intrinsics[4]) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  void *rgba = nullptr, *depth = nullptr;
  if (!d || dsr_exchange_slot_ptrs(x, rank, slot, &rgba, &depth)) return fail(DSR_E_ARG, "bad exchange slot");
  if (!e) {  // not visible in this frame: an empty layer
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipMemsetAsync(depth, 0, (size_t)x->P * 4, d->stream));
    return DSR_OK;
  }
  if (e->device != d->device) return fail(DSR_E_ARG, "the engine does not live on the rank's GPU");
  if (e->P != x->P) return fail(DSR_E_ARG, "image size differs from the exchange's");
  int st = dsr_wait_for_stream(e, d->stream);  // the previous gather / composite is done with this slot
  if (st) return st;
  if ((st = render_common(e, type, pose_m, intrinsics, rgba, depth, true))) return st;
  return dsr_stream_wait_for_engine(e, d->stream);
}

int dsr_exchange_gather(dsr_exchange *x) {
  if (!x) return fail(DSR_E_ARG, "null exchange");
  if (!x->useRccl) return DSR_OK;  // one GPU: every layer is where the composite reads it
  RcclApi *api = rccl_api();
  int prev = 0;
  (void)hipGetDevice(&prev);
  RCCL_TRY(api, api->GroupStart());
  ncclResult_t r = ncclSuccess;
  for (auto &d : x->devs) {
    if (hipSetDevice(d.device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
    r = api->AllGather(d.all + (size_t)d.group * x->chunkBytes, d.all, x->chunkBytes, ncclUint8, d.comm, d.stream);  // in place
    if (r != ncclSuccess) break;
  }
  const ncclResult_t r2 = api->GroupEnd();
  (void)hipSetDevice(prev);
  if (r != ncclSuccess) return fail(DSR_E_DEVICE, std::string("ncclAllGather: ") + api->GetErrorString(r));
  if (r2 != ncclSuccess) return fail(DSR_E_DEVICE, std::string("ncclGroupEnd: ") + api->GetErrorString(r2));
  return DSR_OK;
}

int dsr_exchange_target_ptrs(dsr_exchange *x, int rank, void **rgba_dev, void **depth_dev) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  if (!d) return fail(DSR_E_ARG, "bad exchange rank");
  int st = exchange_target(x, d);
  if (st) return st;
  if (rgba_dev) *rgba_dev = d->targetRgba;
  if (depth_dev) *depth_dev = d->targetDepth;
  return DSR_OK;
}

int dsr_exchange_clear_target(dsr_exchange *x, int rank) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  if (!d) return fail(DSR_E_ARG, "bad exchange rank");
  int st = exchange_target(x, d);
  if (st) return st;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipMemsetAsync(d->targetRgba, 0, (size_t)x->P * 4, d->stream));
  HIP_TRY(hipMemsetAsync(d->targetDepth, 0, (size_t)x->P * 4, d->stream));
  return DSR_OK;
}

int dsr_exchange_composite(dsr_exchange *x, int root_rank, dsr_engine *target_engine, void *target_rgba_dev, void *target_depth_dev,
                           const int32_t *ranks, const int32_t *slots, const int32_t *track_ids, int n_layers, float tint_strength,
                           int dim_background) {
  dsr_exchange::Dev *d = local_dev(x, root_rank);
  if (!d || n_layers < 0 || (n_layers > 0 && (!ranks || !slots || !track_ids))) return fail(DSR_E_ARG, "bad composite arguments");
  if (n_layers > kMaxCompositeLayers) return fail(DSR_E_ARG, "too many layers (max 64)");
  int st = DSR_OK;
  if (!target_depth_dev) {
    if ((st = exchange_target(x, d))) return st;
    target_rgba_dev = d->targetRgba; target_depth_dev = d->targetDepth;
  }
  if (target_engine) {
    if (target_engine->device != d->device) return fail(DSR_E_ARG, "the target's engine does not live on the root's GPU");
    if ((st = dsr_stream_wait_for_engine(target_engine, d->stream))) return st;  // its render of the target
  }
  const void *rp[kMaxCompositeLayers], *dp[kMaxCompositeLayers];
  for (int l = 0; l < n_layers; ++l) {
    void *r = nullptr, *dd = nullptr;
    if ((st = dsr_exchange_layer_ptrs(x, root_rank, ranks[l], slots[l], &r, &dd))) return st;
    rp[l] = r; dp[l] = dd;
  }
  if (n_layers > 0 &&
      (st = dsr_composite_layer_ptrs_dev(d->device, d->stream, target_rgba_dev, target_depth_dev, target_rgba_dev ? rp : nullptr, dp,
                                         track_ids, n_layers, x->P, tint_strength, dim_background)))
    return st;
  if (target_engine) return dsr_wait_for_stream(target_engine, d->stream);  // its next render of the target waits for the composite
  return DSR_OK;
}

int dsr_exchange_gather_and_composite(dsr_exchange *x, int root_rank, dsr_engine *target_engine, void *target_rgba_dev,
                                      void *target_depth_dev, const int32_t *ranks, const int32_t *slots, const int32_t *track_ids,
                                      int n_layers, float tint_strength, int dim_background) {
  int st = dsr_exchange_gather(x);
  if (st) return st;
  if (!local_dev(x, root_rank)) return DSR_OK;  // this process does not hold the consumer of the preview
  return dsr_exchange_composite(x, root_rank, target_engine, target_rgba_dev, target_depth_dev, ranks, slots, track_ids, n_layers,
                                tint_strength, dim_background);
}

int dsr_exchange_read_target(dsr_exchange *x, int rank, uint8_t *rgba_out, float *depth_out) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  if (!d) return fail(DSR_E_ARG, "bad exchange rank");
  int st = exchange_target(x, d);
  if (st) return st;
  HIP_TRY(hipSetDevice(d->device));
  if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, d->targetRgba, (size_t)x->P * 4, hipMemcpyDeviceToHost, d->stream));
  if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, d->targetDepth, (size_t)x->P * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return DSR_OK;
}

int dsr_exchange_sync(dsr_exchange *x) {
  if (!x) return fail(DSR_E_ARG, "null exchange");
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (auto &d : x->devs) {
    HIP_TRY(hipSetDevice(d.device));
    HIP_TRY(hipStreamSynchronize(d.stream));
  }
  (void)hipSetDevice(prev);
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

// ---- self-test

__global__ __launch_bounds__(256) void k_selftest_division(unsigned long long n, unsigned long long seed,
                                                           unsigned long long *mismatches) {
  const float y32767 = rcp_refined(32767.0f), y255 = rcp_refined(255.0f);
  unsigned long long bad = 0;
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  // exhaustive small domains
  if (tid < 65536) {
    const float a = (float)(short)(int)(tid - 32768);
    if (__float_as_uint(div_with_rcp(a, 32767.0f, y32767)) != __float_as_uint(a / 32767.0f)) bad++;
  }
  if (tid < 256) {
    const float a = (float)(int)tid;
    if (__float_as_uint(div_with_rcp(a, 255.0f, y255)) != __float_as_uint(a / 255.0f)) bad++;
  }
  for (unsigned long long i = tid; i < n; i += stride) {
    // splitmix64
    unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    // a: sign, exponent in [-40, 40]; b: sign, exponent in [-34, 40] (>= 1e-10: the smallest divisor a
    // call site lets through is the frustum test's),
    // random mantissas; every 16th pair uses small integers (weights) as divisor
    const unsigned ma = (unsigned)(z & 0x7fffffu), mb = (unsigned)((z >> 23) & 0x7fffffu);
    const int ea = (int)((z >> 46) % 81) - 40, eb = (int)((z >> 53) % 75) - 34;
    float a = __uint_as_float(((unsigned)(ea + 127) << 23) | ma);
    float b = __uint_as_float(((unsigned)(eb + 127) << 23) | mb);
    if (z >> 63) a = -a;
    if ((z >> 62) & 1) b = -b;
    if ((i & 15) == 0) b = (float)(1 + (int)((z >> 23) & 0x1ff));
    if ((i & 255) == 1) a = 0.0f;
    const float q = a / b;
    if (!(fabsf(q) == 0.0f || (fabsf(q) >= 1.17549435e-38f && fabsf(q) < 3.0e38f))) continue;  // not tame
    if (__float_as_uint(fdiv_tame(a, b)) != __float_as_uint(q)) bad++;
    const float yb = rcp_refined(b);
    if (__float_as_uint(div_with_rcp(a, b, yb)) != __float_as_uint(q)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

int dsr_selftest_division(int device, uint64_t n, uint64_t seed, uint64_t *mismatches) {
  if (!mismatches) return fail(DSR_E_ARG, "null");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  unsigned long long *d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), 8));
  HIP_TRY(hipMemset(d, 0, 8));
  hipLaunchKernelGGL(k_selftest_division, dim3(4096), dim3(256), 0, 0, (unsigned long long)n, (unsigned long long)seed, d);
  // the divisors the one-correction form is used with: the constants, every integer weight, and
  // the truncation bands of the presets (an engine checks its own mu at creation)
  hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, 32767.0f, d);
  hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, 255.0f, d);
  for (int w = 1; w <= 256; ++w) hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, (float)w, d);
  for (float mu : {0.02f, 0.016f, 0.2f, 0.14f, 0.1f, 0.3f, 0.05f, 0.04f, 0.08f, 0.5f, 1.0f, 4.0f})
    hipLaunchKernelGGL(k_check_short_division, dim3(1024), dim3(256), 0, 0, mu, d);
  unsigned long long h = 0;
  hipError_t err = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(DSR_E_DEVICE, "selftest failed to run");
  *mismatches = h;
  return DSR_OK;
}

#ifdef DSR_RAYCAST_STATS
// measurement builds only (tools/raycast_wave_stats.py): where k_raycast writes its 12 words per wave
int dsr_debug_raycast_stats(void *dev_buf) {
  unsigned int *p = (unsigned int *)dev_buf;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rcStats), &p, sizeof(p)));
  return DSR_OK;
}
#endif

// ---- HBM ceiling probe (roofline harness)

int dsr_measure_copy_bandwidth(int device, uint64_t bytes, int iters, double *gbps_out) {
  if (!gbps_out || bytes < 16 || iters <= 0) return fail(DSR_E_ARG, "bad bandwidth probe arguments");
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  float4 *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a), bytes));
  if (hipMalloc(reinterpret_cast<void **>(&b), bytes) != hipSuccess) { (void)hipFree(a); return fail(DSR_E_NOMEM, "probe buffers"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMemset(a, 1, bytes);
  if (err == hipSuccess) err = hipMemset(b, 2, bytes);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  // The ceiling a copy kernel reaches depends on its launch shape (4.57 TB/s with one fixed shape where the
  // guide's float4 copy reaches 6.29): three grids x plain / non-temporal accesses, `iters` passes each, the BEST is reported.
  float ms = 0.0f;
  if (err == hipSuccess) {
    const size_t n = bytes / 16;
    float best = 0.0f;
    for (int variant = 0; variant < 6 && err == hipSuccess; ++variant) {
      const int grid = 256 * (variant % 3 == 0 ? 4 : variant % 3 == 1 ? 8 : 16);  // 4 / 8 / 16 workgroups per CU, grid-stride
      const bool nt = variant >= 3;
      auto launch = [&]() {
        if (nt) hipLaunchKernelGGL((k_copy16<true>), dim3(grid), dim3(256), 0, 0, (const copy_v4f *)a, (copy_v4f *)b, n);
        else hipLaunchKernelGGL((k_copy16<false>), dim3(grid), dim3(256), 0, 0, (const copy_v4f *)a, (copy_v4f *)b, n);
      };
      launch();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < iters; ++i) launch();
      (void)hipEventRecord(e1, 0);
      err = hipEventSynchronize(e1);
      float t = 0.0f;
      if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
      if (err == hipSuccess && t > 0.0f && (best == 0.0f || t < best)) best = t;
    }
    ms = best;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (err != hipSuccess || !(ms > 0.0f)) return fail(DSR_E_DEVICE, "bandwidth probe failed");
  *gbps_out = 2.0 * (double)(bytes / 16 * 16) * iters / ((double)ms * 1e-3) / 1e9;
  return DSR_OK;
}

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
int dsr_get_view_previews(dsr_engine *e, uint8_t *bgr_out, int16_t *depth_mm_out) {
  CHECK_E(e);
  if (!e->hasView) return fail(DSR_E_NO_VIEW, "no view yet");
  if (e->W != e->Wr || e->H != e->Hr) return fail(DSR_E_ARG, "rgb and depth sizes differ");
  // The previews depend on the VIEW only, and the host asks for them right after queuing the raycast
  // (InfiniTamDriver.h:148-158): they are produced on the GPU's I/O stream, ordered after the last kernel that wrote the view
  // (evView) — not behind the integration and the raycast on the engine's stream —, land in pinned memory and are handed
  // over with one wait for exactly that work.
  hipStream_t io = nullptr;
  int st = io_stream(e, &io);
  if (st) return st;
  if (!e->pvDev) {
    e->pvMmOff = ((size_t)e->P * 3 + 255) / 256 * 256;
    const size_t bytes = e->pvMmOff + (size_t)e->P * 2;
    if ((st = dmalloc(&e->pvDev, bytes))) return st;
    if (hipHostMalloc(reinterpret_cast<void **>(&e->pvPin), bytes, hipHostMallocDefault) != hipSuccess)
      return fail(DSR_E_NOMEM, "pinned preview staging allocation failed");
  }
  if ((st = io_reads_view(e, io))) return st;
  // buffers the caller page-locked (dsr_pin_host_buffer) take the copy directly: no staging copy on the host
  const bool bgrPinned = bgr_out && host_range_pinned(bgr_out, (size_t)e->P * 3);
  const bool mmPinned = depth_mm_out && host_range_pinned(depth_mm_out, (size_t)e->P * 2);
  {
    StreamSwap sw(e, io);
    if (bgr_out) {
      LAUNCH(e, "preview_convert", k_rgba_to_bgr, dim3(div_up(e->P, 256)), dim3(256), (const uchar4 *)e->rgb, e->pvDev, e->P);
      HIP_TRY(hipMemcpyAsync(bgrPinned ? bgr_out : e->pvPin, e->pvDev, (size_t)e->P * 3, hipMemcpyDeviceToHost, io));
    }
    if (depth_mm_out) {
      LAUNCH(e, "preview_convert", k_depth_m_to_mm, dim3(div_up(e->P, 256)), dim3(256), (const float *)e->depth,
             reinterpret_cast<short *>(e->pvDev + e->pvMmOff), e->P);
      HIP_TRY(hipMemcpyAsync(mmPinned ? (uint8_t *)depth_mm_out : e->pvPin + e->pvMmOff, e->pvDev + e->pvMmOff, (size_t)e->P * 2,
                             hipMemcpyDeviceToHost, io));
    }
  }
  HIP_TRY(hipGetLastError());
  if ((st = io_read_done(e, io))) return st;
  HIP_TRY(hipEventSynchronize(e->evViewRead));
  if (bgr_out && !bgrPinned) memcpy(bgr_out, e->pvPin, (size_t)e->P * 3);
  if (depth_mm_out && !mmPinned) memcpy(depth_mm_out, e->pvPin + e->pvMmOff, (size_t)e->P * 2);
  return DSR_OK;
}

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
  if (raycast_result) HIP_TRY(hipMemcpyAsync(raycast_result, rs.raycastResult, P * 16, hipMemcpyDeviceToHost, e->stream));
  if (points) HIP_TRY(hipMemcpyAsync(points, e->pointsMap, P * 16, hipMemcpyDeviceToHost, e->stream));
  if (normals) HIP_TRY(hipMemcpyAsync(normals, e->normalsMap, P * 16, hipMemcpyDeviceToHost, e->stream));
  if (raycast_image) HIP_TRY(hipMemcpyAsync(raycast_image, rs.raycastImage, P * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

// ---- profiling

int dsr_profile_enable(dsr_engine *e, int enable) {
  CHECK_E(e);
  if (!enable) prof_resolve(e);
  e->profiling = enable == 2 ? 2 : (enable != 0);
  return DSR_OK;
}

int dsr_profile_reset(dsr_engine *e) {
  CHECK_E(e);
  prof_resolve(e);
  for (auto &r : e->profRecs) { r.ms = 0; r.launches = 0; }
  // work counters restart as well (decayed-block count is kept)
  unsigned long long zero = 0;
  HIP_TRY(hipMemcpyAsync(e->scene.work + WORK_V_INTEGRATED, &zero, 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->scene.work + WORK_V_EXPECTED, &zero, 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->scene.work + WORK_V_DECAY, &zero, 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(e->integrateStats, 0, (size_t)e->gridIntegrate * kIntegrateWaves * sizeof(uint2), e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return DSR_OK;
}

int dsr_profile_get(dsr_engine *e, dsr_kernel_time *out, int cap) {
  if (!e || !out || cap <= 0) return 0;
  if (set_device(e)) return 0;
  prof_resolve(e);
  unsigned long long work[WORK_COUNT];
  if (hipMemcpy(work, e->scene.work, sizeof work, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  const double P = (double)e->P, E = (double)e->E, B = (double)kBlockBytes;
  // k_integrate's own tallies: lanes that stored their 24 B of depth planes, voxels that got colour
  double storeLanes = 0.0, colourVoxels = 0.0;
  {
    std::vector<uint2> ws((size_t)e->gridIntegrate * kIntegrateWaves);
    if (hipMemcpy(ws.data(), e->integrateStats, ws.size() * sizeof(uint2), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (const uint2 &w : ws) { storeLanes += (double)w.x; colourVoxels += (double)w.y; }
  }
  int n = 0;
  for (auto &r : e->profRecs) {
    if (n >= cap) break;
    if (r.launches == 0) continue;
    dsr_kernel_time &k = out[n++];
    memset(&k, 0, sizeof k);
    strncpy(k.name, r.name.c_str(), sizeof k.name - 1);
    k.total_ms = r.ms; k.launches = r.launches;
    const double L = (double)r.launches;
    // algorithmic bytes, SURVEY.md 8(d) / DESIGN.md "byte model"
    if (r.name == "integrate") {
      const double V = (double)work[WORK_V_INTEGRATED];
      k.bytes = V * (16.0 + 2.0 * B) + L * 8.0 * P;  // the reference's AoS formulation
      // what THIS layout has to move (DESIGN.md "byte model"): per visible block its list id (4 B), hash
      // entry (16 B) and the sdf + w_depth planes (1536 B) read; 24 B written back per lane that updated
      // a voxel; per colour voxel ONE 4-byte word (r, g, b, w_color) read and written; the depth and RGB frames (8 B per pixel)
      k.bytes_layout = V * (4.0 + 16.0 + 1536.0) + storeLanes * 24.0 + colourVoxels * 8.0 + L * 8.0 * P;
      k.units = V;
      k.store_lanes = storeLanes; k.colour_voxels = colourVoxels;
    }
    else if (r.name == "depth_to_float") k.bytes = L * 6.0 * P;
    else if (r.name == "expected_depth") k.bytes = (double)work[WORK_V_EXPECTED] * 16.0 + L * 8.0 * std::ceil(e->W / 8.0) * std::ceil(e->H / 8.0);
    else if (r.name == "icp_maps") k.bytes = L * P * (16.0 + 16.0 + 16.0 + 4.0);
    else if (r.name == "alloc_commit") k.bytes = L * E / 8.0;
    else if (r.name == "visible_count") k.bytes = L * E * 1.0;
    else if (r.name == "visible_write") k.bytes = L * E * 1.0;
    else if (r.name == "decay_blocks") k.bytes = (double)work[WORK_V_DECAY] * (16.0 + 2.0 * B);
    else k.bytes = 0.0;
  }
  return n;
}

}  // extern "C"
