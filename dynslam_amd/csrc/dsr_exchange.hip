// dsr_exchange.hip — the multi-GPU side of the C ABI (include/dsr.h "multi-GPU"): the layer buffers of the fused preview, the
// RCCL all-gather between the GPUs (librccl is loaded on first use, never linked), and the compositing entry points
// (CompositeInstances / CompositeDepth, InstanceReconstructor.cpp:851-990).
#include <rccl/rccl.h>  // types only

#include <dlfcn.h>
#include <unistd.h>

#include "dsr_internal.h"
#include "k_composite.h"

namespace {

// ---- RCCL, loaded on first use: the library itself does not link librccl (a host without multi-GPU needs never pays for it,
// and a process that already holds a copy — PyTorch ships its own — keeps using that one)
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
    RCCL_SYM(Send, "ncclSend") RCCL_SYM(Recv, "ncclRecv")
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
    bool targetClearPending = false;        // dsr_exchange_clear_target: folded into the next composite over this target
    ncclComm_t comm = nullptr;
  };
  std::vector<Dev> devs;                    // local GPUs
  std::vector<int> devOfRank;               // index into devs, -1: a rank of another process
  bool useRccl = false;
  // which collective dsr_exchange_gather runs (dsr_exchange_set_collective): every GPU gets every layer (in-place all-gather),
  // or only the GPU of rootGroup does (the composite has one consumer: 1/N of the bytes on every link but the root's)
  bool gatherToRoot = false;
  int rootGroup = 0;
  // measurement (dsr_exchange_timing): HIP events around the collective and around the composite, resolved when asked for
  bool timing = false;
  struct Timed { hipEvent_t a, b; int kind; int dev; };
  std::vector<Timed> pending;
  double gatherMs = 0.0, compositeMs = 0.0;
  int gatherN = 0, compositeN = 0;
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
  for (auto &t : x->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
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

extern "C" {

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
  // (two pixels per lane: 4.1-5.0 us for eight layers at 1242x375 against 5.0-6.3 with four — more waves in flight beat wider
  //  loads, profiles/r06d_composite_px*.json)
  hipLaunchKernelGGL((k_composite<false, 2>), dim3((n_pixels + 511) / 512), dim3(256), 0, (hipStream_t)hip_stream, c,
                     (uchar4 *)target_rgba_dev, (float *)target_depth_dev, (const uchar4 *)layers_rgba_dev,
                     (const float *)layers_depth_dev, none);
  HIP_TRY(hipGetLastError());
  return DSR_OK;
}

static int composite_layer_ptrs(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                const void *const *layer_rgba_ptrs, const void *const *layer_depth_ptrs,
                                const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength,
                                int dim_background, int clear_target);
int dsr_composite_layer_ptrs_dev(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                 const void *const *layer_rgba_ptrs, const void *const *layer_depth_ptrs,
                                 const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength,
                                 int dim_background) {
  return composite_layer_ptrs(device, hip_stream, target_rgba_dev, target_depth_dev, layer_rgba_ptrs, layer_depth_ptrs, track_ids,
                              n_layers, n_pixels, tint_strength, dim_background, 0);
}
static int composite_layer_ptrs(int device, void *hip_stream, void *target_rgba_dev, void *target_depth_dev,
                                const void *const *layer_rgba_ptrs, const void *const *layer_depth_ptrs,
                                const int32_t *track_ids, int n_layers, int n_pixels, float tint_strength,
                                int dim_background, int clear_target) {
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
  CompositeP c = composite_params(track_ids, n_layers, n_pixels, tint_strength, dim_background);
  c.clearTarget = clear_target;
  hipLaunchKernelGGL((k_composite<true, 2>), dim3((n_pixels + 511) / 512), dim3(256), 0, (hipStream_t)hip_stream, c,
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
  if ((st = dsr_internal::device_alloc(&tD, P))) { cleanup(); return st; }
  if (L && (st = dsr_internal::device_alloc(&lD, P * L))) { cleanup(); return st; }
  if (target_rgba && (st = dsr_internal::device_alloc(&tR, P))) { cleanup(); return st; }
  if (target_rgba && L && (st = dsr_internal::device_alloc(&lR, P * L))) { cleanup(); return st; }
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
  if ((st = dsr_internal::engine_render(e, type, pose_m, intrinsics, rgba, depth, true))) return st;
  return dsr_stream_wait_for_engine(e, d->stream);
}

static void timed_begin(dsr_exchange *x, dsr_exchange::Dev &d, int kind, size_t devIndex) {
  if (!x->timing) return;
  dsr_exchange::Timed t{nullptr, nullptr, kind, (int)devIndex};
  if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
  (void)hipEventRecord(t.a, d.stream);
  x->pending.push_back(t);
}
static void timed_end(dsr_exchange *x, dsr_exchange::Dev &d, int kind, size_t devIndex) {
  if (!x->timing) return;
  for (auto it = x->pending.rbegin(); it != x->pending.rend(); ++it)
    if (it->kind == kind && it->dev == (int)devIndex) { (void)hipEventRecord(it->b, d.stream); return; }
}

int dsr_exchange_set_collective(dsr_exchange *x, int gather_to_root, int root_rank) {
  if (!x || root_rank < 0 || root_rank >= x->nRanks) return fail(DSR_E_ARG, "bad exchange arguments");
  x->gatherToRoot = gather_to_root != 0;
  x->rootGroup = x->groupOfRank[root_rank];
  return DSR_OK;
}

int dsr_exchange_gather(dsr_exchange *x) {
  if (!x) return fail(DSR_E_ARG, "null exchange");
  if (!x->useRccl) return DSR_OK;  // one GPU: every layer is where the composite reads it
  RcclApi *api = rccl_api();
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (size_t k = 0; k < x->devs.size(); ++k) { (void)hipSetDevice(x->devs[k].device); timed_begin(x, x->devs[k], 0, k); }
  {
    const ncclResult_t r0 = api->GroupStart();
    if (r0 != ncclSuccess) {  // (the caller's device comes back also on this path: ADVICE r5)
      (void)hipSetDevice(prev);
      return fail(DSR_E_DEVICE, std::string("ncclGroupStart: ") + api->GetErrorString(r0));
    }
  }
  ncclResult_t r = ncclSuccess;
  for (auto &d : x->devs) {
    if (hipSetDevice(d.device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
    if (!x->gatherToRoot) {
      r = api->AllGather(d.all + (size_t)d.group * x->chunkBytes, d.all, x->chunkBytes, ncclUint8, d.comm, d.stream);  // in place
    } else if (d.group == x->rootGroup) {  // (communicator rank == group, both for ncclCommInitAll and for rank mode)
      for (int g = 0; g < x->groups && r == ncclSuccess; ++g)
        if (g != d.group) r = api->Recv(d.all + (size_t)g * x->chunkBytes, x->chunkBytes, ncclUint8, g, d.comm, d.stream);
    } else {
      r = api->Send(d.all + (size_t)d.group * x->chunkBytes, x->chunkBytes, ncclUint8, x->rootGroup, d.comm, d.stream);
    }
    if (r != ncclSuccess) break;
  }
  const ncclResult_t r2 = api->GroupEnd();
  for (size_t k = 0; k < x->devs.size(); ++k) { (void)hipSetDevice(x->devs[k].device); timed_end(x, x->devs[k], 0, k); }
  (void)hipSetDevice(prev);
  if (r != ncclSuccess) return fail(DSR_E_DEVICE, std::string("exchange collective: ") + api->GetErrorString(r));
  if (r2 != ncclSuccess) return fail(DSR_E_DEVICE, std::string("ncclGroupEnd: ") + api->GetErrorString(r2));
  return DSR_OK;
}

int dsr_exchange_timing(dsr_exchange *x, int enable, double *gather_ms, double *composite_ms, int32_t *n_gathers, int32_t *n_composites) {
  if (!x) return fail(DSR_E_ARG, "null exchange");
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (auto &t : x->pending) {  // resolve what has been recorded so far
    float ms = 0.0f;
    (void)hipSetDevice(x->devs[t.dev].device);
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
      if (t.kind == 0) { x->gatherMs += ms; x->gatherN++; } else { x->compositeMs += ms; x->compositeN++; }
    }
    (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
  }
  x->pending.clear();
  (void)hipSetDevice(prev);
  if (gather_ms) *gather_ms = x->gatherMs;
  if (composite_ms) *composite_ms = x->compositeMs;
  if (n_gathers) *n_gathers = x->gatherN;
  if (n_composites) *n_composites = x->compositeN;
  x->gatherMs = x->compositeMs = 0.0; x->gatherN = x->compositeN = 0;
  x->timing = enable != 0;
  return DSR_OK;
}

static int materialise_target_clear(dsr_exchange *x, dsr_exchange::Dev *d);
int dsr_exchange_target_ptrs(dsr_exchange *x, int rank, void **rgba_dev, void **depth_dev) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  if (!d) return fail(DSR_E_ARG, "bad exchange rank");
  int st = exchange_target(x, d);
  if (st) return st;
  if ((st = materialise_target_clear(x, d))) return st;  // (the caller is about to look at the buffers)
  if (rgba_dev) *rgba_dev = d->targetRgba;
  if (depth_dev) *depth_dev = d->targetDepth;
  return DSR_OK;
}

int dsr_exchange_clear_target(dsr_exchange *x, int rank) {
  dsr_exchange::Dev *d = local_dev(x, rank);
  if (!d) return fail(DSR_E_ARG, "bad exchange rank");
  int st = exchange_target(x, d);
  if (st) return st;
  // Folded into the next composite over this target (k_composite's clearTarget: the target is not read, every pixel is written):
  // two memsets per frame on the exchange's stream ran NEXT to the following frame's fusion kernels and slowed them
  // (k_batch_alloc_mark 21 -> 76 us under a 54 us fill, profiles/r06g_batch_step_timeline.json).  Anything else that looks at
  // the target first (dsr_exchange_read_target, _target_ptrs) performs the clear.
  d->targetClearPending = true;
  return DSR_OK;
}
static int materialise_target_clear(dsr_exchange *x, dsr_exchange::Dev *d) {
  if (!d->targetClearPending) return DSR_OK;
  d->targetClearPending = false;
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
  // gather-to-root delivers the layers to ONE group's buffer: a composite anywhere else would blend stale layers
  if (x->useRccl && x->gatherToRoot && x->groupOfRank[root_rank] != x->rootGroup)
    return fail(DSR_E_ARG, "the exchange gathers to another root (dsr_exchange_set_collective): this rank does not receive the layers");
  int st = DSR_OK;
  if (!target_depth_dev) {
    if ((st = exchange_target(x, d))) return st;
    target_rgba_dev = d->targetRgba; target_depth_dev = d->targetDepth;
  }
  // a pending dsr_exchange_clear_target of the exchange's own target: done by the composite itself, or — nothing to composite —
  // by the memsets it stands for
  int clearTarget = 0;
  if (d->targetClearPending && target_depth_dev == d->targetDepth && (target_rgba_dev == d->targetRgba || !target_rgba_dev)) {
    if (n_layers > 0 && target_rgba_dev) { clearTarget = 1; d->targetClearPending = false; }
    else if ((st = materialise_target_clear(x, d))) return st;
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
  const size_t devIndex = (size_t)(d - x->devs.data());
  if (n_layers > 0) {
    HIP_TRY(hipSetDevice(d->device));
    timed_begin(x, *d, 1, devIndex);
    if ((st = composite_layer_ptrs(d->device, d->stream, target_rgba_dev, target_depth_dev, target_rgba_dev ? rp : nullptr, dp,
                                   track_ids, n_layers, x->P, tint_strength, dim_background, clearTarget)))
      return st;
    timed_end(x, *d, 1, devIndex);
  }
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
  if ((st = materialise_target_clear(x, d))) return st;
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

}  // extern "C"
