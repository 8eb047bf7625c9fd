// dsr_view.hip — the VIEW side of the engine and the edges of the path (split from dsr_engine.hip in round 6): frames in
// (pinned double-buffered upload on the GPU's I/O stream, BGR / RGBA + int16 ingest, bilateral filter, SetView), the view
// pipeline's buffer hand-over, the instance view split (mask staging ring, cut-out / blanking kernels, the cross-GPU copy), the
// host's layout conversions and depth ingest (InfiniTamDriver.cpp:81-144, DepthProvider.h:94-137), and the two previews of
// PrepareNextStep.  Kernels: k_edges.h.  Shared with dsr_engine.hip through dsr_internal.h (namespace dsr_internal).
#include "dsr_internal.h"
#include "k_edges.h"

using namespace dsr_internal;

namespace dsr_internal {
std::mutex g_ioMutex;
hipStream_t g_ioStream[64] = {};                // per GPU: uploads, previews and view read-backs of every engine on it
}

namespace dsr_internal {

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
int make_event(hipEvent_t *ev, bool hostWaits) {
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

// `direct`: the cut-out kernel writes the instance's ONE view buffer itself (no double buffering, no transfer pair)
void cutout_write_region(const dsr_engine *instance, bool direct, int x0, int y0, int w, int h, int wr[4]) {
  const int W = instance->W, H = instance->H;
  wr[0] = 0; wr[1] = 0; wr[2] = W; wr[3] = H;
  if (!direct || !instance->blankValid) return;
  const int b[4] = {std::max(0, x0), std::max(0, y0), std::min(W, x0 + w), std::min(H, y0 + h)};
  const int *o = instance->blankBox;
  const bool bEmpty = b[0] >= b[2] || b[1] >= b[3], oEmpty = o[0] >= o[2] || o[1] >= o[3];
  if (bEmpty && oEmpty) { wr[2] = 0; wr[3] = 0; return; }
  if (bEmpty) { for (int k = 0; k < 4; ++k) wr[k] = o[k]; return; }
  if (oEmpty) { for (int k = 0; k < 4; ++k) wr[k] = b[k]; return; }
  wr[0] = std::min(b[0], o[0]); wr[1] = std::min(b[1], o[1]); wr[2] = std::max(b[2], o[2]); wr[3] = std::max(b[3], o[3]);
}
void cutout_written(dsr_engine *instance, bool direct, int x0, int y0, int w, int h) {  // (after begin_view_replace, which invalidates)
  instance->blankValid = direct;
  instance->blankBox[0] = std::max(0, x0); instance->blankBox[1] = std::max(0, y0);
  instance->blankBox[2] = std::min(instance->W, x0 + w); instance->blankBox[3] = std::min(instance->H, y0 + h);
}

struct ViewTarget { uchar4 *rgb; float *depth; };

// `ws` is about to REPLACE e's whole view (ingest, SetView, a cut-out from another engine's view): -> the buffers to write
int begin_view_replace(dsr_engine *e, hipStream_t ws, ViewTarget *t) {
  int st = before_view_write(e, ws);  // readers on the I/O stream
  if (st) return st;
  e->viewBox[0] = 0; e->viewBox[1] = 0; e->viewBox[2] = e->W; e->viewBox[3] = e->H;  // (a cut-out narrows it afterwards)
  e->blankValid = false;  // (... and says what it left blank)
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
  e->blankValid = false;  // (a blanked silhouette is rgb 0, not the cut-out's blank)
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

}  // namespace dsr_internal

namespace {

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

}  // namespace

extern "C" {

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

// -> device-side address of the staged mask (see the ring's description in dsr_engine); `mask_slot_used` must be called
// after the kernel that reads it has been enqueued
// The ring's slots hold at least `n` bytes.  Growing (rare: a mask larger than any before) drains the stream and REPLACES the
// ring, so a call that stages several masks for one kernel sizes it for the largest of them BEFORE it stages the first — a
// mask staged earlier would otherwise point into freed pinned memory.
static int ensure_mask_ring(dsr_engine *e, size_t n) {
  if (e->maskSlotBytes >= n) return DSR_OK;
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
  return DSR_OK;
}
static int upload_mask(dsr_engine *e, const uint8_t *mask, int box_w, int box_h, const uint8_t **devOut, int *slotOut) {
  const size_t n = (size_t)box_w * box_h;
  { int st = ensure_mask_ring(e, n); if (st) return st; }
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
  if (!maskDev && blank && !rmaskDev) {  // two host masks for one kernel: the ring must hold the larger one before the first is staged
    int st = ensure_mask_ring(main_engine, std::max((size_t)box_w * box_h, (size_t)rbw * rbh));
    if (st) return st;
  }
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
  const bool direct = !peer && !instance->pipelinedView;
  int wr[4];
  cutout_write_region(instance, direct, x0, y0, box_w, box_h, wr);
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
             dstDepth, e->W, e->H, maskDev, x0, y0, box_w, box_h, rmaskDev, rx0, ry0, rbw, rbh, make_int4(wr[0], wr[1], wr[2], wr[3]));
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
  if (blank) cutout_written(instance, direct, x0, y0, box_w, box_h);  // (the extract-only kernel writes the whole frame)
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
  if (!e->pvPin) {
    e->pvMmOff = ((size_t)e->P * 3 + 255) / 256 * 256;
    const size_t bytes = e->pvMmOff + (size_t)e->P * 2;
    if (hipHostMalloc(reinterpret_cast<void **>(&e->pvPin), bytes, hipHostMallocDefault) != hipSuccess)
      return fail(DSR_E_NOMEM, "pinned preview staging allocation failed");
  }
  if ((st = io_reads_view(e, io))) return st;
  // ONE conversion kernel for both previews (k_edges.h k_previews).  Buffers the caller page-locked (dsr_pin_host_buffer: the
  // reference keeps its previews in cv::Mat members) receive them directly, others through the engine's own pinned pair and one
  // memcpy on the host.  How the bytes cross the host link depends on what else the GPU is doing (profiles/r05h_through_shim.log):
  //  * an instance-sized or mid-sized volume: the kernel STORES STRAIGHT INTO the page-locked host memory — no scratch, no copy
  //    command (round 4's two kernels + two D2H copies were four commands, each handed over between the compute queue and the
  //    copy engine): configs[2] through the C++ host 477-499 -> 664-667 frames/s;
  //  * a map-sized volume (>= 2^20 blocks) that has the GPU to ITSELF: waves that sit on host-link stores take slots and memory
  //    queues from its half-millisecond integration (configs[1] through the host 851-858 -> 824-827 frames/s with direct stores),
  //    so the kernel converts into HBM and the copy engine moves the bytes, as before (852-856).  Next to instance drivers the map
  //    stores directly as well: its copies queue behind theirs on the I/O stream otherwise (configs[2] 535-543 instead of 664).
  const bool direct = !(e->noBlocks >= (1 << 20) && (e->device >= 64 || g_enginesOnDevice[e->device].load(std::memory_order_relaxed) <= 1));
  const bool bgrPinned = bgr_out && host_range_pinned(bgr_out, (size_t)e->P * 3) && ((uintptr_t)bgr_out & 3) == 0;
  const bool mmPinned = depth_mm_out && host_range_pinned(depth_mm_out, (size_t)e->P * 2) && ((uintptr_t)depth_mm_out & 7) == 0;
  uint8_t *bgrHost = bgr_out ? (bgrPinned ? bgr_out : e->pvPin) : nullptr;
  uint8_t *mmHost = depth_mm_out ? (mmPinned ? (uint8_t *)depth_mm_out : e->pvPin + e->pvMmOff) : nullptr;
  void *bgrDst = nullptr, *mmDst = nullptr;
  if (direct) {
    if (bgrHost) HIP_TRY(hipHostGetDevicePointer(&bgrDst, bgrHost, 0));
    if (mmHost) HIP_TRY(hipHostGetDevicePointer(&mmDst, mmHost, 0));
  } else {
    if (!e->pvDev && (st = dmalloc(&e->pvDev, e->pvMmOff + (size_t)e->P * 2))) return st;
    bgrDst = bgrHost ? e->pvDev : nullptr;
    mmDst = mmHost ? e->pvDev + e->pvMmOff : nullptr;
  }
  {
    StreamSwap sw(e, io);
    LAUNCH(e, "preview_convert", k_previews, dim3(div_up(div_up(e->P, 4), 256)), dim3(256), (const uchar4 *)e->rgb, (const float *)e->depth,
           (uint32_t *)bgrDst, (short *)mmDst, e->P);
    if (!direct) {
      if (bgrHost) HIP_TRY(hipMemcpyAsync(bgrHost, e->pvDev, (size_t)e->P * 3, hipMemcpyDeviceToHost, io));
      if (mmHost) HIP_TRY(hipMemcpyAsync(mmHost, e->pvDev + e->pvMmOff, (size_t)e->P * 2, hipMemcpyDeviceToHost, io));
    }
  }
  HIP_TRY(hipGetLastError());
  if ((st = io_read_done(e, io))) return st;
  HIP_TRY(hipEventSynchronize(e->evViewRead));
  if (bgr_out && !bgrPinned) memcpy(bgr_out, e->pvPin, (size_t)e->P * 3);
  if (depth_mm_out && !mmPinned) memcpy(depth_mm_out, e->pvPin + e->pvMmOff, (size_t)e->P * 2);
  return DSR_OK;
}

}  // extern "C"
