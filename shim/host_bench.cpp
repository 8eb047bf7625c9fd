// shim/host_bench.cpp — a C++ host that drives the engines the way DynSLAM does, frame by frame,
// through the ITMLib names of shim/ITMLib.h, and reports what the call sequence sustains
// ("through-shim" frames/s) next to a digest of everything it produced.
//
// Two builds of the SAME main():
//   * default: `HostDriver` below — our own driver class written against the shim, raw BGR / int16
//     buffers, the layout conversions of InfiniTamDriver.cpp:81-144 done on the GPU
//     (dynslam_shim::CvToItm / ItmToCv / ItmDepthToCv).  bench.py's `through_shim` leg runs this one.
//   * -DDSR_HOST_REFERENCE_DRIVER (tests/test_reference_compiles.py; needs /root/reference at BUILD
//     time only): the reference's UNMODIFIED `dynslam::drivers::InfiniTamDriver`
//     (src/DynSLAM/InfiniTamDriver.{h,cpp}, compiled from where they lie against shim/ITMLib.h and
//     the stand-in third-party headers of tests/stubs/) — the proof that the reference's host code
//     runs on the HIP engines unchanged.
//
// usage: host_bench frames.bin W H fx fy cx cy n_frames warmup voxel mu blocks buckets excess [decay_max_w decay_min_age]
//        [--masks masks.bin n_instances]   (our HostDriver only) BASELINE configs[2]: every frame's instance silhouettes are
//        cut out of the static view ON THE GPU (dsr_view_extract_silhouette / dsr_view_remove_silhouette — what replaces
//        ProcessSilhouette_CPU / RemoveSilhouette_CPU and their D2H/H2D round trip, InstanceReconstructor.cpp:59-197) and fused
//        into their own volumes (0.035 m, mu 1.0, 7142 blocks: InstanceReconstructor.cpp:372-379) by further HostDrivers.
//        [--devices d0,d1,...]  (with --masks) one volume per GPU through the C ABI: the static map lives on d0, instance k on
//        d[1 + k mod (n-1)] (n = 1: everything on d0) — ITMLibSettings::deviceIndex, what dynslam_shim::PlaceVolume derives from
//        DSR_DEVICES for the reference's unmodified InstanceReconstructor::InitializeReconstruction (:363-392); the view split
//        crosses GPUs inside dsr_view_extract_silhouette; the fused preview (CompositeInstances, :933-990) is served by a
//        dsr_exchange: every volume raycast on its own GPU into its slot, ONE RCCL all-gather, composite on d0.  Devices may repeat
//        ("0,0": two ranks on one GPU — the exchange degenerates to buffers in place).
//        [--preview]  the fused preview inside the timed loop, every frame (without --devices: GetImage + GetFloatImage per volume
//        to the host and dsr_composite_instances — the reference's own flow)
//   masks.bin: per frame int32 n; per mask int32 k, x0, y0, bw, bh; float rel[16] (camera->object, ROW-major); u8 mask[bw*bh]
//   frames.bin: per frame  BGR u8[H*W*3], depth int16[H*W] (mm), pose float[16] (camera->world, ROW-major);
//               then float[16]: model-view matrix (world->camera, row-major) of the final free-view render
// prints one line: key=value ... (frames_per_s over the frames after `warmup`, FNV-1a digest of the
// final free-view colour + float depth renders, the view depth and the previews)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include <sched.h>  // sched_getcpu: where the host thread ran (diagnostic keys of the output line)

#ifdef DSR_HOST_REFERENCE_DRIVER
#include "InfiniTamDriver.h"  // the reference's own header (-I /root/reference/src/DynSLAM)
using RefDriver = dynslam::drivers::InfiniTamDriver;
#else
#include "ITMLib.h"
#endif

namespace {

uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

#ifndef DSR_HOST_REFERENCE_DRIVER
// Our own host-side driver: same sequence of ITMLib calls as the reference's InfiniTamDriver
// (InfiniTamDriver.h:79-300), raw buffers instead of cv::Mat.
class HostDriver : public ITMMainEngine {
 public:
  HostDriver(const ITMLibSettings *s, const ITMRGBDCalib *c, Vector2i size, bool decay, int decayMaxW, int decayMinAge)
      : ITMMainEngine(s, c, size, size), rgb_(size, true, true), rawDepth_(size, true, true), previewMm_((size_t)size.x * size.y),
        previewBgr_((size_t)size.x * size.y * 3), decay_(decay), decayMaxW_(decayMaxW), decayMinAge_(decayMinAge) {
    // the previews live as long as the driver (the reference keeps them in cv::Mat members): page-locked once, every frame's
    // read-back then lands in them directly
    if (!std::getenv("DSR_HOST_PAGEABLE_PREVIEWS")) {
      pinned_ = dsr_pin_host_buffer(previewBgr_.data(), previewBgr_.size()) == DSR_OK &&
                dsr_pin_host_buffer(previewMm_.data(), previewMm_.size() * sizeof(short)) == DSR_OK;
    }
  }
  ~HostDriver() {
    if (pinned_) { dsr_unpin_host_buffer(previewBgr_.data()); dsr_unpin_host_buffer(previewMm_.data()); }
  }
  void UpdateView(const unsigned char *bgr, const short *depthMm) {  // InfiniTamDriver.cpp:211-224
    static const bool twoStep = std::getenv("DSR_HOST_TWO_STEP_UPDATE") != nullptr;  // A/B: the reference's two-step form
    if (twoStep) {
      dynslam_shim::CvToItm(bgr, rawDepth_.noDims.y, rawDepth_.noDims.x, &rgb_);
      std::memcpy(rawDepth_.GetData(MEMORYDEVICE_CPU), depthMm, rawDepth_.dataSize * sizeof(short));
      viewBuilder->UpdateView(&view, &rgb_, &rawDepth_, settings->useBilateralFilter, settings->modelSensorNoise);
    } else {
      // CvToItm + viewBuilder->UpdateView in one call: the BGR frame goes up as it is, the conversion runs in the ingest kernel
      viewBuilder->UpdateViewBgr(&view, bgr, depthMm, rawDepth_.noDims);
    }
  }
  void SetPose(const Matrix4f &invM) { trackingState->pose_d->SetInvM(invM); }  // .h:131-134
  void Integrate() {                                                         // .h:137-146
    denseMapper->SetFusionWeightParams(weights_);
    denseMapper->ProcessFrame(view, trackingState, scene, renderState_live);
  }
  void PrepareNextStep() {  // .h:148-158 (incl. the two preview conversions)
    if (static_cast<ITMRenderState_VH *>(renderState_live)->noVisibleBlocks > 0) {
      trackingController->Prepare(trackingState, view, renderState_live);
      if (view->owner == GetDsrEngine() && !view->deviceStale)  // the view is current in HBM: both previews from there, one sync
        ITMLib::Engine::dsr_throw(dsr_get_view_previews(GetDsrEngine(), previewBgr_.data(), previewMm_.data()));
      else {
        dynslam_shim::ItmToCv(*view->rgb, previewBgr_.data());
        dynslam_shim::ItmDepthToCv(*view->depth, previewMm_.data());
      }
    }
  }
  void Decay() { if (decay_) denseMapper->Decay(scene, renderState_live, decayMaxW_, decayMinAge_, false); }  // .h:201-206
  // the engine's device view has just been written by dsr_view_extract_silhouette: give the host-side ITMView that
  // names it (InstanceReconstructor.cpp:580 SetView) without uploading stale host buffers over it
  void AdoptDeviceView() {
    if (!view) view = new ITMView(viewBuilder->GetCalib(), rgb_.noDims, rawDepth_.noDims, true);
    view->bind(GetDsrEngine());
    view->deviceStale = false;
  }
  size_t GetUsedMemoryBytes() const {                                                                       // .h:241-244
    return sizeof(ITMVoxel) * SDF_BLOCK_SIZE3 * (size_t)(scene->index.getNumAllocatedVoxelBlocks() - scene->localVBA.lastFreeBlockId);
  }
  size_t GetSavedDecayMemoryBytes() const { return denseMapper->GetDecayedBlockCount() * sizeof(ITMVoxel) * SDF_BLOCK_SIZE3; }
  void Render(ITMUChar4Image *out, ITMFloatImage *outF, const Matrix4f &M) {  // InfiniTamDriver.cpp:165-209
    ITMPose pose; pose.SetM(M); pose.Coerce();
    ITMIntrinsics intr = viewBuilder->GetCalib()->intrinsics_d;
    if (out) GetImage(out, nullptr, InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, &pose, &intr);
    if (outF) GetImage(nullptr, outF, InfiniTAM_IMAGE_FREECAMERA_DEPTH, &pose, &intr);
  }

 private:
  ITMUChar4Image rgb_;
  ITMShortImage rawDepth_;
  std::vector<short> previewMm_;
  std::vector<unsigned char> previewBgr_;
  WeightParams weights_;
  bool decay_;
  int decayMaxW_, decayMinAge_;
  bool pinned_ = false;
};
#endif

}  // namespace

int main(int argc, char **argv) {
  if (argc < 15) {
    fprintf(stderr, "usage: %s frames.bin W H fx fy cx cy n_frames warmup voxel mu blocks buckets excess [decay_max_w decay_min_age]\n", argv[0]);
    return 2;
  }
  const char *path = argv[1];
  const int W = atoi(argv[2]), H = atoi(argv[3]), nFrames = atoi(argv[8]), warmup = atoi(argv[9]);
  const float fx = (float)atof(argv[4]), fy = (float)atof(argv[5]), cx = (float)atof(argv[6]), cy = (float)atof(argv[7]);
  const char *masksPath = nullptr;
  int nInstances = 0;
  std::vector<int> devices;
  bool previewEveryFrame = false;
  for (int a = argc - 1; a >= 15; a--) {  // trailing options, stripped from the positional list
    if (strcmp(argv[a], "--preview") == 0 && a == argc - 1) { previewEveryFrame = true; argc = a; }
    else if (a + 1 < argc && strcmp(argv[a], "--devices") == 0 && a + 2 == argc) {
      for (const char *p = argv[a + 1]; *p;) { char *end = nullptr; devices.push_back((int)strtol(p, &end, 10)); if (end == p) break; p = (*end == ',') ? end + 1 : end; }
      argc = a;
    }
  }
  for (int a = 15; a + 2 < argc + 0; a++)
    if (strcmp(argv[a], "--masks") == 0) { masksPath = argv[a + 1]; nInstances = atoi(argv[a + 2]); argc = a; break; }
  const bool decay = argc > 16;
  const int decayMaxW = decay ? atoi(argv[15]) : 0, decayMinAge = decay ? atoi(argv[16]) : 0;
  const size_t P = (size_t)W * H;

  // The host thread next to the GPU's NUMA node BEFORE the frame buffers are allocated and touched (first touch places them): the
  // same command line read 223-275 or 439-460 frames/s at configs[2] depending on which process started it —
  // a child inherits its parent's CPU affinity and memory policy.  DSR_HOST_NO_PIN=1: leave the thread where it is (A/B).
  if (!std::getenv("DSR_HOST_NO_PIN")) (void)dsr_pin_host_thread(devices.empty() ? -1 : devices[0]);
  ITMLibSettings settings;
  settings.sceneParams.voxelSize = (float)atof(argv[10]); settings.sceneParams.mu = (float)atof(argv[11]);
  settings.sceneParams.maxW = 100; settings.sceneParams.viewFrustum_min = 0.2f; settings.sceneParams.viewFrustum_max = 30.0f;
  settings.sdfLocalBlockNum = atol(argv[12]); settings.hashBucketNum = atoi(argv[13]); settings.excessListSize = atoi(argv[14]);

  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); return 2; }
  std::vector<std::vector<unsigned char>> bgr(nFrames);
  std::vector<std::vector<short>> dep(nFrames);
  std::vector<float> poses((size_t)nFrames * 16);
  for (int i = 0; i < nFrames; i++) {
    bgr[i].resize(P * 3); dep[i].resize(P);
    if (fread(bgr[i].data(), 1, P * 3, f) != P * 3 || fread(dep[i].data(), 2, P, f) != P || fread(&poses[(size_t)i * 16], 4, 16, f) != 16) {
      fprintf(stderr, "%s: short read at frame %d\n", path, i);
      return 2;
    }
  }
  float renderM[16];  // trailer: the model-view matrix (world->camera, row-major) of the final free-view render
  if (fread(renderM, 4, 16, f) != 16) { fprintf(stderr, "%s: render pose missing\n", path); return 2; }
  fclose(f);
  struct MaskRec { int k, x0, y0, bw, bh; float rel[16]; std::vector<unsigned char> bits; };
  std::vector<std::vector<MaskRec>> masks(nFrames);
  if (masksPath) {
#ifdef DSR_HOST_REFERENCE_DRIVER
    fprintf(stderr, "--masks needs the HostDriver build (the reference's driver splits views on the CPU)\n");
    return 2;
#endif
    FILE *mf = fopen(masksPath, "rb");
    if (!mf) { perror(masksPath); return 2; }
    for (int i = 0; i < nFrames; i++) {
      int n = 0;
      if (fread(&n, 4, 1, mf) != 1) { fprintf(stderr, "%s: short read\n", masksPath); return 2; }
      masks[i].resize(n);
      for (auto &m : masks[i]) {
        int hdr[5];
        if (fread(hdr, 4, 5, mf) != 5 || fread(m.rel, 4, 16, mf) != 16) { fprintf(stderr, "%s: short read\n", masksPath); return 2; }
        m.k = hdr[0]; m.x0 = hdr[1]; m.y0 = hdr[2]; m.bw = hdr[3]; m.bh = hdr[4];
        m.bits.resize((size_t)m.bw * m.bh);
        if (fread(m.bits.data(), 1, m.bits.size(), mf) != m.bits.size()) { fprintf(stderr, "%s: short read\n", masksPath); return 2; }
      }
    }
    fclose(mf);
  }

  try {
#ifdef DSR_HOST_REFERENCE_DRIVER
    Eigen::Matrix<double, 3, 4> proj;  // CreateItmCalib (InfiniTamDriver.cpp:49-79) reads fx, fy, cx, cy from P2
    proj(0, 0) = fx; proj(1, 1) = fy; proj(0, 2) = cx; proj(1, 2) = cy; proj(2, 2) = 1.0;
    ITMRGBDCalib *calib = dynslam::drivers::CreateItmCalib(proj, Eigen::Vector2i(W, H));
    RefDriver drv(&settings, calib, Vector2i(W, H), Vector2i(W, H), dynslam::VoxelDecayParams(decay, decayMinAge, decayMaxW), false);
    cv::Mat3b rgbCv(H, W);
    cv::Mat1s depthCv(H, W);
#else
    ITMRGBDCalib calibStore, *calib = &calibStore;
    calib->intrinsics_rgb.SetFrom(fx, fy, cx, cy, (float)W, (float)H);
    calib->intrinsics_d = calib->intrinsics_rgb;
    Matrix4f identity; identity.setIdentity();
    calib->trafo_rgb_to_depth.SetFrom(identity);
    calib->disparityCalib.SetFrom(1.0f / 1000.0f, 0.0f, ITMDisparityCalib::TRAFO_AFFINE);
    // one volume per GPU: rank 0 = the static map on devices[0], instance k on rank 1 + k mod (n - 1)
    const int nRanks = devices.empty() ? 1 : (int)devices.size();
    auto rankOfInstance = [&](int k) { return nRanks > 1 ? 1 + k % (nRanks - 1) : 0; };
    if (!devices.empty()) settings.deviceIndex = devices[0];
    HostDriver drv(&settings, calib, Vector2i(W, H), decay, decayMaxW, decayMinAge);
    // one volume per tracked instance (InstanceReconstructor.cpp:363-389)
    std::vector<ITMLibSettings> instSettings(std::max(1, nInstances), settings);
    std::vector<std::unique_ptr<HostDriver>> inst;
    std::vector<int> slotOfInstance(nInstances, 0), perRank(nRanks, 0);
    for (int k = 0; k < nInstances; k++) {
      ITMLibSettings &is = instSettings[k];
      is.sceneParams.voxelSize = 0.035f; is.sceneParams.mu = 1.0f;
      is.sdfLocalBlockNum = 7142; is.hashBucketNum = 0x100000; is.excessListSize = 0x20000;
      if (!devices.empty()) is.deviceIndex = devices[rankOfInstance(k)];
      slotOfInstance[k] = perRank[rankOfInstance(k)]++;
      inst.emplace_back(new HostDriver(&is, calib, Vector2i(W, H), false, 0, 0));
    }
    // the fused preview (CompositeInstances, InstanceReconstructor.cpp:933-990): through the exchange with --devices, else the
    // reference's own flow (every volume's colour + depth to the host, composited there by dsr_composite_instances)
    dsr_exchange *xch = nullptr;
    if (!devices.empty() && nInstances > 0) {
      std::vector<int32_t> dv(devices.begin(), devices.end());
      const int slots = std::max(1, *std::max_element(perRank.begin(), perRank.end()));
      ITMLib::Engine::dsr_throw(dsr_exchange_create(dv.data(), nRanks, slots, W * H, &xch));
    }
    std::vector<unsigned char> compRgba(P * 4, 0), layerRgba;
    std::vector<float> compDepth(P, 0.0f), layerDepth;
    auto fusedPreview = [&](const Matrix4f &M, const std::vector<MaskRec> &frameMasks, bool readBack) {
      // object -> camera of the preview camera = the model view composed with the instance's pose (:923,968)
      std::vector<int32_t> ranks, slots, tids;
      std::vector<Matrix4f> poses;
      for (const auto &m : frameMasks) {
        Matrix4f rel, relInv;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) rel.at(c, r) = m.rel[r * 4 + c];
        rel.inv(relInv);
        poses.push_back(relInv); ranks.push_back(rankOfInstance(m.k)); slots.push_back(slotOfInstance[m.k]); tids.push_back(1 + m.k);
      }
      for (size_t a = 0; a + 1 < tids.size(); a++)  // ascending track id, the order of the host's loop over its tracks
        for (size_t b = a + 1; b < tids.size(); b++)
          if (tids[b] < tids[a]) { std::swap(tids[a], tids[b]); std::swap(ranks[a], ranks[b]); std::swap(slots[a], slots[b]); std::swap(poses[a], poses[b]); }
      if (xch) {
        void *tr = nullptr, *td = nullptr;
        ITMLib::Engine::dsr_throw(dsr_exchange_target_ptrs(xch, 0, &tr, &td));
        ITMLib::Engine::dsr_throw(dsr_wait_for_stream(drv.GetDsrEngine(), dsr_exchange_stream(xch, 0)));
        ITMLib::Engine::dsr_throw(dsr_get_image_dev(drv.GetDsrEngine(), DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, M.m, nullptr, tr, td));
        for (size_t l = 0; l < tids.size(); l++) {
          const int k = tids[l] - 1;
          ITMLib::Engine::dsr_throw(dsr_exchange_render_slot(xch, ranks[l], slots[l], inst[k]->GetDsrEngine(), DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME,
                                                             poses[l].m, nullptr));
        }
        ITMLib::Engine::dsr_throw(dsr_exchange_gather_and_composite(xch, 0, drv.GetDsrEngine(), nullptr, nullptr, ranks.data(), slots.data(), tids.data(),
                                                                    (int)tids.size(), 1.0f, 1));
        if (readBack) ITMLib::Engine::dsr_throw(dsr_exchange_read_target(xch, 0, compRgba.data(), compDepth.data()));
      } else {
        ITMLib::Engine::dsr_throw(dsr_get_image(drv.GetDsrEngine(), DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, M.m, nullptr, compRgba.data(), compDepth.data()));
        layerRgba.resize(P * 4 * std::max<size_t>(1, tids.size())); layerDepth.resize(P * std::max<size_t>(1, tids.size()));
        for (size_t l = 0; l < tids.size(); l++)
          ITMLib::Engine::dsr_throw(dsr_get_image(inst[tids[l] - 1]->GetDsrEngine(), DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, poses[l].m, nullptr,
                                                  layerRgba.data() + l * P * 4, layerDepth.data() + l * P));
        ITMLib::Engine::dsr_throw(dsr_composite_instances(compRgba.data(), compDepth.data(), layerRgba.data(), layerDepth.data(), tids.data(),
                                                          (int)tids.size(), (int)P, 1.0f, 1));
      }
    };
#endif
    // where the host's time goes, per call site, over the timed frames (diagnostic keys host_ms_*: a clock read per call)
    enum { H_UPDATE, H_SPLIT, H_INST_INTEGRATE, H_INST_PREPARE, H_INTEGRATE, H_PREPARE, H_N };
    double hostMs[H_N] = {0, 0, 0, 0, 0, 0};
    auto lap = [&](std::chrono::steady_clock::time_point &from, int slot, bool timed) {
      const auto now = std::chrono::steady_clock::now();
      if (timed) hostMs[slot] += std::chrono::duration<double, std::milli>(now - from).count();
      from = now;
    };
    const int cpuFirst = sched_getcpu();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nFrames; i++) {
      if (i == warmup) t0 = std::chrono::steady_clock::now();
      const bool timed = i >= warmup;
      auto tl = std::chrono::steady_clock::now();
      const float *T = &poses[(size_t)i * 16];
#ifdef DSR_HOST_REFERENCE_DRIVER
      std::memcpy(rgbCv.data, bgr[i].data(), P * 3);
      std::memcpy(depthCv.data, dep[i].data(), P * 2);
      Eigen::Matrix4f pose;
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose(r, c) = T[r * 4 + c];
      drv.UpdateView(rgbCv, depthCv);
      drv.SetPose(pose);
      lap(tl, H_UPDATE, timed);
#else
      Matrix4f invM;
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) invM.at(c, r) = T[r * 4 + c];
      drv.UpdateView(bgr[i].data(), dep[i].data());
      lap(tl, H_UPDATE, timed);
      for (const auto &m : masks[i]) {  // the view split of InstanceReconstructor::ProcessFrame (:238-263), on the GPU
        HostDriver &id = *inst[m.k];
        // cut-out + blanking of this instance in ONE launch (dsr_view_split_silhouette; the two-call form gives the same views)
        ITMLib::Engine::dsr_throw(dsr_view_split_silhouette(drv.GetDsrEngine(), id.GetDsrEngine(), m.bits.data(), m.x0, m.y0, m.bw, m.bh,
                                                            m.bits.data(), m.x0, m.y0, m.bw, m.bh));
        Matrix4f rel;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) rel.at(c, r) = m.rel[r * 4 + c];
        id.AdoptDeviceView();
        id.SetPose(rel);
        lap(tl, H_SPLIT, timed);
        id.Integrate();
        lap(tl, H_INST_INTEGRATE, timed);
        id.PrepareNextStep();
        lap(tl, H_INST_PREPARE, timed);
      }
      drv.SetPose(invM);
#endif
      drv.Integrate();
      lap(tl, H_INTEGRATE, timed);
      drv.PrepareNextStep();
      drv.Decay();
      lap(tl, H_PREPARE, timed);
#ifndef DSR_HOST_REFERENCE_DRIVER
      if (previewEveryFrame && masksPath) {
        Matrix4f Mv;
        invM.inv(Mv);
        fusedPreview(Mv, masks[i], false);
      }
#endif
    }
    // GetUsedMemoryBytes reads the free-list head from the device: it also drains the stream
    size_t instUsed = 0;
#ifndef DSR_HOST_REFERENCE_DRIVER
    for (auto &id : inst) instUsed += id->GetUsedMemoryBytes();
#endif
    const size_t used = drv.GetUsedMemoryBytes();
#ifndef DSR_HOST_REFERENCE_DRIVER
    if (xch) ITMLib::Engine::dsr_throw(dsr_exchange_sync(xch));
#endif
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // a free-view colour + depth render from the pose in the file's trailer
    ITMUChar4Image out(Vector2i(W, H), true, true);
    ITMFloatImage outF(Vector2i(W, H), true, true);
    Matrix4f M;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M.at(c, r) = renderM[r * 4 + c];
#ifdef DSR_HOST_REFERENCE_DRIVER
    pangolin::OpenGlMatrix mv;
    for (int k = 0; k < 16; k++) mv.m[k] = M.m[k];
    drv.GetImage(&out, dynslam::PreviewType::kColor, mv);
    drv.GetFloatImage(&outF, dynslam::PreviewType::kDepth, mv);
#else
    drv.Render(&out, &outF, M);
#endif
    drv.GetView()->depth->UpdateHostFromDevice();
    uint64_t h = fnv(out.GetData(MEMORYDEVICE_CPU), P * 4);
    h = fnv(outF.GetData(MEMORYDEVICE_CPU), P * 4, h);
    h = fnv(drv.GetView()->depth->GetData(MEMORYDEVICE_CPU), P * 4, h);
    // the host's preview conversions of the renders (InfiniTamDriver.cpp:108-144)
    std::vector<short> mm(P);
    std::vector<unsigned char> pbgr(P * 3);
#ifdef DSR_HOST_REFERENCE_DRIVER
    cv::Mat1s mmCv(H, W);
    cv::Mat3b bgrCv(H, W);
    dynslam::drivers::ItmDepthToCv(outF, &mmCv);  // the reference's CPU loops
    dynslam::drivers::ItmToCv(out, &bgrCv);
    std::memcpy(mm.data(), mmCv.data, P * 2);
    std::memcpy(pbgr.data(), bgrCv.data, P * 3);
#else
    dynslam_shim::ItmDepthToCv(outF, mm.data());
    dynslam_shim::ItmToCv(out, pbgr.data());
#endif
    h = fnv(mm.data(), P * 2, h);
    h = fnv(pbgr.data(), P * 3, h);
    unsigned long long compositeHash = 0;
#ifndef DSR_HOST_REFERENCE_DRIVER
    if (masksPath && nFrames > 0) {  // the fused preview of the last frame, from its own camera
      Matrix4f lastInv, lastM;
      const float *T = &poses[(size_t)(nFrames - 1) * 16];
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) lastInv.at(c, r) = T[r * 4 + c];
      lastInv.inv(lastM);
      fusedPreview(lastM, masks[nFrames - 1], true);
      compositeHash = fnv(compDepth.data(), P * 4, fnv(compRgba.data(), P * 4));
    }
    if (xch) dsr_exchange_destroy(xch);
#endif
    printf("driver=%s frames=%d timed=%d frames_per_s=%.3f ms_per_frame=%.4f used_bytes=%zu saved_bytes=%zu instances=%d inst_used_bytes=%zu composite_hash=%016llx ranks=%d hash=%016llx",
#ifdef DSR_HOST_REFERENCE_DRIVER
           "reference",
#else
           "shim",
#endif
           nFrames, nFrames - warmup, (nFrames - warmup) / secs, 1e3 * secs / (nFrames - warmup), used, drv.GetSavedDecayMemoryBytes(),
           nInstances, instUsed, compositeHash, (int)(devices.empty() ? 1 : devices.size()), (unsigned long long)h);
    {
      const int nTimed = nFrames - warmup > 0 ? nFrames - warmup : 1;
      printf(" host_ms_update=%.4f host_ms_split=%.4f host_ms_inst_integrate=%.4f host_ms_inst_prepare=%.4f host_ms_integrate=%.4f host_ms_prepare=%.4f cpu=%d-%d\n",
             hostMs[H_UPDATE] / nTimed, hostMs[H_SPLIT] / nTimed, hostMs[H_INST_INTEGRATE] / nTimed, hostMs[H_INST_PREPARE] / nTimed,
             hostMs[H_INTEGRATE] / nTimed, hostMs[H_PREPARE] / nTimed, cpuFirst, sched_getcpu());
    }
  } catch (const std::exception &ex) {
    fprintf(stderr, "error: %s\n", ex.what());
    return 1;
  }
  return 0;
}
