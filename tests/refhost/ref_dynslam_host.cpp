// Test host: the reference's COMPLETE per-frame pipeline on the engine library, every DynSLAM class from the
// reference's own, unmodified sources (compiled where they lie under /root/reference by tests/test_reference_pipeline.py):
//
//   dynslam::Input + PrecomputedDepthProvider        (Input.cpp, PrecomputedDepthProvider.cpp: PPM frames, PFM disparity)
//   PrecomputedSegmentationProvider + Mask           (per-detection result / mask text dumps)
//   VisoSparseSFProvider                             (over the SCRIPTED libviso2 stand-in, tests/stubs/libviso2)
//   dynslam::DynSlam::ProcessFrame                   (DynSlam.cpp:16-176)
//   InstanceReconstructor / InstanceTracker / Track  (view split on the CPU, SetView, per-instance InfiniTamDriver,
//                                                     Reap, CompositeInstances, SaveObjectToMesh)
//   drivers::InfiniTamDriver : ITMMainEngine         (shim/ITMLib.h -> include/dsr.h)
//   eval::Evaluation::LogMemoryUse                   (GetUsedMemoryBytes / GetSavedDecayMemoryBytes per frame)
//
// Only this file is ours: it builds the objects like BuildDynSlamKittiOdometry does (DynSLAMGUI.cpp:1100-1270, a GUI unit
// that is not compiled), installs the odometry script, runs N frames and dumps what the engines hold.
//
// usage: ref_dynslam_host <dataset_root> <n_frames> <out.bin> [voxel_size] [decay: 0|1] [evaluate: 0|1] [blocks buckets excess]
//   evaluate = 1: Evaluation::EvaluateFrame after every frame (DynSlam.cpp:153-159) against <dataset_root>/velodyne/%06d.bin
//   <dataset_root>/synthetic.txt : W H fx fy cx cy baseline
//   <dataset_root>/viso/%06d.bin : the script of frame k >= 1 (tests/refhost/make_dataset.py)
// stdout (last line): key=value ...   out.bin: the renders listed in that line, raw, in order.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "DynSlam.h"
#include "Evaluation/Evaluation.h"
#include "InstRecLib/VisoSparseSFProvider.h"
#include "PrecomputedDepthProvider.h"

// flags DEFINEd in the GUI unit (DynSLAMGUI.cpp:36-80) and DECLAREd by the units linked here
DEFINE_bool(semantic_evaluation, false, "");
DEFINE_int32(evaluation_delay, 0, "");
DEFINE_int32(max_decay_weight, 1, "");
DEFINE_int32(fusion_every, 1, "");

namespace {

uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

struct FrameScript {
  double delta[16];  // previous camera -> current camera, row-major
  std::vector<Matcher::p_match> matches;
  std::map<int, std::vector<double>> object_motion;  // object id (carried in p_match::i1c) -> {rx, ry, rz, tx, ty, tz}
};

bool load_script(const std::string &root, int frame, FrameScript &s) {
  FILE *f = fopen(dynslam::utils::Format("%s/viso/%06d.bin", root.c_str(), frame).c_str(), "rb");
  if (!f) return false;
  int32_t nObj = 0, nMatch = 0;
  bool ok = fread(s.delta, 8, 16, f) == 16 && fread(&nObj, 4, 1, f) == 1;
  s.object_motion.clear();
  for (int i = 0; ok && i < nObj; i++) {
    int32_t id = 0, present = 0;
    double tr[6];
    ok = fread(&id, 4, 1, f) == 1 && fread(&present, 4, 1, f) == 1 && fread(tr, 8, 6, f) == 6;
    if (ok && present) s.object_motion[id] = std::vector<double>(tr, tr + 6);
  }
  ok = ok && fread(&nMatch, 4, 1, f) == 1;
  s.matches.resize(ok ? nMatch : 0);
  static_assert(sizeof(Matcher::p_match) == 48, "p_match is 12 packed 4-byte fields");
  ok = ok && fread(s.matches.data(), sizeof(Matcher::p_match), s.matches.size(), f) == s.matches.size();
  fclose(f);
  return ok;
}

// digest of everything an engine holds: hash table + every voxel block
uint64_t engine_digest(dsr_engine *e, dsr_stats *st) {
  dsr_get_stats(e, st);
  std::vector<dsr_hash_entry> ht((size_t)st->no_total_entries);
  dsr_dump_hash_table(e, ht.data());
  uint64_t h = fnv(ht.data(), ht.size() * sizeof(dsr_hash_entry));
  const int chunk = 4096;
  std::vector<dsr_voxel> vox((size_t)chunk * 512);
  for (int b = 0; b < st->num_allocated_voxel_blocks; b += chunk) {
    const int n = std::min(chunk, st->num_allocated_voxel_blocks - b);
    dsr_dump_voxel_blocks(e, b, n, vox.data());
    h = fnv(vox.data(), (size_t)n * 512 * sizeof(dsr_voxel), h);
  }
  return h;
}

}  // namespace

int main(int argc, char **argv) {
  using namespace dynslam;
  using dynslam::utils::Format;
  if (argc < 4) { fprintf(stderr, "usage: %s dataset_root n_frames out.bin [voxel_size] [decay]\n", argv[0]); return 2; }
  const std::string root = argv[1];
  const int nFrames = atoi(argv[2]);
  const char *outPath = argv[3];
  const float voxel = argc > 4 ? (float)atof(argv[4]) : 0.05f;
  const bool decay = argc > 5 && atoi(argv[5]) != 0;
  const bool evaluate = argc > 6 && atoi(argv[6]) != 0;
  const long blocks = argc > 9 ? atol(argv[7]) : 0;  // table sizes for small voxels (default: upstream's constants)
  const int buckets = argc > 9 ? atoi(argv[8]) : 0, excess = argc > 9 ? atoi(argv[9]) : 0;

  int W = 0, H = 0;
  double fx = 0, fy = 0, cx = 0, cy = 0, baseline = 0;
  {
    FILE *f = fopen((root + "/synthetic.txt").c_str(), "r");
    if (!f || fscanf(f, "%d %d %lf %lf %lf %lf %lf", &W, &H, &fx, &fy, &cx, &cy, &baseline) != 7) { fprintf(stderr, "bad synthetic.txt\n"); return 2; }
    fclose(f);
  }

  try {
    // ---- what BuildDynSlamKittiOdometry builds (DynSLAMGUI.cpp:1153-1268) ----------------------------------------------
    Input::Config cfg = Input::KittiOdometryDispnetConfig();  // PFM disparity maps, read_depth = false
    cfg.fname_format = "%06d.ppm";
    Eigen::Matrix34d proj;
    proj(0, 0) = fx; proj(1, 1) = fy; proj(0, 2) = cx; proj(1, 2) = cy; proj(2, 2) = 1.0;
    Eigen::Matrix34d projRight = proj;
    projRight(0, 3) = -fx * baseline;
    Eigen::Matrix4d veloToCam = Eigen::Matrix4d::Identity();
    Eigen::Vector2i frameSize(W, H);
    VoxelDecayParams decayParams(decay, /* min_decay_age */ 3, /* max_decay_weight */ FLAGS_max_decay_weight);
    StereoCalibration stereo((float)baseline, (float)fx);
    Input *input = new Input(root, cfg, nullptr, frameSize, stereo, 0, 1.0f);
    DepthProvider *depth = new PrecomputedDepthProvider(input, root + "/" + cfg.depth_folder, cfg.depth_fname_format, cfg.read_depth, 0,
                                                        cfg.min_depth_m, cfg.max_depth_m);
    input->SetDepthProvider(depth);

    ITMLibSettings *settings = new ITMLibSettings();  // like the GUI: the defaults (shim/ITMLib.h: 5 cm voxels, outdoor frustum) ...
    settings->sceneParams.voxelSize = voxel;          // ... unless the test asks for another voxel size
    settings->sceneParams.mu = 4.0f * voxel;
    if (blocks > 0) { settings->sdfLocalBlockNum = blocks; settings->hashBucketNum = buckets; settings->excessListSize = excess; }
    drivers::InfiniTamDriver *driver = new drivers::InfiniTamDriver(
        settings, drivers::CreateItmCalib(proj, frameSize), drivers::ToItmVec(input->GetRgbSize()), drivers::ToItmVec(input->GetDepthSize()),
        decayParams, false);
    auto *segmentation = new instreclib::segmentation::PrecomputedSegmentationProvider(root + "/" + cfg.segmentation_folder, 0, 1.0f);

    VisualOdometryStereo::parameters sfParams;
    sfParams.base = baseline;
    sfParams.calib.cu = cx; sfParams.calib.cv = cy; sfParams.calib.f = fx;
    auto *sparseSF = new instreclib::VisoSparseSFProvider(sfParams);

    auto *evaluation = new eval::Evaluation(root, input, veloToCam, proj, projRight, (float)baseline, W, H, voxel, false, true, false,
                                            /* separate static / dynamic: the only mode Evaluation.cpp:153 supports */ true);
    FLAGS_enable_evaluation = evaluate;  // LogMemoryUse runs every frame either way
    Vector2i inputShape(W, H);
    DynSlam *dynSlam = new DynSlam(driver, segmentation, sparseSF, evaluation, inputShape, proj.cast<float>(), projRight.cast<float>(),
                                   (float)baseline, false, /* dynamic_mode */ true, FLAGS_fusion_every);

    // ---- the odometry script --------------------------------------------------------------------------------------------
    FrameScript script;
    VisoScript::get().process = [&](int call, Matrix &trDelta, std::vector<Matcher::p_match> &matches) {
      if (call == 0 || !load_script(root, call, script)) { matches.clear(); trDelta = Matrix::eye(4); return false; }
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) trDelta.val[r][c] = script.delta[r * 4 + c];
      matches = script.matches;
      return true;
    };
    VisoScript::get().estimate = [&](const std::vector<Matcher::p_match> &m, const std::vector<double> &) {
      if (m.empty()) return std::vector<double>();
      auto it = script.object_motion.find(m[0].i1c);
      return it == script.object_motion.end() ? std::vector<double>() : it->second;
    };

    // ---- the run ----------------------------------------------------------------------------------------------------------
    for (int i = 0; i < nFrames; i++) {
      dynSlam->ProcessFrame(input);
      if (getenv("REF_HOST_VERBOSE")) {
        cv::Mat3b *rgb; cv::Mat1s *dep;
        input->GetCvImages(&rgb, &dep);
        size_t nz = 0;
        for (int k = 0; k < W * H; k++) nz += reinterpret_cast<short *>(dep->data)[k] > 0;
        dsr_stats fs;
        dsr_get_stats(driver->GetDsrEngine(), &fs);
        fprintf(stderr, "[host] frame %d: input depth valid %zu / %d, visible blocks %d, free head %d, status %d\n", i, nz, W * H,
                fs.no_visible_blocks, fs.last_free_block_id, fs.sticky_status);
      }
    }

    // ---- what the engines hold --------------------------------------------------------------------------------------------
    FILE *out = fopen(outPath, "wb");
    if (!out) { perror(outPath); return 2; }
    const size_t P = (size_t)W * H;
    Eigen::Matrix4f M = dynSlam->GetPoseHistory().back();  // world -> camera of the last frame
    float colMajor[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) colMajor[c * 4 + r] = M(r, c);
    pangolin::OpenGlMatrix mv = pangolin::OpenGlMatrix::ColMajor4x4(colMajor);

    std::vector<unsigned char> colourStatic(P * 4), colourFused(P * 4);
    std::vector<float> depthStatic(P), depthFused(P);
    std::memcpy(colourStatic.data(), dynSlam->GetStaticMapRaycastPreview(mv, PreviewType::kColor, false), P * 4);
    std::memcpy(depthStatic.data(), dynSlam->GetStaticMapRaycastDepthPreview(mv, false), P * 4);
    std::memcpy(colourFused.data(), dynSlam->GetStaticMapRaycastPreview(mv, PreviewType::kColor, true), P * 4);
    std::memcpy(depthFused.data(), dynSlam->GetStaticMapRaycastDepthPreview(mv, true), P * 4);
    fwrite(colourStatic.data(), 1, P * 4, out); fwrite(depthStatic.data(), 4, P, out);
    fwrite(colourFused.data(), 1, P * 4, out); fwrite(depthFused.data(), 4, P, out);
    // the static view after the instances were cut out of it (GetStaticRgbPreview / GetStaticDepthPreview)
    fwrite(dynSlam->GetStaticRgbPreview()->data, 1, P * 3, out);
    fwrite(dynSlam->GetStaticDepthPreview()->data, 2, P, out);

    dsr_stats st;
    const uint64_t staticDigest = engine_digest(driver->GetDsrEngine(), &st);
    std::string line = Format("frames=%d width=%d height=%d static_digest=%016llx static_used_blocks=%d static_decayed=%lld static_memory_bytes=%zu ",
                              dynSlam->GetCurrentFrameNo(), W, H, (unsigned long long)staticDigest,
                              st.num_allocated_voxel_blocks - 1 - st.last_free_block_id, (long long)st.decayed_block_count, dynSlam->GetStaticMapMemoryBytes());
    line += Format("colour_static=%016llx depth_static=%016llx colour_fused=%016llx depth_fused=%016llx ",
                   (unsigned long long)fnv(colourStatic.data(), P * 4), (unsigned long long)fnv(depthStatic.data(), P * 4),
                   (unsigned long long)fnv(colourFused.data(), P * 4), (unsigned long long)fnv(depthFused.data(), P * 4));

    // every preview type the GUI cycles through (PreviewType.h; InfiniTamDriver.cpp:16-34 maps them onto GetImage types)
    {
      const PreviewType types[] = {PreviewType::kGray, PreviewType::kNormal, PreviewType::kWeight, PreviewType::kLatestRaycast, PreviewType::kDepth};
      const char *names[] = {"gray", "normal", "weight", "latest_raycast", "depth_as_colour"};
      for (int k = 0; k < 5; k++) {
        const unsigned char *img = dynSlam->GetStaticMapRaycastPreview(mv, types[k], k % 2 == 0);
        size_t lit = 0;
        for (size_t q = 0; q < P; q++) lit += (img[q * 4] | img[q * 4 + 1] | img[q * 4 + 2]) != 0;
        line += Format("preview_%s=%016llx:lit%zu ", names[k], (unsigned long long)fnv(img, P * 4), lit);
      }
    }

    auto &tracker = dynSlam->GetInstanceReconstructor()->GetInstanceTracker();
    line += Format("tracks=%d ", tracker.GetActiveTrackCount());
    std::vector<int> reconstructed;
    for (const auto &kv : tracker.GetActiveTracks()) {
      const instreclib::reconstruction::Track &track = kv.second;
      dsr_stats ist;
      std::memset(&ist, 0, sizeof(ist));
      uint64_t dig = 0;
      if (track.HasReconstruction()) {
        dig = engine_digest(const_cast<instreclib::reconstruction::Track &>(track).GetReconstruction()->GetDsrEngine(), &ist);
        reconstructed.push_back(kv.first);
      }
      const auto &bb = track.GetLastFrame().instance_view.GetInstanceDetection().GetCopyBoundingBox();
      line += Format("track%d=%s:frames%zu:recon%d:digest%016llx:used%d:bbox%d,%d,%d,%d ", kv.first, track.GetStateLabel().c_str(), track.GetSize(),
                     track.HasReconstruction() ? 1 : 0, (unsigned long long)dig,
                     track.HasReconstruction() ? ist.num_allocated_voxel_blocks - 1 - ist.last_free_block_id : 0, bb.r.x0, bb.r.y0, bb.r.x1, bb.r.y1);
    }
    // per reconstructed object: its volume raycast (colour + float depth) from the pose of its last fused frame
    // (Track::GetFramePose: object frame -> last camera, what FuseFrame handed to SetPose), then the latest instance view
    // (what FuseFrame fused: the cut-out depth) — the two must agree where both are valid
    ITMUChar4Image objColour(Vector2i(W, H), true, true);
    ITMFloatImage objDepth(Vector2i(W, H), true, true);
    for (int id : reconstructed) {
      instreclib::reconstruction::Track &track = tracker.GetTrack(id);
      auto pose = track.GetFramePose(track.GetSize() - 1);
      pangolin::OpenGlMatrix objectView = pangolin::OpenGlMatrix::ColMajor4x4(pose.Get().data());
      track.GetReconstruction()->GetImage(&objColour, PreviewType::kColor, objectView);
      track.GetReconstruction()->GetFloatImage(&objDepth, PreviewType::kDepth, objectView);
      fwrite(objColour.GetData(MEMORYDEVICE_CPU), 1, P * 4, out);
      fwrite(objDepth.GetData(MEMORYDEVICE_CPU), 4, P, out);
      line += Format("object%d_raycast=%016llx object%d_raycast_depth=%016llx ", id, (unsigned long long)fnv(objColour.GetData(MEMORYDEVICE_CPU), P * 4), id,
                     (unsigned long long)fnv(objDepth.GetData(MEMORYDEVICE_CPU), P * 4));
      const float *d = dynSlam->GetObjectDepthPreview(id);
      fwrite(d, 4, P, out);
      line += Format("object%d_view_depth=%016llx ", id, (unsigned long long)fnv(d, P * 4));
      // the same object through DynSlam's own preview call (GUI path), from the last camera
      line += Format("object%d_preview=%016llx ", id, (unsigned long long)fnv(dynSlam->GetObjectRaycastPreview(id, mv, PreviewType::kColor), P * 4));
    }
    fclose(out);

    // ---- the on-disk outputs (DynSlam.cpp:188-212, InstanceReconstructor.cpp:736-763) --------------------------------------
    dynSlam->SaveStaticMap("synthetic", depth->GetName());
    for (int id : reconstructed) dynSlam->SaveDynamicObject("synthetic", depth->GetName(), id);
    // a gap of two frames triggers Reap on the next ProcessReconstructions; ForceDynamicObjectCleanup does it now
    if (!reconstructed.empty()) {
      dynSlam->ForceDynamicObjectCleanup(reconstructed[0]);
      dsr_stats ist;
      engine_digest(tracker.GetTrack(reconstructed[0]).GetReconstruction()->GetDsrEngine(), &ist);
      line += Format("object%d_used_after_reap=%d ", reconstructed[0], ist.num_allocated_voxel_blocks - 1 - ist.last_free_block_id);
    }
    // what InstanceTracker::PruneTracks does when a track expires after 50 frames without a detection
    // (InstanceTracker.cpp:37-58): the track's engine dies, the host keeps (and may still sync) the views of its frames
    if (!reconstructed.empty()) {
      instreclib::reconstruction::Track &t = tracker.GetTrack(reconstructed.back());
      ITMView *lastView = t.GetLastFrame().instance_view.GetView();
      t.GetReconstruction()->SetView(nullptr);
      t.GetReconstruction().reset();  // ~InfiniTamDriver -> ~ITMMainEngine -> dsr_engine_destroy
      lastView->rgb->UpdateHostFromDevice();  // must not reach the dead engine
      lastView->depth->UpdateHostFromDevice();
      line += Format("pruned_track=%d pruned_view_detached=%d pruned_has_reconstruction=%d ", reconstructed.back(), lastView->owner == nullptr ? 1 : 0,
                     t.HasReconstruction() ? 1 : 0);
    }
    // end of sequence (DynSLAMGUI.cpp's "decay catch-up" button): GC of every queued visible list, then wait for mesh jobs
    dynSlam->StaticMapDecayCatchup();
    dynSlam->WaitForJobs();
    dsr_get_stats(driver->GetDsrEngine(), &st);
    line += Format("static_decayed_after_catchup=%lld static_saved_decay_bytes=%zu ", (long long)st.decayed_block_count,
                   dynSlam->GetStaticMapSavedDecayMemoryBytes());
    {  // the GUI's memory read-out (DynSLAMGUI.cpp:909-915)
      size_t freeBytes = 0, totalBytes = 0;
      cudaMemGetInfo(&freeBytes, &totalBytes);
      fprintf(stderr, "[host] device memory: %.1f GiB free of %.1f GiB\n", freeBytes / 1073741824.0, totalBytes / 1073741824.0);
      if (totalBytes == 0 || freeBytes > totalBytes) throw std::runtime_error("cudaMemGetInfo returned nonsense");
    }
    printf("%s\n", line.c_str());
    fflush(stdout);
    delete dynSlam;
    delete input;
    return 0;
  } catch (const std::exception &ex) {
    fprintf(stderr, "ref_dynslam_host: %s\n", ex.what());
    return 1;
  }
}
