// shim/example_host.cpp — a minimal host written against the ITMLib names of shim/ITMLib.h,
// following the call pattern of dynslam::drivers::InfiniTamDriver (InfiniTamDriver.h:79-300):
// subclass ITMMainEngine, reach into its protected members, drive
// UpdateView -> SetPose -> Integrate -> PrepareNextStep -> Decay -> GetImage.
// It is the compile/run check of the drop-in boundary (OpenCV/Pangolin/Eigen are not installed,
// so the cv::Mat conversions of InfiniTamDriver.cpp:81-163 are replaced by raw buffers).
//
// usage: example_host W H frames [out.obj]   -> prints counters and an FNV-1a hash of the outputs;
//        with out.obj also meshes the scene the way InstanceReconstructor::SaveObjectToMesh does
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ITMLib.h"

class MiniDriver : public ITMMainEngine {
 public:
  MiniDriver(const ITMLibSettings *settings, const ITMRGBDCalib *calib, Vector2i size)
      : ITMMainEngine(settings, calib, size, size), rgb_itm_(new ITMUChar4Image(size, true, true)),
        raw_depth_itm_(new ITMShortImage(size, true, true)) {}
  ~MiniDriver() override { delete rgb_itm_; delete raw_depth_itm_; }

  void UpdateView(const Vector4u *rgba, const short *depth_mm) {  // InfiniTamDriver.cpp:211-224
    size_t n = (size_t)rgb_itm_->noDims.x * rgb_itm_->noDims.y;
    memcpy(rgb_itm_->GetData(MEMORYDEVICE_CPU), rgba, n * sizeof(Vector4u));
    memcpy(raw_depth_itm_->GetData(MEMORYDEVICE_CPU), depth_mm, n * sizeof(short));
    this->viewBuilder->UpdateView(&view, rgb_itm_, raw_depth_itm_, settings->useBilateralFilter, settings->modelSensorNoise);
  }
  void SetPose(const Matrix4f &inv_m) { this->trackingState->pose_d->SetInvM(inv_m); }  // .h:131-134
  void Integrate() {                                                                  // .h:137-146
    WeightParams wp; wp.depthWeighting = false;
    this->denseMapper->SetFusionWeightParams(wp);
    this->denseMapper->ProcessFrame(this->view, this->trackingState, this->scene, this->renderState_live);
  }
  void PrepareNextStep() {                                                            // .h:148-158
    ITMRenderState_VH *rs = (ITMRenderState_VH *)this->renderState_live;
    if (rs->noVisibleBlocks > 0) this->trackingController->Prepare(this->trackingState, this->view, this->renderState_live);
  }
  void Decay(int maxW, int minAge) { denseMapper->Decay(scene, renderState_live, maxW, minAge, false); }  // .h:201-206
  size_t GetUsedMemoryBytes() const {                                                 // .h:241-244
    int used = scene->index.getNumAllocatedVoxelBlocks() - scene->localVBA.lastFreeBlockId;
    return sizeof(ITMVoxel) * SDF_BLOCK_SIZE3 * used;
  }
  size_t GetSavedDecayMemoryBytes() const { return denseMapper->GetDecayedBlockCount() * sizeof(ITMVoxel) * SDF_BLOCK_SIZE3; }
  int NoVisibleBlocks() const { return ((ITMRenderState_VH *)renderState_live)->noVisibleBlocks; }

 private:
  ITMUChar4Image *rgb_itm_;
  ITMShortImage *raw_depth_itm_;
};

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char **argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 160, H = argc > 2 ? atoi(argv[2]) : 120, frames = argc > 3 ? atoi(argv[3]) : 3;
  ITMLibSettings settings;
  settings.sceneParams.voxelSize = 0.05f; settings.sceneParams.mu = 0.2f; settings.sceneParams.maxW = 100;
  settings.sceneParams.viewFrustum_min = 0.2f; settings.sceneParams.viewFrustum_max = 30.0f;
  settings.sdfLocalBlockNum = 20000; settings.hashBucketNum = 0x8000; settings.excessListSize = 0x2000;
  ITMRGBDCalib calib;  // CreateItmCalib (InfiniTamDriver.cpp:49-79)
  calib.intrinsics_rgb.SetFrom(150.0f, 150.0f, W / 2.0f - 0.5f, H / 2.0f - 0.5f, (float)W, (float)H);
  calib.intrinsics_d = calib.intrinsics_rgb;
  Matrix4f identity; identity.setIdentity();
  calib.trafo_rgb_to_depth.SetFrom(identity);
  calib.disparityCalib.SetFrom(1.0f / 1000.0f, 0.0f, ITMDisparityCalib::TRAFO_AFFINE);

  try {
    MiniDriver drv(&settings, &calib, Vector2i(W, H));
    std::vector<Vector4u> rgba((size_t)W * H);
    std::vector<short> depth((size_t)W * H);
    for (int f = 0; f < frames; f++) {
      // a tilted plane with a step, deterministic colours
      for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
          int d = 1800 + 3 * x + 2 * y + ((x / 40) & 1) * 150 - 20 * f;
          depth[(size_t)y * W + x] = (short)((x % 53 == 0) ? 0 : d);
          rgba[(size_t)y * W + x] = Vector4u((uchar)(x * 255 / W), (uchar)(y * 255 / H), (uchar)((x + y + 13 * f) & 255), 255);
        }
      Matrix4f inv_m; inv_m.setIdentity();
      inv_m.at(3, 0) = 0.02f * f; inv_m.at(3, 2) = 0.05f * f;  // translation column
      drv.UpdateView(rgba.data(), depth.data());
      drv.SetPose(inv_m);
      drv.Integrate();
      drv.PrepareNextStep();
      drv.Decay(1, 1);
    }
    ITMUChar4Image out(Vector2i(W, H), true, true);
    ITMFloatImage outf(Vector2i(W, H), true, true);
    drv.GetImage(&out, nullptr, ITMMainEngine::InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME);
    drv.GetImage(nullptr, &outf, ITMMainEngine::InfiniTAM_IMAGE_FREECAMERA_DEPTH);
    drv.GetView()->depth->UpdateHostFromDevice();
    uint64_t h = fnv(out.GetData(MEMORYDEVICE_CPU), (size_t)W * H * 4);
    h = fnv(outf.GetData(MEMORYDEVICE_CPU), (size_t)W * H * 4, h);
    h = fnv(drv.GetView()->depth->GetData(MEMORYDEVICE_CPU), (size_t)W * H * 4, h);
    // the host's preview conversions (InfiniTamDriver.cpp:108-144) through the GPU
    std::vector<short> preview_mm((size_t)W * H);
    std::vector<unsigned char> preview_bgr((size_t)W * H * 3);
    dynslam_shim::ItmDepthToCv(outf, preview_mm.data());
    dynslam_shim::ItmToCv(out, preview_bgr.data());
    h = fnv(preview_mm.data(), (size_t)W * H * 2, h);
    h = fnv(preview_bgr.data(), (size_t)W * H * 3, h);
    unsigned triangles = 0;
    if (argc > 4) {  // InstanceReconstructor.cpp:749-757
      auto *meshing_engine = new ITMMeshingEngine_CUDA<ITMVoxel, ITMVoxelIndex>(settings.sdfLocalBlockNum);
      ITMMesh *mesh = new ITMMesh(MEMORYDEVICE_CUDA, settings.sdfLocalBlockNum);
      meshing_engine->MeshScene(mesh, drv.GetScene());
      mesh->WriteOBJ(argv[4]);
      triangles = mesh->noTotalTriangles;
      delete mesh;
      delete meshing_engine;
    }
    printf("visible=%d used_bytes=%zu saved_bytes=%zu hash=%016llx triangles=%u\n", drv.NoVisibleBlocks(), drv.GetUsedMemoryBytes(),
           drv.GetSavedDecayMemoryBytes(), (unsigned long long)h, triangles);
  } catch (const std::exception &ex) {
    fprintf(stderr, "error: %s\n", ex.what());
    return 1;
  }
  return 0;
}
