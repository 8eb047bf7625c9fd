// oracle/ref_edges.cpp — the REFERENCE'S OWN code for the edges of the path, behind a C ABI.  TEST INFRASTRUCTURE.
//
// The engines of the hot path are an empty submodule of /root/reference, but the host loops right
// before and after the path ARE in its tree.  This file #includes the reference's unmodified
// InstRecLib/InstanceReconstructor.cpp (file-scope templates ProcessSilhouette_CPU / RemoveSilhouette_CPU and the
// free functions CompositeDepth / CompositeColor become visible to the wrappers below) and is linked with the
// unmodified InfiniTamDriver.cpp (CvToItm, ItmToCv, FloatDepthmapToShort) and the other translation units they
// need — compiled from where they lie against shim/ITMLib.h and the stand-in third-party headers of tests/stubs/
// (oracle/Makefile target `_ref`, output only into oracle/_ref/, which is git-ignored and travels to the GPU box).
// tests/test_reference_edges.py checks the oracle's restatements and the HIP kernels against THESE functions.
#include "InstRecLib/InstanceReconstructor.cpp"  // -I /root/reference/src/DynSLAM

#include "DepthProvider.h"
#include "PrecomputedDepthProvider.h"

namespace {
using instreclib::utils::BoundingBox;
using instreclib::utils::Mask;

Mask make_mask(const unsigned char *mask, int x0, int y0, int bw, int bh) {
  cv::Mat1b *m = new cv::Mat1b(bh, bw);  // owned by the Mask
  std::memcpy(m->data, mask, (size_t)bw * bh);
  return Mask(BoundingBox(x0, y0, x0 + bw - 1, y0 + bh - 1), m);
}

// DepthProvider is abstract: the disparity -> depth template (DepthProvider.h:94-137) is what is under test
class ProbeDepthProvider : public dynslam::DepthProvider {
 public:
  ProbeDepthProvider(float min_m, float max_m) : DepthProvider(false, min_m, max_m) {}
  void DisparityMapFromStereo(const cv::Mat &, const cv::Mat &, cv::Mat &) override {}
  const std::string &GetName() const override { static std::string n = "probe"; return n; }
};
}  // namespace

extern "C" {

// InstanceReconstructor.cpp:59-133
int ref_process_silhouette(const unsigned char *src_rgba, const float *src_depth, unsigned char *dst_rgba, float *dst_depth,
                           int W, int H, const unsigned char *mask, int x0, int y0, int bw, int bh) {
  Mask m = make_mask(mask, x0, y0, bw, bh);
  instreclib::reconstruction::ProcessSilhouette_CPU<float>(
      reinterpret_cast<Vector4u *>(const_cast<unsigned char *>(src_rgba)), const_cast<float *>(src_depth),
      reinterpret_cast<Vector4u *>(dst_rgba), dst_depth, Eigen::Vector2i(W, H), m, m);
  return 0;
}
// InstanceReconstructor.cpp:135-170
int ref_remove_silhouette(unsigned char *rgba, float *depth, int W, int H, const unsigned char *mask, int x0, int y0, int bw,
                          int bh) {
  Mask m = make_mask(mask, x0, y0, bw, bh);
  instreclib::reconstruction::RemoveSilhouette_CPU<float>(reinterpret_cast<Vector4u *>(rgba), depth, Eigen::Vector2i(W, H), m);
  return 0;
}
// InstanceReconstructor.cpp:875-908 with the tint of track `track_id` (:977: kMatplotlib2Palette[id % size])
int ref_composite_color(unsigned char *target_rgba, float *target_depth, const unsigned char *inst_rgba, const float *inst_depth,
                        int W, int H, int track_id, float tint_strength) {
  ITMUChar4Image tc(Vector2i(W, H), true, false), ic(Vector2i(W, H), true, false);
  ITMFloatImage td(Vector2i(W, H), true, false), id(Vector2i(W, H), true, false);
  const size_t P = (size_t)W * H;
  std::memcpy(tc.GetData(MEMORYDEVICE_CPU), target_rgba, P * 4); std::memcpy(ic.GetData(MEMORYDEVICE_CPU), inst_rgba, P * 4);
  std::memcpy(td.GetData(MEMORYDEVICE_CPU), target_depth, P * 4); std::memcpy(id.GetData(MEMORYDEVICE_CPU), inst_depth, P * 4);
  const auto &pal = instreclib::reconstruction::InstanceReconstructor::kMatplotlib2Palette;
  instreclib::reconstruction::CompositeColor(&tc, &td, &ic, &id, pal[track_id % pal.size()], tint_strength);
  std::memcpy(target_rgba, tc.GetData(MEMORYDEVICE_CPU), P * 4); std::memcpy(target_depth, td.GetData(MEMORYDEVICE_CPU), P * 4);
  return 0;
}
// InstanceReconstructor.cpp:851-871
int ref_composite_depth(float *target_depth, const float *source_depth, int W, int H) {
  ITMFloatImage td(Vector2i(W, H), true, false), sd(Vector2i(W, H), true, false);
  const size_t P = (size_t)W * H;
  std::memcpy(td.GetData(MEMORYDEVICE_CPU), target_depth, P * 4); std::memcpy(sd.GetData(MEMORYDEVICE_CPU), source_depth, P * 4);
  instreclib::reconstruction::CompositeDepth(&td, &sd);
  std::memcpy(target_depth, td.GetData(MEMORYDEVICE_CPU), P * 4);
  return 0;
}
// DepthProvider.h:94-137
int ref_depth_from_disparity(const float *disparity, short *depth_mm_out, int W, int H, float baseline_m, float focal_px, float scale,
                             float min_depth_m, float max_depth_m) {
  cv::Mat_<float> disp(H, W);
  std::memcpy(disp.data, disparity, (size_t)W * H * 4);
  cv::Mat1s out(H, W);
  ProbeDepthProvider p(min_depth_m, max_depth_m);
  dynslam::StereoCalibration calib(baseline_m, focal_px);
  p.DepthFromDisparityMap<float>(disp, calib, out, scale);
  std::memcpy(depth_mm_out, out.data, (size_t)W * H * 2);
  return 0;
}
// PrecomputedDepthProvider::GetDepth (PrecomputedDepthProvider.h:46-68, .cpp:22-75): the reference's own
// ReadPrecomputed + clamp / disparity conversion; only the two file parsers underneath (cv::FileStorage, pfmLib's
// ReadFilePFM — both absent) are the stand-ins of tests/stubs/, which call dsr_read_depth_xml / dsr_read_pfm.
// -> 0 ok, 1 std::runtime_error (message in err)
int ref_precomputed_get_depth(const char *folder, const char *fname_format, int input_is_depth, float min_depth_m, float max_depth_m,
                              int frame_idx, float baseline_m, float focal_px, float scale, short *depth_mm_out, int W, int H,
                              char *err, int err_cap) {
  try {
    dynslam::PrecomputedDepthProvider p(nullptr, folder, fname_format, input_is_depth != 0, 0, min_depth_m, max_depth_m);
    dynslam::StereoCalibration calib(baseline_m, focal_px);
    cv::Mat1s out(H, W);
    p.GetDepth(frame_idx, calib, out, scale);
    if (out.rows != H || out.cols != W) { snprintf(err, err_cap, "size %dx%d", out.cols, out.rows); return 1; }
    std::memcpy(depth_mm_out, out.data, (size_t)W * H * 2);
    return 0;
  } catch (const std::exception &ex) {
    snprintf(err, err_cap, "%s", ex.what());
    return 1;
  }
}
// InfiniTamDriver.cpp:81-100, :108-120, :128-139
int ref_cv_to_itm(const unsigned char *bgr, unsigned char *rgba_out, int W, int H) {
  cv::Mat3b m(H, W);
  std::memcpy(m.data, bgr, (size_t)W * H * 3);
  ITMUChar4Image img(Vector2i(W, H), true, false);
  dynslam::drivers::CvToItm(m, &img);
  std::memcpy(rgba_out, img.GetData(MEMORYDEVICE_CPU), (size_t)W * H * 4);
  return 0;
}
int ref_itm_to_cv(const unsigned char *rgba, unsigned char *bgr_out, int W, int H) {
  ITMUChar4Image img(Vector2i(W, H), true, false);
  std::memcpy(img.GetData(MEMORYDEVICE_CPU), rgba, (size_t)W * H * 4);
  cv::Mat3b m(H, W);
  dynslam::drivers::ItmToCv(img, &m);
  std::memcpy(bgr_out, m.data, (size_t)W * H * 3);
  return 0;
}
int ref_float_depthmap_to_short(const float *depth_m, short *mm_out, int W, int H) {
  cv::Mat1s m(H, W);
  dynslam::drivers::FloatDepthmapToShort(depth_m, m);
  std::memcpy(mm_out, m.data, (size_t)W * H * 2);
  return 0;
}

}  // extern "C"
