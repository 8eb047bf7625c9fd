// shim/ITMLib.h — header-only C++ shim: the ITMLib / ORUtils names DynSLAM's host uses,
// implemented on top of the C ABI of include/dsr.h (libdsr_hip.so).
//
// Purpose: src/DynSLAM/InfiniTamDriver.{h,cpp}, InstRecLib/InstanceReconstructor.cpp and
// DynSlam.{h,cpp} include "../InfiniTAM/InfiniTAM/ITMLib/Engine/ITMMainEngine.h"
// (InfiniTamDriver.h:13) and then reach into ITMMainEngine's protected members
// (InfiniTamDriver.h:115-156,190,203,242-248,283).  Pointing that include at this header
// gives them the same class / member / method names; every engine operation forwards 1:1 to a
// dsr_* call.  Only what those call sites touch is provided.
//
// Memory model: images hold HOST buffers; the device copies live inside the engine.
//   UpdateDeviceFromHost() -> marks the view dirty; the next engine call uploads it
//                             (dsr_set_view_float);
//   UpdateHostFromDevice() -> dsr_get_view;
// which is exactly the round trip InstanceReconstructor.cpp:180-197,262-263 performs.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <future>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/dsr.h"

#ifndef SDF_BLOCK_SIZE
#define SDF_BLOCK_SIZE DSR_BLOCK_SIZE
#define SDF_BLOCK_SIZE3 DSR_BLOCK_SIZE3
#endif

typedef unsigned char uchar;

enum MemoryDeviceType { MEMORYDEVICE_CPU, MEMORYDEVICE_CUDA };

namespace ORUtils {

template <class T> struct Vector2 {
  union { struct { T x, y; }; struct { T width, height; }; T v[2]; };
  Vector2() : x(0), y(0) {}
  Vector2(T a, T b) : x(a), y(b) {}
  const T *getValues() const { return v; }
  T *getValues() { return v; }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  bool operator==(const Vector2 &o) const { return x == o.x && y == o.y; }
};
template <class T> struct Vector3 {
  union { struct { T x, y, z; }; struct { T r, g, b; }; T v[3]; };
  Vector3() : x(0), y(0), z(0) {}
  Vector3(T a, T b_, T c) : x(a), y(b_), z(c) {}
  const T *getValues() const { return v; }
  T *getValues() { return v; }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
};
template <class T> struct Vector4 {
  union { struct { T x, y, z, w; }; struct { T r, g, b, a; }; T v[4]; };
  Vector4() : x(0), y(0), z(0), w(0) {}
  Vector4(T a_, T b_, T c, T d) : x(a_), y(b_), z(c), w(d) {}
  const T *getValues() const { return v; }  // DynSlam.h:102,118
  T *getValues() { return v; }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
};

// column-major 4x4, m[col*4 + row] (InfiniTamDriver.cpp:146-163)
template <class T> struct Matrix4 {
  T m[16];
  Matrix4() { std::memset(m, 0, sizeof m); }
  Matrix4(T a0, T a1, T a2, T a3, T a4, T a5, T a6, T a7, T a8, T a9, T a10, T a11, T a12, T a13, T a14, T a15) {
    T t[16] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15};
    std::memcpy(m, t, sizeof m);
  }
  void setIdentity() { std::memset(m, 0, sizeof m); m[0] = m[5] = m[10] = m[15] = 1; }
  T &at(int x, int y) { return m[x * 4 + y]; }
  const T &at(int x, int y) const { return m[x * 4 + y]; }
  T &operator()(int x, int y) { return at(x, y); }
  const T &operator()(int x, int y) const { return at(x, y); }
  friend Matrix4 operator*(const Matrix4 &l, const Matrix4 &r) {
    Matrix4 o;
    for (int x = 0; x < 4; x++)
      for (int y = 0; y < 4; y++) {
        T s = 0;
        for (int k = 0; k < 4; k++) s += l.m[k * 4 + y] * r.m[x * 4 + k];
        o.m[x * 4 + y] = s;
      }
    return o;
  }
  // general inverse via the engine's own convention is done inside dsr_set_pose_*; the host
  // only needs inv() for rigid transforms (InfiniTamDriver.h:120-122): Gauss-Jordan.
  bool inv(Matrix4 &out) const {
    double a[4][8];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = at(c, r); a[r][4 + c] = (r == c); }
    for (int i = 0; i < 4; i++) {
      int p = i;
      for (int r = i + 1; r < 4; r++) if (std::abs(a[r][i]) > std::abs(a[p][i])) p = r;
      if (a[p][i] == 0) return false;
      for (int c = 0; c < 8; c++) std::swap(a[i][c], a[p][c]);
      double d = a[i][i];
      for (int c = 0; c < 8; c++) a[i][c] /= d;
      for (int r = 0; r < 4; r++) if (r != i) { double f = a[r][i]; for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c]; }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out.at(c, r) = (T)a[r][4 + c];
    return true;
  }
};

template <class T> class MemoryBlock {
 public:
  enum MemoryCopyDirection { CPU_TO_CPU, CPU_TO_CUDA, CUDA_TO_CPU, CUDA_TO_CUDA };
  size_t dataSize = 0;
  explicit MemoryBlock(size_t n = 0) : dataSize(n), data_(n) {}
  T *GetData(MemoryDeviceType) { return data_.data(); }
  const T *GetData(MemoryDeviceType) const { return data_.data(); }
  void Clear(unsigned char v = 0) { if (!data_.empty()) std::memset(static_cast<void *>(data_.data()), v, data_.size() * sizeof(T)); }
  void SetFrom(const MemoryBlock *src, MemoryCopyDirection) { data_ = src->data_; dataSize = src->dataSize; }
 protected:
  std::vector<T> data_;
};

template <class T> class Image : public MemoryBlock<T> {
 public:
  Vector2<int> noDims;
  // hooks installed by ITMView so that the host<->device calls reach the engine
  void (*toHost)(void *ctx) = nullptr;
  void (*toDevice)(void *ctx) = nullptr;
  void *ctx = nullptr;
  Image(Vector2<int> dims, bool /*allocate_CPU*/, bool /*allocate_CUDA*/) : MemoryBlock<T>((size_t)dims.x * dims.y), noDims(dims) {}
  Image(Vector2<int> dims, MemoryDeviceType) : Image(dims, true, false) {}
  void ChangeDims(Vector2<int> d) { if (!(d == noDims)) { noDims = d; this->data_.assign((size_t)d.x * d.y, T()); this->dataSize = this->data_.size(); } }
  void UpdateHostFromDevice() { if (toHost) toHost(ctx); }
  void UpdateDeviceFromHost() { if (toDevice) toDevice(ctx); }
};

}  // namespace ORUtils

typedef ORUtils::Vector2<int> Vector2i;
typedef ORUtils::Vector2<float> Vector2f;
typedef ORUtils::Vector3<float> Vector3f;
typedef ORUtils::Vector4<float> Vector4f;
typedef ORUtils::Vector4<uchar> Vector4u;
typedef ORUtils::Matrix4<float> Matrix4f;
typedef ORUtils::Image<Vector4u> ITMUChar4Image;
typedef ORUtils::Image<float> ITMFloatImage;
typedef ORUtils::Image<short> ITMShortImage;

struct ITMVoxel { short sdf; uchar w_depth; uchar clr[3]; uchar w_color; uchar _pad; };  // ITMVoxel_s_rgb, 8 bytes
struct ITMVoxelIndex {};
static_assert(sizeof(ITMVoxel) == sizeof(dsr_voxel), "voxel size");

// ITMSafeCall(err) (ITMLib/Utils/ITMCUDAUtils.h: print + exit on a CUDA error; DynSlam.cpp:165,171) and the two CUDA
// runtime names DynSLAM's host calls directly (DynSlam.cpp:165-166: device-wide sync + error poll after every frame;
// DynSLAMGUI.cpp:912: cudaMemGetInfo; upstream they arrive through ITMLib's headers).  They map onto the library's device-wide sync; errors of
// individual engine calls are already raised as exceptions where they happen.
typedef int cudaError_t;
enum { cudaSuccess = 0 };
inline cudaError_t cudaDeviceSynchronize() { return dsr_device_synchronize(); }
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline cudaError_t cudaMemGetInfo(size_t *free_bytes, size_t *total_bytes) {  // DynSLAMGUI.cpp:912: the memory read-out
  uint64_t f = 0, t = 0;
  const int st = dsr_device_mem_info(-1, &f, &t);
  *free_bytes = (size_t)f; *total_bytes = (size_t)t;
  return st;
}
inline void ITMSafeCallImpl(int err, const char *file, int line) {
  if (err != 0) throw std::runtime_error(std::string(file) + ":" + std::to_string(line) + ": " + dsr_last_error());
}
#define ITMSafeCall(x) ITMSafeCallImpl((x), __FILE__, __LINE__)

namespace ITMLib {
namespace Objects {

struct ITMIntrinsics {
  struct { Vector4f all; float &fx() { return all.x; } } projectionParamsSimple;
  Vector2i size;
  void SetFrom(float fx, float fy, float cx, float cy, float sizeX, float sizeY) {
    projectionParamsSimple.all = Vector4f(fx, fy, cx, cy);
    size = Vector2i((int)sizeX, (int)sizeY);
  }
};
struct ITMExtrinsics {
  Matrix4f calib, calib_inv;
  ITMExtrinsics() { calib.setIdentity(); calib_inv.setIdentity(); }
  void SetFrom(const Matrix4f &m) { calib = m; m.inv(calib_inv); }
};
struct ITMDisparityCalib {
  enum TrafoType { TRAFO_KINECT, TRAFO_AFFINE };
  Vector2f params; TrafoType type = TRAFO_AFFINE;
  void SetFrom(float a, float b, TrafoType t) { params = Vector2f(a, b); type = t; }
};
struct ITMRGBDCalib {
  ITMIntrinsics intrinsics_rgb, intrinsics_d;
  ITMExtrinsics trafo_rgb_to_depth;
  ITMDisparityCalib disparityCalib;
};

// Defaults: DynSLAM's GUI never sets the scene parameters (DynSLAMGUI.cpp:1214-1219), it runs on the FORK's
// ITMLibSettings defaults, which are not in the reference tree.  What the tree does pin: the experiment artefacts are
// named after `driver_settings->sceneParams.voxelSize` and read "voxelsize-0.0500" (Evaluation.h:66-72,
// notebooks/DepthAnalysis.ipynb:47-51), and depth is fused up to 20 m (Input.h:71-72) — so the fork's defaults are
// 5 cm voxels and an outdoor frustum, not upstream's indoor 5 mm / 3 m.  mu keeps upstream's 4 voxels.
struct ITMSceneParams {
  float mu = 0.2f; int maxW = 100; float voxelSize = 0.05f;
  float viewFrustum_min = 0.2f, viewFrustum_max = 30.0f; bool stopIntegratingAtMaxW = false;
};
struct ITMLibSettings {
  enum DeviceType { DEVICE_CPU, DEVICE_CUDA, DEVICE_METAL };
  DeviceType deviceType = DEVICE_CUDA;
  ITMSceneParams sceneParams;
  bool useSwapping = false, useBilateralFilter = false, modelSensorNoise = false, createMeshingEngine = true;
  long sdfLocalBlockNum = DSR_DEFAULT_LOCAL_BLOCK_NUM;  // fork (InstanceReconstructor.cpp:379)
  int maxWDynamic = 10;                                 // fork (DynSLAMGUI.cpp:1217-1219)
  // engine table sizes (upstream compile-time constants)
  int hashBucketNum = DSR_DEFAULT_BUCKET_NUM, excessListSize = DSR_DEFAULT_EXCESS_LIST_SIZE;
  // one volume per GPU: the HIP device the engine built from these settings lives on; -1 = the placement
  // policy below (dynslam_shim::PlaceVolume), so that the reference's UNMODIFIED InstanceReconstructor::InitializeReconstruction
  // (InstanceReconstructor.cpp:363-392: `new InfiniTamDriver(settings, ...)` per track) lands every track's volume on its own GPU
  int deviceIndex = -1;
  std::string groundTruthPoseFpath; int groundTruthPoseOffset = 0;
};

class ITMPose {
 public:
  ITMPose() { M.setIdentity(); invMSet.setIdentity(); }
  const Matrix4f &GetM() const { return M; }
  Matrix4f GetInvM() const { if (fromInv) return invMSet; Matrix4f r; M.inv(r); return r; }
  void SetM(const Matrix4f &m) { M = m; fromInv = false; }
  // pose_d->SetInvM(...) is how the host sets poses (InfiniTamDriver.h:131-134): the matrix is
  // handed to the engine unchanged, which derives M with ORUtils' own inverse
  void SetInvM(const Matrix4f &im) { invMSet = im; fromInv = true; im.inv(M); }
  void SetFrom(const ITMPose *p) { *this = *p; }
  void Coerce() {}  // re-orthonormalisation: the engine takes M as given
  int apply(dsr_engine *e) const { return fromInv ? dsr_set_pose_inv_m(e, invMSet.m) : dsr_set_pose_m(e, M.m); }
 private:
  Matrix4f M, invMSet;
  bool fromInv = false;
};

struct ITMTrackingState { ITMPose *pose_d = new ITMPose; ~ITMTrackingState() { delete pose_d; } };

struct ITMRenderState { virtual ~ITMRenderState() {} };
// noVisibleBlocks is read by the host once per frame (InfiniTamDriver.h:150); it lives in device memory,
// so the value is fetched (one stream synchronisation) when it is READ, not after every engine call
struct ITMRenderState_VH : ITMRenderState {
  struct LazyCount {
    dsr_engine *e = nullptr;
    mutable int cached = 0;
    mutable bool valid = true;
    operator int() const {
      // known on the host already when the status of the fusion call was fetched (one read-back for both), else 12 bytes
      if (!valid && e) { int32_t n = 0; if (dsr_get_no_visible_blocks(e, &n) == DSR_OK) cached = n; valid = true; }
      return cached;
    }
    LazyCount &operator=(int v) { cached = v; valid = true; return *this; }
    void invalidate(dsr_engine *engine) { e = engine; valid = false; }
  } noVisibleBlocks;
};

class ITMView {
 public:
  ITMRGBDCalib *calib;
  ITMUChar4Image *rgb;
  ITMFloatImage *depth;
  dsr_engine *owner = nullptr;  // engine whose device view mirrors this host view (may be null): set with bind()
  bool deviceStale = true;      // host buffers are newer than the engine's copy
  // takes ownership of calib (InstanceReconstructor.cpp:782-785)
  ITMView(const ITMRGBDCalib *c, Vector2i imgSize_rgb, Vector2i imgSize_d, bool /*useGPU*/)
      : calib(new ITMRGBDCalib(*c)), rgb(new ITMUChar4Image(imgSize_rgb, true, true)), depth(new ITMFloatImage(imgSize_d, true, true)) {
    rgb->ctx = depth->ctx = this;
    rgb->toHost = depth->toHost = [](void *v) { static_cast<ITMView *>(v)->pull(); };
    rgb->toDevice = depth->toDevice = [](void *v) { static_cast<ITMView *>(v)->deviceStale = true; };
  }
  ~ITMView() { bind(nullptr); delete calib; delete rgb; delete depth; }
  void pull() {
    if (owner && !deviceStale)
      dsr_get_view(owner, reinterpret_cast<uint8_t *>(rgb->GetData(MEMORYDEVICE_CPU)), depth->GetData(MEMORYDEVICE_CPU));
  }
  // Views outlive engines in the host (a track keeps the views of its frames, InstanceView.h, while its InfiniTamDriver is
  // reaped or pruned): the engine a view mirrors is tracked here so that a dying engine can detach its views instead
  // of leaving them with a dangling handle.
  void bind(dsr_engine *e) {
    if (owner == e) return;
    auto &reg = registry();
    if (owner) {
      auto range = reg.equal_range(owner);
      for (auto it = range.first; it != range.second; ++it) if (it->second == this) { reg.erase(it); break; }
    }
    owner = e;
    if (e) reg.emplace(e, this);
  }
  static void detach_all(dsr_engine *e) {  // called by the engine's destructor
    auto &reg = registry();
    auto range = reg.equal_range(e);
    for (auto it = range.first; it != range.second; ++it) { it->second->owner = nullptr; it->second->deviceStale = true; }
    reg.erase(range.first, range.second);
  }

 private:
  static std::unordered_multimap<dsr_engine *, ITMView *> &registry() { static std::unordered_multimap<dsr_engine *, ITMView *> r; return r; }
};

}  // namespace Objects

namespace Engine {
struct WeightParams { bool depthWeighting = false; };

inline void dsr_throw(int st) {
  if (st == DSR_OK) return;
  // block exhaustion is a runtime_error in the fork (caught at InstanceReconstructor.cpp:662-671)
  throw std::runtime_error(std::string("dsr: ") + dsr_last_error());
}
}  // namespace Engine
}  // namespace ITMLib

using namespace ITMLib::Objects;
using ITMLib::Engine::WeightParams;

// Device placement of a volume.  DSR_DEVICES="0,1,2,3" lists the GPUs this host may use (unset: everything on the current
// device, as the reference).  The static map — any volume larger than an instance's — takes the first; instance volumes
// (sdfLocalBlockNum <= 16384; the reference's are 7142, InstanceReconstructor.cpp:379) go round-robin over the OTHERS
// (BASELINE configs[3]: "static map on GPU0 + 7 instance volumes on GPUs 1-7"), or over all of them with
// DSR_INSTANCES_ON_ALL_DEVICES=1 (north_star's N concurrent instance volumes).  An explicit ITMLibSettings::deviceIndex wins.
namespace dynslam_shim {
inline const std::vector<int> &Devices() {
  static const std::vector<int> devices = [] {
    std::vector<int> d;
    const char *env = std::getenv("DSR_DEVICES");
    for (const char *p = env; p && *p;) {
      char *end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) break;
      d.push_back((int)v);
      p = (*end == ',') ? end + 1 : end;
    }
    return d;
  }();
  return devices;
}
inline int PlaceVolume(const ITMLib::Objects::ITMLibSettings *s) {
  if (s->deviceIndex >= 0) return s->deviceIndex;
  const std::vector<int> &d = Devices();
  if (d.empty()) return -1;
  if (s->sdfLocalBlockNum > 16384 || d.size() == 1) return d[0];
  static int next = 0;
  const bool all = std::getenv("DSR_INSTANCES_ON_ALL_DEVICES") != nullptr;
  const int n = all ? (int)d.size() : (int)d.size() - 1;
  return d[(all ? 0 : 1) + (next++ % n)];
}
}  // namespace dynslam_shim

// The layout loops of InfiniTamDriver.cpp:81-144 on the GPU, for hosts without OpenCV types at hand:
// raw buffers in, raw buffers out (a cv::Mat3b is `rows*cols` packed BGR triples, a cv::Mat1s `short`s).
namespace dynslam_shim {
inline void CvToItm(const unsigned char *bgr, int rows, int cols, ITMUChar4Image *out_itm) {  // :81-100
  out_itm->ChangeDims(Vector2i(cols, rows));
  ITMLib::Engine::dsr_throw(dsr_bgr_to_rgba(bgr, reinterpret_cast<uint8_t *>(out_itm->GetData(MEMORYDEVICE_CPU)), rows * cols));
}
inline void ItmToCv(const ITMUChar4Image &itm, unsigned char *bgr_out) {  // :108-120
  ITMLib::Engine::dsr_throw(dsr_rgba_to_bgr(reinterpret_cast<const uint8_t *>(itm.GetData(MEMORYDEVICE_CPU)), bgr_out, itm.noDims.x * itm.noDims.y));
}
inline void FloatDepthmapToShort(const float *pixels, short *out_mm, int n) {  // :128-139
  ITMLib::Engine::dsr_throw(dsr_depth_m_to_mm(pixels, out_mm, n));
}
inline void ItmDepthToCv(const ITMFloatImage &itm, short *out_mm) {  // :141-144
  FloatDepthmapToShort(itm.GetData(MEMORYDEVICE_CPU), out_mm, itm.noDims.x * itm.noDims.y);
}
}  // namespace dynslam_shim

// scene facade: only the counters the host reads (InfiniTamDriver.h:241-244)
template <class TVoxel, class TIndex> struct ITMScene {
  dsr_engine *e = nullptr;
  const ITMSceneParams *sceneParams = nullptr;
  struct Index { ITMScene *s; int getNumAllocatedVoxelBlocks() const { dsr_stats st; dsr_get_stats(s->e, &st); return st.num_allocated_voxel_blocks; } } index{this};
  struct VBA { ITMScene *s; struct Proxy { ITMScene *s; operator int() const { dsr_stats st; dsr_get_stats(s->e, &st); return st.last_free_block_id; } } lastFreeBlockId; } localVBA{this, {this}};
};

class ITMMainEngine;

// ITMMesh / ITMMeshingEngine facades (InstanceReconstructor.cpp:749-757).  The triangles stay on the
// device inside the dsr engine that meshed them; ITMMesh is the handle the host writes out.
class ITMMesh {
 public:
  struct Triangle { Vector3f p0, p1, p2; };
  MemoryDeviceType memoryType;
  uint noTotalTriangles = 0;
  uint noMaxTriangles;
  ITMMesh(MemoryDeviceType type, long maxBlocks) : memoryType(type), noMaxTriangles((uint)(maxBlocks * 32)) {}
  ~ITMMesh() { if (e_) dsr_mesh_free(e_); }
  void WriteOBJ(const char *fileName) {
    if (!e_) throw std::runtime_error("ITMMesh::WriteOBJ: MeshScene has not been run");
    ITMLib::Engine::dsr_throw(dsr_mesh_write_obj(e_, fileName));
  }
  // copies triangles [first, first + count) to host memory (upstream: triangles->GetData(MEMORYDEVICE_CPU))
  void GetTriangles(Triangle *out, uint first, uint count) {
    if (!e_) throw std::runtime_error("ITMMesh: MeshScene has not been run");
    ITMLib::Engine::dsr_throw(dsr_mesh_get(e_, reinterpret_cast<dsr_triangle *>(out), first, count));
  }
  void bind(dsr_engine *e, uint64_t n) { e_ = e; noTotalTriangles = (uint)n; }

 private:
  dsr_engine *e_ = nullptr;
};

template <class TVoxel, class TIndex> class ITMMeshingEngine {
 public:
  virtual ~ITMMeshingEngine() {}
  virtual void MeshScene(ITMMesh *mesh, const ITMScene<TVoxel, TIndex> *scene) {
    uint64_t n = 0;
    ITMLib::Engine::dsr_throw(dsr_mesh_scene(scene->e, &n));
    mesh->bind(scene->e, n);
  }
};
template <class TVoxel, class TIndex> class ITMMeshingEngine_CUDA : public ITMMeshingEngine<TVoxel, TIndex> {
 public:
  explicit ITMMeshingEngine_CUDA(long /*maxBlocks*/ = 0) {}
};
template <class TVoxel, class TIndex> class ITMMeshingEngine_CPU : public ITMMeshingEngine<TVoxel, TIndex> {
 public:
  explicit ITMMeshingEngine_CPU(long /*maxBlocks*/ = 0) {}
};

// ITMDenseMapper facade (InfiniTamDriver.h:138-145,203,248,283)
template <class TVoxel, class TIndex> class ITMDenseMapper {
 public:
  explicit ITMDenseMapper(dsr_engine *e) : e_(e) {}
  void SetFusionWeightParams(const WeightParams &p) { ITMLib::Engine::dsr_throw(dsr_set_fusion_weight_params(e_, p.depthWeighting)); }
  void ProcessFrame(ITMView *view, ITMTrackingState *ts, ITMScene<TVoxel, TIndex> *, ITMRenderState *rs) {
    push_view(view);
    ITMLib::Engine::dsr_throw(ts->pose_d->apply(e_));
    int st = dsr_process_frame(e_);
    static_cast<ITMRenderState_VH *>(rs)->noVisibleBlocks.invalidate(e_);
    ITMLib::Engine::dsr_throw(st);
  }
  void Decay(ITMScene<TVoxel, TIndex> *, ITMRenderState *rs, int maxWeight, int minAge, bool forceAllVoxels) {
    ITMLib::Engine::dsr_throw(dsr_decay(e_, maxWeight, minAge, forceAllVoxels));
    static_cast<ITMRenderState_VH *>(rs)->noVisibleBlocks.invalidate(e_);
  }
  size_t GetDecayedBlockCount() const { dsr_stats s; dsr_get_stats(e_, &s); return (size_t)s.decayed_block_count; }
  void ResetScene(ITMScene<TVoxel, TIndex> *) { ITMLib::Engine::dsr_throw(dsr_reset_scene(e_)); }
  void push_view(ITMView *view) {
    if (view->owner != e_ || view->deviceStale) {
      ITMLib::Engine::dsr_throw(dsr_set_view_float(e_, reinterpret_cast<const uint8_t *>(view->rgb->GetData(MEMORYDEVICE_CPU)),
                                                   view->depth->GetData(MEMORYDEVICE_CPU)));
      view->bind(e_); view->deviceStale = false;
    }
  }
 private:
  dsr_engine *e_;
};

// ITMTrackingController facade: Prepare only (InfiniTamDriver.h:152); Track() (ICP) is not on the path
class ITMTrackingController {
 public:
  explicit ITMTrackingController(dsr_engine *e) : e_(e) {}
  void Prepare(ITMTrackingState *ts, const ITMView *, ITMRenderState *) {
    ITMLib::Engine::dsr_throw(ts->pose_d->apply(e_));
    ITMLib::Engine::dsr_throw(dsr_prepare(e_));
  }
  void Track(ITMTrackingState *, const ITMView *) { throw std::runtime_error("ICP tracking is outside the dsr hot path (DynSLAM uses libviso2 poses)"); }
 private:
  dsr_engine *e_;
};

// ITMViewBuilder facade (InfiniTamDriver.cpp:177,222-223)
class ITMViewBuilder {
 public:
  ITMViewBuilder(dsr_engine *e, const ITMRGBDCalib *c) : e_(e), calib_(c) {}
  const ITMRGBDCalib *GetCalib() const { return calib_; }
  void UpdateView(ITMView **view, ITMUChar4Image *rgb, ITMShortImage *rawDepth, bool /*useBilateral*/, bool /*modelSensorNoise*/ = false) {
    if (*view == nullptr) *view = new ITMView(calib_, rgb->noDims, rawDepth->noDims, true);
    ITMLib::Engine::dsr_throw(dsr_update_view(e_, reinterpret_cast<const uint8_t *>(rgb->GetData(MEMORYDEVICE_CPU)), rawDepth->GetData(MEMORYDEVICE_CPU)));
    (*view)->bind(e_); (*view)->deviceStale = false;
    // keep the host colour copy current; the converted depth is fetched on UpdateHostFromDevice()
    (*view)->rgb->SetFrom(rgb, ORUtils::MemoryBlock<Vector4u>::CPU_TO_CPU);
  }
  // InfiniTamDriver::UpdateView as a whole (InfiniTamDriver.cpp:211-224) for hosts that hold the frame as OpenCV does — packed
  // BGR + int16 mm: CvToItm's loop runs in the engine's ingest kernel (dsr_update_view_bgr), the host copies of the view are
  // fetched when somebody asks for them (UpdateHostFromDevice)
  void UpdateViewBgr(ITMView **view, const unsigned char *bgr, const short *rawDepthMm, Vector2i size) {
    if (*view == nullptr) *view = new ITMView(calib_, size, size, true);
    ITMLib::Engine::dsr_throw(dsr_update_view_bgr(e_, bgr, rawDepthMm));
    (*view)->bind(e_); (*view)->deviceStale = false;
  }
 private:
  dsr_engine *e_;
  const ITMRGBDCalib *calib_;
};

struct IITMVisualisationEngine {};

class ITMMainEngine {
 public:
  enum GetImageType {
    InfiniTAM_IMAGE_ORIGINAL_RGB = DSR_IMAGE_ORIGINAL_RGB, InfiniTAM_IMAGE_ORIGINAL_DEPTH = DSR_IMAGE_ORIGINAL_DEPTH,
    InfiniTAM_IMAGE_SCENERAYCAST = DSR_IMAGE_SCENERAYCAST, InfiniTAM_IMAGE_FREECAMERA_SHADED = DSR_IMAGE_FREECAMERA_SHADED,
    InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME = DSR_IMAGE_FREECAMERA_COLOUR_FROM_VOLUME,
    InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL = DSR_IMAGE_FREECAMERA_COLOUR_FROM_NORMAL,
    InfiniTAM_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT = DSR_IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT,
    InfiniTAM_IMAGE_FREECAMERA_DEPTH = DSR_IMAGE_FREECAMERA_DEPTH
  };

  // ITMMainEngine(settings, calib, imgSize_rgb, imgSize_d) (InfiniTamDriver.h:84-91)
  ITMMainEngine(const ITMLibSettings *settings_, const ITMRGBDCalib *calib, Vector2i imgSize_rgb, Vector2i imgSize_d)
      : settings(settings_), imgSize_(imgSize_d) {
    dsr_settings s; dsr_default_settings(&s);
    s.voxel_size = settings->sceneParams.voxelSize; s.mu = settings->sceneParams.mu; s.max_w = settings->sceneParams.maxW;
    s.view_frustum_min = settings->sceneParams.viewFrustum_min; s.view_frustum_max = settings->sceneParams.viewFrustum_max;
    s.stop_integrating_at_max_w = settings->sceneParams.stopIntegratingAtMaxW;
    s.sdf_local_block_num = (int32_t)settings->sdfLocalBlockNum;
    s.hash_bucket_num = settings->hashBucketNum; s.excess_list_size = settings->excessListSize;
    s.use_swapping = settings->useSwapping; s.use_bilateral_filter = settings->useBilateralFilter; s.sync_status = 1;
    s.device = deviceIndex_ = dynslam_shim::PlaceVolume(settings);
    dsr_calib c; std::memset(&c, 0, sizeof c);
    auto fill = [](dsr_intrinsics &o, const ITMIntrinsics &i, Vector2i sz) {
      o.fx = i.projectionParamsSimple.all.x; o.fy = i.projectionParamsSimple.all.y; o.cx = i.projectionParamsSimple.all.z;
      o.cy = i.projectionParamsSimple.all.w; o.width = sz.x; o.height = sz.y; };
    fill(c.rgb, calib->intrinsics_rgb, imgSize_rgb); fill(c.depth, calib->intrinsics_d, imgSize_d);
    std::memcpy(c.trafo_rgb_to_depth, calib->trafo_rgb_to_depth.calib.m, sizeof c.trafo_rgb_to_depth);
    c.disparity_calib[0] = calib->disparityCalib.params.x; c.disparity_calib[1] = calib->disparityCalib.params.y;
    ITMLib::Engine::dsr_throw(dsr_engine_create(&s, &c, &engine_));
    scene = new ITMScene<ITMVoxel, ITMVoxelIndex>(); scene->e = engine_; scene->sceneParams = &settings->sceneParams;
    denseMapper = new ITMDenseMapper<ITMVoxel, ITMVoxelIndex>(engine_);
    trackingController = new ITMTrackingController(engine_);
    viewBuilder = new ITMViewBuilder(engine_, calib);
    trackingState = new ITMTrackingState();
    renderState_live = new ITMRenderState_VH();
    visualisationEngine = new IITMVisualisationEngine();
    view = nullptr;
  }
  virtual ~ITMMainEngine() {
    delete view;  // the host nulls it first when it does not own it (InstanceTracker.cpp:44-50)
    delete scene; delete denseMapper; delete trackingController; delete viewBuilder; delete trackingState;
    delete renderState_live; delete visualisationEngine;
    ITMView::detach_all(engine_);  // views kept by the host (track frames) must not keep the dead handle
    dsr_engine_destroy(engine_);
  }

  ITMView *GetView() { return view; }
  ITMScene<ITMVoxel, ITMVoxelIndex> *GetScene() { return scene; }
  Vector2i GetImageSize() const { return imgSize_; }
  dsr_engine *GetDsrEngine() { return engine_; }
  int GetDeviceIndex() const { return deviceIndex_; }  // -1: the device that was current when the engine was built

  // ITMMainEngine::GetImage(out, outFloat, type, pose, intrinsics) (InfiniTamDriver.cpp:178-183,202-207)
  void GetImage(ITMUChar4Image *out, ITMFloatImage *outFloat, GetImageType type, ITMPose *pose = nullptr, ITMIntrinsics *intrinsics = nullptr) {
    if (view == nullptr) return;
    denseMapper->push_view(view);
    float intr[4];
    if (intrinsics) { intr[0] = intrinsics->projectionParamsSimple.all.x; intr[1] = intrinsics->projectionParamsSimple.all.y; intr[2] = intrinsics->projectionParamsSimple.all.z; intr[3] = intrinsics->projectionParamsSimple.all.w; }
    if (out) out->Clear();
    ITMLib::Engine::dsr_throw(dsr_get_image(engine_, (int)type, pose ? pose->GetM().m : nullptr, intrinsics ? intr : nullptr,
                                            out ? reinterpret_cast<uint8_t *>(out->GetData(MEMORYDEVICE_CPU)) : nullptr,
                                            outFloat ? outFloat->GetData(MEMORYDEVICE_CPU) : nullptr));
  }
  // ITMMainEngine::SaveSceneToMesh (DynSlam.cpp:188-196)
  void SaveSceneToMesh(const char *objFileName) { ITMLib::Engine::dsr_throw(dsr_save_scene_to_mesh(engine_, objFileName)); }

 protected:
  const ITMLibSettings *settings;
  ITMScene<ITMVoxel, ITMVoxelIndex> *scene;
  ITMView *view;
  ITMViewBuilder *viewBuilder;
  ITMTrackingState *trackingState;
  ITMTrackingController *trackingController;
  ITMDenseMapper<ITMVoxel, ITMVoxelIndex> *denseMapper;
  ITMRenderState *renderState_live;
  IITMVisualisationEngine *visualisationEngine;
  std::future<void> write_result;

 private:
  dsr_engine *engine_ = nullptr;
  Vector2i imgSize_;
  int deviceIndex_ = -1;
};
