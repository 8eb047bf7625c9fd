// Stand-in for <opencv2/opencv.hpp>: see tests/stubs/README.md.  cv::Mat / Mat_<T> / Vec / Size_ / Rect_, the
// FileStorage reader and the three image functions the host's input side calls (imread of binary PPM / PGM,
// resize nearest / linear, cvtColor RGB2GRAY).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <iostream>
#include <sstream>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH_MASK 7
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16SC1 CV_MAKETYPE(CV_16S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {

template <class T, int N> struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; i++) val[i] = T(); }
  Vec(T a, T b) { static_assert(N == 2, ""); val[0] = a; val[1] = b; }
  Vec(T a, T b, T c) { static_assert(N == 3, ""); val[0] = a; val[1] = b; val[2] = c; }
  Vec(T a, T b, T c, T d) { static_assert(N == 4, ""); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  T &operator[](int i) { return val[i]; }
  const T &operator[](int i) const { return val[i]; }
};
typedef Vec<uchar, 3> Vec3b;
typedef Vec<float, 3> Vec3f;

template <class T> struct Size_ {
  T width, height;
  Size_() : width(0), height(0) {}
  Size_(T w, T h) : width(w), height(h) {}
  bool operator==(const Size_ &o) const { return width == o.width && height == o.height; }
};
template <class T> std::ostream &operator<<(std::ostream &o, const Size_<T> &s) { return o << "[" << s.width << " x " << s.height << "]"; }
typedef Size_<int> Size2i;
typedef Size2i Size;

template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<int> Point2i;
typedef Point2i Point;
template <class T> struct Rect_ { T x, y, width, height; Rect_() : x(0), y(0), width(0), height(0) {} Rect_(T a, T b, T w, T h) : x(a), y(b), width(w), height(h) {} };
typedef Rect_<int> Rect;

template <class T> struct DataType;
template <> struct DataType<uchar> { enum { type = CV_8UC1 }; };
template <> struct DataType<short> { enum { type = CV_16SC1 }; };
template <> struct DataType<float> { enum { type = CV_32FC1 }; };
template <> struct DataType<Vec3b> { enum { type = CV_8UC3 }; };

class Mat {
 public:
  int rows = 0, cols = 0;
  uchar *data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void *external) { create_header(r, c, type); data = static_cast<uchar *>(external); }  // no copy, not owned
  void create_header(int r, int c, int type) {
    static const int depth_bytes[] = {1, 1, 2, 2, 4, 4, 8, 0};
    type_ = type; rows = r; cols = c;
    elem_ = (size_t)depth_bytes[type & CV_MAT_DEPTH_MASK] * (1 + (type >> CV_CN_SHIFT));
    buf_.reset();
  }
  void create(int r, int c, int type) {
    if (data && r == rows && c == cols && type == type_) return;  // cv::Mat::create keeps a matching buffer
    create_header(r, c, type);
    buf_ = std::make_shared<std::vector<uchar>>((size_t)r * c * elem_);
    data = buf_->data();
  }
  int channels() const { return 1 + (type_ >> CV_CN_SHIFT); }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows * cols == 0; }
  Size size() const { return Size(cols, rows); }
  size_t elemSize() const { return elem_; }
  template <class T> T &at(int i, int j) { assert(sizeof(T) == elem_); return reinterpret_cast<T *>(data)[(size_t)i * cols + j]; }
  template <class T> const T &at(int i, int j) const { assert(sizeof(T) == elem_); return reinterpret_cast<const T *>(data)[(size_t)i * cols + j]; }
  Mat clone() const { Mat m(rows, cols, type_); if (data) std::memcpy(m.data, data, (size_t)rows * cols * elem_); return m; }

 protected:
  int type_ = 0;
  size_t elem_ = 0;
  std::shared_ptr<std::vector<uchar>> buf_;  // shared like cv::Mat's reference-counted header
};

template <class T> class Mat_ : public Mat {
 public:
  Mat_() {}
  Mat_(int r, int c) : Mat(r, c, DataType<T>::type) {}
  explicit Mat_(Size s) : Mat(s.height, s.width, DataType<T>::type) {}
  Mat_(int r, int c, T *external) : Mat(r, c, DataType<T>::type, external) {}
  Mat_(const Mat &m) : Mat(m) { assert(m.empty() || m.type() == DataType<T>::type); }
  T &operator()(int i, int j) { return this->template at<T>(i, j); }
  const T &operator()(int i, int j) const { return this->template at<T>(i, j); }
};
typedef Mat_<uchar> Mat1b;
typedef Mat_<short> Mat1s;
typedef Mat_<float> Mat1f;
typedef Mat_<Vec3b> Mat3b;

// ---- image functions of the input side (Input.cpp:113-141, PrecomputedSegmentationProvider.cpp:173-174, Mask.cpp:38,
// DynSlam.cpp:67-68).  Semantics follow OpenCV's documentation; linear resize agrees with OpenCV's fixed-point 8-bit
// path to within rounding.
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { COLOR_RGB2GRAY = 7, COLOR_BGR2GRAY = 6 };
enum { IMREAD_COLOR = 1 };

// binary PPM (P6) / PGM (P5), maxval 255, chosen by signature like cv::imread does  ->  CV_8UC3 in BGR order
inline Mat imread(const std::string &path, int = IMREAD_COLOR) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return Mat();
  char magic[3] = {0, 0, 0};
  int w = 0, h = 0, maxv = 0;
  Mat out;
  if (fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && (magic[1] == '6' || magic[1] == '5') && magic[0] == 'P' &&
      maxv == 255 && w > 0 && h > 0) {
    fgetc(f);  // the single whitespace after maxval
    int cn = magic[1] == '6' ? 3 : 1;
    std::vector<uchar> raw((size_t)w * h * cn);
    if (fread(raw.data(), 1, raw.size(), f) == raw.size()) {
      out.create(h, w, CV_8UC3);
      for (size_t i = 0; i < (size_t)w * h; i++) {
        uchar r = raw[i * cn], g = raw[i * cn + (cn == 3 ? 1 : 0)], b = raw[i * cn + (cn == 3 ? 2 : 0)];
        out.data[i * 3] = b; out.data[i * 3 + 1] = g; out.data[i * 3 + 2] = r;
      }
    }
  }
  fclose(f);
  return out;
}

inline void resize(const Mat &src, Mat &dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR) {
  int w = dsize.width, h = dsize.height;
  if (w <= 0 || h <= 0) { w = (int)std::lround(src.cols * fx); h = (int)std::lround(src.rows * fy); }
  Mat in = (src.data == dst.data) ? src.clone() : src;
  const size_t es = in.elemSize();
  const int depth = in.type() & CV_MAT_DEPTH_MASK, cn = in.channels();
  dst.create(h, w, in.type());
  const double sx = (double)in.cols / w, sy = (double)in.rows / h;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uchar *o = dst.data + ((size_t)y * w + x) * es;
      if (interpolation == INTER_NEAREST || depth != CV_8U) {
        int ix = std::min((int)std::floor(x * sx), in.cols - 1), iy = std::min((int)std::floor(y * sy), in.rows - 1);
        std::memcpy(o, in.data + ((size_t)iy * in.cols + ix) * es, es);
      } else {
        double fxs = (x + 0.5) * sx - 0.5, fys = (y + 0.5) * sy - 0.5;
        int x0 = (int)std::floor(fxs), y0 = (int)std::floor(fys);
        double ax = fxs - x0, ay = fys - y0;
        int x1 = std::min(std::max(x0 + 1, 0), in.cols - 1), y1 = std::min(std::max(y0 + 1, 0), in.rows - 1);
        x0 = std::min(std::max(x0, 0), in.cols - 1); y0 = std::min(std::max(y0, 0), in.rows - 1);
        for (int c = 0; c < cn; c++) {
          auto px = [&](int yy, int xx) { return (double)in.data[((size_t)yy * in.cols + xx) * es + c]; };
          double v = (1 - ay) * ((1 - ax) * px(y0, x0) + ax * px(y0, x1)) + ay * ((1 - ax) * px(y1, x0) + ax * px(y1, x1));
          o[c] = (uchar)std::min(255.0, std::max(0.0, std::floor(v + 0.5)));
        }
      }
    }
}

inline void cvtColor(const Mat &src, Mat &dst, int code) {
  assert((code == COLOR_RGB2GRAY || code == COLOR_BGR2GRAY) && src.type() == CV_8UC3);
  dst.create(src.rows, src.cols, CV_8UC1);
  const int ri = code == COLOR_RGB2GRAY ? 0 : 2, bi = 2 - ri;
  for (size_t i = 0; i < (size_t)src.rows * src.cols; i++) {  // OpenCV's 14-bit fixed-point luma
    const uchar *p = src.data + i * 3;
    dst.data[i] = (uchar)((p[ri] * 4899 + p[1] * 9617 + p[bi] * 1868 + (1 << 13)) >> 14);
  }
}

// cv::FileStorage, read side only, for `fs["depth-frame"] >> mat` (PrecomputedDepthProvider.cpp:37-41): the node
// reader is the library under test (dsr_read_depth_xml: the OpenCV FileStorage XML dump of a CV_16SC1 matrix)
extern "C" int dsr_read_depth_xml(const char *path, int16_t *out, int capacity, int *width, int *height);
class FileStorage {
 public:
  enum Mode { READ = 0 };
  struct Node {
    std::string path, name;
    void operator>>(Mat &out) const {
      int w = 0, h = 0;
      out = Mat();
      if (name != "depth-frame") return;
      dsr_read_depth_xml(path.c_str(), nullptr, 0, &w, &h);  // size query
      if (w <= 0 || h <= 0) return;
      Mat m(h, w, CV_16SC1);
      if (dsr_read_depth_xml(path.c_str(), reinterpret_cast<int16_t *>(m.data), w * h, &w, &h) == 0) out = m;
    }
  };
  FileStorage(const std::string &path, int) : path_(path) { FILE *f = fopen(path.c_str(), "rb"); opened_ = f != nullptr; if (f) fclose(f); }
  bool isOpened() const { return opened_; }
  Node operator[](const char *name) const { return Node{path_, name}; }

 private:
  std::string path_;
  bool opened_ = false;
};

}  // namespace cv
