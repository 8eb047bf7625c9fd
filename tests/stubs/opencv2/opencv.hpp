// Stand-in for <opencv2/opencv.hpp>: see tests/stubs/README.md.  cv::Mat / Mat_<T> / Vec / Size_ / Rect_ only.
#pragma once
#include <algorithm>
#include <cassert>
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
  void create(int r, int c, int type) {
    static const int depth_bytes[] = {1, 1, 2, 2, 4, 4, 8, 0};
    type_ = type; rows = r; cols = c;
    elem_ = (size_t)depth_bytes[type & CV_MAT_DEPTH_MASK] * (1 + (type >> CV_CN_SHIFT));
    buf_ = std::make_shared<std::vector<uchar>>((size_t)r * c * elem_);
    data = buf_->data();
  }
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
  Mat_(const Mat &m) : Mat(m) { assert(m.empty() || m.type() == DataType<T>::type); }
  T &operator()(int i, int j) { return this->template at<T>(i, j); }
  const T &operator()(int i, int j) const { return this->template at<T>(i, j); }
};
typedef Mat_<uchar> Mat1b;
typedef Mat_<short> Mat1s;
typedef Mat_<float> Mat1f;
typedef Mat_<Vec3b> Mat3b;

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
