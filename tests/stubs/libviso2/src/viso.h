// Stand-in for libviso2's matrix.h / viso.h (an empty submodule of the reference; DynSLAM builds against a fork with
// getRawMatches() and an initial estimate for estimateMotion()).  Visual odometry is outside the hot path: the class
// is SCRIPTED — the test host installs callbacks (VisoScript) that supply each frame's ego-motion, raw matches and
// per-object motion from the synthetic ground truth — while the container types behave like the library's
// (row-major Matrix in one contiguous block, Rx*Ry*Rz transformation vectors).  See tests/stubs/README.md.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <vector>
#include "matcher.h"
typedef double FLOAT;

class Matrix {
 public:
  int32_t m = 0, n = 0;
  FLOAT **val = nullptr;  // val[i] points into one contiguous m*n block (callers read val[0] as a flat array)
  Matrix() {}
  Matrix(int32_t m_, int32_t n_) { allocate(m_, n_); }
  Matrix(const Matrix &o) { allocate(o.m, o.n); for (int i = 0; i < m * n; i++) val[0][i] = o.val[0][i]; }
  Matrix &operator=(const Matrix &o) {
    if (this != &o) { release(); allocate(o.m, o.n); for (int i = 0; i < m * n; i++) val[0][i] = o.val[0][i]; }
    return *this;
  }
  ~Matrix() { release(); }
  static Matrix eye(int32_t k) { Matrix r(k, k); for (int i = 0; i < k; i++) r.val[i][i] = 1; return r; }
  Matrix operator~() const { Matrix r(n, m); for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) r.val[j][i] = val[i][j]; return r; }
  Matrix operator*(const Matrix &o) const {
    Matrix r(m, o.n);
    for (int i = 0; i < m; i++) for (int j = 0; j < o.n; j++) for (int k = 0; k < n; k++) r.val[i][j] += val[i][k] * o.val[k][j];
    return r;
  }
  static Matrix inv(const Matrix &a) {  // Gauss-Jordan with partial pivoting
    const int k = a.m;
    Matrix w(a), r = eye(k);
    for (int i = 0; i < k; i++) {
      int p = i;
      for (int q = i + 1; q < k; q++) if (std::fabs(w.val[q][i]) > std::fabs(w.val[p][i])) p = q;
      for (int c = 0; c < k; c++) { std::swap(w.val[i][c], w.val[p][c]); std::swap(r.val[i][c], r.val[p][c]); }
      const FLOAT d = w.val[i][i];
      for (int c = 0; c < k; c++) { w.val[i][c] /= d; r.val[i][c] /= d; }
      for (int q = 0; q < k; q++) if (q != i) { const FLOAT f = w.val[q][i]; for (int c = 0; c < k; c++) { w.val[q][c] -= f * w.val[i][c]; r.val[q][c] -= f * r.val[i][c]; } }
    }
    return r;
  }

 private:
  void allocate(int32_t m_, int32_t n_) {
    m = m_; n = n_;
    if (m * n <= 0) { val = nullptr; return; }
    val = new FLOAT *[m];
    val[0] = new FLOAT[(size_t)m * n]();
    for (int i = 1; i < m; i++) val[i] = val[0] + (size_t)i * n;
  }
  void release() { if (val) { delete[] val[0]; delete[] val; val = nullptr; } m = n = 0; }
};

// what the scripted odometry "computes": installed by the test host
struct VisoScript {
  // call number (0 = first frame) -> success; fills the ego-motion (previous camera -> current camera) and the raw matches
  std::function<bool(int, Matrix &, std::vector<Matcher::p_match> &)> process;
  // (matches of one object, initial estimate) -> {rx, ry, rz, tx, ty, tz} or {} when no motion can be found
  std::function<std::vector<double>(const std::vector<Matcher::p_match> &, const std::vector<double> &)> estimate;
  static VisoScript &get() { static VisoScript s; return s; }
};

class VisualOdometry {
 public:
  struct calibration { double f = 1, cu = 0, cv = 0; };
  struct bucketing { int max_features = 2, bucket_width = 50, bucket_height = 50; };
  struct parameters { calibration calib; bucketing bucket; };
  virtual ~VisualOdometry() {}
  Matrix getMotion() { return Tr_delta; }
  std::vector<Matcher::p_match> getMatches() { return p_matched; }
  std::vector<Matcher::p_match> getRawMatches() { return p_matched; }
  std::vector<int32_t> getInlierIndices() { return {}; }
  // libviso2: Tr = [Rx(rx) * Ry(ry) * Rz(rz) | t]
  static Matrix transformationVectorToMatrix(std::vector<double> tr) {
    const double sx = std::sin(tr[0]), cx = std::cos(tr[0]), sy = std::sin(tr[1]), cy = std::cos(tr[1]);
    const double sz = std::sin(tr[2]), cz = std::cos(tr[2]);
    Matrix T(4, 4);
    T.val[0][0] = +cy * cz;                T.val[0][1] = -cy * sz;                T.val[0][2] = +sy;      T.val[0][3] = tr[3];
    T.val[1][0] = +sx * sy * cz + cx * sz; T.val[1][1] = -sx * sy * sz + cx * cz; T.val[1][2] = -sx * cy; T.val[1][3] = tr[4];
    T.val[2][0] = -cx * sy * cz + sx * sz; T.val[2][1] = +cx * sy * sz + sx * cz; T.val[2][2] = +cx * cy; T.val[2][3] = tr[5];
    T.val[3][3] = 1;
    return T;
  }

 protected:
  Matrix Tr_delta = Matrix::eye(4);
  std::vector<Matcher::p_match> p_matched;
};
