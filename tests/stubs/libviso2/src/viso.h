// Stand-in for libviso2's viso.h / matrix.h: names only (visual odometry is outside the hot path).
#pragma once
#include <vector>
#include "matcher.h"
typedef double FLOAT;
class Matrix {
 public:
  int m = 0, n = 0;
  FLOAT **val = nullptr;
  Matrix() {}
  Matrix(int m_, int n_) : m(m_), n(n_) { val = new FLOAT *[m]; for (int i = 0; i < m; i++) { val[i] = new FLOAT[n]; for (int j = 0; j < n; j++) val[i][j] = 0; } }
  static Matrix eye(int k) { Matrix r(k, k); for (int i = 0; i < k; i++) r.val[i][i] = 1; return r; }
  static Matrix inv(const Matrix &a) { return a; }
  Matrix operator~() const { Matrix r(n, m); for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) r.val[j][i] = val[i][j]; return r; }
  Matrix operator*(const Matrix &o) const { Matrix r(m, o.n); for (int i = 0; i < m; i++) for (int j = 0; j < o.n; j++) for (int k = 0; k < n; k++) r.val[i][j] += val[i][k] * o.val[k][j]; return r; }
};
class VisualOdometry {
 public:
  struct calibration { double f = 1, cu = 0, cv = 0; };
  struct bucketing { int max_features = 2, bucket_width = 50, bucket_height = 50; };
  struct parameters { calibration calib; bucketing bucket; };
  virtual ~VisualOdometry() {}
  Matrix getMotion() { return Matrix::eye(4); }
  std::vector<Matcher::p_match> getMatches() { return {}; }
  std::vector<int32_t> getInlierIndices() { return {}; }
  virtual std::vector<double> estimateMotion(std::vector<Matcher::p_match>) { return {}; }
  static Matrix transformationVectorToMatrix(std::vector<double>) { return Matrix::eye(4); }
};
