#pragma once
#include "viso.h"
class VisualOdometryStereo : public VisualOdometry {
 public:
  struct parameters : VisualOdometry::parameters { double base = 1.0; int ransac_iters = 200; double inlier_threshold = 2.0; bool reweighting = true; };
  explicit VisualOdometryStereo(parameters p) : param(p) {}
  bool process(unsigned char *, unsigned char *, int32_t *, bool = false) { return true; }
  std::vector<double> estimateMotion(std::vector<Matcher::p_match> m) override { return VisualOdometry::estimateMotion(m); }
  parameters param;
};
