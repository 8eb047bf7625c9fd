// Stand-in for libviso2's viso_stereo.h (DynSLAM's fork): see viso.h.
#pragma once
#include "viso.h"
class VisualOdometryStereo : public VisualOdometry {
 public:
  struct parameters : VisualOdometry::parameters { double base = 1.0; int ransac_iters = 200; double inlier_threshold = 2.0; bool reweighting = true; };
  explicit VisualOdometryStereo(parameters p) : param(p) {}
  // one stereo pair; false on the first frame (no previous pair) or when the script says so
  bool process(unsigned char *, unsigned char *, int32_t *, bool = false) {
    auto &s = VisoScript::get();
    const int call = n_calls_++;
    if (!s.process) { Tr_delta = Matrix::eye(4); p_matched.clear(); return call > 0; }
    return s.process(call, Tr_delta, p_matched);
  }
  std::vector<double> estimateMotion(std::vector<Matcher::p_match> matches, const std::vector<double> &initial = std::vector<double>()) {
    auto &s = VisoScript::get();
    return s.estimate ? s.estimate(matches, initial) : std::vector<double>();
  }
  parameters param;

 private:
  int n_calls_ = 0;
};
