// Stand-in for <pangolin/pangolin.h>: the OpenGL model-view matrix type the driver converts poses from
// (InfiniTamDriver.cpp:36-47).  See tests/stubs/README.md.
#pragma once
namespace pangolin {
struct OpenGlMatrix {
  double m[16];  // column-major, like OpenGL
  OpenGlMatrix() { for (int i = 0; i < 16; i++) m[i] = 0.0; }
  template <class P> static OpenGlMatrix ColMajor4x4(const P *col_major) { OpenGlMatrix r; for (int i = 0; i < 16; i++) r.m[i] = (double)col_major[i]; return r; }
  friend OpenGlMatrix operator*(const OpenGlMatrix &l, const OpenGlMatrix &rr) {
    OpenGlMatrix o;
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) { double s = 0; for (int k = 0; k < 4; k++) s += l.m[k * 4 + r] * rr.m[c * 4 + k]; o.m[c * 4 + r] = s; }
    return o;
  }
  void SetIdentity() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
};
inline OpenGlMatrix IdentityMatrix() { OpenGlMatrix r; r.SetIdentity(); return r; }
}  // namespace pangolin
