// Stand-in for <pangolin/pangolin.h>: the OpenGL model-view matrix type the driver converts poses from
// (InfiniTamDriver.cpp:36-47).  See tests/stubs/README.md.
#pragma once
namespace pangolin {
struct OpenGlMatrix {
  double m[16];  // column-major, like OpenGL
  OpenGlMatrix() { for (int i = 0; i < 16; i++) m[i] = 0.0; }
  void SetIdentity() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
};
inline OpenGlMatrix IdentityMatrix() { OpenGlMatrix r; r.SetIdentity(); return r; }
}  // namespace pangolin
