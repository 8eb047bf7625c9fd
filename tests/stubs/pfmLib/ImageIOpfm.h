// Stand-in for pfmLib's ImageIOpfm.h (an empty submodule of the reference): ReadFilePFM as the reference calls it
// (PrecomputedDepthProvider.cpp:31), implemented with the library under test (dsr_read_pfm).  See tests/stubs/README.md.
#pragma once
#include <stdexcept>
#include <string>
#include <opencv2/opencv.hpp>
extern "C" int dsr_read_pfm(const char *path, float *out, int capacity, int *width, int *height);
inline int ReadFilePFM(cv::Mat &im, const std::string &path) {
  int w = 0, h = 0;
  dsr_read_pfm(path.c_str(), nullptr, 0, &w, &h);  // size query
  if (w <= 0 || h <= 0) { im = cv::Mat(); return 0; }
  im.create(h, w, CV_32FC1);
  if (dsr_read_pfm(path.c_str(), reinterpret_cast<float *>(im.data), w * h, &w, &h) != 0) { im = cv::Mat(); return 0; }
  return 1;
}
