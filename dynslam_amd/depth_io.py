"""Precomputed depth / disparity maps from disk — the "fed by the ELAS/DispNet depth map" edge of the path.

Mirror of `dynslam::PrecomputedDepthProvider` (src/DynSLAM/PrecomputedDepthProvider.{h,cpp}):
  ReadPrecomputed(frame_idx)  .cpp:22-75   OpenCV-XML (int16 depth, ELAS) or PFM (float disparity, DispNet),
                                          chosen by the file name's ending, + the input_is_depth clamp
  GetDepth(frame_idx, calib)  .h:46-68     depth map as-is, or DepthFromDisparityMap for float disparity
File parsing is host code inside libdsr_hip.so (`dsr_read_depth_xml`, `dsr_read_pfm`); the clamp and the
disparity -> depth conversion run on the GPU (`dsr_clip_depth_mm`, `dsr_depth_from_disparity`).  No CPU fallback.
"""
import ctypes as C

import numpy as np

from ._capi import DSR_OK


class DepthIOError(RuntimeError):
    """std::runtime_error of ReadPrecomputed (.cpp:40,43,47-52)."""


def _check(api, st):
    if st != DSR_OK:
        msg = api.last_error()
        raise DepthIOError(msg.decode() if msg else f"dsr status {st}")


def read_depth_xml(path, api=None):
    """-> int16 [rows, cols] (mm) from an OpenCV FileStorage XML node "depth-frame"."""
    from .engine import load_hip_api
    api = api or load_hip_api()
    w, h = C.c_int(0), C.c_int(0)
    st = api.read_depth_xml(str(path).encode(), None, 0, C.byref(w), C.byref(h))  # size query
    if w.value <= 0 or h.value <= 0:
        _check(api, st)
    out = np.empty((h.value, w.value), np.int16)
    _check(api, api.read_depth_xml(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h)))
    return out


def read_pfm(path, api=None):
    """-> float32 [rows, cols], top row first, from a single-channel PFM file."""
    from .engine import load_hip_api
    api = api or load_hip_api()
    w, h = C.c_int(0), C.c_int(0)
    st = api.read_pfm(str(path).encode(), None, 0, C.byref(w), C.byref(h))
    if w.value <= 0 or h.value <= 0:
        _check(api, st)
    out = np.empty((h.value, w.value), np.float32)
    _check(api, api.read_pfm(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h)))
    return out


class PrecomputedDepthProvider:
    """PrecomputedDepthProvider(input, folder, fname_format, input_is_depth, frame_offset, min_depth_m, max_depth_m)
    without the `Input` back-pointer: frames are addressed by index."""

    kMetersToMillimeters = 1000.0

    def __init__(self, folder, fname_format, input_is_depth, min_depth_m, max_depth_m, api=None):
        from .engine import load_hip_api
        self.api = api or load_hip_api()
        self.folder, self.fname_format = str(folder), fname_format
        self.input_is_depth = bool(input_is_depth)
        self.min_depth_m, self.max_depth_m = float(min_depth_m), float(max_depth_m)

    def GetName(self):
        return "precomputed-dispnet" if self.fname_format.endswith("pfm") else "precomputed-elas"

    def ReadPrecomputed(self, frame_idx):
        path = self.folder + "/" + (self.fname_format % frame_idx)
        out = read_pfm(path, self.api) if path.endswith(".pfm") else read_depth_xml(path, self.api)
        if self.input_is_depth:
            if out.dtype == np.int16:
                _check(self.api, self.api.clip_depth_mm(out.ctypes.data_as(C.c_void_p), out.size, self.max_depth_m))
            else:  # float branch of the clamp (.cpp:63-68): depth > max_depth_mm_f -> 0
                out[out > np.float32(self.max_depth_m * self.kMetersToMillimeters)] = 0.0
        return out

    def GetDepth(self, frame_idx, baseline_m, focal_px, scale=1.0):
        """-> int16 mm depth map (what InfiniTamDriver::UpdateView takes)."""
        m = self.ReadPrecomputed(frame_idx)
        if self.input_is_depth:
            if m.dtype != np.int16:
                raise DepthIOError("a depth map read directly must be int16 (cv::Mat1s)")
            return m
        if m.dtype != np.float32:
            raise DepthIOError("Unsupported.")  # .h:60-61: int16 disparity
        out = np.empty(m.shape, np.int16)
        _check(self.api, self.api.depth_from_disparity(m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), m.size,
                                                       float(baseline_m), float(focal_px), float(scale), self.min_depth_m,
                                                       self.max_depth_m))
        return out
