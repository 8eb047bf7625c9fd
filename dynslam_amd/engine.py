"""Host-side mirror of the reference's engine boundary, on top of the C ABI.

`EngineCore` is a thin numpy-facing wrapper around the `dsr_*` entry points of
include/dsr.h.  `InfiniTamDriver` re-creates the method names and argument
meaning of the reference's `dynslam::drivers::InfiniTamDriver`
(src/DynSLAM/InfiniTamDriver.h:79-300) so that tests read like calls the
reference host makes.

The product path is the HIP library `dynslam_amd/csrc/libdsr_hip.so`; loading
fails loudly when it is missing (there is NO CPU fallback).
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import (BLOCK_SIZE3, DSR_E_OUT_OF_BLOCKS, DSR_OK, Calib, KernelTime,
                    Settings, Stats)

HASH_ENTRY_DTYPE = np.dtype([("pos", "<i2", (3,)), ("pad", "<i2"), ("offset", "<i4"), ("ptr", "<i4")])
VOXEL_DTYPE = np.dtype([("sdf", "<i2"), ("w_depth", "u1"), ("clr", "u1", (3,)), ("w_color", "u1"), ("pad", "u1")])
assert HASH_ENTRY_DTYPE.itemsize == 16 and VOXEL_DTYPE.itemsize == 8

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("DSR_HIP_LIB") or os.path.join(_HERE, "csrc", "libdsr_hip.so")  # (override: kernel experiments, tools/bench_variants.py)

_hip_api = None


class DsrError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"dsr status {status}: {message}")
        self.status = status


class OutOfBlocksError(DsrError):
    """The fork's std::runtime_error on block exhaustion
    (caught at InstanceReconstructor.cpp:662-671)."""


def load_hip_api():
    """Load libdsr_hip.so and bind every symbol of include/dsr.h.  No fallback."""
    global _hip_api
    if _hip_api is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError(
                f"{HIP_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        _capi.preload_hip_runtime()
        lib = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL)
        older = bool(os.environ.get("DSR_HIP_LIB")) and bool(os.environ.get("DSR_HIP_LIB_OLDER_ABI"))  # A/B tools only
        api = _capi.bind(lib, "dsr_", allow_missing=older)
        if api.abi_version() != _capi.ABI_VERSION and not older:
            raise ImportError("libdsr_hip.so ABI version mismatch")
        _hip_api = api
    return _hip_api


def default_settings(api=None, **overrides):
    api = api or load_hip_api()
    s = Settings()
    api.default_settings(C.byref(s))
    for k, v in overrides.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def make_calib(fx, fy, cx, cy, width, height):
    """CreateItmCalib (InfiniTamDriver.cpp:49-79): identical rgb/depth intrinsics,
    identity extrinsic, affine disparity calib (0.001, 0)."""
    c = Calib()
    for intr in (c.rgb, c.depth):
        intr.fx, intr.fy, intr.cx, intr.cy = fx, fy, cx, cy
        intr.width, intr.height = width, height
    ident = np.eye(4, dtype=np.float32).T.reshape(-1)
    c.trafo_rgb_to_depth[:] = ident.tolist()
    c.disparity_calib[0] = 1.0 / 1000.0
    c.disparity_calib[1] = 0.0
    return c


def _colmajor(m):
    """4x4 math matrix -> float[16] column-major (ORUtils::Matrix4f::m)."""
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(4, 4).T).reshape(-1)
    return a


class PoseArg:
    """A 4x4 pose converted ONCE to the float[16] column-major buffer the C ABI takes.  Every entry point that takes a pose accepts
    one in place of the matrix: a host that calls per frame and per volume (bench.py's loop, ShardedScene) then pays a pointer
    hand-over instead of a numpy transpose + a ctypes view per call (~10 us each in CPython, 16 of them per step of 8 volumes)."""
    __slots__ = ("buf", "ptr")

    def __init__(self, m):
        self.buf = (C.c_float * 16)(*_colmajor(m).tolist())
        self.ptr = C.cast(self.buf, C.c_void_p)


def _pose_arg(m):
    """-> (object to keep alive during the call, c_void_p)"""
    if isinstance(m, PoseArg):
        return m, m.ptr
    a = _colmajor(m)
    return a, _ptr(a)


def _from_colmajor(buf):
    return np.asarray(buf, dtype=np.float32).reshape(4, 4).T.copy()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Exchange:
    """The fused-preview exchange of include/dsr.h (`dsr_exchange_*`): layer buffers on every GPU this process drives, the RCCL
    all-gather between them (called by the library itself) and the composite.  `devices`: one process drives all GPUs, rank r
    on devices[r];  `unique_id` + `world_size` + `rank` + `device`: one process per GPU (the id comes from
    `Exchange.unique_id()` on rank 0 and travels to the others by the caller's means)."""

    def __init__(self, n_pixels, slots_per_rank, devices=None, unique_id=None, world_size=None, rank=None, device=-1, api=None):
        self.api = api or load_hip_api()
        self.P, self.slots = int(n_pixels), int(slots_per_rank)
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            self._check(self.api.exchange_create(arr, len(devices), self.slots, self.P, C.byref(h)))
            self.n_ranks = len(devices)
        else:
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
            self._check(self.api.exchange_create_rank(buf, int(world_size), int(rank), int(device), self.slots, self.P, C.byref(h)))
            self.n_ranks = int(world_size)
        self._h = h

    @staticmethod
    def unique_id(api=None):
        api = api or load_hip_api()
        buf = (C.c_uint8 * 128)()
        if api.exchange_unique_id(buf) != DSR_OK:
            msg = api.last_error()
            raise DsrError(2, msg.decode() if msg else "exchange_unique_id")
        return bytes(buf)

    def _check(self, status):
        if status != DSR_OK:
            msg = self.api.last_error()
            raise DsrError(status, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None):
            self.api.exchange_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self, rank):
        return self.api.exchange_stream(self._h, int(rank))

    def slot_ptrs(self, rank, slot):
        r, d = C.c_void_p(), C.c_void_p()
        self._check(self.api.exchange_slot_ptrs(self._h, int(rank), int(slot), C.byref(r), C.byref(d)))
        return r.value, d.value

    def layer_ptrs(self, on_rank, rank, slot):
        r, d = C.c_void_p(), C.c_void_p()
        self._check(self.api.exchange_layer_ptrs(self._h, int(on_rank), int(rank), int(slot), C.byref(r), C.byref(d)))
        return r.value, d.value

    def target_ptrs(self, rank):
        r, d = C.c_void_p(), C.c_void_p()
        self._check(self.api.exchange_target_ptrs(self._h, int(rank), C.byref(r), C.byref(d)))
        return r.value, d.value

    def render_slot(self, rank, slot, engine, image_type=_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=None, intrinsics=None):
        keep, pm_ptr = _pose_arg(pose_m) if pose_m is not None else (None, None)
        intr = np.ascontiguousarray(intrinsics, dtype=np.float32) if intrinsics is not None else None
        self._check(self.api.exchange_render_slot(self._h, int(rank), int(slot), engine._h if engine is not None else None, int(image_type),
                                                  pm_ptr, _ptr(intr) if intr is not None else None))

    def gather(self):
        self._check(self.api.exchange_gather(self._h))

    def set_collective(self, gather_to_root, root_rank=0):
        """0: in-place all-gather; 1: gather to `root_rank`'s GPU only (dsr_exchange_set_collective)."""
        self._check(self.api.exchange_set_collective(self._h, int(bool(gather_to_root)), int(root_rank)))

    def timing(self, enable=True):
        """-> {gather_ms, composite_ms, n_gathers, n_composites} since the last call; `enable` switches the event timing on / off."""
        g, c, ng, nc = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        self._check(self.api.exchange_timing(self._h, int(bool(enable)), C.byref(g), C.byref(c), C.byref(ng), C.byref(nc)))
        return {"gather_ms": g.value, "composite_ms": c.value, "n_gathers": ng.value, "n_composites": nc.value}

    def clear_target(self, rank):
        self._check(self.api.exchange_clear_target(self._h, int(rank)))

    def gather_and_composite(self, root_rank, layers, target_engine=None, target_rgba_ptr=None, target_depth_ptr=None, tint_strength=1.0,
                             dim_background=True, gather=True):
        """layers: [(rank, slot, track id)] in compositing order (ascending track id)."""
        n = len(layers)
        key = tuple(layers)
        hit = getattr(self, "_layer_cache", None)
        if hit is None or hit[0] != key:  # (the same layer set frame after frame: the three ctypes arrays are built once)
            hit = self._layer_cache = (key, (C.c_int32 * max(1, n))(*[int(l[0]) for l in layers]),
                                       (C.c_int32 * max(1, n))(*[int(l[1]) for l in layers]), (C.c_int32 * max(1, n))(*[int(l[2]) for l in layers]))
        _, ranks, slots, tids = hit
        fn = self.api.exchange_gather_and_composite if gather else self.api.exchange_composite
        self._check(fn(self._h, int(root_rank), target_engine._h if target_engine is not None else None,
                       C.c_void_p(target_rgba_ptr) if target_rgba_ptr else None, C.c_void_p(target_depth_ptr) if target_depth_ptr else None,
                       ranks, slots, tids, n, float(tint_strength), int(bool(dim_background))))

    def read_target(self, rank, width, height):
        rgba = np.empty((height, width, 4), np.uint8)
        depth = np.empty((height, width), np.float32)
        self._check(self.api.exchange_read_target(self._h, int(rank), _ptr(rgba), _ptr(depth)))
        return rgba, depth

    def sync(self):
        self._check(self.api.exchange_sync(self._h))


class Batch:
    """The instance volumes of one GPU driven together (`dsr_batch_*`, include/dsr.h "volume batch"): `fuse` = silhouette split,
    SetPose, Integrate and PrepareNextStep of every listed instance in 2 + 6 launches, `render` = their preview renders in two.
    `source`: the engine that holds the full frame; `volumes`: 1..8 instance-sized engines on its GPU."""

    def __init__(self, source, volumes, api=None):
        self.api = api or source.api
        self.source, self.volumes = source, list(volumes)
        arr = (C.c_void_p * len(self.volumes))(*[v._h for v in self.volumes])
        h = C.c_void_p()
        self._check(self.api.batch_create(source._h, arr, len(self.volumes), C.byref(h)))
        self._h = h
        self._items = (_capi.BatchItem * 16)()
        self._ritems = (_capi.BatchRenderItem * 8)()
        self._status = (C.c_int32 * 16)()

    def _check(self, status):
        if status != DSR_OK:
            msg = self.api.last_error()
            raise DsrError(status, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None):
            self.api.batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fuse(self, items, want_status=False):
        """items: [(volume index or -1, copy mask (device pointer, w, h) or None, x0, y0, delete mask (device pointer, w, h) or None,
        dx0, dy0, camera->object pose (matrix or PoseArg) or None)] in the host's order.  -> per-item status list if asked for."""
        n = len(items)
        if n > len(self._items):
            self._items = (_capi.BatchItem * n)()
            self._status = (C.c_int32 * n)()
        for it, (vol, cm, x0, y0, dm, dx0, dy0, pose) in zip(self._items, items):
            it.volume = int(vol)
            it.copy_mask_dev, it.box_w, it.box_h = (cm[0], int(cm[1]), int(cm[2])) if cm is not None else (None, 0, 0)
            it.x0, it.y0 = int(x0), int(y0)
            it.delete_mask_dev, it.dbox_w, it.dbox_h = (dm[0], int(dm[1]), int(dm[2])) if dm is not None else (None, 0, 0)
            it.dx0, it.dy0 = int(dx0), int(dy0)
            if pose is not None:
                if isinstance(pose, PoseArg):
                    C.memmove(it.inv_m, pose.buf, 64)
                else:
                    it.inv_m[:] = _colmajor(pose).tolist()
        self._check(self.api.batch_fuse(self._h, self._items, n, self._status if want_status else None))
        return list(self._status[:n]) if want_status else None

    def render(self, items, image_type=_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME):
        """items: [(volume index, object->camera pose (matrix or PoseArg), rgba device pointer or None, depth device pointer or None)]"""
        n = len(items)
        for it, (vol, pose, rp, dp) in zip(self._ritems, items):
            it.volume = int(vol)
            it.rgba_out_dev, it.depth_out_dev = rp, dp
            if isinstance(pose, PoseArg):
                C.memmove(it.pose_m, pose.buf, 64)
            else:
                it.pose_m[:] = _colmajor(pose).tolist()
        self._check(self.api.batch_render(self._h, int(image_type), self._ritems, n))


class EngineCore:
    """One engine handle (an ITMMainEngine: scene + render states + view + pose)."""

    def __init__(self, settings, calib, api=None):
        self.api = api or load_hip_api()
        self.settings = settings
        self.calib = calib
        self.W, self.H = calib.depth.width, calib.depth.height
        self.no_total_entries = settings.hash_bucket_num + settings.excess_list_size
        self.no_blocks = settings.sdf_local_block_num
        h = C.c_void_p()
        self._check(self.api.engine_create(C.byref(settings), C.byref(calib), C.byref(h)))
        self._h = h

    # -- plumbing -----------------------------------------------------------
    def _check(self, status):
        if status == DSR_OK:
            return
        msg = self.api.last_error()
        msg = msg.decode() if msg else ""
        if status == DSR_E_OUT_OF_BLOCKS:
            raise OutOfBlocksError(status, msg)
        raise DsrError(status, msg)

    def close(self):
        if getattr(self, "_h", None):
            self.api.engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.api.sync(self._h))

    def wait_for_stream(self, hip_stream):
        """Engine work queued after this call starts when `hip_stream` (a hipStream_t as int) has drained."""
        self._check(self.api.wait_for_stream(self._h, C.c_void_p(hip_stream)))

    def stream_wait_for_engine(self, hip_stream):
        """Work queued on `hip_stream` after this call starts when the engine's queued work has finished."""
        self._check(self.api.stream_wait_for_engine(self._h, C.c_void_p(hip_stream)))

    def reset_scene(self):
        self._check(self.api.reset_scene(self._h))

    # -- view ---------------------------------------------------------------
    def update_view(self, rgba, depth_mm):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        depth_mm = np.ascontiguousarray(depth_mm, dtype=np.int16)
        assert rgba.shape == (self.H, self.W, 4) and depth_mm.shape == (self.H, self.W)
        self._check(self.api.update_view(self._h, _ptr(rgba), _ptr(depth_mm)))

    def update_view_bgr(self, bgr, depth_mm):
        """InfiniTamDriver::UpdateView from the host's own cv::Mat3b layout (packed BGR) + int16 mm: converted in the ingest kernel."""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_mm = np.ascontiguousarray(depth_mm, dtype=np.int16)
        assert bgr.shape == (self.H, self.W, 3) and depth_mm.shape == (self.H, self.W)
        self._check(self.api.update_view_bgr(self._h, _ptr(bgr), _ptr(depth_mm)))

    def update_view_dev(self, rgba_dev_ptr, depth_mm_dev_ptr):
        self._check(self.api.update_view_dev(self._h, C.c_void_p(rgba_dev_ptr), C.c_void_p(depth_mm_dev_ptr)))

    def set_view_float(self, rgba, depth_m):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        depth_m = np.ascontiguousarray(depth_m, dtype=np.float32)
        assert rgba.shape == (self.H, self.W, 4) and depth_m.shape == (self.H, self.W)
        self._check(self.api.set_view_float(self._h, _ptr(rgba), _ptr(depth_m)))

    def set_view_float_dev(self, rgba_dev_ptr, depth_m_dev_ptr):
        self._check(self.api.set_view_float_dev(self._h, C.c_void_p(rgba_dev_ptr), C.c_void_p(depth_m_dev_ptr)))

    def get_view(self):
        rgba = np.empty((self.H, self.W, 4), np.uint8)
        depth = np.empty((self.H, self.W), np.float32)
        self._check(self.api.get_view(self._h, _ptr(rgba), _ptr(depth)))
        return rgba, depth

    def extract_silhouette(self, instance, mask, x0, y0):
        """ProcessSilhouette (InstanceReconstructor.cpp:59-133): `instance`'s view := this
        engine's view under the bbox-local uint8 mask placed at (x0, y0)."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.api.view_extract_silhouette(self._h, instance._h, _ptr(mask), int(x0), int(y0),
                                                     mask.shape[1], mask.shape[0]))

    def extract_silhouette_dev(self, instance, mask_dev_ptr, x0, y0, box_w, box_h):
        """... with the mask already in HBM: no copy, no synchronisation (dsr_view_extract_silhouette_dev)."""
        self._check(self.api.view_extract_silhouette_dev(self._h, instance._h, C.c_void_p(mask_dev_ptr), int(x0), int(y0),
                                                         int(box_w), int(box_h)))

    def remove_silhouette_dev(self, mask_dev_ptr, x0, y0, box_w, box_h):
        self._check(self.api.view_remove_silhouette_dev(self._h, C.c_void_p(mask_dev_ptr), int(x0), int(y0), int(box_w), int(box_h)))

    def split_silhouette(self, instance, mask, x0, y0, delete_mask=None, dx0=None, dy0=None):
        """ProcessSilhouette + RemoveSilhouette of one instance in one launch (dsr_view_split_silhouette); `delete_mask`
        defaults to the copy mask at the same place."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        dm = mask if delete_mask is None else np.ascontiguousarray(delete_mask, dtype=np.uint8)
        dx0, dy0 = (x0 if dx0 is None else dx0), (y0 if dy0 is None else dy0)
        self._check(self.api.view_split_silhouette(self._h, instance._h, _ptr(mask), int(x0), int(y0), mask.shape[1], mask.shape[0],
                                                   _ptr(dm), int(dx0), int(dy0), dm.shape[1], dm.shape[0]))

    def split_silhouette_dev(self, instance, mask_dev_ptr, x0, y0, box_w, box_h, delete_mask_dev_ptr=None, dx0=None, dy0=None,
                             dbox_w=None, dbox_h=None):
        """... with the masks already in HBM."""
        if delete_mask_dev_ptr is None:
            delete_mask_dev_ptr, dx0, dy0, dbox_w, dbox_h = mask_dev_ptr, x0, y0, box_w, box_h
        self._check(self.api.view_split_silhouette_dev(self._h, instance._h, C.c_void_p(mask_dev_ptr), int(x0), int(y0), int(box_w),
                                                       int(box_h), C.c_void_p(delete_mask_dev_ptr), int(dx0), int(dy0), int(dbox_w),
                                                       int(dbox_h)))

    def share_stream(self, owner):
        """Queue this engine's work on `owner`'s stream from now on (dsr_engine_share_stream): one volume next to its view engine
        on a GPU of their own needs no cross-stream event per frame."""
        self._check(self.api.engine_share_stream(self._h, owner._h))

    def remove_silhouette(self, mask, x0, y0):
        """RemoveSilhouette (InstanceReconstructor.cpp:135-170)."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.api.view_remove_silhouette(self._h, _ptr(mask), int(x0), int(y0), mask.shape[1], mask.shape[0]))

    # -- pose ---------------------------------------------------------------
    def set_pose_inv_m(self, inv_m):
        keep, ptr = _pose_arg(inv_m)
        self._check(self.api.set_pose_inv_m(self._h, ptr))

    def set_pose_m(self, m):
        a = _colmajor(m)
        self._check(self.api.set_pose_m(self._h, _ptr(a)))

    def get_pose(self):
        m = np.empty(16, np.float32)
        im = np.empty(16, np.float32)
        self._check(self.api.get_pose(self._h, _ptr(m), _ptr(im)))
        return _from_colmajor(m), _from_colmajor(im)

    # -- fusion -------------------------------------------------------------
    def set_fusion_weight_params(self, depth_weighting):
        self._check(self.api.set_fusion_weight_params(self._h, int(bool(depth_weighting))))

    def process_frame(self):
        self._check(self.api.process_frame(self._h))

    def allocate_scene_from_depth(self):
        self._check(self.api.allocate_scene_from_depth(self._h))

    def integrate_into_scene(self):
        self._check(self.api.integrate_into_scene(self._h))

    def prepare(self):
        self._check(self.api.prepare(self._h))

    def decay(self, max_weight, min_age, force_all_voxels=False):
        self._check(self.api.decay(self._h, int(max_weight), int(min_age), int(bool(force_all_voxels))))

    # -- rendering ----------------------------------------------------------
    def get_image(self, image_type, pose_m=None, intrinsics=None, want_rgba=True, want_depth=False):
        rgba = np.zeros((self.H, self.W, 4), np.uint8) if want_rgba else None
        depth = np.zeros((self.H, self.W), np.float32) if want_depth else None
        pm = _colmajor(pose_m) if pose_m is not None else None
        intr = np.ascontiguousarray(intrinsics, dtype=np.float32) if intrinsics is not None else None
        self._check(self.api.get_image(
            self._h, int(image_type), _ptr(pm) if pm is not None else None,
            _ptr(intr) if intr is not None else None,
            _ptr(rgba) if rgba is not None else None, _ptr(depth) if depth is not None else None))
        return rgba, depth

    def get_image_dev(self, image_type, pose_m, intrinsics, rgba_dev_ptr, depth_dev_ptr):
        keep, pm_ptr = _pose_arg(pose_m) if pose_m is not None else (None, None)
        intr = np.ascontiguousarray(intrinsics, dtype=np.float32) if intrinsics is not None else None
        self._check(self.api.get_image_dev(
            self._h, int(image_type), pm_ptr,
            _ptr(intr) if intr is not None else None,
            C.c_void_p(rgba_dev_ptr) if rgba_dev_ptr else None,
            C.c_void_p(depth_dev_ptr) if depth_dev_ptr else None))

    # -- statistics and parity dumps -----------------------------------------
    def get_stats(self):
        st = Stats()
        self._check(self.api.get_stats(self._h, C.byref(st)))
        return st

    def dump_hash_table(self):
        out = np.empty(self.no_total_entries, HASH_ENTRY_DTYPE)
        self._check(self.api.dump_hash_table(self._h, _ptr(out)))
        return out

    def dump_visible_list(self, freeview=False):
        ids = np.empty(self.no_blocks, np.int32)
        n = C.c_int32(0)
        self._check(self.api.dump_visible_list(self._h, int(freeview), _ptr(ids), C.byref(n)))
        return ids[: n.value].copy()

    def dump_visible_types(self):
        out = np.empty(self.no_total_entries, np.uint8)
        self._check(self.api.dump_visible_types(self._h, _ptr(out)))
        return out

    def dump_voxel_blocks(self, first_block=0, n_blocks=None):
        if n_blocks is None:
            n_blocks = self.no_blocks - first_block
        out = np.empty((n_blocks, BLOCK_SIZE3), VOXEL_DTYPE)
        self._check(self.api.dump_voxel_blocks(self._h, int(first_block), int(n_blocks), _ptr(out)))
        return out

    def dump_allocation_lists(self):
        v = np.empty(self.no_blocks, np.int32)
        x = np.empty(self.settings.excess_list_size, np.int32)
        self._check(self.api.dump_allocation_lists(self._h, _ptr(v), _ptr(x)))
        return v, x

    def dump_swap_state(self):
        st = np.empty(self.no_total_entries, np.uint8)
        hs = np.empty(self.no_total_entries, np.uint8)
        self._check(self.api.dump_swap_state(self._h, _ptr(st), _ptr(hs)))
        return st, hs

    def dump_stored_block(self, entry):
        out = np.empty(BLOCK_SIZE3, VOXEL_DTYPE)
        present = C.c_int(0)
        self._check(self.api.dump_stored_block(self._h, int(entry), _ptr(out), C.byref(present)))
        return out if present.value else None

    def dump_render_state(self, freeview=False):
        mw, mh = (self.W + 7) // 8, (self.H + 7) // 8
        out = {
            "minmax": np.empty((mh, mw, 2), np.float32),
            "raycast_result": np.empty((self.H, self.W, 4), np.float32),
            "points": np.empty((self.H, self.W, 4), np.float32),
            "normals": np.empty((self.H, self.W, 4), np.float32),
            "raycast_image": np.empty((self.H, self.W, 4), np.uint8),
        }
        self._check(self.api.dump_render_state(
            self._h, int(freeview), _ptr(out["minmax"]), _ptr(out["raycast_result"]), _ptr(out["points"]),
            _ptr(out["normals"]), _ptr(out["raycast_image"])))
        return out

    # -- profiling ------------------------------------------------------------
    def profile_enable(self, enable=True):
        """True / 1: HIP events around every kernel; 2: around integrate and raycast only."""
        self._check(self.api.profile_enable(self._h, 2 if enable == 2 and enable is not True else int(bool(enable))))

    # ---- meshing (ITMMeshingEngine::MeshScene / ITMMesh::WriteOBJ / ITMMainEngine::SaveSceneToMesh)
    def mesh_scene(self):
        """Marching cubes over the allocated blocks; returns the triangles as float32 [n, 3, 3] (metres)."""
        n = C.c_uint64(0)
        self._check(self.api.mesh_scene(self._h, C.byref(n)))
        out = np.empty((n.value, 3, 3), np.float32)
        if n.value:
            self._check(self.api.mesh_get(self._h, out.ctypes.data_as(C.c_void_p), 0, n.value))
        return out

    def mesh_write_obj(self, path):
        self._check(self.api.mesh_write_obj(self._h, str(path).encode()))

    def mesh_free(self):
        self._check(self.api.mesh_free(self._h))

    def save_scene_to_mesh(self, path):
        self._check(self.api.save_scene_to_mesh(self._h, str(path).encode()))

    def profile_reset(self):
        self._check(self.api.profile_reset(self._h))

    def profile_get(self):
        buf = (KernelTime * 64)()
        n = self.api.profile_get(self._h, buf, 64)
        return [dict(name=buf[i].name.decode(), total_ms=buf[i].total_ms, launches=buf[i].launches,
                     bytes=buf[i].bytes, bytes_layout=buf[i].bytes_layout, units=buf[i].units,
                     store_lanes=buf[i].store_lanes, colour_voxels=buf[i].colour_voxels) for i in range(n)]


class VoxelDecayParams:
    """src/DynSLAM/VoxelDecayParams.h"""

    def __init__(self, enabled=True, min_decay_age=200, max_decay_weight=1):
        self.enabled = enabled
        self.min_decay_age = min_decay_age
        self.max_decay_weight = max_decay_weight


class PreviewType:
    """src/DynSLAM/PreviewType.h"""
    kDepth, kGray, kColor, kNormal, kWeight, kLatestRaycast = range(6)


_PREVIEW_TO_IMAGE = {  # GetItmVisualization, InfiniTamDriver.cpp:16-34
    PreviewType.kDepth: _capi.IMAGE_FREECAMERA_DEPTH,
    PreviewType.kGray: _capi.IMAGE_FREECAMERA_SHADED,
    PreviewType.kColor: _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME,
    PreviewType.kNormal: _capi.IMAGE_FREECAMERA_COLOUR_FROM_NORMAL,
    PreviewType.kWeight: _capi.IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT,
    PreviewType.kLatestRaycast: _capi.IMAGE_SCENERAYCAST,
}


class InfiniTamDriver:
    """Mirror of dynslam::drivers::InfiniTamDriver (InfiniTamDriver.h:79-300).

    Same method names, argument meaning and error behaviour:
      UpdateView(rgb, raw_depth)  .cpp:211-224   (rgb here is RGBA uint8, depth int16 mm)
      SetPose(new_pose)           .h:131-135     (camera->world, sets pose_d.invM)
      Integrate()                 .h:137-146     (raises OutOfBlocksError like the fork throws)
      PrepareNextStep()           .h:148-158
      Decay()/DecayCatchup()/Reap(w)  .h:201-235
      GetImage/GetFloatImage      .cpp:165-209   (silent no-op before the first frame)
      GetUsedMemoryBytes/GetSavedDecayMemoryBytes  .h:241-250
      Reset()                     .h:282-284
      SaveSceneToMesh(path)       ITMMainEngine (DynSlam.cpp:188-196); WaitForMeshDump() .h:252-255
    """

    def __init__(self, settings, calib, voxel_decay_params=None, use_depth_weighting=False, api=None):
        self.core = EngineCore(settings, calib, api=api)
        self.voxel_decay_params = voxel_decay_params or VoxelDecayParams(False, 0, 0)
        self.use_depth_weighting = bool(use_depth_weighting)
        self._has_view = False
        self._last_egomotion = np.eye(4, dtype=np.float32)

    def UpdateView(self, rgba_image, raw_depth_image):
        self.core.update_view(rgba_image, raw_depth_image)
        self._has_view = True

    def SetView(self, rgba_image, depth_m):
        """SetView(ITMView*) with the already converted instance view
        (InstanceReconstructor.cpp:580)."""
        self.core.set_view_float(rgba_image, depth_m)
        self._has_view = True

    def SetPose(self, new_pose):
        _, old_inv = self.core.get_pose()
        self._last_egomotion = np.linalg.inv(old_inv) @ np.asarray(new_pose, np.float32)
        self.core.set_pose_inv_m(new_pose)

    def GetPose(self):
        return self.core.get_pose()[1]

    def GetLastEgomotion(self):
        return self._last_egomotion

    def Integrate(self):
        self.core.set_fusion_weight_params(self.use_depth_weighting)
        self.core.process_frame()

    def PrepareNextStep(self):
        self.core.prepare()

    def Decay(self):
        p = self.voxel_decay_params
        if p.enabled:
            self.core.decay(p.max_decay_weight, p.min_decay_age, False)

    def DecayCatchup(self):
        p = self.voxel_decay_params
        if p.enabled:
            for _ in range(p.min_decay_age):
                self.core.decay(p.max_decay_weight, 0, False)

    def Reap(self, max_decay_weight):
        if self.voxel_decay_params.enabled:
            self.core.decay(max_decay_weight, 0, True)

    def GetImage(self, preview_type, model_view=None):
        if not self._has_view:
            return None
        if preview_type == PreviewType.kDepth:
            return None  # "Cannot preview depth normally anymore." .cpp:171-175
        rgba, _ = self.core.get_image(_PREVIEW_TO_IMAGE[preview_type], pose_m=model_view)
        return rgba

    def GetFloatImage(self, preview_type, model_view=None):
        if not self._has_view:
            return None
        if preview_type != PreviewType.kDepth:
            return None  # "Can only preview depth as float." .cpp:196-199
        _, depth = self.core.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=model_view, want_rgba=False,
                                       want_depth=True)
        return depth

    def GetVoxelSizeBytes(self):
        return self.core.get_stats().voxel_bytes

    def GetUsedMemoryBytes(self):
        st = self.core.get_stats()
        num_used_blocks = st.num_allocated_voxel_blocks - st.last_free_block_id
        return st.voxel_bytes * st.block_voxels * num_used_blocks

    def GetSavedDecayMemoryBytes(self):
        st = self.core.get_stats()
        return st.decayed_block_count * st.voxel_bytes * st.block_voxels

    def IsDecayEnabled(self):
        return self.voxel_decay_params.enabled

    def IsUsingDepthWeights(self):
        return self.use_depth_weighting

    def Reset(self):
        self.core.reset_scene()

    def SaveSceneToMesh(self, path):
        """ITMMainEngine::SaveSceneToMesh as called by DynSlam::SaveStaticMap (DynSlam.cpp:188-196)
        and, per instance, InstanceReconstructor::SaveObjectToMesh (InstanceReconstructor.cpp:736-763)."""
        self.core.save_scene_to_mesh(path)

    def WaitForMeshDump(self):
        """InfiniTamDriver.h:252-255 joins the fork's asynchronous dump thread; the dump here is
        synchronous, so there is nothing to wait for."""
        return None
