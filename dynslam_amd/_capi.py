"""ctypes declarations of the C ABI in include/dsr.h.

`bind(lib, prefix)` attaches argtypes/restypes for every entry point the header
declares; the product binds `dsr_` from libdsr_hip.so (engine.py), the test
infrastructure binds the same signatures with the prefix `orc_` from the CPU checker library.
"""
import ctypes as C
import os
from types import SimpleNamespace

VIEW_PIPELINE_AUTO, VIEW_PIPELINE_OFF, VIEW_PIPELINE_PER_ENGINE, VIEW_PIPELINE_SHARED = 0, 1, 2, 3  # dsr_settings.view_pipeline
ABI_VERSION = 5  # == DSR_ABI_VERSION of include/dsr.h (tests/test_capi_symbols.py compares the header too)
BLOCK_SIZE = 8
BLOCK_SIZE3 = 512

DSR_OK = 0
DSR_E_ARG = 1
DSR_E_DEVICE = 2
DSR_E_OUT_OF_BLOCKS = 3
DSR_E_NO_VIEW = 4
DSR_E_NOMEM = 5
DSR_E_IO = 6

IMAGE_ORIGINAL_RGB = 0
IMAGE_ORIGINAL_DEPTH = 1
IMAGE_SCENERAYCAST = 2
IMAGE_FREECAMERA_SHADED = 3
IMAGE_FREECAMERA_COLOUR_FROM_VOLUME = 4
IMAGE_FREECAMERA_COLOUR_FROM_NORMAL = 5
IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT = 6
IMAGE_FREECAMERA_DEPTH = 7


class HashEntry(C.Structure):
    _fields_ = [("pos", C.c_int16 * 3), ("_pad", C.c_int16), ("offset", C.c_int32), ("ptr", C.c_int32)]


class Voxel(C.Structure):
    _fields_ = [("sdf", C.c_int16), ("w_depth", C.c_uint8), ("clr", C.c_uint8 * 3),
                ("w_color", C.c_uint8), ("_pad", C.c_uint8)]


class Settings(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_float), ("mu", C.c_float), ("max_w", C.c_int32),
        ("view_frustum_min", C.c_float), ("view_frustum_max", C.c_float),
        ("stop_integrating_at_max_w", C.c_int32), ("sdf_local_block_num", C.c_int32),
        ("hash_bucket_num", C.c_int32), ("excess_list_size", C.c_int32),
        ("use_swapping", C.c_int32), ("use_bilateral_filter", C.c_int32),
        ("device", C.c_int32), ("sync_status", C.c_int32), ("view_pipeline", C.c_int32), ("reserved", C.c_int32 * 6),
    ]


class Intrinsics(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class Calib(C.Structure):
    _fields_ = [("rgb", Intrinsics), ("depth", Intrinsics),
                ("trafo_rgb_to_depth", C.c_float * 16), ("disparity_calib", C.c_float * 2)]


class Stats(C.Structure):
    _fields_ = [
        ("num_allocated_voxel_blocks", C.c_int32), ("last_free_block_id", C.c_int32),
        ("last_free_excess_list_id", C.c_int32), ("no_visible_blocks", C.c_int32),
        ("no_total_entries", C.c_int32), ("voxel_bytes", C.c_int32), ("block_voxels", C.c_int32),
        ("sticky_status", C.c_int32), ("decayed_block_count", C.c_int64),
        ("frames_processed", C.c_int64), ("no_visible_blocks_freeview", C.c_int32),
        ("host_store_slots", C.c_int32), ("host_store_capacity_slots", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("launches", C.c_int64),
                ("bytes", C.c_double), ("bytes_layout", C.c_double), ("units", C.c_double),
                ("store_lanes", C.c_double), ("colour_voxels", C.c_double)]


class BatchItem(C.Structure):  # dsr_batch_item
    _fields_ = [("volume", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32), ("box_w", C.c_int32), ("box_h", C.c_int32),
                ("dx0", C.c_int32), ("dy0", C.c_int32), ("dbox_w", C.c_int32), ("dbox_h", C.c_int32), ("reserved", C.c_int32),
                ("copy_mask_dev", C.c_void_p), ("delete_mask_dev", C.c_void_p), ("inv_m", C.c_float * 16)]


class BatchRenderItem(C.Structure):  # dsr_batch_render_item
    _fields_ = [("volume", C.c_int32), ("reserved", C.c_int32), ("rgba_out_dev", C.c_void_p), ("depth_out_dev", C.c_void_p),
                ("pose_m", C.c_float * 16)]


assert C.sizeof(HashEntry) == 16 and C.sizeof(Voxel) == 8 and C.sizeof(BatchItem) == 120 and C.sizeof(BatchRenderItem) == 88

_P = C.c_void_p
_H = C.c_void_p  # dsr_engine*
_X = C.c_void_p  # dsr_exchange*

# name -> (restype, argtypes); exactly the entry points declared in include/dsr.h
SIGNATURES = {
    "abi_version": (C.c_int, []),
    "default_settings": (None, [C.POINTER(Settings)]),
    "last_error": (C.c_char_p, []),
    "engine_create": (C.c_int, [C.POINTER(Settings), C.POINTER(Calib), C.POINTER(_H)]),
    "engine_destroy": (None, [_H]),
    "reset_scene": (C.c_int, [_H]),
    "sync": (C.c_int, [_H]),
    "device_synchronize": (C.c_int, []),
    "device_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "wait_for_stream": (C.c_int, [_H, _P]),
    "stream_wait_for_engine": (C.c_int, [_H, _P]),
    "engine_share_stream": (C.c_int, [_H, _H]),
    "pin_host_thread": (C.c_int, [C.c_int]),
    "batch_create": (C.c_int, [_H, C.POINTER(_H), C.c_int, C.POINTER(C.c_void_p)]),
    "batch_destroy": (None, [C.c_void_p]),
    "batch_fuse": (C.c_int, [C.c_void_p, C.POINTER(BatchItem), C.c_int, C.POINTER(C.c_int32)]),
    "batch_render": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(BatchRenderItem), C.c_int]),
    "update_view": (C.c_int, [_H, _P, _P]),
    "update_view_dev": (C.c_int, [_H, _P, _P]),
    "update_view_bgr": (C.c_int, [_H, _P, _P]),
    "set_view_float": (C.c_int, [_H, _P, _P]),
    "set_view_float_dev": (C.c_int, [_H, _P, _P]),
    "get_view": (C.c_int, [_H, _P, _P]),
    "get_view_previews": (C.c_int, [_H, _P, _P]),
    "get_no_visible_blocks": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "pin_host_buffer": (C.c_int, [_P, C.c_size_t]),
    "unpin_host_buffer": (C.c_int, [_P]),
    "set_pose_inv_m": (C.c_int, [_H, _P]),
    "set_pose_m": (C.c_int, [_H, _P]),
    "get_pose": (C.c_int, [_H, _P, _P]),
    "set_fusion_weight_params": (C.c_int, [_H, C.c_int]),
    "process_frame": (C.c_int, [_H]),
    "allocate_scene_from_depth": (C.c_int, [_H]),
    "integrate_into_scene": (C.c_int, [_H]),
    "prepare": (C.c_int, [_H]),
    "decay": (C.c_int, [_H, C.c_int, C.c_int, C.c_int]),
    "get_image": (C.c_int, [_H, C.c_int, _P, _P, _P, _P]),
    "get_image_dev": (C.c_int, [_H, C.c_int, _P, _P, _P, _P]),
    "depth_from_disparity": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "depth_from_disparity_dev": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "read_depth_xml": (C.c_int, [C.c_char_p, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "read_pfm": (C.c_int, [C.c_char_p, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "clip_depth_mm": (C.c_int, [_P, C.c_int, C.c_float]),
    "clip_depth_mm_dev": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_float]),
    "bgr_to_rgba": (C.c_int, [_P, _P, C.c_int]),
    "bgr_to_rgba_dev": (C.c_int, [C.c_int, _P, _P, _P, C.c_int]),
    "rgba_to_bgr": (C.c_int, [_P, _P, C.c_int]),
    "rgba_to_bgr_dev": (C.c_int, [C.c_int, _P, _P, _P, C.c_int]),
    "depth_m_to_mm": (C.c_int, [_P, _P, C.c_int]),
    "depth_m_to_mm_dev": (C.c_int, [C.c_int, _P, _P, _P, C.c_int]),
    "view_extract_silhouette": (C.c_int, [_H, _H, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "view_remove_silhouette": (C.c_int, [_H, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "view_extract_silhouette_dev": (C.c_int, [_H, _H, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "view_remove_silhouette_dev": (C.c_int, [_H, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "view_split_silhouette": (C.c_int, [_H, _H, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "view_split_silhouette_dev": (C.c_int, [_H, _H, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "composite_layer_ptrs_dev": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_int]),
    "composite_instances_dev": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_int]),
    "composite_instances": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_int]),
    "exchange_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_X)]),
    "exchange_unique_id": (C.c_int, [_P]),
    "exchange_create_rank": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_X)]),
    "exchange_destroy": (None, [_X]),
    "exchange_stream": (C.c_void_p, [_X, C.c_int]),
    "exchange_slot_ptrs": (C.c_int, [_X, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    "exchange_layer_ptrs": (C.c_int, [_X, C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    "exchange_render_slot": (C.c_int, [_X, C.c_int, C.c_int, _H, C.c_int, _P, _P]),
    "exchange_gather": (C.c_int, [_X]),
    "exchange_set_collective": (C.c_int, [_X, C.c_int, C.c_int]),
    "exchange_timing": (C.c_int, [_X, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "exchange_composite": (C.c_int, [_X, C.c_int, _H, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_int]),
    "exchange_gather_and_composite": (C.c_int, [_X, C.c_int, _H, _P, _P, _P, _P, _P, C.c_int, C.c_float, C.c_int]),
    "exchange_target_ptrs": (C.c_int, [_X, C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    "exchange_clear_target": (C.c_int, [_X, C.c_int]),
    "exchange_read_target": (C.c_int, [_X, C.c_int, _P, _P]),
    "exchange_sync": (C.c_int, [_X]),
    "get_stats": (C.c_int, [_H, C.POINTER(Stats)]),
    "dump_hash_table": (C.c_int, [_H, _P]),
    "dump_visible_list": (C.c_int, [_H, C.c_int, _P, C.POINTER(C.c_int32)]),
    "dump_visible_types": (C.c_int, [_H, _P]),
    "dump_voxel_blocks": (C.c_int, [_H, C.c_int, C.c_int, _P]),
    "dump_allocation_lists": (C.c_int, [_H, _P, _P]),
    "dump_render_state": (C.c_int, [_H, C.c_int, _P, _P, _P, _P, _P]),
    "dump_swap_state": (C.c_int, [_H, _P, _P]),
    "dump_stored_block": (C.c_int, [_H, C.c_int, _P, C.POINTER(C.c_int)]),
    "selftest_division": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "measure_copy_bandwidth": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "measure_copy_bandwidth_spread": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "profile_enable": (C.c_int, [_H, C.c_int]),
    "profile_reset": (C.c_int, [_H]),
    "profile_get": (C.c_int, [_H, C.POINTER(KernelTime), C.c_int]),
    "mesh_scene": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "mesh_get": (C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_uint64]),
    "mesh_write_obj": (C.c_int, [_H, C.c_char_p]),
    "mesh_free": (C.c_int, [_H]),
    "save_scene_to_mesh": (C.c_int, [_H, C.c_char_p]),
}


def preload_hip_runtime():
    """If PyTorch-ROCm is installed, load ITS copy of libamdhip64.so.7 before libdsr_hip.so is
    dlopen'ed, without importing torch.  Both the wheel and /opt/rocm ship a HIP runtime with
    the same SONAME; whichever is loaded first serves the whole process, and a process that ends
    up with /opt/rocm's runtime plus the wheel's HSA runtime sees no device.  With the wheel's
    copy loaded first, `import torch` before or after the engine both work."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def bind(lib, prefix, allow_missing=False):
    """Return a namespace of typed functions `prefix + name` looked up in `lib`.

    Raises AttributeError when the library does not export a declared symbol (`allow_missing`: measurement tools that load
    an OLDER build of the library for an A/B skip the entry points it lacks).
    """
    ns = SimpleNamespace()
    for name, (res, args) in SIGNATURES.items():
        if allow_missing and not hasattr(lib, prefix + name):
            continue
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = args
        setattr(ns, name, fn)
    ns.lib = lib
    ns.prefix = prefix
    return ns
