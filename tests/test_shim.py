"""The C++ drop-in boundary: shim/ITMLib.h (ITMLib names over the C ABI) + shim/example_host.cpp
(a host written like InfiniTamDriver).  CPU: it must compile and link against libdsr_hip.so.
GPU: running it must reproduce what the same call sequence gives through the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim")
EXE = os.path.join(SHIM, "example_host")


def build_example():
    lib_dir = os.path.join(ROOT, "dynslam_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", SHIM, os.path.join(SHIM, "example_host.cpp"), "-o", EXE,
           "-L", lib_dir, "-ldsr_hip", f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_shim_host_compiles_and_links():
    exe = build_example()
    assert os.path.exists(exe)
    # every undefined dsr_* symbol of the host resolves in libdsr_hip.so
    out = subprocess.check_output(["nm", "-u", exe]).decode()
    used = sorted({l.split()[-1] for l in out.splitlines() if " dsr_" in l or l.strip().startswith("U dsr_")})
    assert "dsr_process_frame" in used and "dsr_get_image" in used and "dsr_decay" in used
    assert "dsr_mesh_scene" in used and "dsr_mesh_write_obj" in used


def fnv(data, h=1469598103934665603):
    for b in bytes(data):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_shim_host_matches_python_mirror(hip_api, tmp_path):
    from dynslam_amd import _capi
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    W, H, frames = 96, 64, 3
    exe = build_example()
    obj = tmp_path / "shim.obj"
    out = subprocess.check_output([exe, str(W), str(H), str(frames), str(obj)]).decode().strip()
    got = dict(kv.split("=") for kv in out.split())
    e = EngineCore(default_settings(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                                    sdf_local_block_num=20000, hash_bucket_num=0x8000, excess_list_size=0x2000),
                   make_calib(150.0, 150.0, W / 2.0 - 0.5, H / 2.0 - 0.5, W, H))
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    for f in range(frames):
        d = 1800 + 3 * xs + 2 * ys + ((xs // 40) & 1) * 150 - 20 * f
        d = np.where(xs % 53 == 0, 0, d).astype(np.int16)
        rgba = np.stack([(xs * 255 // W), (ys * 255 // H), (xs + ys + 13 * f) & 255, np.full_like(xs, 255)], -1).astype(np.uint8)
        T = np.eye(4, dtype=np.float32)
        T[0, 3] = np.float32(0.02) * np.float32(f)
        T[2, 3] = np.float32(0.05) * np.float32(f)
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.process_frame()
        e.prepare()
        e.decay(1, 1, False)
    col, _ = e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME)
    _, dep = e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, want_rgba=False, want_depth=True)
    vdepth = e.get_view()[1]
    h = fnv(vdepth.tobytes(), fnv(dep.tobytes(), fnv(col.tobytes())))
    import ctypes as C
    mm = np.empty(W * H, np.int16); bgr = np.empty((W * H, 3), np.uint8)
    assert e.api.depth_m_to_mm(dep.ctypes.data_as(C.c_void_p), mm.ctypes.data_as(C.c_void_p), W * H) == 0
    assert e.api.rgba_to_bgr(col.ctypes.data_as(C.c_void_p), bgr.ctypes.data_as(C.c_void_p), W * H) == 0
    assert np.array_equal(bgr.reshape(H, W, 3), col[..., 2::-1])
    h = fnv(bgr.tobytes(), fnv(mm.tobytes(), h))
    st = e.get_stats()
    assert int(got["visible"]) == st.no_visible_blocks
    assert int(got["used_bytes"]) == 8 * 512 * (st.num_allocated_voxel_blocks - st.last_free_block_id)
    assert int(got["saved_bytes"]) == st.decayed_block_count * 4096
    assert got["hash"] == f"{h:016x}"
    # ITMMeshingEngine::MeshScene + ITMMesh::WriteOBJ through the shim == the same through the mirror
    ref = tmp_path / "mirror.obj"
    tris = e.mesh_scene()
    e.mesh_write_obj(ref)
    assert int(got["triangles"]) == len(tris) > 0
    assert obj.read_bytes() == ref.read_bytes()
