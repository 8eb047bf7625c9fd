"""Device functions of the HIP kernels, run on the CPU against the oracle.

The per-pixel / per-block functions of the allocation, visibility, range-image, raycast and shading kernels (alloc_ray,
check_block_visibility, project_single_block, cast_ray, icp_pixel, render_pixel in dynslam_amd/csrc/k_alloc.h / k_raycast.h) are
templates over a small `Ops` policy (float -> int conversion, floor / ceil / sqrt, division by a tame divisor, "any ray of the
wave"), so the very functions the kernels run can be compiled for the host with a one-ray Ops
(tests/hostsim/device_functions_host.hip) and driven over the oracle's table, voxels, views and range images: they must
reproduce the oracle bit for bit.  Here, without a GPU; the same comparisons run on the device in the -m gpu suite.
(It is also how variants of the ray march were verified before any GPU time was spent on them: profiles/r03_raycast_*_variant*.)
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.common import SMALL

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "device_functions_host.hip")
LIB = os.path.join(HERE, "hostsim", "_build", "libdevice_functions_host.so")
CSRC = os.path.join(os.path.dirname(HERE), "dynslam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _lib():
    deps = [SRC] + [os.path.join(CSRC, h) for h in ("k_raycast.h", "k_alloc.h", "k_edges.h", "k_composite.h", "dsr_device.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
            pytest.skip("hipcc not available to build the host stand-in")
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fno-fast-math", "-Wno-unused-function", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.rr_cast_all.restype = C.c_int
    return lib


def _oracle_scene(n_frames, **kw):
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    W, H = 320, 96
    settings = dict(SMALL)
    settings.update(kw)
    sc = StreetScene(W, H)
    o = OracleEngine(oracle_settings(**settings), make_calib(*sc.intrinsics(), W, H))
    for i in range(n_frames):
        rgba, d, T, _ = sc.frame(i)
        o.update_view(rgba, d)
        o.set_pose_inv_m(T)
        o.process_frame()
        o.prepare()
    return sc, o, settings


def _cast(lib, sc, o, settings):
    rs = o.dump_render_state()
    table = o.dump_hash_table()
    vox = o.dump_voxel_blocks()
    vba = np.zeros((o.no_blocks, 4096), np.uint8)  # the library's block layout: the sdf plane comes first
    vba[:, :1024] = np.ascontiguousarray(vox["sdf"]).view(np.uint8).reshape(o.no_blocks, 1024)
    _, inv_m = o.get_pose()
    inv_m = np.ascontiguousarray(inv_m.T.astype(np.float32)).ravel()  # column-major, as the engine holds it
    proj = np.array(sc.intrinsics(), np.float32)
    out = np.zeros((o.H, o.W, 4), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.rr_cast_all(p(inv_m), p(proj), C.c_float(settings["voxel_size"]), C.c_float(settings["mu"]), o.W, o.H,
                         settings["hash_bucket_num"], o.no_total_entries, p(table), p(vba), p(np.ascontiguousarray(rs["minmax"])), p(out))
    assert rc == 0
    return out, rs["raycast_result"]


def _assert_same(got, want):
    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        bad = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1))
        raise AssertionError(f"{len(bad)} rays differ, first {bad[0]}: {got[tuple(bad[0])]} vs {want[tuple(bad[0])]}")


def test_march_equals_oracle():
    lib = _lib()
    sc, o, settings = _oracle_scene(4)
    got, want = _cast(lib, sc, o, settings)
    assert (want[..., 3] > 0).sum() > 0.3 * want[..., 3].size, "the scene must be hit by a good part of the rays"
    _assert_same(got, want)
    o.close()


def test_march_with_long_chains():
    """256 buckets for thousands of blocks: nearly every lookup walks the excess list."""
    lib = _lib()
    sc, o, settings = _oracle_scene(3, hash_bucket_num=256, excess_list_size=0x8000)
    _assert_same(*_cast(lib, sc, o, settings))
    o.close()


def test_march_with_fine_voxels():
    """2 cm voxels: long runs of absent blocks in front of the surface, most trilinear cells straddle blocks."""
    lib = _lib()
    sc, o, settings = _oracle_scene(2, voxel_size=0.02, mu=0.08, sdf_local_block_num=150000, hash_bucket_num=0x40000)
    _assert_same(*_cast(lib, sc, o, settings))
    o.close()


def _full_vba(o):
    """The oracle's voxels (array of structs) in the library's plane-wise block layout (dsr_device.h): sdf, w_depth, colour + w_color."""
    vox = o.dump_voxel_blocks()
    vba = np.zeros((o.no_blocks, 4096), np.uint8)
    vba[:, :1024] = np.ascontiguousarray(vox["sdf"]).view(np.uint8).reshape(o.no_blocks, 1024)
    vba[:, 1024:1536] = vox["w_depth"]
    clr = np.zeros((o.no_blocks, 512, 4), np.uint8)
    clr[..., :3] = vox["clr"]
    clr[..., 3] = vox["w_color"]
    vba[:, 2048:] = clr.reshape(o.no_blocks, 2048)
    return vba


def test_icp_maps_equal_oracle():
    """k_icp_maps' per-pixel function: points, normals and the grey image of the tracking view from its raycast result."""
    lib = _lib()
    sc, o, settings = _oracle_scene(4)
    rs = o.dump_render_state()
    _, inv_m = o.get_pose()
    inv_m = np.ascontiguousarray(inv_m.T.astype(np.float32)).ravel()
    pts, nrm = np.zeros((o.H, o.W, 4), np.float32), np.zeros((o.H, o.W, 4), np.float32)
    grey = np.zeros((o.H, o.W, 4), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.rr_icp_all(p(inv_m), C.c_float(settings["voxel_size"]), o.W, o.H, p(np.ascontiguousarray(rs["raycast_result"])),
                          p(pts), p(nrm), p(grey)) == 0
    assert (rs["points"][..., 3] > 0).sum() > 0.2 * o.W * o.H
    _assert_same(pts, rs["points"])
    _assert_same(nrm, rs["normals"])
    assert np.array_equal(grey, rs["raycast_image"])
    o.close()


@pytest.mark.parametrize("image_type", ["SHADED", "COLOUR_FROM_VOLUME", "COLOUR_FROM_NORMAL", "COLOUR_FROM_DEPTH_WEIGHT", "DEPTH"])
def test_free_view_shading_equals_oracle(image_type):
    """k_render's per-pixel function on the oracle's free-view raycast: every image type + the float depth."""
    from dynslam_amd import _capi
    lib = _lib()
    sc, o, settings = _oracle_scene(4)
    t = getattr(_capi, "IMAGE_FREECAMERA_" + image_type)
    M = np.linalg.inv(sc.pose(2).astype(np.float64)).astype(np.float32)
    want_rgba, want_depth = o.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
    rs = o.dump_render_state(True)
    table, vba = o.dump_hash_table(), _full_vba(o)
    o.set_pose_m(M)  # (done with the live view) -> get_pose() hands back the engine's own cofactor inverse of M, the one it shaded with
    _, inv = o.get_pose()
    rgba = np.zeros((o.H, o.W, 4), np.uint8)
    depth = np.zeros((o.H, o.W), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    Mc = np.ascontiguousarray(M.T).ravel()
    invc = np.ascontiguousarray(inv.T.astype(np.float32)).ravel()
    assert lib.rr_render_all(int(t), p(Mc), p(invc), C.c_float(settings["voxel_size"]), settings["max_w"], o.W, o.H,
                             settings["hash_bucket_num"], o.no_total_entries, p(table), p(vba),
                             p(np.ascontiguousarray(rs["raycast_result"])), p(rgba), p(depth)) == 0
    assert (rs["raycast_result"][..., 3] > 0).sum() > 0.2 * o.W * o.H
    assert np.array_equal(depth.view(np.uint32), want_depth.view(np.uint32))
    if image_type != "DEPTH":
        assert np.array_equal(rgba, want_rgba), f"{(rgba != want_rgba).any(axis=-1).sum()} pixels differ"
    o.close()


@pytest.mark.parametrize("kw", [{}, dict(voxel_size=0.02, mu=0.08, sdf_local_block_num=150000, hash_bucket_num=0x40000)])
def test_range_image_equals_oracle(kw):
    """K6: project_single_block (the function all three range-image kernels call) folded over the visible list on the CPU
    equals the oracle's range image — for the tracking view and for a free view."""
    from dynslam_amd import _capi
    lib = _lib()
    sc, o, settings = _oracle_scene(3, **kw)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    proj = np.array(sc.intrinsics(), np.float32)
    mw, mh = (o.W + 7) // 8, (o.H + 7) // 8

    def check(M, freeview):
        table = o.dump_hash_table()
        ids = o.dump_visible_list(freeview)
        ids = ids[table["ptr"][ids] >= 0]
        pos = np.ascontiguousarray(table["pos"][ids].astype(np.int16))
        got = np.zeros((mh, mw, 2), np.float32)
        Mc = np.ascontiguousarray(M.T.astype(np.float32)).ravel()
        assert lib.rr_range_image(p(Mc), p(proj), C.c_float(settings["voxel_size"]), o.W, o.H, p(pos), len(ids), p(got)) == 0
        want = o.dump_render_state(freeview)["minmax"]
        assert (want[..., 0] < want[..., 1]).sum() > 0.2 * mw * mh
        _assert_same(got, want)

    M_live, _ = o.get_pose()
    check(M_live, False)
    M = np.linalg.inv(sc.pose(1).astype(np.float64)).astype(np.float32)
    o.get_image(_capi.IMAGE_FREECAMERA_SHADED, pose_m=M)
    check(M, True)
    o.close()


def test_free_view_visible_list_equals_oracle():
    """K5: check_block_visibility (also the previous-frame re-test of K0b) over the whole table on the CPU == the oracle's
    free-view visible list, for a camera that sees only part of the map."""
    from dynslam_amd import _capi
    lib = _lib()
    sc, o, settings = _oracle_scene(5)
    T = sc.pose(1).astype(np.float64)
    yaw = np.eye(4)
    a = np.deg2rad(25.0)
    yaw[0, 0], yaw[0, 2], yaw[2, 0], yaw[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    M = np.linalg.inv(T @ yaw).astype(np.float32)
    o.get_image(_capi.IMAGE_FREECAMERA_SHADED, pose_m=M)
    want = o.dump_visible_list(True)
    table = o.dump_hash_table()
    ids = np.zeros(o.no_total_entries, np.int32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    Mc = np.ascontiguousarray(M.T).ravel()
    proj = np.array(sc.intrinsics(), np.float32)
    n = lib.rr_freeview_visible(p(Mc), p(proj), C.c_float(settings["voxel_size"]), o.W, o.H, p(table), o.no_total_entries, p(ids))
    allocated = int((table["ptr"] >= 0).sum())
    assert 100 < len(want) < allocated, (len(want), allocated)  # part of the map, not all of it
    assert n == len(want) and np.array_equal(ids[:n], want)
    o.close()


def test_allocation_ray_walk_equals_oracle():
    """K1: the blocks the depth rays of a frame ask for (alloc_ray + the kernel's running additions, on the CPU) are exactly
    the blocks the oracle ends up allocating for that view from an empty scene."""
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    lib = _lib()
    W, H = 320, 96
    sc = StreetScene(W, H)
    o = OracleEngine(oracle_settings(**SMALL), make_calib(*sc.intrinsics(), W, H))
    rgba, d, T, _ = sc.frame(0)
    o.update_view(rgba, d)
    o.set_pose_inv_m(T)
    # one entry per bucket / chain tail and frame (the last writer wins): blocks that collide wait for the next frame.  Repeating
    # the allocation from the same view converges to everything the rays ask for.
    counts = []
    for _ in range(12):
        o.allocate_scene_from_depth()
        table = o.dump_hash_table()
        counts.append(int((table["ptr"] >= 0).sum()))
        if len(counts) > 1 and counts[-1] == counts[-2]:
            break
    assert counts[0] < counts[-1] and counts[-1] == counts[-2], counts
    want = np.unique(table["pos"][table["ptr"] >= 0].astype(np.int16), axis=0)
    depth = np.ascontiguousarray(o.get_view()[1].astype(np.float32))
    _, inv_m = o.get_pose()
    inv_c = np.ascontiguousarray(inv_m.T.astype(np.float32)).ravel()
    proj = np.array(sc.intrinsics(), np.float32)
    out = np.zeros((SMALL["sdf_local_block_num"], 3), np.int16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.rr_alloc_blocks(p(inv_c), p(proj), C.c_float(SMALL["voxel_size"]), C.c_float(SMALL["mu"]), C.c_float(SMALL["view_frustum_min"]),
                            C.c_float(SMALL["view_frustum_max"]), W, H, p(depth), p(out), len(out))
    assert 1000 < n <= len(out)
    got = out[:n]  # sorted lexicographically by the harness, like np.unique's rows
    assert len(want) == n and np.array_equal(got, want)
    o.close()

