"""-m gpu: BASELINE.json's full size (1242x375, 5 mm voxels, 2^23-entry table) — the oracle
only checks the first two frames here (seconds with OpenMP); beyond that the engine state is
checked through size-independent properties of the data structure."""
import os

import numpy as np
import pytest

from dynslam_amd import _capi
from dynslam_amd.engine import EngineCore, default_settings, make_calib
from dynslam_amd.synth import StreetScene

pytestmark = pytest.mark.gpu

W, H = 1242, 375
KW = dict(voxel_size=0.005, mu=0.02, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
          sdf_local_block_num=1 << 21, hash_bucket_num=1 << 22, excess_list_size=1 << 20)


from dynslam_amd.invariants import check_structure, np_hash  # noqa: E402,F401  (moved: bench.py's configs[4] leg uses them too)


def test_full_size_oracle_two_frames_then_properties(hip_api):
    from oracle.oracle import OracleEngine, oracle_settings
    sc = StreetScene(W, H)
    calib = make_calib(*sc.intrinsics(), W, H)
    g = EngineCore(default_settings(**KW), calib)
    o = OracleEngine(oracle_settings(**KW), calib, threads=16)
    for i in range(2):
        rgba, d, T, _ = sc.frame(i)
        for e in (g, o):
            e.update_view(rgba, d)
            e.set_pose_inv_m(T)
            e.process_frame()
            e.prepare()
    assert np.array_equal(g.dump_hash_table(), o.dump_hash_table())
    assert np.array_equal(g.dump_visible_list(), o.dump_visible_list())
    n_used = (1 << 21) - 1 - o.get_stats().last_free_block_id
    assert n_used > 300_000
    ho = o.dump_hash_table()
    used_ptrs = np.sort(ho["ptr"][ho["ptr"] >= 0])
    lo, hi = int(used_ptrs[0]), int(used_ptrs[-1]) + 1
    assert np.array_equal(g.dump_voxel_blocks(lo, hi - lo), o.dump_voxel_blocks(lo, hi - lo))
    rg, ro = g.dump_render_state(), o.dump_render_state()
    for k in ("minmax", "raycast_result", "points", "normals", "raycast_image"):
        assert np.array_equal(rg[k], ro[k]), k
    o.close()
    # continue on the GPU only
    for i in range(2, 8):
        rgba, d, T, _ = sc.frame(i)
        g.update_view(rgba, d)
        g.set_pose_inv_m(T)
        g.process_frame()
        g.prepare()
        g.decay(1, 3, False)
    st, ht, used = check_structure(g, 1 << 21, 1 << 22)
    assert st.sticky_status == 0 and st.decayed_block_count > 0
    # raycast sanity at full size: near rays hit where the analytic scene is (the stereo noise
    # model exceeds mu = 2 cm beyond a few metres, so only the near field is asserted)
    rs = g.dump_render_state()
    z_true = sc.render(7)[0]
    rr = rs["raycast_result"]
    hit = rr[..., 3] > 0
    M = np.linalg.inv(sc.pose(7).astype(np.float64))
    cam_z = ((rr[..., :3].astype(np.float64) * 0.005) @ M[:3, :3].T + M[:3, 3])[..., 2]
    near = np.isfinite(z_true) & (z_true < 8.0)
    assert (hit & near).sum() > 0.5 * near.sum()
    assert np.median(np.abs(cam_z[hit & near] - z_true[hit & near])) < 0.05


def test_full_size_split_calls_equal_process_frame(hip_api):
    """checksum of checksums: ProcessFrame == AllocateSceneFromDepth + IntegrateIntoScene, and
    re-fusing an identical frame allocates nothing on the third pass."""
    sc = StreetScene(W, H)
    calib = make_calib(*sc.intrinsics(), W, H)
    a = EngineCore(default_settings(**KW), calib)
    b = EngineCore(default_settings(**KW), calib)
    for i in range(3):
        rgba, d, T, _ = sc.frame(i)
        for e in (a, b):
            e.update_view(rgba, d)
            e.set_pose_inv_m(T)
        a.process_frame()
        b.allocate_scene_from_depth()
        b.integrate_into_scene()
    assert np.array_equal(a.dump_hash_table(), b.dump_hash_table())
    sa = a.get_stats()
    n = (1 << 21) - 1 - sa.last_free_block_id
    va, vb = a.dump_voxel_blocks(0, 4096), b.dump_voxel_blocks(0, 4096)
    assert np.array_equal(va, vb)
    ha = a.dump_hash_table()
    some = np.sort(ha["ptr"][ha["ptr"] >= 0])[:: max(1, n // 64)][:64]
    for ptr in some.tolist():
        assert np.array_equal(a.dump_voxel_blocks(ptr, 1), b.dump_voxel_blocks(ptr, 1))
    # idempotence of allocation on an unchanged view
    a.allocate_scene_from_depth()
    mid = a.get_stats().last_free_block_id
    a.allocate_scene_from_depth()
    a.allocate_scene_from_depth()
    assert a.get_stats().last_free_block_id <= mid
    last = a.get_stats().last_free_block_id
    a.allocate_scene_from_depth()
    assert a.get_stats().last_free_block_id == last
    check_structure(a, 1 << 21, 1 << 22)


def test_full_size_free_view_and_reset(hip_api):
    sc = StreetScene(W, H)
    calib = make_calib(*sc.intrinsics(), W, H)
    g = EngineCore(default_settings(**KW), calib)
    for i in range(3):
        rgba, d, T, _ = sc.frame(i)
        g.update_view(rgba, d); g.set_pose_inv_m(T); g.process_frame(); g.prepare()
    pose = np.linalg.inv(sc.pose(2).astype(np.float64)).astype(np.float32)
    # the live raycast and a free-view render from the same pose see the same geometry
    _, dep = g.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=pose, want_rgba=False, want_depth=True)
    live = g.dump_render_state()["raycast_result"][..., 3] > 0
    assert ((dep > 0) == live).mean() > 0.995
    col, _ = g.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=pose)
    rgba, _, _, _ = sc.frame(2)
    m = (col[..., 3] == 255)
    assert m.mean() > 0.5
    err = np.abs(col[m][:, :3].astype(int) - rgba[m][:, :3].astype(int))
    assert np.median(err) <= 30  # the fused colour resembles the input image (checker texture + depth noise)
    g.reset_scene()
    st = g.get_stats()
    assert st.last_free_block_id == (1 << 21) - 1 and st.no_visible_blocks == 0
    assert (g.dump_hash_table()["ptr"] == -2).all()


def test_byte_tallies_of_k_integrate_equal_an_independent_count(hip_api, oracle_lib):
    """`roofline.achieved` prices a launch of k_integrate with bytes the kernel tallies ITSELF (lanes that wrote their 24 B of sdf +
    w_depth back, voxels that went through the colour update: dsr_profile_get's store_lanes / colour_voxels).  The oracle counts
    the same two quantities independently while it fuses the same full-size frames (orc_debug_integrate_stats: x rows of a block
    with any updated voxel, voxels that pass the colour gate) — they must agree exactly, and so must the visible blocks."""
    import ctypes as C
    from oracle.oracle import OracleEngine, oracle_settings
    sc = StreetScene(W, H)
    calib = make_calib(*sc.intrinsics(), W, H)
    kw = dict(KW)
    g = EngineCore(default_settings(**kw), calib)
    o = OracleEngine(oracle_settings(**kw), calib, threads=os.cpu_count() or 1)
    stats_fn = oracle_lib.lib.orc_debug_integrate_stats
    stats_fn.restype = C.c_int
    stats_fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    out = (C.c_longlong * 8)()
    stats_fn(1, 1, out)
    g.profile_enable(2)
    g.profile_reset()
    for i in range(2):
        rgba, d, T, _ = sc.frame(i)
        for e in (g, o):
            e.update_view(rgba, d); e.set_pose_inv_m(T); e.process_frame()
    g.sync()
    rec = next(r for r in g.profile_get() if r["name"] == "integrate")
    stats_fn(0, 1, out)
    blocks, upd_voxels, colour_voxels, rows = out[0], out[1], out[2], out[7]
    assert rec["launches"] == 2 and blocks > 300_000
    assert int(rec["units"]) == blocks
    assert int(rec["store_lanes"]) == rows and rows * 8 >= upd_voxels > 0
    assert int(rec["colour_voxels"]) == colour_voxels > 0
    # ... and the layout-true byte model built from them
    P = W * H
    assert rec["bytes_layout"] == blocks * (4 + 16 + 1536) + rows * 24 + colour_voxels * 8 + 2 * 8 * P
    g.close(); o.close()
