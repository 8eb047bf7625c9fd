"""-m gpu parity tests proper: HIP engine (through the C ABI) vs the CPU oracle on the same
seeded inputs.  Bar: BIT-EXACT — hash table, free lists, visible lists, every voxel (sdf,
weights, colour) and every raycast / shading output.  (north_star allows 1e-4 on TSDF
values; the engine is built so that 0 is reached: no FMA contraction, correctly rounded
divide/sqrt, defined conversions.)"""
import numpy as np
import pytest

from dynslam_amd import _capi
from tests.common import RENDER_TYPES, assert_render_equal, assert_scene_equal, feed, make_pair

pytestmark = pytest.mark.gpu


def test_sequence_bit_exact(hip_api):
    sc, g, o = make_pair()
    for i in range(6):
        feed((g, o), sc, i)
        assert_scene_equal(g, o, voxels=(i in (0, 5)))
        assert_render_equal(g, o)


def test_allocation_only_first_frame(hip_api):
    sc, g, o = make_pair()
    rgba, d, T, _ = sc.frame(0)
    for e in (g, o):
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.allocate_scene_from_depth()
    assert_scene_equal(g, o)
    assert np.array_equal(g.get_view()[1], o.get_view()[1])


def test_hash_collisions_and_excess_list(hip_api):
    # 256 buckets for thousands of blocks: nearly every allocation goes through the
    # excess list, many same-frame bucket conflicts (last raster writer wins).
    sc, g, o = make_pair(hash_bucket_num=256, excess_list_size=0x8000)
    for i in range(4):
        feed((g, o), sc, i)
        assert_scene_equal(g, o, voxels=False)
    # one allocation per chain tail per frame: 3 frames x 256 chains go through the excess list
    assert o.get_stats().last_free_excess_list_id <= 0x8000 - 1 - 3 * 256
    assert_scene_equal(g, o)
    assert_render_equal(g, o)


def test_out_of_blocks_and_excess_exhaustion(hip_api):
    sc, g, o = make_pair(sdf_local_block_num=3000, hash_bucket_num=1024, excess_list_size=700)
    seen = False
    for i in range(4):
        r = feed((g, o), sc, i, ignore_oob=True)
        assert r[0] == r[1]
        seen |= r[0]
        assert_scene_equal(g, o, voxels=False)
    assert seen, "test must exhaust the block array"
    assert g.get_stats().last_free_block_id == -1
    assert_scene_equal(g, o)
    assert_render_equal(g, o)


def test_instance_volume_params(hip_api):
    # InstanceReconstructor.cpp:372-379: mu = 1.0, voxel 0.035, 7142 blocks
    sc, g, o = make_pair(W=256, H=80, voxel_size=0.035, mu=1.0, sdf_local_block_num=7142,
                         view_frustum_max=12.0)
    for i in range(3):
        feed((g, o), sc, i, ignore_oob=True)
    assert_scene_equal(g, o)
    assert_render_equal(g, o)
    # the fused-preview pair (colour + float depth from one call) of a 7142-block volume: the small-volume paths
    M = np.linalg.inv(sc.pose(1).astype(np.float64)).astype(np.float32)
    gc, gd = g.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
    oc, od = o.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
    assert np.array_equal(gd, od) and np.array_equal(gc, oc) and (gd > 0).any()
    assert np.array_equal(g.dump_visible_list(True), o.dump_visible_list(True))
    assert_render_equal(g, o, freeview=True)
    # ... and straight into HBM buffers of the caller (dsr_get_image_dev: the shading kernel writes them itself)
    import torch
    c_t = torch.zeros((80 * 256, 4), dtype=torch.uint8, device="cuda")
    d_t = torch.zeros((80 * 256,), dtype=torch.float32, device="cuda")
    g.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, M, None, c_t.data_ptr(), d_t.data_ptr())
    g.sync()
    assert np.array_equal(c_t.cpu().numpy().reshape(80, 256, 4), oc) and np.array_equal(d_t.cpu().numpy().reshape(80, 256), od)


def test_depth_weighting_and_stop_at_max_w(hip_api):
    sc, g, o = make_pair(max_w=3, stop_integrating_at_max_w=1)
    for e in (g, o):
        e.set_fusion_weight_params(True)
    for i in range(5):
        feed((g, o), sc, i)
    assert_scene_equal(g, o)


def test_revisit_same_pose_no_new_blocks(hip_api):
    sc, g, o = make_pair(scene_kw=dict(noise_px=0.0))
    feed((g, o), sc, 0)
    before = g.get_stats().last_free_block_id
    feed((g, o), sc, 0)
    feed((g, o), sc, 0)
    assert_scene_equal(g, o)
    # the second pass may allocate same-frame collision losers, the third must not
    mid = g.get_stats().last_free_block_id
    feed((g, o), sc, 0)
    assert g.get_stats().last_free_block_id == mid <= before


@pytest.mark.parametrize("image_type", RENDER_TYPES)
def test_free_view_render_types(hip_api, image_type):
    sc, g, o = make_pair()
    for i in range(4):
        feed((g, o), sc, i)
    pose = np.linalg.inv(sc.pose(2).astype(np.float64)).astype(np.float32)  # world->camera of an older frame
    want_depth = image_type == _capi.IMAGE_FREECAMERA_DEPTH
    ig, dg = g.get_image(image_type, pose_m=pose, want_rgba=True, want_depth=want_depth)
    io, do = o.get_image(image_type, pose_m=pose, want_rgba=True, want_depth=want_depth)
    assert np.array_equal(g.dump_visible_list(True), o.dump_visible_list(True))
    assert_render_equal(g, o, freeview=True)
    assert np.array_equal(ig, io)
    if want_depth:
        assert np.array_equal(dg, do)
        assert (dg > 0).mean() > 0.3
    else:
        assert ig[..., :3].any()


def test_instance_volume_list_path_through_gc_exhaustion_and_reset(hip_api):
    """Round 6: an instance-sized volume keeps its allocated entries as a sorted list (k_small.h list path) and falls back to the
    bit-plane sweeps — rebuilding the list — after a GC pass, with an exhausted block array and after a reset; its tracking render
    is deferred and goes out paired with the preview render into device buffers.  One sequence that crosses every transition, the
    whole state and both render states against the oracle after every frame."""
    import torch
    # 600 blocks: exhausted within the first frames (entries that are visible without owning a block), freed again by the GC
    sc, g, o = make_pair(W=256, H=80, voxel_size=0.035, mu=1.0, sdf_local_block_num=600, view_frustum_max=12.0,
                         scene_kw=dict(noise_px=0.4))
    out_rgba = torch.zeros((80 * 256, 4), dtype=torch.uint8, device="cuda")
    out_depth = torch.zeros((80 * 256,), dtype=torch.float32, device="cuda")
    seen_oob, freed = False, 0
    for i in range(14):
        r = feed((g, o), sc, i, ignore_oob=True)   # update_view, pose, process_frame, prepare (deferred on the GPU side)
        assert r[0] == r[1]
        seen_oob |= r[0]
        M = np.linalg.inv(sc.pose(max(0, i - 1)).astype(np.float64)).astype(np.float32)
        # the preview into device buffers: pairs with the deferred tracking render
        g.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, M, None, out_rgba.data_ptr(), out_depth.data_ptr())
        oc, od = o.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
        g.sync()
        assert np.array_equal(out_depth.cpu().numpy().reshape(80, 256), od)
        assert np.array_equal(out_rgba.cpu().numpy().reshape(80, 256, 4), oc)
        assert_scene_equal(g, o, voxels=False)
        assert_render_equal(g, o)
        assert_render_equal(g, o, freeview=True)
        assert np.array_equal(g.dump_visible_list(True), o.dump_visible_list(True))
        if i in (3, 4, 8):           # GC passes: the list is invalidated, the next frame sweeps and rebuilds it
            for e in (g, o):
                e.decay(2 if i < 8 else 100, 0, i == 8)   # frame 8: Reap with a weight nothing exceeds — every block goes
            assert_scene_equal(g, o, voxels=False)
            freed = max(freed, o.get_stats().decayed_block_count)
        if i == 10:                  # a reset in the middle: everything starts over (the first raycasts are full-frame again)
            for e in (g, o):
                e.reset_scene()
    assert seen_oob, "the sequence must exhaust the block array"
    assert freed > 100, "the GC passes must free blocks (the list is invalidated through them)"
    assert_scene_equal(g, o)
    g.close(); o.close()


def test_scene_raycast_and_original_rgb(hip_api):
    sc, g, o = make_pair()
    feed((g, o), sc, 0)
    for t in (_capi.IMAGE_SCENERAYCAST, _capi.IMAGE_ORIGINAL_RGB):
        assert np.array_equal(g.get_image(t)[0], o.get_image(t)[0])


def test_decay_and_reap(hip_api):
    sc, g, o = make_pair(scene_kw=dict(noise_px=0.6))
    for i in range(8):
        feed((g, o), sc, i)
        for e in (g, o):
            e.decay(1, 3, False)
        assert_scene_equal(g, o, voxels=False)
    assert o.get_stats().decayed_block_count > 0
    assert_scene_equal(g, o)
    # tombstones get re-used by later allocations
    for i in range(8, 11):
        feed((g, o), sc, i)
        for e in (g, o):
            e.decay(2, 0, False)
    assert_scene_equal(g, o)
    for e in (g, o):
        e.decay(3, 0, True)  # Reap
    assert_scene_equal(g, o)
    assert_render_equal(g, o)


def test_decay_fifo_grows_and_drains(hip_api):
    """The GC FIFO is a ring of bit planes (one bit per hash entry per queued frame): raising min_age
    mid-sequence re-allocates the ring with the queued lists in order, lowering it (DecayCatchup:
    min_age 0, InfiniTamDriver.h:210-226) drains one list per call, a reset empties it."""
    sc, g, o = make_pair(scene_kw=dict(noise_px=0.6))
    ages = [2, 2, 2, 6, 6, 6, 6, 6, 1, 0, 0, 0, 0]
    for i, age in enumerate(ages):
        feed((g, o), sc, min(i, 7))
        for e in (g, o):
            e.decay(1, age, False)
        assert_scene_equal(g, o, voxels=False)
    assert o.get_stats().decayed_block_count > 0
    assert_scene_equal(g, o)
    for e in (g, o):
        e.reset_scene()
    for i in range(3):
        feed((g, o), sc, i)
        for e in (g, o):
            e.decay(1, 1, False)
    assert_scene_equal(g, o)


def test_reset_scene(hip_api):
    sc, g, o = make_pair()
    feed((g, o), sc, 0)
    for e in (g, o):
        e.reset_scene()
    assert_scene_equal(g, o)
    st = g.get_stats()
    assert st.last_free_block_id == 40000 - 1 and st.no_visible_blocks == 0
    feed((g, o), sc, 1)
    assert_scene_equal(g, o)


def test_bilateral_filter_view(hip_api):
    # expf differs between libm and the GPU: tolerance 1e-5 relative on the filtered depth
    sc, g, o = make_pair(use_bilateral_filter=1)
    rgba, d, T, _ = sc.frame(0)
    for e in (g, o):
        e.update_view(rgba, d)
    dg, do = g.get_view()[1], o.get_view()[1]
    inner = (slice(2, -2), slice(2, -2))
    assert np.allclose(dg[inner], do[inner], rtol=1e-5, atol=1e-6)


def test_no_view_errors(hip_api):
    from dynslam_amd.engine import DsrError
    sc, g, o = make_pair()
    for e in (g, o):
        with pytest.raises(DsrError) as ei:
            e.process_frame()
        assert ei.value.status == _capi.DSR_E_NO_VIEW


def test_division_selftest(hip_api):
    """The shared-reciprocal division of the integrate path == the device's IEEE divide."""
    import ctypes as C
    bad = C.c_uint64(123)
    assert hip_api.selftest_division(0, 200_000_000, 12345, C.byref(bad)) == 0
    assert bad.value == 0


def test_device_resident_view_equals_host_view(hip_api):
    """dsr_update_view_dev / dsr_set_view_float_dev (the entry bench.py and the multi-GPU path use: inputs
    already in HBM) give the same view and the same scene as the host-buffer entry points — through
    the fused ingest kernel (16-byte aligned inputs) and through the copy fallback (misaligned)."""
    import torch
    sc, g, o = make_pair()
    sc2, g2, _o2 = make_pair()
    _o2.close()
    dev = torch.device("cuda", 0)
    for i in range(3):
        rgba, d, T, _ = sc.frame(i)
        for e in (g, o):
            e.update_view(rgba, d)
        if i == 1:  # misaligned device pointers: the copy fallback
            rb = torch.empty(rgba.size + 16, dtype=torch.uint8, device=dev)
            db = torch.empty(d.size + 8, dtype=torch.int16, device=dev)
            r_dev = rb[4:4 + rgba.size]; r_dev.copy_(torch.from_numpy(rgba.reshape(-1)))
            d_dev = db[1:1 + d.size]; d_dev.copy_(torch.from_numpy(d.reshape(-1)))
            assert r_dev.data_ptr() % 16 != 0 and d_dev.data_ptr() % 16 != 0
        else:
            r_dev, d_dev = torch.from_numpy(rgba).to(dev), torch.from_numpy(d).to(dev)
        torch.cuda.synchronize()
        g2.update_view_dev(r_dev.data_ptr(), d_dev.data_ptr())
        for e in (g, o, g2):
            e.set_pose_inv_m(T)
            e.process_frame()
            e.prepare()
        vr, vd = g.get_view()
        vr2, vd2 = g2.get_view()
        assert np.array_equal(vr, vr2) and np.array_equal(vd.view(np.uint32), vd2.view(np.uint32))
        assert np.array_equal(vr, rgba)
    assert_scene_equal(g, o)
    assert_scene_equal(g2, o)
    # float view from device memory (instance views: InstanceReconstructor.cpp:580)
    rgba, d, T, _ = sc.frame(3)
    depth_m = np.where(d > 0, d.astype(np.float32) * np.float32(0.001), np.float32(-1.0)).astype(np.float32)
    r_dev, f_dev = torch.from_numpy(rgba).to(dev), torch.from_numpy(depth_m).to(dev)
    torch.cuda.synchronize()
    g2.set_view_float_dev(r_dev.data_ptr(), f_dev.data_ptr())
    for e in (g, o):
        e.set_view_float(rgba, depth_m)
    for e in (g, o, g2):
        e.set_pose_inv_m(T)
        e.process_frame()
    assert_scene_equal(g2, o)
    for e in (g, o, g2):
        e.close()


@pytest.mark.parametrize("env", [dict(), dict(DSR_OVERLAP_EXPECTED="1"), dict(DSR_OVERLAP_EXPECTED="1", DSR_PIPELINED_VIEW="1"),
                                 dict(DSR_SMALL_VOLUME="1"), dict(DSR_SMALL_VOLUME="1", DSR_PIPELINED_VIEW="2")])
def test_asynchronous_loop_bit_exact(hip_api, monkeypatch, env):
    """The steady loop UpdateView(dev) -> ProcessFrame -> Prepare with sync_status = 0, as bench.py
    drives it: nothing is read back and the host never waits in between; the final state must be
    the oracle's, several times over (a missing dependency would show as a flaky difference).  With the side stream forced on
    (a small volume does not get one by itself) the range image runs under the integration: the hazard evExpected guards.
    DSR_SMALL_VOLUME: the one-workgroup kernels of instance-sized volumes (k_small.h) on this volume."""
    import torch
    from tests.common import assert_render_equal
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = torch.device("cuda", 0)
    for rep in range(3):
        sc, g, o = make_pair(sync_status=0)
        frames = [sc.frame(i) for i in range(8)]
        r_dev = [torch.from_numpy(f[0]).to(dev) for f in frames]
        d_dev = [torch.from_numpy(f[1]).to(dev) for f in frames]
        torch.cuda.synchronize()
        for i, (rgba, d, T, _) in enumerate(frames):
            g.update_view_dev(r_dev[i].data_ptr(), d_dev[i].data_ptr())
            g.set_pose_inv_m(T)
            g.process_frame()
            g.prepare()
            o.update_view(rgba, d); o.set_pose_inv_m(T); o.process_frame(); o.prepare()
        g.sync()
        assert_scene_equal(g, o)
        assert_render_equal(g, o)
        g.close(); o.close()


@pytest.mark.parametrize("size", [(203, 77), (1226 // 2, 370 // 2)])
def test_odd_image_sizes(hip_api, size):
    """Widths / heights that are not multiples of 4, 8 or 16: partial raycast tiles, partial cells
    of the range image, the scalar tails of the view ingest (host and device entry points)."""
    import torch
    W, H = size
    sc, g, o = make_pair(W=W, H=H)
    dev = torch.device("cuda", 0)
    for i in range(3):
        rgba, d, T, _ = sc.frame(i)
        o.update_view(rgba, d)
        if i == 1:
            r_dev, d_dev = torch.from_numpy(rgba).to(dev), torch.from_numpy(d).to(dev)
            torch.cuda.synchronize()
            g.update_view_dev(r_dev.data_ptr(), d_dev.data_ptr())
        else:
            g.update_view(rgba, d)
        for e in (g, o):
            e.set_pose_inv_m(T); e.process_frame(); e.prepare()
        vr, vd = g.get_view(); wr, wd = o.get_view()
        assert np.array_equal(vr, wr) and np.array_equal(vd.view(np.uint32), wd.view(np.uint32))
    assert_scene_equal(g, o)
    assert_render_equal(g, o)
    for t in RENDER_TYPES[:2] + RENDER_TYPES[-1:]:
        want_d = t == _capi.IMAGE_FREECAMERA_DEPTH
        a = g.get_image(t, want_rgba=not want_d, want_depth=want_d)
        b = o.get_image(t, want_rgba=not want_d, want_depth=want_d)
        assert np.array_equal(a[1] if want_d else a[0], b[1] if want_d else b[0])
    assert np.array_equal(g.mesh_scene().view(np.uint32), o.mesh_scene().view(np.uint32))
    g.close(); o.close()


def test_free_view_cache(hip_api):
    """Several image types from ONE pose (what a DynSLAM redraw asks for) reuse the free-view visible
    list, range image and raycast; a change of the scene (fusion, voxel GC, reset), of the pose or of
    the intrinsics must invalidate them.  Every image is compared with the oracle, which always
    recomputes."""
    sc, g, o = make_pair()
    for i in range(3):
        feed((g, o), sc, i)

    def same_images(pose, intr=None):
        for t in RENDER_TYPES + RENDER_TYPES[:2]:
            want_depth = t == _capi.IMAGE_FREECAMERA_DEPTH
            a = g.get_image(t, pose_m=pose, intrinsics=intr, want_rgba=True, want_depth=want_depth)
            b = o.get_image(t, pose_m=pose, intrinsics=intr, want_rgba=True, want_depth=want_depth)
            assert np.array_equal(a[0], b[0]), t
            if want_depth:
                assert np.array_equal(a[1], b[1])
        assert_render_equal(g, o, freeview=True)

    pose_a = np.linalg.inv(sc.pose(1).astype(np.float64)).astype(np.float32)
    pose_b = np.linalg.inv(sc.pose(2).astype(np.float64)).astype(np.float32)
    same_images(pose_a)
    same_images(pose_b)                                   # new pose
    fx, fy, cx, cy = sc.intrinsics()
    same_images(pose_b, np.array([fx * 0.8, fy * 0.8, cx, cy], np.float32))  # new intrinsics
    feed((g, o), sc, 3)                                   # the scene changed
    same_images(pose_b)
    for e in (g, o):
        e.decay(2, 0, True)                               # voxel GC changed it again
    same_images(pose_b)
    for e in (g, o):
        e.reset_scene()
    feed((g, o), sc, 4)
    same_images(pose_b)
    g.close(); o.close()


@pytest.mark.parametrize("env", [dict(DSR_GRID_INTEGRATE="1"), dict(DSR_GRID_INTEGRATE="37", DSR_GRID_EXPECTED="1", DSR_GRID_DECAY="3"),
                                 dict(DSR_GRID_INTEGRATE="16384", DSR_GRID_EXPECTED="257", DSR_GRID_DECAY="32768"),
                                 # the paths of instance-sized volumes (k_small.h: commit + visible list + range image in ONE workgroup,
                                 # the free-view list from the allocated bits, the raycast that shades its own pixels) forced onto this
                                 # 40000-block volume, and explicitly off; the large-volume paths are what the other cases run
                                 dict(DSR_SMALL_VOLUME="1"), dict(DSR_SMALL_VOLUME="1", DSR_GRID_INTEGRATE="5"), dict(DSR_SMALL_VOLUME="0"),
                                 # the side stream forced onto this small volume: range image under the integration; view stream forms
                                 dict(DSR_OVERLAP_EXPECTED="1"), dict(DSR_OVERLAP_EXPECTED="1", DSR_SMALL_VOLUME="1"),
                                 dict(DSR_PIPELINED_VIEW="0"), dict(DSR_PIPELINED_VIEW="1", DSR_SMALL_VOLUME="1"),
                                 dict(DSR_GRID_INTEGRATE="3")])
def test_results_do_not_depend_on_the_launch_geometry(hip_api, monkeypatch, env):
    """The tuning knobs an engine reads from the environment at creation (grid sizes of k_integrate, of the range-image
    kernel and of the GC kernel: tools/bench_variants.py sweeps them) change how the work is split over waves — the colour
    list of k_integrate is appended in a different order, the range image is folded by other workgroups — never a bit of
    the result."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc, g, o = make_pair()
    for i in range(5):
        feed((g, o), sc, i)
        for e in (g, o):
            e.decay(1, 2, False)
        assert_scene_equal(g, o, voxels=(i in (0, 4)))
        assert_render_equal(g, o)
    # a free-view render goes through the free-view visible list, the range image, the raycast and the shading
    M = np.linalg.inv(sc.pose(2).astype(np.float64)).astype(np.float32)
    gc, gd = g.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
    oc, od = o.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
    assert np.array_equal(gd, od) and np.array_equal(gc, oc)
    assert np.array_equal(g.dump_visible_list(True), o.dump_visible_list(True))
    assert_render_equal(g, o, freeview=True)


def test_non_finite_depth_in_a_float_view(hip_api):
    """+inf / huge / NaN / -inf depths handed over through SetView (dsr_set_view_float) fuse exactly as in the
    reference's arithmetic: +inf and huge depths update the voxels along the pixel's ray with sdf = +1 and never take
    colour, -inf is an invalid pixel, NaN propagates the same way on both sides."""
    sc, g, o = make_pair()
    feed((g, o), sc, 0)
    rgba, d, T, _ = sc.frame(1)
    depth = np.where((d <= 0) | (d > 32000), -1.0, d.astype(np.float32) * np.float32(0.001)).astype(np.float32)
    rng = np.random.default_rng(5)
    special = np.array([np.inf, 1e35, 3e38, -np.inf, np.nan, 1e30, 2e30], np.float32)
    ys, xs = rng.integers(0, depth.shape[0], 600), rng.integers(0, depth.shape[1], 600)
    depth[ys, xs] = special[rng.integers(0, len(special), 600)]
    depth[40:44, 100:140] = np.inf  # a patch: whole blocks see nothing but +inf
    for e in (g, o):
        e.set_view_float(rgba, depth)
        e.set_pose_inv_m(T)
        e.process_frame()
        e.prepare()
    assert_scene_equal(g, o)
    assert_render_equal(g, o)
    # the stored view differs only where documented: values above 1e30 read back as 1e30 (include/dsr.h)
    gv, ov = g.get_view()[1], o.get_view()[1]
    big = ov > 1e30
    assert big.any() and np.all(gv[big] == np.float32(1e30))
    assert np.array_equal(gv[~big], ov[~big], equal_nan=True)


def test_pose_or_list_change_between_fusion_and_prepare(hip_api):
    """The live view's range image is computed speculatively under the integration (side stream, dsr_engine.hip); Prepare must
    not take it when the camera moved or the visible list changed (voxel GC) between Integrate and PrepareNextStep."""
    sc, g, o = make_pair()
    for i in range(3):
        feed((g, o), sc, i)
    rgba, d, T, _ = sc.frame(3)
    T2 = sc.pose(5)
    for e in (g, o):
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.process_frame()
        e.set_pose_inv_m(T2)   # the tracker refined the pose after fusion
        e.prepare()
    assert_render_equal(g, o)
    rgba, d, T, _ = sc.frame(4)
    for e in (g, o):
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.process_frame()
        e.decay(3, 0, True)    # GC rewrote the visible list
        e.prepare()
    assert_scene_equal(g, o)
    assert_render_equal(g, o)
    # ... and the plain sequence afterwards still takes the speculative image and matches
    feed((g, o), sc, 5)
    assert_render_equal(g, o)


@pytest.mark.parametrize("size", [(320, 96), (203, 77)])
def test_update_view_bgr_equals_the_two_step_form(hip_api, size):
    """dsr_update_view_bgr (InfiniTamDriver::UpdateView as a whole: CvToItm's BGR -> RGBA loop inside the ingest kernel) gives the
    view and the scene of CvToItm on the host followed by dsr_update_view — also for pixel counts that are not multiples of 4."""
    W, H = size
    sc, g, o = make_pair(W=W, H=H)
    sc2, g2, o2 = make_pair(W=W, H=H)
    o2.close()
    for i in range(3):
        rgba, d, T, _ = sc.frame(i)
        rgba = rgba.copy(); rgba[..., 3] = 255  # CvToItm sets alpha to 255 (InfiniTamDriver.cpp:94)
        bgr = np.ascontiguousarray(rgba[..., 2::-1])
        g.update_view(rgba, d); o.update_view(rgba, d)
        g2.update_view_bgr(bgr, d)
        for e in (g, o, g2):
            e.set_pose_inv_m(T); e.process_frame(); e.prepare()
        va, vb, vo = g.get_view(), g2.get_view(), o.get_view()
        assert np.array_equal(va[0], vo[0]) and np.array_equal(vb[0], vo[0])
        assert np.array_equal(va[1].view(np.uint32), vo[1].view(np.uint32)) and np.array_equal(vb[1].view(np.uint32), vo[1].view(np.uint32))
    assert_scene_equal(g, o)
    assert_scene_equal(g2, o)
    for e in (g, o, g2):
        e.close()


@pytest.mark.parametrize("pipelined_view", ["0", "1", "2"])
def test_host_buffer_frames_pipelined_without_waiting(hip_api, monkeypatch, pipelined_view):
    """Host-buffer frames (dsr_update_view: pinned double-buffered staging, upload on the I/O stream, landing buffer, ingest on the
    engine's stream) handed over back to back with sync_status = 0 — the host never waits, overwrites its own buffers right after
    every call, and asks for previews and the view in between (I/O-stream readers of the view) — the final state is the
    oracle's, several times over."""
    from tests.common import assert_render_equal
    import ctypes as C
    monkeypatch.setenv("DSR_PIPELINED_VIEW", pipelined_view)  # "1": view operations on the view stream, the view double buffered
    for rep in range(3):
        sc, g, o = make_pair(sync_status=0)
        frames = [sc.frame(i) for i in range(9)]
        W, H = g.W, g.H
        scratch_c = np.empty((H, W, 4), np.uint8); scratch_d = np.empty((H, W), np.int16)
        for i, (rgba, d, T, _) in enumerate(frames):
            scratch_c[...] = rgba; scratch_d[...] = d
            g.update_view(scratch_c, scratch_d)
            scratch_c[...] = 0; scratch_d[...] = 0  # the caller's buffers are free as soon as the call returns
            g.set_pose_inv_m(T); g.process_frame(); g.prepare()
            o.update_view(rgba, d); o.set_pose_inv_m(T); o.process_frame(); o.prepare()
            if i % 3 == 1:  # previews / read-backs of the view while the fusion of this frame is still in flight
                bgr = np.zeros((H, W, 3), np.uint8); mm = np.zeros((H, W), np.int16)
                assert hip_api.get_view_previews(g._h, bgr.ctypes.data_as(C.c_void_p), mm.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(bgr, rgba[..., 2::-1])
                vr, vd = g.get_view()
                assert np.array_equal(vr, rgba) and np.array_equal(vd.view(np.uint32), o.get_view()[1].view(np.uint32))
        g.sync()
        assert_scene_equal(g, o)
        assert_render_equal(g, o)
        g.close(); o.close()


def test_published_status_and_visible_count(hip_api, monkeypatch):
    """The status word dsr_process_frame returns and noVisibleBlocks are PUBLISHED by the allocation's last kernel into pinned
    host memory (the host does not wait for the integration): same values as the read-back path (DSR_NO_PUBLISHED_STATUS), an
    exhausted block array is still reported by the frame that caused it, and the next frame starts clean."""
    import ctypes as C
    from dynslam_amd.engine import OutOfBlocksError
    n = C.c_int32(-1)
    results = []
    for published in (True, False):
        if not published:
            monkeypatch.setenv("DSR_NO_PUBLISHED_STATUS", "1")
        sc, g, o = make_pair(sdf_local_block_num=1500)
        seen = []
        for i in range(4):
            rgba, d, T, _ = sc.frame(i)
            flags = []
            for e in (g, o):
                e.update_view(rgba, d); e.set_pose_inv_m(T)
                try:
                    e.process_frame(); flags.append(False)
                except OutOfBlocksError:
                    flags.append(True)
            assert flags[0] == flags[1]
            assert hip_api.get_no_visible_blocks(g._h, C.byref(n)) == 0 and n.value == o.get_stats().no_visible_blocks
            seen.append((flags[0], n.value))
        assert any(f for f, _ in seen)
        assert_scene_equal(g, o)
        results.append(seen)
        g.close(); o.close()
    assert results[0] == results[1]


def test_mask_staging_ring_reuse(hip_api):
    """the pinned, device-mapped mask ring (32 slots) reused more than twice over with a DIFFERENT mask every time and
    no synchronisation in between — every kernel must have read its own mask (coherent mapping, slot events)."""
    sc, g, o = make_pair()
    rgba, d, T, _ = sc.frame(0)
    for e in (g, o):
        e.update_view(rgba, d)
    rng = np.random.default_rng(5)
    W, H = g.W, g.H
    for i in range(81):
        bw, bh = int(rng.integers(8, 40)), int(rng.integers(6, 24))
        x0, y0 = int(rng.integers(-10, W - 10)), int(rng.integers(-5, H - 5))
        mask = (rng.random((bh, bw)) < 0.35).astype(np.uint8)
        for e in (g, o):
            e.remove_silhouette(mask, x0, y0)
    vg, vo = g.get_view(), o.get_view()
    assert np.array_equal(vg[0], vo[0]) and np.array_equal(vg[1].view(np.uint32), vo[1].view(np.uint32))
    assert (vo[1] == 0).mean() > 0.2
    g.close(); o.close()
