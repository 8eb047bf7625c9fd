"""Host logic of the InfiniTamDriver mirror (dynslam_amd/engine.py), run on the oracle
library so that it needs no GPU: same method names / error behaviour as
src/DynSLAM/InfiniTamDriver.{h,cpp}."""
import numpy as np
import pytest

from dynslam_amd.engine import OutOfBlocksError, PreviewType, VoxelDecayParams, make_calib
from dynslam_amd.synth import StreetScene

W, H = 160, 48


def make_driver(decay=None, **kw):
    from oracle.oracle import oracle_driver, oracle_settings
    base = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                sdf_local_block_num=20000, hash_bucket_num=0x8000, excess_list_size=0x2000)
    base.update(kw)
    sc = StreetScene(W, H)
    return sc, oracle_driver(oracle_settings(**base), make_calib(*sc.intrinsics(), W, H), decay)


def step(drv, sc, i):
    rgba, d, T, _ = sc.frame(i)
    drv.UpdateView(rgba, d)
    drv.SetPose(T)
    drv.Integrate()
    drv.PrepareNextStep()


def test_get_image_before_first_frame_is_a_silent_noop(oracle_lib):
    sc, drv = make_driver()
    assert drv.GetImage(PreviewType.kGray) is None          # InfiniTamDriver.cpp:168,185
    assert drv.GetFloatImage(PreviewType.kDepth) is None


def test_preview_type_rules(oracle_lib):
    sc, drv = make_driver()
    step(drv, sc, 0)
    assert drv.GetImage(PreviewType.kDepth) is None          # "Cannot preview depth normally anymore."
    assert drv.GetFloatImage(PreviewType.kGray) is None      # "Can only preview depth as float."
    for t in (PreviewType.kGray, PreviewType.kColor, PreviewType.kNormal, PreviewType.kWeight, PreviewType.kLatestRaycast):
        img = drv.GetImage(t)
        assert img.shape == (H, W, 4) and img.any()
    dep = drv.GetFloatImage(PreviewType.kDepth)
    assert dep.shape == (H, W) and (dep > 0).mean() > 0.3 and dep.max() < 30.0


def test_memory_stats_formulas(oracle_lib):
    sc, drv = make_driver()
    # quirk kept from InfiniTamDriver.h:241-244: N - lastFreeBlockId == 1 when empty
    assert drv.GetUsedMemoryBytes() == 8 * 512 * 1
    step(drv, sc, 0)
    st = drv.core.get_stats()
    assert drv.GetUsedMemoryBytes() == 8 * 512 * (20000 - st.last_free_block_id)
    assert drv.GetVoxelSizeBytes() == 8 and drv.GetSavedDecayMemoryBytes() == 0


def test_pose_and_egomotion(oracle_lib):
    sc, drv = make_driver()
    T0, T1 = sc.pose(0), sc.pose(3)
    drv.SetPose(T0)
    drv.SetPose(T1)
    assert np.allclose(drv.GetPose(), T1, atol=1e-5)
    assert np.allclose(drv.GetLastEgomotion(), np.linalg.inv(T0) @ T1, atol=1e-4)


def test_decay_disabled_is_a_noop_and_enabled_frees(oracle_lib):
    sc, off = make_driver(decay=VoxelDecayParams(False, 1, 1))
    sc, on = make_driver(decay=VoxelDecayParams(True, 1, 1), )
    sc = StreetScene(W, H, noise_px=0.7)
    for i in range(4):
        for drv in (off, on):
            step(drv, sc, i)
            drv.Decay()
    assert off.GetSavedDecayMemoryBytes() == 0 and not off.IsDecayEnabled()
    assert on.GetSavedDecayMemoryBytes() > 0
    before = on.GetSavedDecayMemoryBytes()
    on.Reap(2)
    assert on.GetSavedDecayMemoryBytes() >= before
    off.Reap(2)
    assert off.GetSavedDecayMemoryBytes() == 0
    on.DecayCatchup()


def test_out_of_blocks_raises_like_the_fork(oracle_lib):
    sc, drv = make_driver(sdf_local_block_num=500)
    rgba, d, T, _ = sc.frame(0)
    drv.UpdateView(rgba, d)
    drv.SetPose(T)
    with pytest.raises(OutOfBlocksError):   # InstanceReconstructor.cpp:662-671 catches runtime_error
        drv.Integrate()
    drv.PrepareNextStep()                   # the volume stays usable
    drv.Reset()
    assert drv.core.get_stats().last_free_block_id == 499


def test_instance_view_path(oracle_lib):
    """SetView with an already converted, masked view (InstanceReconstructor.cpp:238-263,580)."""
    sc, drv = make_driver(voxel_size=0.035, mu=1.0, sdf_local_block_num=7142, view_frustum_max=12.0)
    rgba, d, T, _ = sc.frame(0)
    depth_m = np.where(d > 0, d.astype(np.float32) * np.float32(0.001), np.float32(0.0))
    mask = np.zeros((H, W), bool); mask[10:40, 40:100] = True
    inst_rgb = np.where(mask[..., None], rgba, 255).astype(np.uint8)   # dest initialised to 255 / 0
    inst_depth = np.where(mask, depth_m, np.float32(0.0)).astype(np.float32)
    drv.SetView(inst_rgb, inst_depth)
    drv.SetPose(T)
    try:
        drv.Integrate()
    except OutOfBlocksError:
        pass
    drv.PrepareNextStep()
    assert drv.core.get_stats().no_visible_blocks > 0
    got_rgb, got_depth = drv.core.get_view()
    assert np.array_equal(got_rgb, inst_rgb) and np.array_equal(got_depth, inst_depth)
