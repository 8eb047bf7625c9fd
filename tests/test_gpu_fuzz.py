"""Randomised differential test: one seeded sequence of engine calls — frames from poses that jump, voxel GC passes, resets,
free-view renders from perturbed cameras, with or without a tracking render in between — on the HIP engine and on the CPU oracle,
the complete engine state compared after every call.  Settings are drawn per seed so that the sequences cross the corners the
hand-written parity tests reach one at a time: a table of a few hundred buckets (long excess chains, the excess list running out),
a block array that runs out, instance-sized volumes (the one-workgroup kernels with their sorted list, its merge and its
fall-back), map-sized ones, host swapping, odd image sizes.

The suite runs a few seeds; `DSR_FUZZ_SEEDS=a:b` runs seeds a..b-1 (a soak on the GPU box: `profiles/r06x_fuzz_*.log`)."""
import os

import numpy as np
import pytest

from dynslam_amd import _capi
from tests.common import RENDER_TYPES, assert_render_equal, assert_scene_equal, make_pair


def _seeds():
    spec = os.environ.get("DSR_FUZZ_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    # one of every kind of settings: map-sized, instance-sized (300 / 7142 blocks), tiny tables, swapping; 188, 398: the sequences
    # that found round 6's list path skipping entries which are visible without owning a block (k_small.h `ghost`); 835: a frame
    # without visible blocks after frames without a Prepare
    return [1, 4, 5, 12, 13, 24, 188, 398, 835]


def _draw_settings(rng):
    W, H = [(256, 80), (320, 96), (251, 83), (192, 64)][rng.integers(4)]
    kind = rng.integers(4)
    if kind == 0:      # instance-sized volume (k_small.h), upstream's table
        kw = dict(voxel_size=0.035, mu=1.0, sdf_local_block_num=int(rng.choice([300, 2000, 7142])), hash_bucket_num=0x100000,
                  excess_list_size=0x20000, view_frustum_max=float(rng.choice([8.0, 12.0, 30.0])))
    elif kind == 1:    # instance-sized volume behind a tiny table: chains, the excess list runs out
        kw = dict(voxel_size=0.05, mu=float(rng.choice([0.2, 0.4])), sdf_local_block_num=int(rng.choice([1500, 9000, 16000])),
                  hash_bucket_num=int(rng.choice([0x100, 0x400, 0x1000])), excess_list_size=int(rng.choice([0x40, 0x400, 0x4000])))
    elif kind == 2:    # map-sized volume (the multi-workgroup kernels)
        kw = dict(voxel_size=float(rng.choice([0.05, 0.08])), mu=float(rng.choice([0.2, 0.32])), sdf_local_block_num=int(rng.choice([17000, 40000])),
                  hash_bucket_num=int(rng.choice([0x2000, 0x10000])), excess_list_size=int(rng.choice([0x200, 0x4000])))
    else:              # host swapping
        kw = dict(voxel_size=0.05, mu=0.2, sdf_local_block_num=int(rng.choice([12000, 40000])), hash_bucket_num=0x10000,
                  excess_list_size=0x4000, use_swapping=1)
    kw["max_w"] = int(rng.choice([3, 100]))
    return W, H, kw, kind


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _seeds())
def test_random_call_sequences_equal_the_oracle(hip_api, seed):
    from dynslam_amd.engine import OutOfBlocksError
    rng = np.random.default_rng(1000 + seed)
    W, H, kw, kind = _draw_settings(rng)
    sc, g, o = make_pair(W=W, H=H, scene_kw=dict(noise_px=float(rng.choice([0.0, 0.4, 0.8]))), **kw)
    swapping = bool(kw.get("use_swapping"))
    frame, fed, log = int(rng.integers(0, 4)), 0, []
    try:
        for step in range(int(rng.integers(10, 18))):
            op = rng.choice(["frame", "frame", "frame", "decay", "render", "reset"], p=[0.3, 0.2, 0.15, 0.15, 0.15, 0.05])
            if fed == 0:
                op = "frame"
            if op == "frame":
                frame = max(0, frame + int(rng.choice([1, 1, 1, 2, 5, -3])))   # mostly forward, jumps both ways
                prepare = bool(rng.random() < 0.8)
                log.append(("frame", frame, prepare))
                rgba, d, T, _ = sc.frame(frame)
                if rng.random() < 0.15:
                    d = d.copy(); d[:, : W // 3] = 0                            # a third of the image without depth
                raised = []
                for e in (g, o):
                    e.update_view(rgba, d)
                    e.set_pose_inv_m(T)
                    try:
                        e.process_frame(); raised.append(False)
                    except OutOfBlocksError:
                        raised.append(True)
                    if prepare:
                        e.prepare()
                assert raised[0] == raised[1], log
                fed += 1
                assert_scene_equal(g, o, voxels=False)
                if prepare:
                    # Prepare() is skipped without visible blocks and the range image "keeps its previous contents": on the
                    # instance path that is the image of the last ProcessFrame (k_small.h builds it inside the allocation's
                    # kernel), in the serial engine the image of the last Prepare — they differ after frames without a Prepare
                    # (seed 835).  Nothing reads that image before the next frame with visible blocks rebuilds it.
                    empty = o.get_stats().no_visible_blocks == 0
                    assert_render_equal(g, o, skip=("minmax",) if empty else ())
            elif op == "decay" and not swapping:   # (GC and swapping together: tests/test_swapping.py drives the supported order)
                args = (int(rng.choice([1, 2, 5, 100])), int(rng.choice([0, 0, 1, 3])), bool(rng.random() < 0.25))
                log.append(("decay",) + args)
                for e in (g, o):
                    e.decay(*args)
                assert_scene_equal(g, o, voxels=False)
            elif op == "render":
                T = sc.pose(max(0, frame + int(rng.integers(-2, 3)))).astype(np.float64)
                T[:3, 3] += rng.normal(0, 0.05, 3)
                M = np.linalg.inv(T).astype(np.float32)
                t = RENDER_TYPES[rng.integers(len(RENDER_TYPES))]
                log.append(("render", int(t)))
                cg, dg = g.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                co, do = o.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                assert np.array_equal(dg, do) and np.array_equal(cg, co), log
                assert_render_equal(g, o, freeview=True)
                assert np.array_equal(g.dump_visible_list(True), o.dump_visible_list(True)), log
            elif op == "reset":
                log.append(("reset",))
                for e in (g, o):
                    e.reset_scene()
                fed = 0
        assert_scene_equal(g, o)   # every voxel
        if swapping:
            sg, so = g.dump_swap_state(), o.dump_swap_state()
            assert np.array_equal(sg[0], so[0]) and np.array_equal(sg[1], so[1]), log
    except AssertionError as ex:
        raise AssertionError(f"seed {seed} kind {kind} {W}x{H} {kw}\ncalls: {log}\n{ex}") from None
    finally:
        g.close(); o.close()
