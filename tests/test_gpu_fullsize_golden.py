"""-m gpu: the BENCHED configurations at BASELINE.json's full size against committed oracle digests
(tests/golden/fullsize_digests.json, made offline by tests/golden/make_golden_fullsize.py with the
CPU oracle) — frame by frame, without running the oracle on the GPU box:

  bench_5mm         bench.py's default workload with its exact table sizes, frames 0..24
                    (= the driver's `--steps 20 --warmup 5` run)
  cfg5_4mm_gc_swap  the `4mm` preset with voxel GC + host swapping (BASELINE configs[4])
  cfg2_instances    static map + 4 instance volumes, masks split on the device (configs[2])

Bit-exact: every digest is a SHA-256 over the raw arrays (hash table, visible list, voxels,
range image, raycast points, ICP maps, swap state, host store)."""
import json
import os

import pytest

from tests.golden.make_golden_fullsize import CASES, case_frames, run_case

pytestmark = pytest.mark.gpu
_PATH = os.path.join(os.path.dirname(__file__), "golden", "fullsize_digests.json")
GOLD = json.load(open(_PATH))["cases"] if os.path.exists(_PATH) else {}


def _diff(got, want, path=""):
    out = []
    if isinstance(want, dict):
        for k in want:
            out += _diff(got.get(k) if isinstance(got, dict) else None, want[k], f"{path}/{k}")
    elif isinstance(want, list):
        for i, w in enumerate(want):
            out += _diff(got[i] if isinstance(got, list) and i < len(got) else None, w, f"{path}[{i}]")
    elif got != want:
        out.append(path)
    return out


@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_fullsize_golden(hip_api, name):
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    assert name in GOLD, f"{name} missing from fullsize_digests.json: run tests/golden/make_golden_fullsize.py"

    def hip_factory(settings, calib_args):
        return EngineCore(default_settings(**settings), make_calib(*calib_args))
    got = run_case(hip_factory, CASES[name], case_frames(CASES[name]))
    bad = _diff(got, GOLD[name])
    assert not bad, f"{len(bad)} digests differ from the oracle's, first: {bad[:8]}"


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_fullsize_composite_through_the_exchange_matches_golden(hip_api, devices, monkeypatch):
    """the full-size composite digests of `cfg3_static_plus_7_instances` were produced through dsr_composite_instances
    (host pointers); the path a multi-GPU run takes — every volume rendered straight into its exchange slot
    (dsr_exchange_render_slot), the gather, k_composite<true> over per-layer pointers into the gathered buffer
    (dsr_exchange_gather_and_composite) — was only compared at 320x96.  Here the same 5 mm map + 7 instance volumes are fused at
    1242x375 and frames 5, 9 and 14 are composited THROUGH the exchange: the committed oracle digests, bit for bit.  `devices`:
    one rank holding all slots, and three ranks sharing the box's GPU (ranks of one GPU exchange nothing: layers in place), the
    latter with a real RCCL communicator forced (DSR_EXCHANGE_FORCE_RCCL)."""
    import numpy as np
    from dynslam_amd import _capi
    from dynslam_amd.engine import EngineCore, Exchange, OutOfBlocksError, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    from tests.golden.make_golden_fullsize import INSTANCE, W, H, _h
    name = "cfg3_static_plus_7_instances"
    case, gold = CASES[name], GOLD[name]
    if len(devices) > 1:
        monkeypatch.setenv("DSR_EXCHANGE_FORCE_RCCL", "1")
    frames = case_frames(case)
    sc = StreetScene(W, H, n_instances=case["instances"])
    calib = make_calib(*sc.intrinsics(), W, H)
    e = EngineCore(default_settings(**case["settings"]), calib)
    inst = [EngineCore(default_settings(**INSTANCE), calib) for _ in range(case["instances"])]
    n_ranks = len(devices)
    slots = -(-case["instances"] // n_ranks)
    x = Exchange(W * H, slots, devices=devices)
    where = {k: (k % n_ranks, k // n_ranks) for k in range(case["instances"])}  # instance -> (rank, slot)
    checked = 0
    for i in range(case["frames"]):
        rgba, d, T, masks = frames[i]
        e.update_view(rgba, d)
        for k, x0, y0, mask, rel in masks:
            e.split_silhouette(inst[k], mask, x0, y0)
            inst[k].set_pose_inv_m(rel)
            inst[k].process_frame()
            inst[k].prepare()
        e.set_pose_inv_m(T)
        try:
            e.process_frame()
        except OutOfBlocksError:
            pass
        e.prepare()
        if i not in (5, 9, 14):
            continue
        M = np.linalg.inv(np.asarray(T, np.float64)).astype(np.float32)
        visible = {k: rel for k, _, _, _, rel in masks}
        for k in range(case["instances"]):  # every slot is rendered or emptied, as ShardedScene does
            r, s = where[k]
            x.render_slot(r, s, inst[k] if k in visible else None,
                          pose_m=np.linalg.inv(np.asarray(visible[k], np.float64)).astype(np.float32) if k in visible else None)
        tr, td = x.target_ptrs(0)
        e.wait_for_stream(x.stream(0))
        e.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, M, None, tr, td)
        layers = sorted((1 + k, where[k][0], where[k][1]) for k in visible)
        x.gather_and_composite(0, [(r, s, t) for t, r, s in layers], target_engine=e, tint_strength=1.0, dim_background=True)
        c_rgba, c_depth = x.read_target(0, W, H)
        rec = gold["frames"][i]
        assert rec["composite_layers"] == len(masks)
        assert _h(c_rgba) == rec["composite_rgba"] and _h(c_depth) == rec["composite_depth"], f"frame {i}: composite through the exchange differs"
        checked += 1
    assert checked == 3
    x.close()
    for ie in inst:
        ie.close()
    e.close()
