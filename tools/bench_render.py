#!/usr/bin/env python3
"""Free-view rendering cost (ITMMainEngine::GetImage: FindVisibleBlocks + CreateExpectedDepths +
RenderImage) on the bench's 5 mm map: per image type, device-resident outputs, HIP-event kernel table.
Usage: python tools/bench_render.py [--frames 30] [--preset 5mm]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_frames, settings_kwargs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--preset", default="5mm")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
W, H = 1242, 375
frames = make_frames(W, H, args.frames)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dynslam_amd import _capi  # noqa: E402
from dynslam_amd.engine import EngineCore, default_settings, make_calib  # noqa: E402
from dynslam_amd.synth import StreetScene  # noqa: E402
sc = StreetScene(W, H)
dev = torch.device("cuda", 0)
e = EngineCore(default_settings(**settings_kwargs(args.preset), device=0, sync_status=0), make_calib(*sc.intrinsics(), W, H))
for rgba, d, T, _ in frames:
    e.update_view(rgba, d); e.set_pose_inv_m(T); e.process_frame(); e.prepare()
e.sync()
pose = np.linalg.inv(frames[-3][2].astype(np.float64)).astype(np.float32)  # a view two frames back
pose2 = np.linalg.inv(frames[-4][2].astype(np.float64)).astype(np.float32)
rgba_out = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
depth_out = torch.empty((H, W), dtype=torch.float32, device=dev)
res = {}
for name, t, want_d in (("shaded", _capi.IMAGE_FREECAMERA_SHADED, False), ("colour", _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, False),
                        ("normal", _capi.IMAGE_FREECAMERA_COLOUR_FROM_NORMAL, False),
                        ("weight", _capi.IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT, False),
                        ("depth", _capi.IMAGE_FREECAMERA_DEPTH, True)):
    call = lambda ps: e.get_image_dev(t, ps, None, 0 if want_d else rgba_out.data_ptr(), depth_out.data_ptr() if want_d else 0)
    call(pose); e.sync()
    e.profile_enable(True); e.profile_reset()
    t0 = time.perf_counter()
    for k in range(args.reps):  # a new pose every call: visible list, range image and raycast recomputed
        call(pose2 if k % 2 == 0 else pose)
    e.sync()
    dt = (time.perf_counter() - t0) / args.reps
    prof = {r["name"]: round(1e3 * r["total_ms"] / max(1, r["launches"]), 1) for r in e.profile_get()}
    e.profile_enable(False)
    call(pose); e.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):  # same pose, unchanged scene: only the shading runs
        call(pose)
    e.sync()
    dt_same = (time.perf_counter() - t0) / args.reps
    res[name] = {"ms_per_call": round(dt * 1e3, 3), "ms_per_call_same_pose": round(dt_same * 1e3, 3), "kernels_us": prof}
print(json.dumps({"workload": f"GetImage on the {args.preset} map after {args.frames} frames, 1242x375", "types": res}))
