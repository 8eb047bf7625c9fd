#!/usr/bin/env python3
"""Where does the frame of ONE instance volume go?  (~0.1 ms fuse + ~0.4 ms preview, launch bound.)

One "view" engine holding the full frame + one instance volume (0.035 m, mu 1.0, 7142 blocks, upstream's table sizes), driven
exactly as ShardedScene drives them: silhouette split (mask in HBM) -> SetPose -> ProcessFrame -> Prepare -> fused-preview render
(colour + depth).  Reports, per frame:
  * host time of every API call (perf_counter around the call, nothing waited for): what the CPU pays to ENQUEUE the frame;
  * wall time per frame with the stream drained once at the end of the run (host and GPU overlapped) and with a sync per frame
    (the dependent chain: enqueue + GPU latency);
  * GPU time per kernel (HIP events, --profile-all style) and the number of launches.
Usage (GPU box):  python tools/bench_instance_frame.py [--frames 200] [--host-masks]
"""
import argparse
import json
import os
import sys
import time
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--host-masks", action="store_true", help="upload the mask per call (the reference host's way: synchronises)")
    ap.add_argument("--two-renders", action="store_true", help="colour and depth as two get_image calls (round 2's preview)")
    ap.add_argument("--two-step-split", action="store_true", help="extract + remove as two calls (round 4's view split)")
    ap.add_argument("--share-stream", action="store_true", help="the instance queues its work on the view engine's stream (dsr_engine_share_stream)")
    ap.add_argument("--reset-every", type=int, default=0, help="ResetScene of the instance volume every N frames: the sequence has 16 unique "
                    "frames, so without it every frame after the first pass re-fuses blocks that exist (no allocation); with 16 every pass "
                    "allocates again")
    args = ap.parse_args()
    import bench
    W, H = 1242, 375
    n_unique = 16  # instance 0 of the synthetic street is overtaken by the camera around frame 19
    frames = bench.make_frames(W, H, n_unique, 1)
    import torch
    dev = torch.device("cuda", 0)
    from dynslam_amd import _capi
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    calib = make_calib(*StreetScene(W, H).intrinsics(), W, H)
    kinds = bench.volume_settings("5mm")
    view = EngineCore(default_settings(**kinds["view"], device=0, sync_status=0), calib)
    inst = EngineCore(default_settings(**kinds["instance"], device=0, sync_status=0), calib)
    if args.share_stream:
        inst.share_stream(view)
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]
    masks = []
    for f in frames:
        m = [x for x in f[3] if x[0] == 0]
        assert m, "instance 0 must be visible in every frame"
        k, x0, y0, mk, rel = m[0]
        from dynslam_amd.engine import PoseArg  # (poses converted once: the tool measures the engine's enqueue cost, not numpy's)
        masks.append((x0, y0, mk, torch.from_numpy(np.ascontiguousarray(mk)).to(dev), PoseArg(rel) if hasattr(view, "share_stream") else rel,
                      PoseArg(np.linalg.inv(rel.astype(np.float64)).astype(np.float32))))
    out_rgba = torch.zeros((W * H, 4), dtype=torch.uint8, device=dev)
    out_depth = torch.zeros((W * H,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    host = defaultdict(float)

    timed_calls = [True]

    def call(name, fn, *a):
        if not timed_calls[0]:
            fn(*a)
            return
        t = time.perf_counter()
        fn(*a)
        host[name] += time.perf_counter() - t

    def frame(i):
        j = i % n_unique
        if args.reset_every and i % args.reset_every == 0:
            call("reset_scene", inst.reset_scene)
        x0, y0, mk, mk_dev, rel, pose_m = masks[j]
        call("update_view_dev", view.update_view_dev, rgb[j].data_ptr(), dep[j].data_ptr())
        if not args.two_step_split and hasattr(view, "split_silhouette_dev"):
            if args.host_masks:
                call("split_silhouette", view.split_silhouette, inst, mk, x0, y0)
            else:
                call("split_silhouette", view.split_silhouette_dev, inst, mk_dev.data_ptr(), x0, y0, mk.shape[1], mk.shape[0])
        elif args.host_masks:
            call("extract_silhouette", view.extract_silhouette, inst, mk, x0, y0)
            call("remove_silhouette", view.remove_silhouette, mk, x0, y0)
        else:
            call("extract_silhouette", view.extract_silhouette_dev, inst, mk_dev.data_ptr(), x0, y0, mk.shape[1], mk.shape[0])
            call("remove_silhouette", view.remove_silhouette_dev, mk_dev.data_ptr(), x0, y0, mk.shape[1], mk.shape[0])
        call("set_pose", inst.set_pose_inv_m, rel)
        call("process_frame", inst.process_frame)
        call("prepare", inst.prepare)
        if args.two_renders:
            call("get_image(colour)", inst.get_image_dev, _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m, None, out_rgba.data_ptr(), None)
            call("get_image(depth)", inst.get_image_dev, _capi.IMAGE_FREECAMERA_DEPTH, pose_m, None, None, out_depth.data_ptr())
        else:
            call("get_image(colour+depth)", inst.get_image_dev, _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m, None,
                 out_rgba.data_ptr(), out_depth.data_ptr())

    def drain():
        view.sync(); inst.sync(); torch.cuda.synchronize()

    for i in range(n_unique):  # warm-up: one pass over the sequence
        frame(i)
    drain()
    res = {"frames": args.frames, "host_masks": args.host_masks, "two_renders": args.two_renders, "two_step_split": args.two_step_split,
           "share_stream": args.share_stream, "reset_every": args.reset_every, "lib": os.environ.get("DSR_HIP_LIB", "default")}
    # (a) free-running: host enqueues ahead, one drain at the end
    host.clear()
    t0 = time.perf_counter()
    for i in range(args.frames):
        frame(n_unique + i)
    t_enq = time.perf_counter() - t0
    drain()
    t_all = time.perf_counter() - t0
    res["free_running"] = {"us_per_frame": round(1e6 * t_all / args.frames, 1), "host_enqueue_us_per_frame": round(1e6 * t_enq / args.frames, 1),
                           "host_us_per_call": {k: round(1e6 * v / args.frames, 1) for k, v in host.items()}}
    # (a') the same without the per-call clock reads (two perf_counter calls + a dict update per API call are host time too)
    timed_calls[0] = False
    t0 = time.perf_counter()
    for i in range(args.frames):
        frame(n_unique + i)
    t_enq = time.perf_counter() - t0
    drain()
    t_all = time.perf_counter() - t0
    timed_calls[0] = True
    res["free_running_untimed_calls"] = {"us_per_frame": round(1e6 * t_all / args.frames, 1), "host_enqueue_us_per_frame": round(1e6 * t_enq / args.frames, 1)}
    # (b) a drain per frame: enqueue + the GPU's dependent chain
    t0 = time.perf_counter()
    for i in range(args.frames):
        frame(n_unique + i)
        drain()
    res["sync_per_frame"] = {"us_per_frame": round(1e6 * (time.perf_counter() - t0) / args.frames, 1)}
    # (c) GPU time per kernel
    for e in (view, inst):
        e.profile_enable(True); e.profile_reset()
    for i in range(args.frames):
        frame(n_unique + i)
    drain()
    kern = {}
    for tag, e in (("view", view), ("inst", inst)):
        for r in e.profile_get():
            kern[f"{tag}:{r['name']}"] = {"launches_per_frame": round(r["launches"] / args.frames, 2), "us_per_frame": round(1e3 * r["total_ms"] / args.frames, 2)}
        e.profile_enable(False)
    res["gpu_kernels"] = kern
    res["gpu_us_per_frame"] = round(sum(v["us_per_frame"] for v in kern.values()), 1)
    res["launches_per_frame"] = round(sum(v["launches_per_frame"] for v in kern.values()), 1)
    st = inst.get_stats()
    res["instance_visible_blocks"] = st.no_visible_blocks
    res["instance_allocated_blocks"] = 7142 - 1 - st.last_free_block_id
    print(json.dumps(res))
    view.close(); inst.close()


if __name__ == "__main__":
    main()
