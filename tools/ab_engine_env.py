#!/usr/bin/env python3
"""A/B of engine variants selected by environment variables (read at engine creation) on the bench workload, all variants inside
ONE process / one gpurun call: per variant the HIP-event averages of k_integrate and of the frame's raycast, the wall time per
frame, and SHA-256 digests of the final raycast result + ICP maps and of the voxels in use — every variant must produce the bits
of the first.
usage: python tools/ab_engine_env.py 'DSR_INTEGRATE_XLDS=0' 'DSR_INTEGRATE_XLDS=1' 'DSR_INTEGRATE_XLDS=1,DSR_RAYCAST_SPLIT=64' ...
       [--frames 25] [--warmup 5] [--preset 5mm] [--repeat 2]"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="5mm")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    a = ap.parse_args()
    from bench import make_frames, settings_kwargs
    frames = make_frames(a.width, a.height, a.frames, 0)
    import torch
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    from tests.golden.make_golden_fullsize import voxel_digest
    dev = torch.device("cuda", 0)
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]
    torch.cuda.synchronize()
    calib = make_calib(*StreetScene(a.width, a.height).intrinsics(), a.width, a.height)
    kw = settings_kwargs(a.preset)
    out = []
    touched = set()
    for rep in range(a.repeat):
        for v in a.variants:
            for k in touched:
                os.environ.pop(k, None)
            for kv in [x for x in v.split(",") if x]:
                k, val = kv.split("=")
                os.environ[k] = val
                touched.add(k)
            e = EngineCore(default_settings(**kw, device=0, sync_status=0), calib)
            t0 = None
            for i in range(a.frames):
                if i == a.warmup:
                    e.sync(); e.profile_enable(2); e.profile_reset(); t0 = time.perf_counter()
                e.update_view_dev(rgb[i].data_ptr(), dep[i].data_ptr())
                e.set_pose_inv_m(frames[i][2]); e.process_frame(); e.prepare()
            e.sync()
            wall = (time.perf_counter() - t0) / (a.frames - a.warmup)
            prof = {r["name"]: 1e3 * r["total_ms"] / max(1, r["launches"]) for r in e.profile_get()}
            e.profile_enable(False)
            rec = {"variant": v, "rep": rep, "integrate_us": round(prof.get("integrate", 0.0), 1), "raycast_us": round(prof.get("raycast", 0.0), 1), "raycast_tail_us": round(prof.get("raycast_tail", 0.0), 1),
                   "raycast_tail_us": round(prof.get("raycast_tail", 0.0), 1), "ms_per_frame": round(1e3 * wall, 4)}
            if rep == 0:
                rs = e.dump_render_state()
                ht = e.dump_hash_table()
                rec["digest"] = hashlib.sha256(rs["raycast_result"].tobytes() + rs["points"].tobytes() + rs["normals"].tobytes()).hexdigest()[:16]
                rec["voxels"] = voxel_digest(e, ht["ptr"][ht["ptr"] >= 0])[:16]
            print(json.dumps(rec), flush=True)
            out.append(rec)
            e.close()
    first = [r for r in out if r["rep"] == 0]
    ok = all(r["digest"] == first[0]["digest"] and r["voxels"] == first[0]["voxels"] for r in first)
    print(json.dumps({"all_bit_identical": ok}), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
