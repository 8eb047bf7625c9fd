#!/usr/bin/env python3
"""A/B of the two-kernel raycast (k_raycast cut after K loop trips + k_raycast_tail, 8 lanes per ray) on the bench workload, all
variants inside ONE process / one gpurun call (boxes of the pool differ by a few per cent): per K the HIP-event average of the
frame's raycast (both kernels) and of k_integrate, and the SHA-256 of the final raycast result — every variant must produce the
bits of K = 0 (one kernel).
usage: python tools/ab_raycast_split.py [--splits 0,32,48,64,96,128] [--tail-grids 2048] [--frames 25] [--warmup 5] [--preset 5mm]"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", default="0,24,32,48,64,96,128,0")
    ap.add_argument("--tail-grids", default="2048")
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="5mm")
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    a = ap.parse_args()
    from bench import make_frames, settings_kwargs
    frames = make_frames(a.width, a.height, a.frames, 0)
    import torch
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    dev = torch.device("cuda", 0)
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]
    torch.cuda.synchronize()
    calib = make_calib(*StreetScene(a.width, a.height).intrinsics(), a.width, a.height)
    kw = settings_kwargs(a.preset)
    out = []
    for grid in [int(g) for g in a.tail_grids.split(",")]:
        for K in [int(k) for k in a.splits.split(",")]:
            os.environ["DSR_RAYCAST_SPLIT"] = str(K)
            os.environ["DSR_GRID_RAYCAST_TAIL"] = str(grid)
            e = EngineCore(default_settings(**kw, device=0, sync_status=0), calib)
            for i in range(a.frames):
                if i == a.warmup:
                    e.sync(); e.profile_enable(2); e.profile_reset()
                e.update_view_dev(rgb[i].data_ptr(), dep[i].data_ptr())
                e.set_pose_inv_m(frames[i][2]); e.process_frame(); e.prepare()
            e.sync()
            prof = {r["name"]: 1e3 * r["total_ms"] / max(1, r["launches"]) for r in e.profile_get()}
            e.profile_enable(False)
            rs = e.dump_render_state()
            digest = hashlib.sha256(rs["raycast_result"].tobytes() + rs["points"].tobytes() + rs["normals"].tobytes()).hexdigest()[:16]
            import ctypes as C
            cnt = (C.c_uint32 * 2)()
            e.api.lib.dsr_debug_tail_rays(e._h, cnt)
            rec = {"split_trips": K, "tail_grid": grid, "tail_rays": max(cnt[0], cnt[1]), "raycast_us": round(prof.get("raycast", 0.0), 1), "tail_us": round(prof.get("raycast_tail", 0.0), 1),
                   "integrate_us": round(prof.get("integrate", 0.0), 1), "digest": digest}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            e.close()
    ref = out[0]["digest"]
    ok = all(r["digest"] == ref for r in out)
    print(json.dumps({"all_bit_identical": ok, "best": min(out, key=lambda r: r["raycast_us"])}), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
