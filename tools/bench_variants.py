#!/usr/bin/env python3
"""Kernel-variant harness: ONE process, the bench workload's frames generated once, one engine per
variant (the library reads DSR_* environment switches at engine creation), integrate / raycast timed by
the engine's HIP events over frames warmup..n.  usage: python tools/bench_variants.py "GRID=8192" "GRID=16384" ...   (an experiment adds its own DSR_* switch here)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KEYS = {"GRID": "DSR_GRID_INTEGRATE", "GRIDX": "DSR_GRID_EXPECTED", "GRIDD": "DSR_GRID_DECAY", "OVERLAP": "DSR_OVERLAP_EXPECTED"}  # switches the library reads at engine creation


def main():
    from bench import make_frames, settings_kwargs
    n, warm = 15, 5
    frames = make_frames(1242, 375, n)
    import torch
    dev = torch.device("cuda", 0)
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    sc = StreetScene(1242, 375)
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]
    torch.cuda.synchronize()
    for spec in sys.argv[1:] or [""]:
        for k in KEYS.values():
            os.environ.pop(k, None)
        for kv in spec.split():
            k, v = kv.split("=")
            os.environ[KEYS[k]] = v
        e = EngineCore(default_settings(**settings_kwargs("5mm"), device=0, sync_status=0), make_calib(*sc.intrinsics(), 1242, 375))
        for i in range(n):
            if i == warm:
                e.sync(); e.profile_enable(1 if os.environ.get("DSR_VARIANTS_PROFILE_ALL") else 2); e.profile_reset()
            e.update_view_dev(rgb[i].data_ptr(), dep[i].data_ptr())
            e.set_pose_inv_m(frames[i][2])
            e.process_frame()
            e.prepare()
        e.sync()
        prof = {r["name"]: round(1e3 * r["total_ms"] / max(1, r["launches"]), 1) for r in e.profile_get()}
        st = e.get_stats()
        rec = {"variant": spec, "integrate_us": prof.get("integrate"), "raycast_us": prof.get("raycast"),
               "visible": st.no_visible_blocks, "status": st.sticky_status}
        if os.environ.get("DSR_VARIANTS_PROFILE_ALL"):
            rec["all_us"] = prof
        print(json.dumps(rec), flush=True)
        e.close()


if __name__ == "__main__":
    main()
