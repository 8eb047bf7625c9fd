#!/usr/bin/env python3
"""BASELINE configs[4]: 4541 frames (KITTI-odometry seq 00's length), 4 mm voxels, voxel GC
(max_weight 1, min_age 200: DynSLAMGUI.cpp:36-42) + host swap-out, one MI355X, sustained.

The synthetic street is driven as a LOOP: a 40 m stretch (50 frames at 0.8 m/frame, the period of the
scene) traversed again and again — block coordinates are `short` (upstream ITMHashEntry), so a straight
3.6 km drive at 4 mm is not representable (32767 blocks x 3.2 cm = 1 km), and a map that only ever grows
would exhaust any voxel array at ~100 k new blocks per frame.  Every lap re-fuses the same surfaces
(weights saturate), blocks behind the camera are swapped out to the pinned host store (<= 4096 per
frame) and swapped back in (merged) when the next lap sees them, noise blocks seen once are freed 200
frames later, their entries become tombstones that later allocations re-use: everything the long run
is meant to stress — FIFO, host slabs, tombstones, free list — is exercised 90 times over.

Prints one JSON line: sustained frames/s (whole run and per 500-frame window), peak HBM in use, pinned
host bytes, decayed blocks, allocated blocks, sticky status, and — every --check-every frames and at
the end — the structural invariants of tests/test_gpu_fullsize.py::check_structure (free-list
accounting, unique pointers, chains reachable, visible list ascending).

usage: python tools/bench_cfg5_sustained.py [--frames 4541] [--check-every 1000] [--min-age 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4541)
    ap.add_argument("--period", type=int, default=50, help="frames per lap (40 m at 0.8 m/frame)")
    ap.add_argument("--preset", default="4mm")
    ap.add_argument("--min-age", type=int, default=200)
    ap.add_argument("--max-weight", type=int, default=1)
    ap.add_argument("--check-every", type=int, default=1000)
    ap.add_argument("--no-swap", action="store_true")
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    a = ap.parse_args()
    from bench import make_frames, settings_kwargs
    frames = make_frames(a.width, a.height, a.period)

    import torch
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    from dynslam_amd.invariants import check_structure
    W, H = a.width, a.height
    sc = StreetScene(W, H)
    kw = settings_kwargs(a.preset)
    if not a.no_swap:
        kw["use_swapping"] = 1
    rgb_dev = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep_dev = [torch.from_numpy(f[1]).to(dev) for f in frames]
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info(dev)
    eng = EngineCore(default_settings(**kw, device=0, sync_status=0), make_calib(*sc.intrinsics(), W, H))
    eng.sync()
    peak_used = total - torch.cuda.mem_get_info(dev)[0]
    windows, checks = [], []
    t_run = 0.0
    t0 = time.perf_counter()
    w0, wf = t0, 0
    for i in range(a.frames):
        j = i % a.period
        eng.update_view_dev(rgb_dev[j].data_ptr(), dep_dev[j].data_ptr())
        eng.set_pose_inv_m(frames[j][2])
        eng.process_frame()
        eng.prepare()
        eng.decay(a.max_weight, a.min_age, False)
        if (i + 1) % 500 == 0 or i + 1 == a.frames:
            eng.sync()
            now = time.perf_counter()
            st = eng.get_stats()
            used = total - torch.cuda.mem_get_info(dev)[0]
            peak_used = max(peak_used, used)
            windows.append({"frames": [wf, i + 1], "frames_per_s": round((i + 1 - wf) / (now - w0), 2),
                            "allocated_blocks": kw["sdf_local_block_num"] - 1 - st.last_free_block_id,
                            "visible_blocks": st.no_visible_blocks, "decayed_blocks": st.decayed_block_count,
                            "host_store_slots": st.host_store_slots, "hbm_used_GB": round(used / 1e9, 2), "status": st.sticky_status})
            print(json.dumps(windows[-1]), file=sys.stderr, flush=True)
            t_run += now - w0
            if a.check_every and ((i + 1) % a.check_every == 0 or i + 1 == a.frames):
                tc = time.perf_counter()
                check_structure(eng, kw["sdf_local_block_num"], kw["hash_bucket_num"])
                checks.append({"frame": i + 1, "ok": True, "seconds": round(time.perf_counter() - tc, 1)})
            w0, wf = time.perf_counter(), i + 1
    st = eng.get_stats()
    ht = eng.dump_hash_table()
    tomb = int(((ht["ptr"] < -1) & (ht["offset"] != 0)).sum())  # freed entries still linking a chain
    out = {"workload": f"configs[4]: {a.frames} frames = {a.frames / a.period:.1f} laps of a {a.period}-frame loop, {W}x{H}, preset {a.preset} "
                       f"(voxel {kw['voxel_size']} m, mu {kw['mu']} m), voxel GC max_weight {a.max_weight} min_age {a.min_age}, "
                       f"host swapping {'on' if not a.no_swap else 'off'}; step = UpdateView + ProcessFrame (+swap) + Prepare + Decay",
           "frames_per_s": round(a.frames / t_run, 2), "ms_per_frame": round(1e3 * t_run / a.frames, 4),
           "frames_per_s_first_500": windows[0]["frames_per_s"], "frames_per_s_last_500": windows[-1]["frames_per_s"],
           "peak_hbm_used_GB": round(peak_used / 1e9, 2), "pinned_host_GB": round(st.host_store_capacity_slots * 4096 / 1e9, 3),
           "host_store_slots": st.host_store_slots, "decayed_block_count": st.decayed_block_count,
           "allocated_blocks": kw["sdf_local_block_num"] - 1 - st.last_free_block_id, "tombstones_in_chains": tomb,
           "status": st.sticky_status, "structure_checks": checks, "windows": windows}
    print(json.dumps(out), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
