#!/usr/bin/env python3
"""What is k_raycast's launch made of?  Per-wave clocks and stage counts of the march, from a MEASUREMENT build of the library
(-DDSR_RAYCAST_STATS: the kernel writes 12 words per wave; built by this tool's --build, never the product build).

All 7.3 k waves of a 1242x375 launch are resident at once (56 VGPRs: 8 waves / SIMD x 1024 SIMDs), so the launch lasts as long as
its slowest waves.  A wave's iteration runs up to three dependent memory stages one after the other — a table read for the lanes
whose block changed and whose look-ahead missed, the voxel read, the trilinear read for the lanes inside the band — and pays each
of them if ANY of its 64 rays needs it.  This tool reports, per wave: duration, iterations, iterations with a table read / a band
read, and the largest number of stages any single ray needs for itself (what a wave of decoupled lanes would run).

usage:  python tools/raycast_wave_stats.py --build          (here: hipcc)
        python tools/raycast_wave_stats.py [--frames 10]    (GPU box)
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "dynslam_amd", "csrc", "libdsr_hip_rcstats.so")


def build():
    import __graft_entry__ as g
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIPCC_FLAGS + ["-DDSR_RAYCAST_STATS", "-o", LIB] + g.HIP_SOURCES  # (one step, every translation unit)
    subprocess.check_call(cmd, cwd=g.CSRC)
    print(LIB)


def pct(a, qs=(50, 90, 99, 100)):
    return {f"p{q}": round(float(np.percentile(a, q)), 2) for q in qs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--dump", default=None, help="write the raw per-wave table to this .npy")
    a = ap.parse_args()
    if a.build:
        return build()
    os.environ["DSR_HIP_LIB"] = LIB
    from bench import make_frames, settings_kwargs
    W, H = 1242, 375
    frames = make_frames(W, H, a.frames)
    import torch
    dev = torch.device("cuda", 0)
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    e = EngineCore(default_settings(**settings_kwargs("5mm"), device=0, sync_status=0), make_calib(*StreetScene(W, H).intrinsics(), W, H))
    lib = C.CDLL(LIB)
    gx, gy = -(-W // 16), -(-H // 16)
    n_waves = gx * gy * 4
    buf = torch.zeros((n_waves, 12), dtype=torch.int32, device=dev)
    assert lib.dsr_debug_raycast_stats(C.c_void_p(buf.data_ptr())) == 0
    for i, f in enumerate(frames):
        if i == a.frames - 1:
            e.sync(); buf.zero_(); torch.cuda.synchronize()
        e.update_view_dev(torch.from_numpy(f[0]).to(dev).data_ptr(), torch.from_numpy(f[1]).to(dev).data_ptr())
        e.set_pose_inv_m(f[2])
        e.process_frame()
        e.prepare()
        e.sync()
    t = buf.cpu().numpy().view(np.uint32).astype(np.int64)
    if a.dump:
        np.save(a.dump, t)
    ran = t[:, 11] != 0
    t = t[ran]
    t0 = t[:, 0] | (t[:, 1] << 32)
    t1 = t[:, 2] | (t[:, 3] << 32)
    us = (t1 - t0) / 100.0  # wall_clock64: 100 MHz
    start = (t0 - t0.min()) / 100.0
    end = (t1 - t0.min()) / 100.0
    it, w_look, own, look, vox, band, sum_iter = (t[:, k] for k in (4, 5, 6, 7, 8, 9, 10))
    lanes, w_band = t[:, 11] & 255, t[:, 11] >> 8
    stages = it + w_look + w_band  # serial memory stages the wave runs today (the voxel stage counted once per iteration)
    full = lanes == 64
    res = {"waves": int(ran.sum()), "full_waves": int(full.sum()), "launch_span_us": round(float(end.max()), 1),
           "wave_start_us": pct(start), "wave_duration_us": {"mean": round(float(us.mean()), 1), **pct(us)},
           "resident_wave_time_over_span": round(float(us.sum() / (end.max() * 8192)), 3),
           "iterations_per_wave": {"mean": round(float(it.mean()), 1), **pct(it)},
           "iterations_with_a_table_read": {"mean": round(float(w_look.mean()), 1), **pct(w_look)},
           "iterations_with_a_band_read": {"mean": round(float(w_band.mean()), 1), **pct(w_band)},
           "stages_today": {"mean": round(float(stages.mean()), 1), **pct(stages)},
           "stages_of_the_neediest_ray": {"mean": round(float(own.mean()), 1), **pct(own)},
           "lane_utilisation": round(float(sum_iter[full].sum() / (64.0 * it[full].sum())), 3)}
    res["rounds_with_a_bucket_head_read"] = {"mean": round(float(vox.mean()), 1), **pct(vox)}   # look-ahead missed
    res["rounds_with_a_chain_entry_read"] = {"mean": round(float(band.mean()), 1), **pct(band)}  # excess-list hops
    ok = stages > 0
    res["us_per_stage_today"] = {"all": round(float(us[ok].sum() / stages[ok].sum()), 3), **pct(us[ok] / stages[ok])}
    # the slowest waves decide the launch: what are they made of?
    order = np.argsort(-us)[:max(1, len(us) // 100)]
    res["slowest_1pct"] = {"duration_us": round(float(us[order].mean()), 1), "iterations": round(float(it[order].mean()), 1),
                           "with_table_read": round(float(w_look[order].mean()), 1), "with_band_read": round(float(w_band[order].mean()), 1),
                           "stages_today": round(float(stages[order].mean()), 1), "stages_of_the_neediest_ray": round(float(own[order].mean()), 1),
                           "max_own_table_reads": round(float(look[order].mean()), 1), "rounds_with_head_read": round(float(vox[order].mean()), 1),
                           "rounds_with_chain_read": round(float(band[order].mean()), 1)}
    # time profile of the launch: waves still running at a few instants
    res["waves_running_at_us"] = {str(int(x)): int(((start <= x) & (end > x)).sum()) for x in np.linspace(0, end.max(), 9)[:-1]}
    print(json.dumps(res))
    e.close()


if __name__ == "__main__":
    main()
