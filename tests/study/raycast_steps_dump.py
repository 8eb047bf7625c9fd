#!/usr/bin/env python3
"""Offline study input (uses the CPU oracle, hence under tests/): per-ray march step counts of the bench workload's steady state, from the CPU oracle
(oracle/dsr_oracle.cpp orc_debug_raycast_steps).  Fuses frames 0..N-1 of bench.py's default workload (1242x375, preset 5mm,
the bench's table sizes) and writes the step count of every ray of the LAST frame's Prepare() raycast to an .npy file,
for tools/raycast_divergence_model.py.

usage: python tests/study/raycast_steps_dump.py [--frames 8] [--preset 5mm] [--out /tmp/raycast_steps.npy]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--preset", default="5mm")
    ap.add_argument("--out", default="/tmp/raycast_steps.npy")
    a = ap.parse_args()
    from bench import make_frames, settings_kwargs
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, load_api, oracle_settings
    W, H = 1242, 375
    frames = make_frames(W, H, a.frames)
    sc = StreetScene(W, H)
    e = OracleEngine(oracle_settings(**settings_kwargs(a.preset)), make_calib(*sc.intrinsics(), W, H), threads=os.cpu_count() or 1)
    lib = load_api()._lib if hasattr(load_api(), "_lib") else C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    steps = np.zeros((H, W), np.int32)
    for i, (rgba, d, T, _) in enumerate(frames):
        t0 = time.time()
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.process_frame()
        if i == a.frames - 1:
            lib.orc_debug_raycast_steps(steps.ctypes.data_as(C.c_void_p), C.c_int(W))
        e.prepare()
        print(f"frame {i}: {time.time() - t0:.1f} s, visible {e.get_stats().no_visible_blocks}", flush=True)
    lib.orc_debug_raycast_steps(None, C.c_int(0))
    np.save(a.out, steps)
    print("rays", steps.size, "mean steps", steps.mean(), "max", steps.max(), "->", a.out)
    e.close()


if __name__ == "__main__":
    main()
