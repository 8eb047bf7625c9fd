#!/usr/bin/env python3
"""Through-shim throughput: the C++ host shim/host_bench (our HostDriver over
shim/ITMLib.h; with --reference the prebuilt tests/refhost/_build/ref_driver_host = the reference's
own InfiniTamDriver class) fed with pageable host frames of the bench workload, every frame paying
what DynSLAM's host pays: BGR->RGBA, H2D of the frame, ProcessFrame, Prepare, the two preview
conversions with their D2H copies, one status synchronisation.
usage: python tools/bench_through_shim.py [--preset 5mm] [--steps 20] [--warmup 5] [--reference] [--decay]"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_frames(path, frames, render_pose_m):
    with open(path, "wb") as f:
        for rgba, d, T, _ in frames:
            f.write(np.ascontiguousarray(rgba[..., 2::-1]).tobytes())
            f.write(np.ascontiguousarray(d, np.int16).tobytes())
            f.write(np.ascontiguousarray(T, np.float32).tobytes())
        f.write(np.ascontiguousarray(render_pose_m, np.float32).tobytes())


def write_masks(path, frames):
    import struct
    with open(path, "wb") as f:
        for fr in frames:
            masks = fr[3]
            f.write(struct.pack("<i", len(masks)))
            for k, x0, y0, mask, rel in masks:
                f.write(struct.pack("<5i", k, x0, y0, mask.shape[1], mask.shape[0]))
                f.write(np.ascontiguousarray(rel, np.float32).tobytes())
                f.write(np.ascontiguousarray(mask, np.uint8).tobytes())


def run(exe, frames, W, H, intr, preset_kw, warmup, decay=None, tmpdir=None, instances=0):
    tmpdir = tmpdir or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    path = os.path.join(tmpdir, f"dsr_frames_{os.getpid()}.bin")
    M = np.linalg.inv(np.asarray(frames[-1][2], np.float64)).astype(np.float32)
    write_frames(path, frames, M)
    try:
        args = [exe, path, str(W), str(H)] + [repr(float(v)) for v in intr] + [str(len(frames)), str(warmup),
                repr(preset_kw["voxel_size"]), repr(preset_kw["mu"]), str(preset_kw["sdf_local_block_num"]),
                str(preset_kw["hash_bucket_num"]), str(preset_kw["excess_list_size"])]
        if decay:
            args += [str(decay[0]), str(decay[1])]
        if instances:  # configs[2]: silhouettes split on the GPU, one volume per instance (shim/host_bench.cpp --masks)
            write_masks(path + ".masks", frames)
            args += ["--masks", path + ".masks", str(instances)]
        out = subprocess.check_output(args, timeout=600).decode().strip()
    finally:
        os.unlink(path)
        if os.path.exists(path + ".masks"):
            os.unlink(path + ".masks")
    return dict(kv.split("=") for kv in out.split())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="5mm")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--decay", action="store_true")
    ap.add_argument("--instances", type=int, default=0, help="configs[2]: this many instance volumes, views split on the GPU")
    a = ap.parse_args()
    from bench import PRESETS, make_frames
    from dynslam_amd.synth import StreetScene
    frames = make_frames(a.width, a.height, a.warmup + a.steps, a.instances)
    exe = os.path.join(ROOT, "tests", "refhost", "_build", "ref_driver_host") if a.reference else os.path.join(ROOT, "shim", "host_bench")
    r = run(exe, frames, a.width, a.height, StreetScene(a.width, a.height).intrinsics(), PRESETS[a.preset], a.warmup,
            (1, 200) if a.decay else None, instances=a.instances)
    print(r)
