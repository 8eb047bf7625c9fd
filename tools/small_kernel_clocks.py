#!/usr/bin/env python3
"""Where do the one-workgroup kernels of an instance volume (k_small.h) spend their time?  Thread 0's 100 MHz clock at every
phase boundary, from a MEASUREMENT build of the library (-DDSR_SMALL_CLOCKS; built by --build, never the product build).

usage:  python tools/small_kernel_clocks.py --build          (here: hipcc)
        python tools/small_kernel_clocks.py [--frames 64]    (GPU box)
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
LIB = os.path.join(ROOT, "dynslam_amd", "csrc", "libdsr_hip_smallclk.so")


def build():
    import __graft_entry__ as g
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIPCC_FLAGS + ["-DDSR_SMALL_CLOCKS", "-o", LIB] + g.HIP_SOURCES
    subprocess.check_call(cmd, cwd=g.CSRC)
    print(LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--twice", action="store_true", help="render the preview twice per frame (second pose perturbed: not served from the "
                    "free-view cache) and report the second launch too: code and data of the first are still in the caches")
    ap.add_argument("--reset-every", type=int, default=0, help="ResetScene of the instance volume every N frames (16: every pass over the "
                    "sequence allocates again; without it every frame after the first pass re-fuses blocks that exist)")
    a = ap.parse_args()
    if a.build:
        return build()
    os.environ["DSR_HIP_LIB"] = LIB
    import bench
    W, H = 1242, 375
    n_unique = 16
    frames = bench.make_frames(W, H, n_unique, 1)
    import torch
    dev = torch.device("cuda", 0)
    from dynslam_amd import _capi
    from dynslam_amd.engine import EngineCore, PoseArg, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    calib = make_calib(*StreetScene(W, H).intrinsics(), W, H)
    kinds = bench.volume_settings("5mm")
    view = EngineCore(default_settings(**kinds["view"], device=0, sync_status=0), calib)
    inst = EngineCore(default_settings(**kinds["instance"], device=0, sync_status=0), calib)
    inst.share_stream(view)
    lib = C.CDLL(LIB)
    buf = torch.zeros((32,), dtype=torch.int64, device=dev)
    assert lib.dsr_debug_small_clocks(C.c_void_p(buf.data_ptr())) == 0
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]
    masks = []
    for f in frames:
        k, x0, y0, mk, rel = [x for x in f[3] if x[0] == 0][0]
        masks.append((x0, y0, mk, torch.from_numpy(np.ascontiguousarray(mk)).to(dev), PoseArg(rel),
                      PoseArg(np.linalg.inv(rel.astype(np.float64)).astype(np.float32))))
    out_rgba = torch.zeros((W * H, 4), dtype=torch.uint8, device=dev)
    out_depth = torch.zeros((W * H,), dtype=torch.float32, device=dev)
    # (on the list path — round 6 — D0 / D / E / G are skipped and "F" is M + H: the merge of the frame's new entries and the dense pass
    #  over the sorted list with the ordered compaction, the stream and the range image)
    names_a = ["lds_init+A(tile scan)", "B(commit)", "D0(touched groups)", "C(apply)", "D(retest prev)", "E", "F(sweep visBits) | M+H(list path)",
               "G(stream+project+fold)", "store range image"]
    names_f = ["lds_init+sweep allocBits", "frustum+scan+project+fold", "ctr", "store range image"]
    acc_a, acc_f, acc_f2, n = np.zeros(9), np.zeros(4), np.zeros(4), 0
    for i in range(n_unique + a.frames):
        j = i % n_unique
        x0, y0, mk, mk_dev, rel, pose_m = masks[j]
        if a.reset_every and i % a.reset_every == 0:
            inst.reset_scene()
        view.update_view_dev(rgb[j].data_ptr(), dep[j].data_ptr())
        view.split_silhouette_dev(inst, mk_dev.data_ptr(), x0, y0, mk.shape[1], mk.shape[0])
        inst.set_pose_inv_m(rel)
        inst.process_frame()
        inst.prepare()
        inst.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m, None, out_rgba.data_ptr(), out_depth.data_ptr())
        view.sync(); inst.sync(); torch.cuda.synchronize()
        if i < n_unique:
            continue
        t = buf.cpu().numpy().astype(np.int64)
        acc_a += np.diff(t[0:10]) / 100.0
        fv = t[16:21]
        acc_f += np.diff(fv) / 100.0
        if a.twice:
            inst.get_image_dev(_capi.IMAGE_FREECAMERA_SHADED, masks[(j + 1) % n_unique][5], None, out_rgba.data_ptr(), out_depth.data_ptr())
            view.sync(); inst.sync(); torch.cuda.synchronize()
            t2 = buf.cpu().numpy().astype(np.int64)
            acc_f2 += np.diff(t2[16:21]) / 100.0
        n += 1
    res = {"frames": n, "reset_every": a.reset_every,
           "small_alloc_visible_us": {k: round(float(v / n), 2) for k, v in zip(names_a, acc_a)},
           "small_alloc_visible_total_us": round(float(acc_a.sum() / n), 2),
           "small_freeview_us": {k: round(float(v / n), 2) for k, v in zip(names_f, acc_f)},
           "small_freeview_total_us": round(float(acc_f.sum() / n), 2),
           "small_freeview_second_launch_us": {k: round(float(v / n), 2) for k, v in zip(names_f, acc_f2)} if a.twice else None,
           "visible_blocks": inst.get_stats().no_visible_blocks}
    print(json.dumps(res))
    view.close(); inst.close()


if __name__ == "__main__":
    main()
