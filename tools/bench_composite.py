#!/usr/bin/env python3
"""k_composite alone: 8 layers at 1242x375 through dsr_composite_layer_ptrs_dev (the exchange's form: one pointer pair per layer),
torch events over many launches.  Scenarios: depth only / colour + dimmed background / colour; layers that cover 0 %, 3 % and
28 % of the frame (the bench's preview_hit_fraction); planes 16-byte aligned or only 8-byte aligned (as in the exchange buffer).
usage (GPU box): python tools/bench_composite.py [--iters 200]
(round 6 measured two against four pixels per lane with it — profiles/r06d_composite_px*.json; the library keeps two)"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--layers", type=int, default=8)
    a = ap.parse_args()
    import torch
    from dynslam_amd.engine import load_hip_api
    api = load_hip_api()
    dev = torch.device("cuda", 0)
    W, H, L = 1242, 375, a.layers
    P = W * H
    rng = np.random.default_rng(7)
    res = {"px_per_lane": 2, "layers": L, "pixels": P}
    stream = torch.cuda.current_stream().cuda_stream
    for cover in (0.0, 0.03, 0.28):
        # every layer a rectangle of `cover / L * 2` of the frame at a random place, depth 5..15 m
        depth = np.zeros((L, H, W), np.float32)
        for l in range(L):
            frac = 2.0 * cover / L
            bw, bh = int(W * np.sqrt(frac)), int(H * np.sqrt(frac))
            if bw and bh:
                x0, y0 = rng.integers(0, W - bw + 1), rng.integers(0, H - bh + 1)
                depth[l, y0:y0 + bh, x0:x0 + bw] = rng.uniform(5, 15, (bh, bw)).astype(np.float32)
        rgba = rng.integers(0, 255, (L, P, 4), dtype=np.uint8)
        for aligned in (True, False):
            # one buffer, layer = depth plane then colour plane (the exchange's layout); `aligned`: planes padded to 16 bytes
            plane = (P * 4 + 15) // 16 * 16 if aligned else P * 4
            buf = torch.zeros((L * 2 * plane + 64,), dtype=torch.uint8, device=dev)
            base = buf.data_ptr()
            base += (-base) % 256
            off0 = base - buf.data_ptr()
            dptr, cptr = [], []
            for l in range(L):
                o = off0 + l * 2 * plane
                buf[o:o + P * 4] = torch.from_numpy(depth[l].reshape(-1).view(np.uint8)).to(dev)
                buf[o + plane:o + plane + P * 4] = torch.from_numpy(rgba[l].reshape(-1)).to(dev)
                dptr.append(buf.data_ptr() + o)
                cptr.append(buf.data_ptr() + o + plane)
            t_rgba0 = torch.from_numpy(rng.integers(0, 255, (P, 4), dtype=np.uint8)).to(dev)
            t_depth0 = torch.from_numpy(rng.uniform(3, 30, P).astype(np.float32)).to(dev)
            t_rgba, t_depth = t_rgba0.clone(), t_depth0.clone()
            ids = (C.c_int32 * L)(*range(1, L + 1))
            DP = (C.c_void_p * L)(*dptr)
            CP = (C.c_void_p * L)(*cptr)
            for name, with_rgba, dim in (("depth_only", False, 0), ("colour_dim", True, 1), ("colour", True, 0)):
                def launch():
                    st = api.composite_layer_ptrs_dev(0, C.c_void_p(stream), C.c_void_p(t_rgba.data_ptr()) if with_rgba else None,
                                                      C.c_void_p(t_depth.data_ptr()), CP if with_rgba else None, DP, ids, L, P, C.c_float(0.4), dim)
                    assert st == 0
                for _ in range(10):
                    launch()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    launch()
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / a.iters
                comp = P * (4.0 * L + (16.0 if with_rgba else 8.0))
                res[f"cover{cover}_{'aligned16' if aligned else 'aligned8'}_{name}"] = {"us": round(us, 2), "GBps_compulsory": round(comp / us / 1e3, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
