#!/usr/bin/env python3
"""Offline study (uses the CPU oracle, hence under tests/): how many of k_integrate's tasks could end early?

k_integrate spends ~650 instructions on a visible block whether or not a single voxel of it is updated.  A block needs no work
when every voxel fails the update condition of computeUpdatedVoxelDepthInfo — its pixel outside the image, no depth there, or the
voxel more than mu BEHIND the measured surface (eta < -mu).  A conservative test per block, exact by construction: project the 8
corners, take the pixel bounding box (+1 px), and the block is skippable if the box is outside the image, or every depth in it is
invalid, or max(depth in box) < z_min(block) - mu - margin.  The depth maxima come from a min/max pyramid of the depth image.

This script fuses the first frames of bench.py's workload with the oracle, samples visible blocks of the last frame, evaluates the
true per-voxel condition (float64 here: a model, not a parity check) and the conservative test, and prints how many blocks have no
update and how many of those the test finds.

usage: python tests/study/integrate_early_out_model.py [--frames 6] [--sample 30000] [--preset 5mm]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--sample", type=int, default=30000)
    ap.add_argument("--preset", default="5mm")
    a = ap.parse_args()
    from bench import make_frames, settings_kwargs
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    W, H = 1242, 375
    kw = settings_kwargs(a.preset)
    kw["sdf_local_block_num"] = min(kw["sdf_local_block_num"], 1 << 21)  # the oracle keeps the voxels in host RAM
    kw["hash_bucket_num"] = min(kw["hash_bucket_num"], 1 << 22)
    kw["excess_list_size"] = min(kw["excess_list_size"], 1 << 20)
    frames = make_frames(W, H, a.frames)
    sc = StreetScene(W, H)
    fx, fy, cx, cy = sc.intrinsics()
    e = OracleEngine(oracle_settings(**kw), make_calib(fx, fy, cx, cy, W, H), threads=os.cpu_count() or 1)
    for i, (rgba, d, T, _) in enumerate(frames):
        t0 = time.time()
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        if i < a.frames - 1:
            e.process_frame()
        else:
            e.allocate_scene_from_depth()  # the visible list k_integrate of this frame would sweep
        print(f"frame {i}: {time.time() - t0:.1f} s", flush=True)
    depth = e.get_view()[1].astype(np.float64)  # metres, <= 0 invalid
    M, _ = e.get_pose()
    M = M.astype(np.float64)
    table = e.dump_hash_table()
    vis = e.dump_visible_list()
    rng = np.random.default_rng(0)
    ids = rng.choice(vis, size=min(a.sample, len(vis)), replace=False)
    pos = table["pos"][ids].astype(np.float64)  # block coordinates
    vs, mu = kw["voxel_size"], kw["mu"]
    # ---- truth: does any voxel of the block pass the update condition?
    g = np.arange(8)
    off = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)[:, ::-1]  # (512, 3) x fastest
    any_update = np.zeros(len(ids), bool)
    n_updated = np.zeros(len(ids), np.int32)
    for s0 in range(0, len(ids), 2000):
        p = (pos[s0:s0 + 2000, None, :] * 8 + off[None]) * vs  # metres
        pc = p @ M[:3, :3].T + M[:3, 3]
        z = pc[..., 2]
        ok = z > 0
        u = fx * pc[..., 0] / np.where(ok, z, 1) + cx
        v = fy * pc[..., 1] / np.where(ok, z, 1) + cy
        ok &= (u >= 1) & (u <= W - 2) & (v >= 1) & (v <= H - 2)
        ui, vi = np.clip((u + 0.5).astype(np.int64), 0, W - 1), np.clip((v + 0.5).astype(np.int64), 0, H - 1)
        dm = depth[vi, ui]
        ok &= dm > 0
        ok &= (dm - z) >= -mu
        any_update[s0:s0 + 2000] = ok.any(axis=1)
        n_updated[s0:s0 + 2000] = ok.sum(axis=1)
    # ---- the conservative block test
    corners = np.stack(np.meshgrid([0, 1], [0, 1], [0, 1], indexing="ij"), -1).reshape(-1, 3)
    pcn = ((pos[:, None, :] + corners[None]) * 8 * vs) @ M[:3, :3].T + M[:3, 3]
    zc = pcn[..., 2]
    zmin = zc.min(axis=1)
    front = zmin > 1e-3
    uc = fx * pcn[..., 0] / np.where(zc > 1e-3, zc, 1e-3) + cx
    vc = fy * pcn[..., 1] / np.where(zc > 1e-3, zc, 1e-3) + cy
    u0, u1 = np.floor(uc.min(axis=1)) - 1, np.ceil(uc.max(axis=1)) + 1
    v0, v1 = np.floor(vc.min(axis=1)) - 1, np.ceil(vc.max(axis=1)) + 1
    outside = front & ((u1 < 1) | (u0 > W - 2) | (v1 < 1) | (v0 > H - 2))
    # max of the valid depths in the box through a summed... plain loops over the sample are fine here
    skip_behind = np.zeros(len(ids), bool)
    skip_invalid = np.zeros(len(ids), bool)
    dv = np.where(depth > 0, depth, -np.inf)
    for k in range(len(ids)):
        if not front[k] or outside[k]:
            continue
        x0, x1 = int(max(0, u0[k])), int(min(W - 1, u1[k]))
        y0, y1 = int(max(0, v0[k])), int(min(H - 1, v1[k]))
        m = dv[y0:y1 + 1, x0:x1 + 1].max()
        if m == -np.inf:
            skip_invalid[k] = True
        elif m < zmin[k] - mu - 1e-4:
            skip_behind[k] = True
    skip = outside | skip_behind | skip_invalid
    none = ~any_update
    print(f"visible blocks {len(vis)}, sampled {len(ids)}; voxels updated per sampled block: mean {n_updated.mean():.0f} of 512")
    print(f"blocks without a single update: {none.mean():.3f}")
    print(f"  found by the conservative test: {(skip & none).sum() / max(1, none.sum()):.3f} of them "
          f"(outside {(outside & none).mean():.3f}, no depth {(skip_invalid & none).mean():.3f}, behind the surface {(skip_behind & none).mean():.3f} of all blocks)")
    print(f"  test says skip but a voxel updates (must be 0): {(skip & any_update).sum()}")
    print(f"blocks with < 64 updated voxels: {(n_updated < 64).mean():.3f}")
    e.close()


if __name__ == "__main__":
    main()
