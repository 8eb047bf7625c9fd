#!/usr/bin/env python3
"""Offline feasibility model of the "batched corridor pre-resolve" raycast, no GPU needed.

Idea under test: per 8x8-pixel tile (= one wave = one cell of the range image), before marching, the 64 lanes resolve the hash
lookups of every voxel block inside the tile's frustum slab [zmin, zmax] into an LDS table (block -> ptr / absent); the march then
probes LDS and only voxel reads go to memory.

Inputs (made with the CPU oracle on bench.py's workload, frame 8 of the 5 mm street):
  /tmp/raycast_steps.npy   per-ray step counts (tests/study/raycast_steps_dump.py)
  /tmp/minmax.npy          the range image of the same frame (OracleEngine.dump_render_state()["minmax"])

Model: a direct-indexed table, 4 B per block, one slice per block along the frustum's principal axis, a square window per slice
that holds the tile's footprint at that depth + 1 block for misalignment (+ 1 more with --pad 2: the march samples round(p) and
the 2x2x2 trilinear cell, and the frustum drifts sideways by up to one block per slice).
"""
import sys

import numpy as np

pad = 2 if "--pad" not in sys.argv else int(sys.argv[sys.argv.index("--pad") + 1])
mm = np.load("/tmp/minmax.npy")
a = np.load("/tmp/raycast_steps.npy")
miss, sat, band = a & 1023, (a >> 10) & 1023, (a >> 20) & 1023
steps = (miss + sat + band).astype(np.int64)
H, W = steps.shape
Hp, Wp = -(-H // 8) * 8, -(-W // 8) * 8


def per_tile(x, fn):
    s = np.zeros((Hp, Wp), np.float64)
    s[:H, :W] = x
    return fn(s.reshape(Hp // 8, 8, Wp // 8, 8), axis=(1, 3))


tmax = per_tile(steps, np.max)
zmin, zmax = mm[..., 0].astype(np.float64), mm[..., 1].astype(np.float64)
valid = zmax > zmin
block, f = 0.04, 707.09
depth = np.where(valid, zmax - zmin, 0)
foot = 8 * zmax / f / block
side = np.ceil(foot) + pad
corr = np.where(valid, (np.ceil(depth / block) + 1) * side ** 2, 0)
look = per_tile(miss + (sat + band) / 2.0, np.sum)  # a lookup per miss step + one per ~2 found steps (4 voxels per step)
tot = tmax.sum()
print(f"frame 8 of the bench workload: {valid.sum()} tiles with a range, slab depth median {np.median(depth[valid]):.2f} m / "
      f"p90 {np.percentile(depth[valid], 90):.2f} m, footprint median {np.median(foot[valid]):.1f} blocks / p90 {np.percentile(foot[valid], 90):.1f}")
print(f"corridor entries per tile (window {pad} blocks wider than the footprint): median {np.median(corr[valid]):.0f}, mean {corr[valid].mean():.0f}, "
      f"p90 {np.percentile(corr[valid], 90):.0f}")
print(f"hash lookups per tile TODAY (model): median {np.median(look[valid]):.0f}, mean {look[valid].mean():.0f}; "
      f"sum {look[valid].sum() / 1e6:.1f} M   vs   corridor entries to resolve: sum {corr[valid].sum() / 1e6:.1f} M")
for cap_kb in (5, 10, 20):
    cap = cap_kb * 1024 // 4
    fit = valid & (corr <= cap)
    waves_per_simd = min(8, int(160 / cap_kb) // 4)
    print(f"  LDS table {cap_kb:2d} KB per wave ({cap} entries, <= {waves_per_simd} waves/SIMD): {fit.sum() / valid.sum():.2f} of the tiles fit, "
          f"carrying {tmax[fit].sum() / tot:.2f} of the wave-iterations")
line = 64  # a bucket head is a scattered 16 B read: a 64 B sector at least
print(f"traffic of the pre-resolve alone at one {line} B sector per entry: {corr[valid].sum() * line / 1e9:.2f} GB "
      f"(the whole kernel fetches 1.33 GB today, profiles/r02f_bench5mm_pmc_traffic.json)")
