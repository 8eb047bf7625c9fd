#!/usr/bin/env python3
"""Offline model of k_raycast's lane divergence (no GPU needed): given the per-ray march step counts of a frame
(tests/study/raycast_steps_dump.py), how many WAVE-ITERATIONS does a mapping of rays to wavefronts cost?  A wave iterates as long
as its longest ray, so cost(mapping) = sum over waves of max(steps of its 64 rays); the kernel is bound by the per-wave
chain of dependent round trips with every wave resident, so launch time is roughly proportional to it.

Mappings compared:
  tile WxH         one wave per WxH pixel tile (the kernel uses 8x8 = one range-image cell)
  refill G,T       persistent waves over groups of G consecutive 8x8 tiles: a lane whose ray has ended takes the next
                   unprocessed pixel of the group as soon as fewer than T lanes are still marching (finished rays are
                   finalised and new rays set up for all idle lanes together); cost = march iterations + `setup` iterations
                   per refill round (the ray set-up and the hit refinement are ~2 march steps of instructions)
  sorted           lower bound: rays sorted by step count, 64 per wave (not realisable: needs the counts in advance)

usage: python tools/raycast_divergence_model.py /tmp/raycast_steps.npy
"""
import sys

import numpy as np


def unpack(a):
    miss, sat, band = a & 1023, (a >> 10) & 1023, (a >> 20) & 1023
    return miss + sat + band, miss, sat, band


def cost_tiles(steps, tw, th):
    H, W = steps.shape
    Hp, Wp = -(-H // th) * th, -(-W // tw) * tw
    s = np.zeros((Hp, Wp), steps.dtype)
    s[:H, :W] = steps
    t = s.reshape(Hp // th, th, Wp // tw, tw).max(axis=(1, 3))
    return int(t.sum())


def tile_order(steps, tw=8, th=8):
    """rays in the order a persistent wave would fetch them: tile after tile (row-major tiles), row-major inside a tile."""
    H, W = steps.shape
    Hp, Wp = -(-H // th) * th, -(-W // tw) * tw
    s = np.full((Hp, Wp), -1, np.int64)
    s[:H, :W] = steps
    return s.reshape(Hp // th, th, Wp // tw, tw).transpose(0, 2, 1, 3).reshape(-1, tw * th)


def cost_refill(steps, group, threshold, setup=2):
    tiles = tile_order(steps)
    total = 0
    for g0 in range(0, len(tiles), group):
        q = tiles[g0:g0 + group].reshape(-1)
        q = q[q >= 0]
        lanes = np.zeros(64, np.int64)  # remaining steps per lane (0 = idle)
        nxt = 0
        while True:
            active = lanes > 0
            na = int(active.sum())
            if (na < threshold or na == 0) and nxt < len(q):
                idle = np.nonzero(~active)[0]
                take = min(len(idle), len(q) - nxt)
                lanes[idle[:take]] = np.maximum(q[nxt:nxt + take], 1)  # a ray costs at least its set-up
                nxt += take
                total += setup
                continue
            if na == 0:
                break
            # march until the next event: either the lane count drops below the threshold (refill possible) or all end
            rem = np.sort(lanes[active])
            if nxt < len(q) and na >= threshold:
                k = rem[na - threshold]  # after k more steps fewer than `threshold` lanes remain
            else:
                k = rem[-1]
            lanes[active] -= k
            lanes = np.maximum(lanes, 0)
            total += int(k)
    return total


def main():
    packed = np.load(sys.argv[1])
    steps, miss, sat, band = unpack(packed)
    n = steps.size
    print(f"rays {n}, mean steps {steps.mean():.2f} (miss {miss.mean():.2f}, saturated {sat.mean():.2f}, band {band.mean():.2f}), "
          f"max {steps.max()}, rays > 150 steps: {(steps > 150).mean() * 100:.1f} %")
    ideal = steps.sum() / 64.0
    base = cost_tiles(steps, 8, 8)
    print(f"ideal (sum / 64)          {ideal:12.0f}  1.00")
    srt = np.sort(steps.reshape(-1))[::-1]
    pad = (-len(srt)) % 64
    srt = np.concatenate([srt, np.zeros(pad, srt.dtype)])
    print(f"sorted (lower bound)      {srt.reshape(-1, 64).max(axis=1).sum():12.0f}  {srt.reshape(-1, 64).max(axis=1).sum() / ideal:.2f}")
    for tw, th in ((8, 8), (16, 4), (4, 16), (32, 2), (64, 1), (2, 32)):
        c = cost_tiles(steps, tw, th)
        print(f"tile {tw:2d}x{th:<2d}               {c:12.0f}  {c / ideal:.2f}  ({c / base:.2f} of 8x8)")
    for group in (4, 16, 64):
        for thr in (16, 32, 48, 56):
            c = cost_refill(steps, group, thr)
            print(f"refill G={group:<3d} T={thr:<2d}        {c:12.0f}  {c / ideal:.2f}  ({c / base:.2f} of 8x8)")


if __name__ == "__main__":
    main()
