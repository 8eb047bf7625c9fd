#!/usr/bin/env python3
"""Known-byte-count workload for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half of a wide coalesced read).  Runs the
engine's own k_reset_vba (writes exactly noBlocks*4096 B, reads nothing) and torch device
copies of 1 GiB (reads 1 GiB + writes 1 GiB each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynslam_amd.engine import EngineCore, default_settings, make_calib

e = EngineCore(default_settings(voxel_size=0.05, mu=0.2, sdf_local_block_num=1 << 18, hash_bucket_num=1 << 16,
                                excess_list_size=1 << 12), make_calib(100, 100, 50, 50, 128, 96))
e.reset_scene(); e.sync()                      # k_reset_vba: 2^18 * 4096 B = 1 GiB written
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
b = torch.empty_like(a)
a.fill_(1.0); torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)                                  # 1 GiB read + 1 GiB written
torch.cuda.synchronize()
print("calibration workload done")
