"""bench.py's host-side logic (no GPU): the pieces of the JSON line the driver records must not depend on luck —
PMC traffic lookup for any --steps, the roofline object, the CPU baseline legs, the frame generator."""
import argparse
import json
import os

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    d = dict(width=1242, height=375, decay=False, swap=False, instances=0, preset="5mm", volumes=0)
    d.update(kw)
    return argparse.Namespace(**d)


def test_pmc_traffic_is_available_for_the_default_workload_and_any_step_count():
    for v in (600000.0, 617000.0):  # visible blocks per launch differ with --steps / --warmup
        traffic, src = bench.pmc_traffic(_args(), "k_integrate", v)
        assert traffic and src and src.startswith("profiles/r") and src.endswith("_bench5mm_pmc_traffic.json")
        per_block = json.load(open(os.path.join(ROOT, src)))["kernels"]["k_integrate"]["hbm_bytes_per_visible_block"]
        assert abs(traffic - per_block * v) < 1.0
    # the newest committed set is the one that is used
    import glob
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench5mm_pmc_traffic.json")))[-1]
    assert bench.pmc_traffic(_args(), "k_integrate", 1.0)[1] == os.path.relpath(newest, ROOT)
    # other workloads have no committed PMC set: null, not a wrong number
    for other in (_args(decay=True), _args(swap=True), _args(instances=4), _args(width=640), _args(volumes=8), _args(preset="4mm")):
        assert bench.pmc_traffic(other, "k_integrate", 1e5) == (None, None)


def test_roofline_object_from_an_engine_profile():
    V, launches, avg_ms = 617000.0, 20, 0.608
    prof = [dict(name="integrate", total_ms=avg_ms * launches, launches=launches, bytes=5.07e9 * launches,
                 bytes_layout=1.684e9 * launches, units=V * launches),
            dict(name="raycast", total_ms=0.476 * launches, launches=launches, bytes=0.0, bytes_layout=0.0, units=0.0)]
    r, kernels = bench.roofline_from_profile(prof, _args(), 4700.0)
    assert r["bound"] == "hbm" and r["kernel"] == "k_integrate" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - 1.684e9 / 0.608e-3 / 1e9) < 1.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert 0 < r["frac"] <= 1 and r["traffic"] and 0 < r["traffic_frac"] < 1 and r["frac"] < r["traffic_frac"]
    assert abs(r["avg_launch_us"] - 608.0) < 0.1 and r["algorithmic_aos"]["GBps"] > r["achieved"]
    assert kernels["raycast"]["avg_us"] == 476.0 and kernels["raycast"]["GBps"] is None
    json.dumps(r)  # serialisable
    assert bench.roofline_from_profile([], _args(), None) == (None, {})


def test_frames_and_cpu_baseline_legs():
    frames = bench.make_frames(96, 32, 3, 2)
    assert len(frames) == 3 and frames[0][0].shape == (32, 96, 4) and frames[0][1].dtype == np.int16
    assert all(m[3].dtype == np.uint8 and m[4].shape == (4, 4) for f in frames for m in f[3])
    cpu = bench.cpu_baseline(frames, 96, 32, "5cm", 0.5)
    assert cpu["kind"] == "port" and cpu["unit"] == "frames/s" and cpu["value"] > 0 and cpu["single_thread_value"] > 0
    assert cpu["cores"] >= 1 and "sample" in cpu


def test_presets_match_the_baseline_configs():
    assert bench.PRESETS["5mm"]["voxel_size"] == 0.005 and bench.PRESETS["4mm"]["voxel_size"] == 0.004
    kw = bench.settings_kwargs("5mm")
    assert kw["mu"] == 0.02 and kw["max_w"] == 100 and kw["view_frustum_max"] == 30.0
