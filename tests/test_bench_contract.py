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
    # the static map of the multi-volume / GC workloads is the same kernel on the same kind of data: the per-block figure applies (and says so)
    for same_map in (_args(decay=True), _args(instances=4), _args(volumes=8)):
        t, src = bench.pmc_traffic(same_map, "k_integrate", 1e5)
        assert t and "configs[1] static map" in src
    # other image sizes / presets / host swapping have no committed PMC set: null, not a wrong number
    for other in (_args(swap=True), _args(width=640), _args(preset="4mm")):
        assert bench.pmc_traffic(other, "k_integrate", 1e5) == (None, None)
    # the second kernel of the frame has its figure too
    t, src = bench.pmc_traffic(_args(), "k_raycast", 617000.0)
    assert t and 0.8e9 < t < 2.5e9


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
    assert kernels["raycast"]["avg_us"] == 476.0
    rc = r["raycast"]  # a roofline object for the second kernel
    assert rc["kernel"] == "k_raycast" and 0 < rc["frac"] < rc["traffic_frac"] < 1 and abs(rc["avg_launch_us"] - 476.0) < 0.1
    assert abs(rc["bytes_per_launch"] - (V * 1040 + 16 * 1242 * 375 + 8 * 156 * 47)) < 1 and kernels["raycast"]["GBps"] == rc["achieved"]
    assert r["guide_copy_GBps"] == 6290.0 and abs(r["frac_of_guide_copy"] - r["achieved"] / 6290.0) < 1e-3
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


# ---------------------------------------------------------------------------------------------------------
# `bench.py --gpus N` (N > 1): the timed part of the configs[3] job — bench.run_volumes, what every rank runs after the
# process group is up — at world size 2 over gloo with the CPU oracle as the engine.  Checks the line the driver will
# record (value = V*K/t in volume-frames/s, max over ranks, the time-sliced one-GPU leg on rank 0 while the other
# ranks wait) and that no rank deadlocks on a collective the others do not enter.

def _volumes_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, load_api, oracle_settings
    W, H, V = 256, 80, world
    args = argparse.Namespace(volumes=V, width=W, height=H, steps=3, warmup=1, preset="5cm", no_profile=False, no_time_sliced=False,
                              decay=False, swap=False, instances=0, no_cpu_baseline=False, cpu_budget_s=1.0)
    frames = [bench._gen_frame((W, H, i, V - 1)) for i in range(args.steps + args.warmup)]
    calib = make_calib(*StreetScene(W, H, n_instances=V - 1).intrinsics(), W, H)
    kinds = bench.volume_settings(args.preset)
    line = bench.run_volumes(args, frames, lambda kind: OracleEngine(oracle_settings(**kinds[kind]), calib), torch.device("cpu"), world, rank,
                             True, host_api=load_api())
    assert (line is not None) == (rank == 0)
    if rank == 0:
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            json.dump(line, f)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 8])  # 8 = the node the driver's scaling run uses: static map + 7 instance volumes
def test_configs3_bench_line_at_world_size_n(tmp_path, world):
    import socket

    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_volumes_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    line = json.load(open(tmp_path / "line.json"))
    assert line["n_gpus"] == world and line["unit"] == "volume-frames/s" and line["scaling"] == "strong" and line["steps"] == 3 and line["warmup"] == 1
    assert line["metric"].startswith("frames/sec TSDF integrate+raycast") and line["higher_is_better"] is True and line["vs_baseline"] is None
    cfg = line["config"]
    assert cfg["volumes"] == world and cfg["volumes_per_rank"] == [1] * world and cfg["workload"].startswith("configs[3]")
    # value = whole-job volume-frames per second = V * K / (max-over-ranks time)
    assert abs(line["value"] - world * 3 / (line["ms_per_step"] * 3 / 1e3)) / line["value"] < 1e-3
    assert abs(cfg["composited_frames_per_s"] * world - line["value"]) / line["value"] < 1e-3
    assert cfg["preview_hit_fraction"] > 0.3 and cfg["status"] == 0 and cfg["static_visible_blocks_last_frame"] > 100
    ts = line["time_sliced_1gpu"]
    assert ts and ts["composited_frames_per_s"] > 0
    # the same workload on ONE GPU and the ratio are top-level keys of the line
    assert line["value_same_workload_1gpu"] == ts["value"] and abs(line["speedup_vs_1gpu"] - line["value"] / ts["value"]) < 2e-3
    assert line["roofline"] is None  # no HIP events on a CPU device
    cpu = line["cpu_baseline"]  # the N > 1 line carries the CPU baseline of ITS workload: the same volumes on the host cores
    assert cpu["kind"] == "port" and cpu["unit"] == "volume-frames/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and str(world) in cpu["sample"]
    assert line["backend"] == "hip"  # run_volumes' default: only bench._test_backend's seam stamps anything else


# ---------------------------------------------------------------------------------------------------------
# The REAL command line, as the driver starts it: `python bench.py --gpus N --steps K --warmup W` with no launcher around it.
# bench.py must fork its N ranks itself and print ONE rank-0 line with n_gpus = N (`--gpus` used to be ignored).
# The device layer is swapped for the CPU oracle over gloo through bench.py's test seam; everything else is the shipped file.

@pytest.mark.parametrize("n", [2, 3])
def test_cli_gpus_n_spawns_its_own_ranks(n):
    import subprocess
    import sys
    env = dict(os.environ, DSR_BENCH_TEST_BACKEND="tests.bench_backend_oracle", PYTHONPATH=ROOT, DSR_BENCH_NO_POOL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                        "--width", "256", "--height", "80", "--preset", "5cm"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE line for the whole job
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong"
    assert line["unit"] == "volume-frames/s" and line["metric"].startswith("frames/sec TSDF integrate+raycast")
    assert line["backend"] == "test-seam:tests.bench_backend_oracle"  # a line made through the seam says so
    cfg = line["config"]
    # the headline leg: north_star's 8 concurrent instance volumes — the SAME 8 for every N, instance k on rank k mod N
    assert cfg["workload"].startswith("north_star scaling workload") and cfg["has_static_map"] is False
    assert cfg["volumes"] == 8 and cfg["volumes_per_rank"] == {2: [4, 4], 3: [3, 3, 2]}[n] and cfg["status"] == 0
    assert abs(line["value"] - 8 * 2 / (line["ms_per_step"] * 2 / 1e3)) / line["value"] < 1e-3
    assert line["time_sliced_1gpu"]["value"] == line["value_same_workload_1gpu"] > 0 and line["speedup_vs_1gpu"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert cfg["preview_hit_fraction"] > 0.001  # the composited preview holds the instances' pixels
    # the second leg, nested: configs[3] — the static map on rank 0 + 7 instance volumes on the other ranks
    c3 = line["configs3"]
    assert c3["config"]["workload"].startswith("configs[3]") and c3["config"]["has_static_map"] is True
    assert c3["config"]["volumes_per_rank"] == {2: [1, 7], 3: [1, 4, 3]}[n] and c3["value"] > 0 and c3["time_sliced_1gpu"]["value"] > 0
    assert c3["config"]["static_visible_blocks_last_frame"] > 100 and c3["cpu_baseline"]["value"] > 0
    # the third leg: one MAP-sized volume per rank, each fusing the whole frame, the same fused preview
    mv = line["map_volumes"]
    assert mv["config"]["workload"].startswith("map volumes") and mv["config"]["volumes"] == n and mv["config"]["volumes_per_rank"] == [1] * n
    assert mv["scaling"] == "weak" and mv["value"] > 0 and mv["time_sliced_1gpu"]["value"] == mv["value_same_workload_1gpu"] > 0
    assert mv["speedup_vs_1gpu"] > 0 and mv["config"]["preview_hit_fraction"] > 0.3 and mv["config"]["status"] == 0
    assert all(v > 100 for v in mv["config"]["instance_visible_blocks_last_frame_rank0"])
    # ... and the line went through the multi-GPU check (the seam's transport is exempt from the RCCL items only)
    assert line["multi_gpu_check"] == {"ok": True, "problems": []}


def test_a_line_that_does_not_show_n_gpus_at_work_is_refused():
    """`bench.py --gpus N` exits non-zero — after printing the line with the reasons — when the product backend's line does not carry
    an N-rank RCCL communicator, a collective that took time, and the same workload measured on one GPU."""
    good = {"n_gpus": 4, "backend": "hip", "value": 40.0, "value_same_workload_1gpu": 10.0, "speedup_vs_1gpu": 4.0,
            "config": {"volumes": 8, "volumes_per_rank": [2, 2, 2, 2], "rccl_ranks": 4, "gather_us": 31.5, "status": 0}}
    assert bench.check_multi_gpu_line(good, 4) == []
    import copy
    for mutate, word in ((lambda d: d["config"].update(rccl_ranks=0), "rccl_ranks"),
                         (lambda d: d["config"].update(rccl_ranks=2), "rccl_ranks"),
                         (lambda d: d["config"].update(gather_us=0.0), "gather_us"),
                         (lambda d: d.update(speedup_vs_1gpu=None), "speedup_vs_1gpu"),
                         (lambda d: d.update(value_same_workload_1gpu=None), "speedup_vs_1gpu"),
                         (lambda d: d.update(n_gpus=1), "n_gpus"),
                         (lambda d: d["config"].update(volumes_per_rank=[8, 0, 0, 0]), "volumes_per_rank"),
                         (lambda d: d["config"].update(status=-3), "status")):
        bad = copy.deepcopy(good)
        mutate(bad)
        problems = bench.check_multi_gpu_line(bad, 4)
        assert problems and any(word in p for p in problems), (word, problems)
    # a line made through the test seam (gloo, the oracle): no RCCL to show, everything else still required
    seam = copy.deepcopy(good)
    seam["backend"] = "test-seam:tests.bench_backend_oracle"
    seam["config"].update(rccl_ranks=0, gather_us=0.0)
    assert bench.check_multi_gpu_line(seam, 4) == []


def test_seam_refuses_modules_outside_tests(monkeypatch):
    """DSR_BENCH_TEST_BACKEND may only name a module under tests/: the oracle itself, or anything else importable, is refused."""
    monkeypatch.setenv("DSR_BENCH_TEST_BACKEND", "oracle.oracle")
    with pytest.raises(RuntimeError, match="only loads modules under tests/"):
        bench._test_backend()
    monkeypatch.setenv("DSR_BENCH_TEST_BACKEND", "tests.bench_backend_oracle")
    assert bench.backend_name(bench._test_backend()) == "test-seam:tests.bench_backend_oracle"
    monkeypatch.delenv("DSR_BENCH_TEST_BACKEND")
    assert bench._test_backend() is None and bench.backend_name(None) == "hip"


def test_cli_rank_failure_does_not_hang():
    """A rank that dies (here: the test backend module cannot be imported in the children) takes the job down with a
    non-zero status instead of leaving the other ranks in a collective."""
    import subprocess
    import sys
    env = dict(os.environ, DSR_BENCH_TEST_BACKEND="tests.no_such_backend", PYTHONPATH=ROOT, DSR_BENCH_NO_POOL="1")
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--width", "96", "--height", "32", "--preset", "5cm"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no_such_backend" in p.stderr


@pytest.mark.gpu
def test_cli_gpus_2_with_hip_engines_on_one_gpu(hip_api):
    """`python bench.py --gpus 2` on the GPU box with the REAL engines: two forked ranks, each with its own HIP engines on
    cuda:0, masks in HBM, renders written into the exchange slots, the collective over gloo (RCCL refuses two ranks on one
    device; its own path runs at world size 1 under torchrun), the HIP composite on rank 0 — the N > 1 code path end to end."""
    import subprocess
    import sys
    env = dict(os.environ, DSR_BENCH_TEST_BACKEND="tests.bench_backend_hip_gloo", PYTHONPATH=ROOT, DSR_BENCH_NO_POOL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--width", "640", "--height", "192", "--preset", "5cm"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["unit"] == "volume-frames/s" and line["config"]["volumes_per_rank"] == [4, 4]
    assert line["backend"] == "test-seam:tests.bench_backend_hip_gloo"
    assert line["config"]["status"] == 0 and line["config"]["preview_hit_fraction"] > 0.001
    assert line["time_sliced_1gpu"]["value"] > 0 and line["configs3"]["config"]["static_visible_blocks_last_frame"] > 100
    assert line["configs3"]["config"]["preview_hit_fraction"] > 0.3


def test_through_shim_legs_are_the_stand_alone_tool(monkeypatch):
    """bench.py's through-shim legs are `tools/bench_through_shim.py` run as processes of their own (the same host
    command read half the rate for configs[2] when bench.py's own process started it): the command lines carry the job's sizes,
    the tool's last line is what the keys are made of, a failing configs[2] leg does not take the configs[1] figure down."""
    import subprocess
    calls = []

    def fake(cmd, **kw):
        calls.append(cmd)
        if "--instances" in cmd:
            if os.environ.get("FAKE_CFG2_FAILS"):
                raise subprocess.CalledProcessError(1, cmd)
            return b"noise\n{'driver': 'shim', 'frames_per_s': '458.312', 'ms_per_frame': '2.1819'}\n"
        return b"{'driver': 'shim', 'frames_per_s': '852.78', 'ms_per_frame': '1.1726'}\n"
    monkeypatch.setattr(subprocess, "check_output", fake)
    monkeypatch.setattr(os.path, "exists", lambda p: True)
    a = bench.parse_args(["--steps", "20", "--warmup", "5"])
    r = bench.through_shim(a, True)
    assert r["frames_per_s"] == 852.78 and r["configs2"]["frames_per_s"] == 458.312 and r["configs2"]["ms_per_frame"] == 2.1819
    assert all(c[1].endswith(os.path.join("tools", "bench_through_shim.py")) for c in calls) and len(calls) == 2
    assert calls[0][calls[0].index("--steps") + 1] == "20" and calls[0][calls[0].index("--warmup") + 1] == "5"
    assert "--instances" not in calls[0] and calls[1][calls[1].index("--instances") + 1] == "4"
    assert "configs2" not in bench.through_shim(a, False)
    monkeypatch.setenv("FAKE_CFG2_FAILS", "1")
    r = bench.through_shim(a, True)
    assert r["frames_per_s"] == 852.78 and r["configs2"]["frames_per_s"] is None and "failed" in r["configs2"]["note"]
    json.dumps(r)


def test_timed_regions_run_with_the_cyclic_collector_paused():
    """A full collection inside the timed loop cost the no-flag run 44 ms: the collector is paused for the region
    and back afterwards, also when the region raises."""
    import gc
    assert gc.isenabled()
    with bench._no_gc():
        assert not gc.isenabled()
    assert gc.isenabled()
    try:
        with bench._no_gc():
            raise RuntimeError("step failed")
    except RuntimeError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with bench._no_gc():
            pass
        assert not gc.isenabled()  # a caller that runs without the collector keeps running without it
    finally:
        gc.enable()
