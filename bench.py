#!/usr/bin/env python3
"""bench.py — frames/s of TSDF integrate + raycast (BASELINE.json metric) on N MI355X.

A "step" is one pass of the hot path over one frame of the synthetic KITTI-like sequence
(dynslam_amd/synth.py): UpdateView (inputs already resident in HBM) -> SetPose ->
ProcessFrame (allocate + integrate) -> Prepare (expected depths + raycast + ICP maps).
At N = 1 the workload is BASELINE.json configs[1]: static map only, 1242x375, 5 mm voxels.

`python bench.py --gpus N` (N > 1, as the driver starts it — no launcher) forks its N ranks itself, one per GPU, RCCL process
group over 127.0.0.1; under `python -m torch.distributed.run` it uses the ranks it is given.  The N > 1 line measures what
north_star names: 8 CONCURRENT INSTANCE VOLUMES (0.035 m voxels, mu 1.0, 7142 blocks) — the SAME eight volumes for every N,
instance k on rank k mod N (N = 2: four per GPU, N = 8: one per GPU; strong scaling) — silhouette split, fusion (allocate +
integrate + raycast) and, inside the timed step, the one real exchange of the path: every rank raycasts its volumes from the
shared camera, the per-volume depth + colour layers are ALL-GATHERED over RCCL (one collective, issued by the library's own
dsr_exchange_*) and rank 0 z-composites them (dynslam_amd/multigpu.py ShardedScene).  value = volume-frames/s = 8*K /
max-rank time; the line also carries, measured on rank 0 after the timed region, the same 8 volumes TIME-SLICED ON ONE GPU
(`value_same_workload_1gpu`, north_star's denominator), their ratio (`speedup_vs_1gpu`) and the CPU oracle fusing + previewing
the same volumes on the host cores (`cpu_baseline`).  Nested under "configs3": the same measurement for BASELINE configs[3] — the
5 mm static map on rank 0 + 7 instance volumes on the other ranks.
`--instance-volumes V` / `--volumes V` run either leg alone with V volumes on any number of GPUs (N = 1: all time-sliced);
`--replicas` runs N independent copies of configs[1] (no collective).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement) with the
extra objects "roofline" (dominant kernel: integrate) and "cpu_baseline" (the CPU oracle on
a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCALING_VOLUMES = 8   # north_star: "8 concurrent instance volumes" — the SAME workload at 1 / 2 / 4 / 8 GPUs (strong scaling)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_GUIDE_COPY_GBS = 6290.0  # ... and the 6.29 TB/s its float4 copy kernel measures (79 %)

PRESETS = {
    # BASELINE.json "5mm voxels": upstream InfiniTAM indoor ratio mu = 4 * voxel
    "5mm": dict(voxel_size=0.005, mu=0.02, sdf_local_block_num=1 << 23, hash_bucket_num=1 << 23,
                excess_list_size=1 << 21),
    "4mm": dict(voxel_size=0.004, mu=0.016, sdf_local_block_num=1 << 24, hash_bucket_num=1 << 24,
                excess_list_size=1 << 22),
    # the reference's own experiments: 5 cm voxels
    "5cm": dict(voxel_size=0.05, mu=0.2, sdf_local_block_num=1 << 18, hash_bucket_num=1 << 20,
                excess_list_size=1 << 17),
    "3.5cm": dict(voxel_size=0.035, mu=0.14, sdf_local_block_num=1 << 19, hash_bucket_num=1 << 20,
                  excess_list_size=1 << 17),
}


def _gen_frame(args):
    from dynslam_amd.synth import StreetScene
    w, h, i, n_inst = args
    sc = StreetScene(w, h, n_instances=n_inst)
    rgba, d, T, inst_id = sc.frame(i)
    masks = []  # (instance, x0, y0, bbox-local uint8 mask, object->camera pose) like MNC detections
    for k in range(n_inst):
        ys, xs = np.nonzero(inst_id == k)
        if len(ys) < 64:
            continue
        y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
        rel = (np.linalg.inv(sc.instance_pose(k, i).astype(np.float64)) @ T.astype(np.float64)).astype(np.float32)
        masks.append((k, int(x0), int(y0), (inst_id[y0:y1, x0:x1] == k).astype(np.uint8), rel))
    return rgba, d, T, masks


def make_frames(w, h, n, n_inst=0):
    procs = min(n, max(1, (os.cpu_count() or 2) - 1), 16)
    if os.environ.get("DSR_BENCH_NO_POOL"):  # under rocprofv3 the forked pool sometimes never returns on this pool of boxes
        procs = 1
    if procs <= 1:
        return [_gen_frame((w, h, i, n_inst)) for i in range(n)]
    with Pool(procs) as pool:
        return pool.map(_gen_frame, [(w, h, i, n_inst) for i in range(n)])


def pmc_traffic(args, kernel, visible_blocks_per_launch):
    """HBM bytes per launch of `kernel` (k_integrate / k_raycast of the static map) from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 + WRITE_SIZE, separate passes, MI355X_MICROARCH.md corrections; summarised by the round's profiling run into
    profiles/).  PMC counters cannot be collected from inside this process.  The traffic of both kernels is proportional to the
    visible blocks they walk, so the profile's bytes PER VISIBLE BLOCK (measured over the profiled launches of the configs[1]
    workload) x this run's visible blocks per launch is reported — valid for any --steps / --warmup, and for the static map of
    the --instances / --volumes / --decay workloads (same preset, image size, kernel and kind of data; the line says so in
    `traffic_source`).  Other presets / image sizes / --swap have no committed set: (None, None), not a wrong number."""
    if args.width != 1242 or args.height != 375 or args.swap:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench{args.preset}_pmc_traffic.json")))
    for f in reversed(files):  # newest set that has the per-block figure
        try:
            ks = json.load(open(f))["kernels"]
            per_block = ks["k_integrate"].get("hbm_bytes_per_visible_block")
            if kernel != "k_integrate" and per_block:
                per_block = ks[kernel]["hbm_bytes"] / ks["k_integrate"]["visible_blocks_per_launch"]
            if per_block:
                src = os.path.relpath(f, ROOT)
                if args.decay or args.instances or getattr(args, "volumes", 0) > 1:
                    src += " (per-visible-block figure of the configs[1] static map)"
                return round(per_block * visible_blocks_per_launch, 0), src
        except Exception:
            continue
    return None, None


class _no_gc:
    """The timed region runs with Python's cyclic collector paused: with torch imported a full collection walks a million objects
    (40+ ms — longer than 35 steps of enqueueing) and where it lands is a matter of allocation counts: the no-flag 45-step run
    read 485 frames/s with ONE such pause inside its timed loop, 880 without (profiles/r04x_bench_default.json: step 4 enqueued
    after 43.75 ms).  Nothing is skipped: collection happens before the region and resumes after it."""
    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()


def settings_kwargs(preset):
    kw = dict(PRESETS[preset])
    kw.update(max_w=100, view_frustum_min=0.2, view_frustum_max=30.0)
    return kw


def _cpu_run(frames, w, h, preset, budget_s, threads):
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    kw = settings_kwargs(preset)
    # the oracle keeps the voxel array in host RAM: size it for the sample only
    kw["sdf_local_block_num"] = min(kw["sdf_local_block_num"], 1 << 20)
    kw["hash_bucket_num"] = min(kw["hash_bucket_num"], 1 << 22)
    kw["excess_list_size"] = min(kw["excess_list_size"], 1 << 20)
    sc = StreetScene(w, h)
    e = OracleEngine(oracle_settings(**kw), make_calib(*sc.intrinsics(), w, h), threads=threads)
    done, t_total = 0, 0.0
    for rgba, d, T, _ in frames:
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        t0 = time.perf_counter()
        try:
            e.process_frame()
        except Exception:
            break
        e.prepare()
        t_total += time.perf_counter() - t0
        done += 1
        if t_total > budget_s or done >= 12:  # the sample's voxel array holds ~12 frames at 5 mm
            break
    e.close()
    return done, t_total


def cpu_baseline(frames, w, h, preset, budget_s):
    """The CPU oracle (kind "port": our restatement of ITMSceneReconstructionEngine_CPU +
    ITMVisualisationEngine_CPU, whose loops are `#pragma omp parallel for` like upstream's; the
    reference's own engines are not in /root/reference) on the first frames of the same sequence:
    once on ALL host cores (`value`, `cores`), once on one thread, each bounded by budget_s."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    d1, t1 = _cpu_run(frames, w, h, preset, budget_s, 1)
    dn, tn = _cpu_run(frames, w, h, preset, budget_s, cores) if cores > 1 else (d1, t1)
    if d1 == 0 or dn == 0:
        return None
    return {"value": round(dn / tn, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "single_thread_value": round(d1 / t1, 4),
            "sample": f"first {dn} frames of the same sequence ({w}x{h}, preset {preset}), allocate+integrate+raycast, "
                      f"oracle/dsr_oracle.cpp with OpenMP on {cores} threads, {tn:.1f} s; single thread: first {d1} frames, {t1:.1f} s"}


def cpu_baseline_volumes(frames, w, h, n_volumes, has_static, preset, budget_s):
    """The CPU oracle on the multi-volume workload: the same silhouette split, fusion and fused preview of the same V volumes,
    one after the other on the host cores (what the reference's _CPU engines would do: InstanceReconstructor.cpp:315-361 is a
    sequential loop over the volumes), OpenMP inside each engine on all cores; bounded by budget_s."""
    import torch
    from dynslam_amd.engine import make_calib
    from dynslam_amd.multigpu import ShardedScene
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, load_api, oracle_settings
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kinds = volume_settings(preset)
    for kind in ("static", "view"):  # the oracle keeps the voxel array in host RAM: size the map for the sample only
        kinds[kind] = dict(kinds[kind], sdf_local_block_num=min(kinds[kind]["sdf_local_block_num"], 1 << 20),
                           hash_bucket_num=min(kinds[kind]["hash_bucket_num"], 1 << 22),
                           excess_list_size=min(kinds[kind]["excess_list_size"], 1 << 20))
    n_inst = n_volumes - 1 if has_static else n_volumes
    calib = make_calib(*StreetScene(w, h).intrinsics(), w, h)
    scene = ShardedScene(lambda kind: OracleEngine(oracle_settings(**kinds[kind]), calib, threads=cores), w, h, n_volumes, 1, 0,
                         torch.device("cpu"), local_only=True, has_static=has_static)
    scene.exchange.host_api = load_api()
    track_ids = {k: 1 + k for k in range(n_inst)}
    done, t_total = 0, 0.0
    for rgba, d, T, masks in frames:
        pose_m = np.linalg.inv(np.asarray(T, np.float64)).astype(np.float32)
        inst_m = {k: np.linalg.inv(np.asarray(rel, np.float64)).astype(np.float32) for k, _, _, _, rel in masks}
        t0 = time.perf_counter()
        try:
            scene.step(rgba, d, T, masks)
        except Exception:
            break
        scene.preview(pose_m, inst_m, track_ids)
        t_total += time.perf_counter() - t0
        done += 1
        if t_total > budget_s or done >= 12:
            break
    scene.close()
    if done == 0:
        return None
    return {"value": round(n_volumes * done / t_total, 4), "unit": "volume-frames/s", "cores": cores, "kind": "port",
            "composited_frames_per_s": round(done / t_total, 4),
            "sample": f"first {done} frames of the same sequence ({w}x{h}): silhouette split + fusion + fused preview of the same "
                      f"{n_volumes} volumes ({'static map + ' if has_static else ''}{n_inst} instance volumes) one after the other, "
                      f"oracle/dsr_oracle.cpp with OpenMP on {cores} threads, {t_total:.1f} s"}


def through_shim(args, with_instances):
    """SURVEY 8d "through-shim" rate: the C++ host shim/host_bench (our driver class over shim/ITMLib.h, the
    ITMLib names DynSLAM's InfiniTamDriver uses) fed with the SAME frames as pageable host buffers; per frame
    it pays what DynSLAM's host pays around the engine: BGR->RGBA conversion, the H2D copy of the frame,
    ProcessFrame, one status / noVisibleBlocks synchronisation, Prepare, the two previews written into the host's page-locked
    buffers.  PCIe inclusive — never `value`.
    Each leg is tools/bench_through_shim.py run as a process of its own (it generates the same frames, writes them to /dev/shm and
    starts the C++ host).  Round 4 saw the identical host_bench command read half the rate for configs[2] when THIS process
    started it (profiles/r04t_*); on round 5's boxes the parent made no difference, with the host thread pinned next to the GPU
    or not (profiles/r05c_where_does_the_host_run.log) — host_bench pins itself (dsr_pin_host_thread) and the leg stays a process
    of its own either way."""
    exe = os.path.join(ROOT, "shim", "host_bench")
    if not os.path.exists(exe):
        return None
    import ast
    import subprocess

    def leg(instances):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_through_shim.py"), "--preset", args.preset, "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--width", str(args.width), "--height", str(args.height)]
        if instances:
            cmd += ["--instances", str(instances)]
        lines = subprocess.check_output(cmd, timeout=900, stderr=subprocess.DEVNULL).decode().strip().splitlines()
        return ast.literal_eval(lines[-1])
    r = leg(0)
    out = {"frames_per_s": float(r["frames_per_s"]), "ms_per_frame": float(r["ms_per_frame"]), "host": "shim/host_bench.cpp (C++)",
           "note": "same frames and table sizes, handed over as pageable host BGR + int16 buffers through shim/ITMLib.h: the frame "
                   "upload (BGR -> RGBA in the ingest kernel), the allocation status and both previews per frame, as "
                   "InfiniTamDriver::UpdateView / Integrate / PrepareNextStep ask for them (PCIe inclusive)"}
    if with_instances:  # configs[2] through the reference's call pattern: the map + 4 instance drivers, every call of every driver
        try:
            r2 = leg(4)
            out["configs2"] = {"frames_per_s": float(r2["frames_per_s"]), "ms_per_frame": float(r2["ms_per_frame"]),
                               "note": "static map + 4 instance volumes (shim/host_bench --masks): GPU view split, per driver and frame one "
                                       "allocation status and two previews back to the host; at --steps 20 the instance volumes are "
                                       "still growing (45 steps read ~12 % more: profiles/r05zz_through_shim.log)"}
        except Exception as ex:
            out["configs2"] = {"frames_per_s": None, "note": f"failed: {ex}"}
    return out


def roofline_from_profile(prof, args, copy_gbs):
    """-> (roofline object of the dominant kernel k_integrate, per-kernel timing dict) from the engine's HIP-event
    profile (dsr_profile_get) of the timed region."""
    roofline = None
    kernels = {}
    for r in prof:
        kernels[r["name"]] = {"ms_total": round(r["total_ms"], 4), "launches": r["launches"],
                              "avg_us": round(1e3 * r["total_ms"] / max(1, r["launches"]), 2),
                              "GBps": round(r["bytes"] / (r["total_ms"] * 1e6), 1) if r["total_ms"] > 0 and r["bytes"] > 0 else None}
        if r["name"] in ("minmax_init", "expected_depth") and os.environ.get("DSR_OVERLAP_EXPECTED", "1") != "0":
            # the live view's range image runs on the engine's side stream UNDER k_integrate: its workgroups only find room as
            # integration workgroups retire, so the events bracket a SPAN about as long as the integration, not a cost
            kernels[r["name"]]["overlapped_with"] = "integrate (side stream): avg_us is a span, not a cost; stand-alone ~6 / ~41 us"
            kernels[r["name"]]["GBps"] = None
        if r["name"] == "integrate" and r["total_ms"] > 0:
            avg_s = r["total_ms"] * 1e-3 / r["launches"]
            v_per_launch = r["units"] / r["launches"]
            layout = r["bytes_layout"] / r["launches"]   # compulsory bytes of the layout in use (DESIGN.md byte model)
            aos = r["bytes"] / r["launches"]             # the reference's AoS formulation
            achieved = layout / avg_s / 1e9
            traffic, traffic_src = pmc_traffic(args, "k_integrate", v_per_launch)
            # north_star's bar is ">= 60 % of MEASURED HBM roofline".  The denominator is the larger of this box's copy probe
            # (dsr_measure_copy_bandwidth_spread: 4 GiB per direction, warm clocks, best launch of each of >= 5 rounds; max /
            # median / min over the rounds reported) and the guide's float4-copy figure: a probe that reads low on a cold or
            # shared box must not flatter the fraction (five driver runs read 4.6-6.2 TB/s from a 1 GiB probe).
            probe = copy_gbs if isinstance(copy_gbs, dict) else ({"max": copy_gbs, "median": copy_gbs, "min": copy_gbs} if copy_gbs else None)
            denom = max(probe["max"], HBM_GUIDE_COPY_GBS) if probe else None
            f_meas = round(achieved / denom, 4) if denom else None
            t_meas = round(traffic / avg_s / 1e9 / denom, 4) if (traffic and denom) else None
            roofline = {"bound": "hbm", "kernel": "k_integrate", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "frac_of_measured_copy": f_meas,
                        # the bar is priced on the COMPULSORY bytes of the layout; the counter traffic includes what the kernel
                        # re-reads or over-fetches (1.3x): a diagnostic of waste, never credited as achievement
                        "target_60pct_of_measured": {"denominator_GBps": denom, "on_compulsory_bytes": f_meas,
                                                     "met": (f_meas >= 0.6) if f_meas is not None else None,
                                                     "traffic_incl_waste": t_meas},
                        "traffic": traffic,
                        # the same launch priced with the bytes it really moved (PMC, profiles/)
                        "traffic_GBps": round(traffic / avg_s / 1e9, 1) if traffic else None,
                        "traffic_frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                        "traffic_source": traffic_src,
                        # the same bytes against what a copy kernel reaches: the guide's figure and this library's own probe
                        # on THIS box (best of 3 grids x plain / non-temporal, dsr_measure_copy_bandwidth)
                        "guide_copy_GBps": HBM_GUIDE_COPY_GBS, "frac_of_guide_copy": round(achieved / HBM_GUIDE_COPY_GBS, 4),
                        "traffic_frac_of_guide_copy": round(traffic / avg_s / 1e9 / HBM_GUIDE_COPY_GBS, 4) if traffic else None,
                        "measured_copy_GBps": probe["max"] if probe else None,
                        "measured_copy_spread_GBps": probe,
                        "avg_launch_us": round(1e6 * avg_s, 2),
                        "bytes_per_launch": round(layout, 0),
                        "visible_blocks_per_launch": round(v_per_launch, 1),
                        "algorithmic_aos": {"bytes_per_launch": round(aos, 0), "GBps": round(aos / avg_s / 1e9, 1),
                                            "note": "SURVEY 8d: V*(16+2*4096)+8P — every voxel of every visible block read "
                                                    "and written as an 8 B struct; NOT what this layout moves, kept for reference"},
                        "note": "achieved = layout-true compulsory bytes / HIP-event duration: per visible block 4 B list id "
                                "+ 16 B hash entry + 1536 B sdf and w_depth planes read, 24 B written per lane that updated a "
                                "voxel, 8 B per colour voxel (one word read + written), 8 B per pixel of the frames (tallied by the kernel itself); "
                                "traffic = rocprofv3 PMC bytes per visible block (profiles/) x this run's visible blocks"}
    ray = next((r for r in prof if r["name"] == "raycast" and r["total_ms"] > 0), None)
    if roofline and ray:
        # second kernel of the frame: compulsory bytes of THIS layout = per visible block its hash entry
        # (16 B) + its sdf plane (1024 B: the raycast never touches weights or colour) + 16 B per pixel written + the range
        # image; R (blocks some ray really enters) <= V, so this is an upper bound of the compulsory traffic
        avg_s = ray["total_ms"] * 1e-3 / ray["launches"]
        V = roofline["visible_blocks_per_launch"]
        P = args.width * args.height
        comp = V * (16.0 + 1024.0) + 16.0 * P + 8.0 * ((args.width + 7) // 8) * ((args.height + 7) // 8)
        traffic, src = pmc_traffic(args, "k_raycast", V)
        roofline["raycast"] = {"bound": "hbm", "kernel": "k_raycast", "achieved": round(comp / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(comp / avg_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "traffic_GBps": round(traffic / avg_s / 1e9, 1) if traffic else None,
                               "traffic_frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                               "traffic_source": src, "avg_launch_us": round(1e6 * avg_s, 2), "bytes_per_launch": round(comp, 0),
                               "note": "compulsory = V*(16 + 1024) + 16*P + range image; the kernel is bound by its per-wave chain of "
                                       "dependent gathers (latency), not by bandwidth: DESIGN.md"}
        kernels["raycast"]["GBps"] = roofline["raycast"]["achieved"]
        denom = roofline["target_60pct_of_measured"]["denominator_GBps"]
        roofline["raycast"]["frac_of_measured_copy"] = round(roofline["raycast"]["achieved"] / denom, 4) if denom else None
        roofline["raycast_frac"] = roofline["raycast"]["frac"]  # first-class next to `frac`
    return roofline, kernels


class _stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the driver reads stdout for ONE JSON line.
    File-descriptor level (the banner comes from C code): fd 1 points at stderr inside the block."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def _test_backend():
    """TEST SEAM.  The product path is fixed: HIP engines (libdsr_hip.so) on cuda:<LOCAL_RANK>, RCCL.  tests/ may name a module
    in DSR_BENCH_TEST_BACKEND that stands in for the device layer — `device(local_rank)`, `DIST_BACKEND`,
    `engine_factory(kinds, calib, local_rank)` and `host_api()` — so that THIS file's command line, rank spawning, collectives
    and JSON line run in the CPU suite (tests/bench_backend_oracle.py: the CPU oracle over gloo).  bench.py itself never imports
    anything under oracle/ outside the cpu_baseline leg, and without the variable a missing GPU / HIP library is an error."""
    name = os.environ.get("DSR_BENCH_TEST_BACKEND")
    if not name:
        return None
    import importlib
    mod = importlib.import_module(name)
    where = os.path.realpath(getattr(mod, "__file__", "") or "")
    if not where.startswith(os.path.realpath(os.path.join(ROOT, "tests")) + os.sep):
        raise RuntimeError(f"DSR_BENCH_TEST_BACKEND={name}: the seam only loads modules under tests/ ({where} is not)")
    return mod


def backend_name(tb):
    """What produced the line: "hip" = the product path; a line made through the test seam says so and names the module."""
    return "hip" if tb is None else f"test-seam:{tb.__name__}"


_PREGENERATED = {}  # n_instances -> frames, filled by the parent of spawn_ranks() before it forks
_JOB_STORE = None  # this rank's handle of the job's rendezvous store when spawn_ranks() started it (rank 0 hosts the store)


def frames_for(args, n_inst):
    """The synthetic frames of a leg: generated once per job (the forked ranks inherit the parent's copy)."""
    if n_inst not in _PREGENERATED:
        _PREGENERATED[n_inst] = make_frames(args.width, args.height, args.warmup + args.steps, n_inst)
    return _PREGENERATED[n_inst]


def _n_inst_of_leg(V, has_static):
    """Moving boxes in the frames of a leg: none for map volumes (whole frames), V - 1 next to a static map, else V."""
    return 0 if has_static == "maps" else (V - 1 if has_static else V)


def multi_gpu_legs(args, world):
    """-> [(n_volumes, has_static)] of a multi-volume job: the headline leg first.
    --instance-volumes V: north_star's scaling workload, V concurrent instance volumes;  --volumes V: configs[3], the static map
    + V-1 instance volumes;  `--gpus N` alone: BOTH with 8 volumes for EVERY N — 8 instance volumes (headline; the same workload at
    1 / 2 / 4 / 8 GPUs) and configs[3] = the static map + 7 instance volumes (nested)."""
    legs = []
    if args.instance_volumes:
        legs.append((args.instance_volumes, False))
    if args.volumes:
        legs.append((args.volumes, True))
    if args.map_volumes:
        legs.append((args.map_volumes, "maps"))
    if not legs and world > 1 and not args.replicas:
        # ... and the case sharding by volume is made for: one MAP-sized volume per GPU ("maps": N volumes at N GPUs)
        legs = [(SCALING_VOLUMES, False)] + ([] if args.no_configs3 else [(SCALING_VOLUMES, True), (world, "maps")])
    return legs


def spawn_ranks(args, legs):
    """`python bench.py --gpus N` exactly as the driver starts it (no torchrun): generate the inputs ONCE, then fork one rank per
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the child's environment, as torch.distributed.run would set them).  The
    parent has not touched HIP or torch, so the fork is safe; the children inherit the frames copy-on-write.  Rank 0 prints the
    line; the parent's exit status is the worst of its ranks and a failing rank takes the others down (no hang on a collective)."""
    import signal
    N = args.gpus
    if args.replicas:
        frames_for(args, args.instances)
    for V, has_static in legs:
        frames_for(args, _n_inst_of_leg(V, has_static))
    # The rendezvous port is chosen by rank 0's OWN store (bound to port 0, i.e. by the kernel) and handed to the parent through
    # a pipe before the other ranks are forked: a port picked here and closed again could be taken by another process before
    # rank 0 binds it.
    sys.stdout.flush()
    pids = {}
    port = None
    rd, wr = os.pipe()
    for r in range(N):
        if r == 1:
            os.close(wr)
            data = os.read(rd, 64)
            os.close(rd)
            if not data:  # rank 0 died before its store was up
                break
            port = int(data.decode())
        pid = os.fork()
        if pid == 0:
            rc = 1
            try:
                import faulthandler
                faulthandler.register(signal.SIGUSR1, all_threads=True)  # `kill -USR1 <rank pid>`: where is it stuck?
                os.environ.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1")
                import datetime
                from torch.distributed import TCPStore
                global _JOB_STORE  # every rank hands ITS handle of the one store to init_process_group (no env:// rendezvous)
                if r == 0:
                    os.close(rd)
                    _JOB_STORE = TCPStore("127.0.0.1", 0, N, is_master=True, wait_for_workers=False, timeout=datetime.timedelta(seconds=600))
                    os.environ["MASTER_PORT"] = str(_JOB_STORE.port)
                    os.write(wr, str(_JOB_STORE.port).encode())
                    os.close(wr)
                else:
                    os.environ["MASTER_PORT"] = str(port)
                    _JOB_STORE = TCPStore("127.0.0.1", port, N, is_master=False, timeout=datetime.timedelta(seconds=600))
                run_rank(args)
                rc = 0
            except BaseException:  # noqa: BLE001 - the child must never return into the parent's stack
                import traceback
                traceback.print_exc()
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(rc)
        pids[pid] = r
    worst = 0
    while pids:
        pid, status = os.wait()
        if pid not in pids:
            continue
        r = pids.pop(pid)
        rc = os.waitstatus_to_exitcode(status)
        if rc != 0:
            worst = worst or (rc if rc > 0 else 1)
            print(f"bench.py: rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr)
            for other in list(pids):
                try:
                    os.kill(other, signal.SIGTERM)
                except ProcessLookupError:
                    pass
    return worst


def main_volumes(args, legs):
    """One rank of a multi-volume job (see multi_gpu_legs): the headline leg's line with the other leg nested under "configs3"."""
    W, H = args.width, args.height
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    frame_sets = [frames_for(args, _n_inst_of_leg(V, has_static)) for V, has_static in legs]  # before HIP / RCCL start (fork)

    import torch
    tb = _test_backend()
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun the process group (RCCL) is used for one rank too
    if tb is None:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
        assert torch.cuda.device_count() > local_rank, f"rank {rank}: no GPU {local_rank} on this node"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = tb.device(local_rank)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        store_kw = dict(store=_JOB_STORE, rank=rank, world_size=world) if _JOB_STORE is not None else {}
        with _stdout_to_stderr():
            if tb is None:
                dist.init_process_group("nccl", device_id=dev, **store_kw)
            else:
                dist.init_process_group(tb.DIST_BACKEND, **store_kw)
            dist.barrier()  # the first collective creates the communicator (and prints RCCL's banner)

    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    calib = make_calib(*StreetScene(W, H).intrinsics(), W, H)
    kinds = volume_settings(args.preset)
    if tb is None:
        from dynslam_amd.engine import EngineCore, default_settings

        def make_engine(kind):
            return EngineCore(default_settings(**kinds[kind], device=local_rank, sync_status=0), calib)
        host_api = None
    else:
        make_engine = tb.engine_factory(kinds, calib, local_rank)
        host_api = tb.host_api()

    out = None
    for (V, has_static), frames in zip(legs, frame_sets):
        a = argparse.Namespace(**vars(args))
        a.volumes = V
        line = run_volumes(a, frames, make_engine, dev, world, rank, use_dist, host_api=host_api, has_static=has_static,
                           backend=backend_name(tb))
        if rank == 0:
            if out is None:
                out = line
            else:  # the second leg rides along under its own key: ONE line per job
                out["map_volumes" if has_static == "maps" else "configs3" if has_static else "instance_volumes"] = {k: line[k] for k in (
                    "value", "unit", "ms_per_step", "scaling", "config", "value_same_workload_1gpu", "speedup_vs_1gpu", "time_sliced_1gpu",
                    "cpu_baseline", "kernels")}
    bad = None
    if rank == 0:
        if world > 1:  # a --gpus N line that does not show N GPUs at work must not pass for a scaling point
            bad = check_multi_gpu_line(out, world)
            out["multi_gpu_check"] = {"ok": not bad, "problems": bad}
        print(json.dumps(out), flush=True)
        if bad:
            print("bench.py: this line is NOT a valid multi-GPU measurement: " + "; ".join(bad), file=sys.stderr, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


def check_multi_gpu_line(line, n):
    """-> list of reasons why `line` (rank 0's line of a `--gpus n` job, n > 1) is not a measurement on n GPUs; empty when it is.
    With the product backend the library's own RCCL communicator must span n ranks and the collective must have taken time on the
    exchange's stream; every backend must report the same workload on one GPU and the ratio, and one rank per GPU with work."""
    bad = []
    cfg = line.get("config") or {}
    if line.get("n_gpus") != n:
        bad.append(f"n_gpus is {line.get('n_gpus')}, not {n}")
    per_rank = cfg.get("volumes_per_rank") or []
    if len(per_rank) != n or sum(per_rank) != cfg.get("volumes") or min(per_rank or [0]) < 1:
        bad.append(f"volumes_per_rank {per_rank} does not put every one of {cfg.get('volumes')} volumes on one of {n} working ranks")
    if not (line.get("value_same_workload_1gpu") and line.get("speedup_vs_1gpu")):
        bad.append("no value_same_workload_1gpu / speedup_vs_1gpu: the same workload was not measured on one GPU")
    if line.get("backend") == "hip":
        if cfg.get("rccl_ranks") != n:
            bad.append(f"rccl_ranks is {cfg.get('rccl_ranks')}: the library's RCCL communicator does not span {n} GPUs")
        if not (cfg.get("gather_us") or 0) > 0:
            bad.append("gather_us is 0: no collective ran on the exchange's stream")
    if cfg.get("status") not in (0, None):
        bad.append(f"engine status {cfg.get('status')}")
    return bad


def volume_settings(preset):
    """Engine settings of the three kinds of engine a rank of a multi-volume job may hold."""
    kw = settings_kwargs(preset)
    inst_kw = dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                   sdf_local_block_num=7142, hash_bucket_num=0x100000, excess_list_size=0x20000)
    view_kw = dict(kw, sdf_local_block_num=64, hash_bucket_num=64, excess_list_size=64)
    # a map-sized volume of the map_volumes leg: the preset's voxels and table, 2^21 blocks (8 GiB) — the 25 frames of the bench
    # allocate 1.84 M — so that the leg's 1-GPU denominator (N such volumes time-sliced on one GPU) fits at N = 8
    map_kw = dict(kw, sdf_local_block_num=min(kw["sdf_local_block_num"], 1 << 21))
    return {"static": kw, "instance": inst_kw, "view": view_kw, "map": map_kw}


def run_volumes(args, frames, make_engine, dev, world, rank, use_dist, host_api=None, has_static=True, backend="hip"):
    """The timed part of a multi-volume job, after the process group and the device are set up: returns the bench line
    (a dict) on rank 0, None elsewhere.  `dev` is this rank's torch device; with a CPU device (tests/test_bench_contract.py
    drives this function over gloo with the CPU oracle as `make_engine`) the frames are handed over as host arrays and
    `host_api` composites; on a GPU everything — frames AND silhouette masks — is resident in HBM before the timed region."""
    import torch
    import torch.distributed as dist
    from dynslam_amd.multigpu import ShardedScene, volumes_of_rank
    V = args.volumes
    maps = has_static == "maps"
    has_static = bool(has_static) and not maps
    n_inst = V - 1 if has_static else V
    W, H, K, Wm = args.width, args.height, args.steps, args.warmup
    on_gpu = dev.type == "cuda"
    masks_in = [f[3] for f in frames]
    if on_gpu:
        rgb_in = [torch.from_numpy(f[0]).to(dev) for f in frames]
        dep_in = [torch.from_numpy(f[1]).to(dev) for f in frames]
        mask_dev = [[torch.from_numpy(np.ascontiguousarray(m)).to(dev) for _, _, _, m, _ in f[3]] for f in frames]
        masks_in = [[(k, x0, y0, (t.data_ptr(), m.shape[1], m.shape[0]), rel) for (k, x0, y0, m, rel), t in zip(f[3], md)]
                    for f, md in zip(frames, mask_dev)]
        torch.cuda.synchronize()
    track_ids = {k: 1 + k for k in range(n_inst)}
    pose_m = [np.linalg.inv(np.asarray(f[2], np.float64)).astype(np.float32) for f in frames]
    inst_m = [{k: np.linalg.inv(np.asarray(rel, np.float64)).astype(np.float32) for k, _, _, _, rel in f[3]} for f in frames]
    if maps:  # every volume is previewed from the frame's camera
        inst_m = [{k: m for k in range(n_inst)} for m in pose_m]
        masks_in = [[] for _ in frames]
    if on_gpu:
        # poses are input data like the frames: converted to the C ABI's float[16] once, not per call inside the timed loop (a numpy
        # transpose + a ctypes view cost CPython ~10 us each — 16 of them per step of 8 volumes; a C++ host pays nothing for this)
        from dynslam_amd.engine import PoseArg
        pose_m = [PoseArg(m) for m in pose_m]
        inst_m = [{k: PoseArg(m) for k, m in d.items()} for d in inst_m]
        masks_in = [[(k, x0, y0, mk, PoseArg(rel)) for k, x0, y0, mk, rel in f] for f in masks_in]

    def run(scene, nranks, preview=True):
        def step(i):
            if on_gpu:
                scene.step(rgb_in[i].data_ptr(), dep_in[i].data_ptr(), frames[i][2], masks_in[i])
            else:
                scene.step(frames[i][0], frames[i][1], frames[i][2], masks_in[i])
            if preview:
                scene.preview(pose_m[i], inst_m[i], track_ids)
            if getattr(args, "sync_every_step", False):
                scene.sync()

        def barrier():
            scene.sync()
            if on_gpu:
                torch.cuda.synchronize()
            if use_dist and nranks == world:
                dist.barrier()
            scene.sync()
            if on_gpu:
                torch.cuda.synchronize()
        for i in range(Wm):
            step(i)
        if getattr(scene, "after_warmup", None):
            scene.after_warmup()
        barrier()
        with _no_gc():
            t0 = time.perf_counter()
            for i in range(Wm, Wm + K):
                step(i)
            t_enq = time.perf_counter() - t0  # host time to enqueue the K steps (nothing has been waited for yet)
            barrier()
            return time.perf_counter() - t0, t_enq

    scene = ShardedScene(make_engine, W, H, V, world, rank, dev, has_static=has_static, maps=maps)
    scene.exchange.host_api = host_api
    prof = []
    probe = scene.static if scene.owns_static else (next(iter(scene.instances.values())) if (scene.instances and rank == 0) else None)
    profiled = probe is not None and not args.no_profile and on_gpu
    if profiled:  # HIP events around integrate + raycast of rank 0's map (or first instance volume)
        probe.profile_enable(2)

        def _reset():
            probe.sync()
            probe.profile_reset()
        scene.after_warmup = _reset
    native = bool(getattr(scene, "native", False))
    if native:  # HIP events around the collective and the composite on the exchange's stream (dsr_exchange_timing)
        prev_after = getattr(scene, "after_warmup", None)

        def _after():
            if prev_after:
                prev_after()
            scene.exchange.x.timing(True)
        scene.after_warmup = _after
    elapsed, t_enq = run(scene, world)
    xt = scene.exchange.x.timing(False) if native else None
    if profiled:
        prof = probe.profile_get()
        probe.profile_enable(False)
    # (the state the line reports is the one the timed region left: read before the diagnostic legs below fuse the frames again)
    stats = scene.static.get_stats() if scene.owns_static else None
    inst_stats = [e.get_stats() for e in scene.instances.values()]
    hit = float((scene.target_depth > 0).float().mean().item()) if rank == 0 else 0.0
    # where a step goes (for the day the N > 1 curve is measured): this rank's fusion chain alone (the same K steps without the
    # preview: no render, no collective, no composite), and — N > 1 — the same workload with the OTHER collective (gather to the
    # consumer's GPU instead of the in-place all-gather)
    scene.after_warmup = None
    # (from EMPTY volumes, like the timed region: frames that allocate cost more than the same frames fused a second time — the
    #  allocation mark's atomics, the merge of the new entries — and a replay would under-report the chain)
    scene.reset()
    chain_s, _ = run(scene, world, preview=False)
    alt = None
    if native and world > 1 and use_dist:
        scene.exchange.x.set_collective(1, 0)
        scene.reset()
        scene.after_warmup = lambda: scene.exchange.x.timing(True)
        alt_s, _ = run(scene, world)
        alt_t = scene.exchange.x.timing(False)
        scene.exchange.x.set_collective(0, 0)
        alt = (alt_s, alt_t)
    scene.close()
    gather_us = 1e3 * xt["gather_ms"] / xt["n_gathers"] if xt and xt["n_gathers"] else 0.0
    composite_us = 1e3 * xt["composite_ms"] / xt["n_composites"] if xt and xt["n_composites"] else 0.0
    alt_elapsed = alt[0] if alt else 0.0
    alt_gather_us = 1e3 * alt[1]["gather_ms"] / alt[1]["n_gathers"] if alt and alt[1]["n_gathers"] else 0.0
    if world > 1:
        t = torch.tensor([elapsed, chain_s, gather_us, composite_us, alt_elapsed, alt_gather_us], dtype=torch.float64,
                         device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, chain_s, gather_us, composite_us, alt_elapsed, alt_gather_us = [float(v) for v in t.tolist()]

    sliced = None
    if rank == 0 and world > 1 and not args.no_time_sliced:  # north_star's denominator: the same V volumes on ONE GPU
        one = ShardedScene(make_engine, W, H, V, 1, 0, dev, local_only=True, has_static=has_static, maps=maps)
        one.exchange.host_api = host_api
        t1, _ = run(one, 1)
        one.close()
        sliced = {"value": round(V * K / t1, 3), "unit": "volume-frames/s", "composited_frames_per_s": round(K / t1, 3),
                  "ms_per_step": round(1e3 * t1 / K, 4),
                  "note": f"the same {V} volumes fused + previewed sequentially on rank 0's GPU, same frames"}
    if rank != 0:
        return None
    cpu = None
    if not getattr(args, "no_cpu_baseline", False) and not maps:  # rank 0's host cores, after the timed region (the other ranks wait at the barrier)
        try:
            cpu = cpu_baseline_volumes(frames, W, H, V, has_static, args.preset, getattr(args, "cpu_budget_s", 12.0))
        except Exception as ex:  # the baseline must never take the bench line down
            cpu = {"value": None, "unit": "volume-frames/s", "cores": 1, "kind": "port", "sample": f"failed: {ex}"}
    roofline, kernels = roofline_from_profile(prof, args, None)  # rank 0's probe engine
    per_rank = [len(volumes_of_rank(r, V, world, has_static)) for r in range(world)]
    if maps:
        what = (f"map volumes: {V} MAP-sized volumes (preset {args.preset} voxels and table, 2^21 blocks each) sharded by volume over "
                f"{world} GPU(s) (volume k on rank k mod N), each fusing the whole frame (configs[1]'s step)")
    elif has_static:
        what = (f"configs[3]: static map (preset {args.preset}) + {n_inst} instance volumes (0.035 m, mu 1.0, 7142 blocks) "
                f"sharded by volume over {world} GPU(s) (the map on rank 0, instance k on rank 1 + k mod (N-1))")
    else:
        what = (f"north_star scaling workload: {n_inst} concurrent instance volumes (0.035 m, mu 1.0, 7142 blocks; "
                f"InstanceReconstructor.cpp:372-379) sharded by volume over {world} GPU(s) (instance k on rank k mod N), no static map")
    volume_rate = world > 1 or not has_static
    value = round(V * K / elapsed, 3) if volume_rate else round(K / elapsed, 3)
    same_1gpu = (sliced["value"] if volume_rate else sliced["composited_frames_per_s"]) if sliced else (value if world == 1 else None)
    return {
        "metric": "frames/sec TSDF integrate+raycast (KITTI 1242x375, 5mm voxels); HBM GB/s vs peak",
        "value": value,
        "unit": "volume-frames/s" if volume_rate else "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        # the SAME V volumes at every N (the driver's 1 / 2 / 4 / 8 curve is one workload): total work fixed = strong scaling
        "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak" if maps else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "backend": backend,
        # north_star: "aggregate at N GPUs over 8 concurrent instance volumes" against the same volumes on ONE GPU
        "value_same_workload_1gpu": same_1gpu,
        "speedup_vs_1gpu": round(value / same_1gpu, 3) if same_1gpu else None,
        "config": {"workload": f"{what}, synthetic KITTI-like street {W}x{H} with {n_inst} moving boxes, frames {Wm}..{Wm + K - 1}; "
                               f"every step = silhouette split (masks resident in HBM) + fusion (allocate, integrate, raycast) of "
                               f"every volume + fused preview: colour and depth raycast of every volume from the frame's camera, "
                               f"ONE RCCL all-gather of the {n_inst} instance layers ({n_inst * W * H * 8 / 1e6:.1f} MB; dsr_exchange_*), z-composite on rank 0",
                   "volumes": V, "volumes_per_rank": per_rank, "has_static_map": bool(has_static),
                   "composited_frames_per_s": round(K / elapsed, 3),
                   "host_enqueue_ms_per_step_rank0": round(1e3 * t_enq / K, 4),
                   # max over ranks: a rank's fusion chain alone (no preview), the collective and the composite on the exchange's
                   # stream (HIP events), and how many GPUs the library's RCCL communicator spans (0: no collective ran)
                   "chain_us_max_rank": round(1e6 * chain_s / K, 1), "gather_us": round(gather_us, 1),
                   "composite_us": round(composite_us, 1), "rccl_ranks": world if (native and use_dist) else 0,
                   "collective": "all-gather (in place)",
                   "gather_to_root": ({"value": round(V * K / alt_elapsed, 3), "ms_per_step": round(1e3 * alt_elapsed / K, 4),
                                       "gather_us": round(alt_gather_us, 1)} if alt_elapsed > 0 else None),
                   "preview_hit_fraction": round(hit, 4),
                   "static_visible_blocks_last_frame": stats.no_visible_blocks if stats else None,
                   "instance_visible_blocks_last_frame_rank0": [s_.no_visible_blocks for s_ in inst_stats],
                   "status": max([stats.sticky_status if stats else 0] + [s_.sticky_status for s_ in inst_stats])},
        "time_sliced_1gpu": sliced,
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
    }


CFG4_PERIOD, CFG4_FRAMES, CFG4_MIN_AGE = 50, 320, 200


def loop_frames_for(args):
    """configs[4]'s lap: 50 frames = the 40 m period of the synthetic street (tools/bench_cfg5_sustained.py)."""
    if "loop" not in _PREGENERATED:
        _PREGENERATED["loop"] = make_frames(args.width, args.height, CFG4_PERIOD)
    return _PREGENERATED["loop"]


def leg_configs2(args, dev, local_rank, calib):
    """BASELINE configs[2], engine-only: the 5 mm map + 4 instance volumes on this GPU — view split, fusion and tracking render of
    every volume per frame (the per-volume calls of the reference's loop, InstanceReconstructor.cpp:315-361), inputs resident in HBM."""
    import torch
    from dynslam_amd.engine import EngineCore, default_settings
    frames = frames_for(args, 4)
    K, Wm = args.steps, args.warmup
    kinds = volume_settings(args.preset)
    eng = EngineCore(default_settings(**kinds["static"], device=local_rank, sync_status=0), calib)
    inst = [EngineCore(default_settings(**kinds["instance"], device=local_rank, sync_status=0), calib) for _ in range(4)]
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]

    def step(i):
        eng.update_view_dev(rgb[i].data_ptr(), dep[i].data_ptr())
        for k, x0, y0, mask, rel in frames[i][3]:
            eng.extract_silhouette(inst[k], mask, x0, y0)
            eng.remove_silhouette(mask, x0, y0)
            inst[k].set_pose_inv_m(rel)
            inst[k].process_frame()
            inst[k].prepare()
        eng.set_pose_inv_m(frames[i][2])
        eng.process_frame()
        eng.prepare()

    def drain():
        for e in inst:
            e.sync()
        eng.sync()
        torch.cuda.synchronize()
    for i in range(Wm):
        step(i)
    drain()
    with _no_gc():
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step(i)
        drain()
        elapsed = time.perf_counter() - t0
    st = eng.get_stats()
    ist = [e.get_stats() for e in inst]
    for e in inst:
        e.close()
    eng.close()
    return {"value": round(K / elapsed, 3), "unit": "frames/s", "ms_per_step": round(1e3 * elapsed / K, 4), "steps": K, "warmup": Wm,
            "config": {"workload": f"configs[2]: static map (preset {args.preset}) + 4 instance volumes (0.035 m, mu 1.0, 7142 blocks) on ONE "
                                   f"GPU, engine-only (inputs resident in HBM, masks as host buffers through the pinned ring), frames {Wm}..{Wm + K - 1}",
                       "static_visible_blocks_last_frame": st.no_visible_blocks,
                       "instance_visible_blocks_last_frame": [s_.no_visible_blocks for s_ in ist],
                       "engine_status": max([st.sticky_status] + [s_.sticky_status for s_ in ist])}}


def leg_configs3_1gpu(args, dev, local_rank, calib):
    """BASELINE configs[3] on ONE GPU (time-sliced): the 5 mm map + 7 instance volumes + the fused preview per step."""
    from dynslam_amd.engine import EngineCore, default_settings
    kinds = volume_settings(args.preset)
    a = argparse.Namespace(**vars(args))
    a.volumes, a.no_cpu_baseline, a.no_profile = SCALING_VOLUMES, True, True
    line = run_volumes(a, frames_for(args, SCALING_VOLUMES - 1),
                       lambda kind: EngineCore(default_settings(**kinds[kind], device=local_rank, sync_status=0), calib),
                       dev, 1, 0, False, has_static=True)
    return {k: line[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config")}


def leg_configs4_short(args, dev, local_rank, calib):
    """A SHORT configs[4] leg: 4 mm voxels, voxel GC (max_weight 1, min_age 200: DynSLAMGUI.cpp:36-42) + host swapping, 320 frames
    of the 50-frame lap — the GC's FIFO is full and freeing blocks for the last 120 — with the structural invariants of the map
    checked at the end (dynslam_amd/invariants.py).  The 4541-frame run is tools/bench_cfg5_sustained.py (profiles/)."""
    import torch
    from dynslam_amd.engine import EngineCore, default_settings
    from dynslam_amd.invariants import check_structure
    frames = loop_frames_for(args)
    kw = dict(settings_kwargs("4mm"), use_swapping=1)
    rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep = [torch.from_numpy(f[1]).to(dev) for f in frames]
    eng = EngineCore(default_settings(**kw, device=local_rank, sync_status=0), calib)
    n_warm = 20

    def step(i):
        j = i % CFG4_PERIOD
        eng.update_view_dev(rgb[j].data_ptr(), dep[j].data_ptr())
        eng.set_pose_inv_m(frames[j][2])
        eng.process_frame()
        eng.prepare()
        eng.decay(1, CFG4_MIN_AGE, False)
    for i in range(n_warm):
        step(i)
    eng.sync()
    with _no_gc():
        t0 = time.perf_counter()
        for i in range(n_warm, CFG4_FRAMES):
            step(i)
        eng.sync()
        elapsed = time.perf_counter() - t0
    st = eng.get_stats()
    t1 = time.perf_counter()
    invariants = "ok"
    try:
        check_structure(eng, kw["sdf_local_block_num"], kw["hash_bucket_num"])
    except AssertionError as ex:
        invariants = f"VIOLATED: {ex}"
    t_check = time.perf_counter() - t1
    eng.close()
    n = CFG4_FRAMES - n_warm
    return {"value": round(n / elapsed, 3), "unit": "frames/s", "ms_per_step": round(1e3 * elapsed / n, 4), "steps": n, "warmup": n_warm,
            "config": {"workload": f"configs[4], short: frames {n_warm}..{CFG4_FRAMES - 1} of laps of a {CFG4_PERIOD}-frame loop, preset 4mm "
                                   f"(voxel {kw['voxel_size']} m, mu {kw['mu']} m, 2^24 blocks), voxel GC max_weight 1 min_age {CFG4_MIN_AGE} every "
                                   f"frame + host swapping; step = UpdateView + ProcessFrame (+ swap in / out) + Prepare + Decay",
                       "allocated_blocks": kw["sdf_local_block_num"] - 1 - st.last_free_block_id, "visible_blocks_last_frame": st.no_visible_blocks,
                       "decayed_blocks": st.decayed_block_count, "host_store_slots": st.host_store_slots, "engine_status": st.sticky_status,
                       "structural_invariants": invariants, "invariants_check_s": round(t_check, 2)}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node to use.  N > 1 without a launcher (RANK unset): bench.py forks the N ranks itself")
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="5mm", choices=sorted(PRESETS))
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--cpu-budget-s", type=float, default=8.0, help="per CPU leg (all cores, one thread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-through-shim", action="store_true",
                    help="skip the C++-host leg (shim/host_bench: the same frames as pageable host buffers through the ITMLib shim)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--profile-all", action="store_true",
                    help="bracket every kernel with HIP events (default: integrate and raycast only; the events "
                         "of a full profile cost ~9 %% of the frame)")
    ap.add_argument("--decay", action="store_true", help="also run voxel GC each frame (min_age 200, max_weight 1)")
    ap.add_argument("--decay-min-age", type=int, default=200, help="min_age of --decay (DynSLAMGUI.cpp:38-42: 200)")
    ap.add_argument("--host-views", action="store_true",
                    help="hand every frame over as pageable HOST buffers (dsr_update_view: H2D copy + stream sync per "
                         "frame, as InfiniTamDriver::UpdateView does): the PCIe-inclusive rate quoted in DESIGN.md, never `value`")
    ap.add_argument("--swap", action="store_true", help="enable host swap-in/out (use_swapping; configs[4])")
    ap.add_argument("--instances", type=int, default=0,
                    help="configs[2]: also reconstruct this many moving instances in their own volumes "
                         "(voxel 0.035, mu 1.0, 7142 blocks: InstanceReconstructor.cpp:372-379), split on the GPU")
    ap.add_argument("--instance-volumes", type=int, default=0,
                    help="north_star's scaling workload: V concurrent INSTANCE volumes (0.035 m, mu 1.0, 7142 blocks), one per GPU "
                         "(instance k on rank k mod N), silhouette split + fusion + fused preview (all-gather + composite) per "
                         "step; reports aggregate volume-frames/s and the same V volumes time-sliced on one GPU "
                         "(the headline of --gpus N > 1, with V = N)")
    ap.add_argument("--volumes", type=int, default=0,
                    help="configs[3]: static map + (V-1) instance volumes sharded one per GPU with the fused-preview "
                         "all-gather + composite in the timed step (nested under \"configs3\" in the --gpus N > 1 line)")
    ap.add_argument("--map-volumes", type=int, default=0,
                    help="V MAP-sized volumes (2^21 blocks of the preset's voxels), volume k on rank k mod N, each fusing the whole frame + the fused preview")
    ap.add_argument("--no-configs3", action="store_true", help="--gpus N > 1: only the instance-volumes leg")
    ap.add_argument("--replicas", action="store_true", help="--gpus N: N independent configs[1] replicas, no collective")
    ap.add_argument("--no-time-sliced", action="store_true", help="skip the 1-GPU time-sliced leg of a multi-volume line")
    ap.add_argument("--sync-every-step", action="store_true",
                    help="diagnostic (multi-volume legs): the host waits for every step before it queues the next — a GUI host's "
                         "pattern; ms_per_step is then a step's LATENCY, not the rate of a full queue")
    ap.add_argument("--no-nested-legs", action="store_true", help="skip configs[2] / configs[3] on one GPU / the short configs[4] leg of the N = 1 line")
    ap.add_argument("--no-scaling-leg", action="store_true",
                    help="N = 1: do not append north_star's scaling workload (8 instance volumes on this GPU) to the line")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # the driver's `python bench.py --gpus N ...`: no launcher around us, so the ranks are ours to start
        sys.exit(spawn_ranks(args, multi_gpu_legs(args, args.gpus)))
    if "RANK" in os.environ and args.gpus != world_env and args.gpus != 1:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world_env} rank(s); using {world_env}", file=sys.stderr)
    run_rank(args)


def run_rank(args):
    """One rank of the job (the only one at N = 1)."""
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    legs = multi_gpu_legs(args, world_env)
    if legs:
        return main_volumes(args, legs)

    # The C++ host (SURVEY 8d "through-shim") first, as processes of their own, before this process has generated a frame or
    # opened the device (the state in which profiles/r04s_through_shim_via_tool.log was measured).
    shim = None
    if (int(os.environ.get("RANK", "0")) == 0 and world_env == 1 and not args.no_through_shim
            and not (args.decay or args.swap or args.instances or args.host_views)):
        try:
            shim = through_shim(args, with_instances=args.preset == "5mm")
        except Exception as ex:
            shim = {"frames_per_s": None, "note": f"failed: {ex}"}

    # synthetic frames: the worker pool forks, which must happen before HIP / RCCL start
    frames = frames_for(args, args.instances)
    # the N = 1 line also carries north_star's scaling workload (8 instance volumes) on this one GPU: the N = 1 point of the curve
    # whose N > 1 points are `python bench.py --gpus N`
    scaling_leg = (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_scaling_leg and args.preset == "5mm"
                   and not (args.decay or args.swap or args.instances or args.host_views))
    frames8 = frames_for(args, SCALING_VOLUMES) if scaling_leg else None
    # ... and BASELINE.json's other configurations as nested legs, each with a `status`, so that the driver's line — not only
    # profiles/ — observes them: configs[2] engine-only, configs[3] time-sliced on this one GPU, and a short
    # configs[4] leg (4 mm, voxel GC + host swapping, structural invariants checked)
    nested_legs = scaling_leg and not args.no_nested_legs
    if nested_legs:
        frames_for(args, 4); frames_for(args, SCALING_VOLUMES - 1); loop_frames_for(args)


    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        store_kw = dict(store=_JOB_STORE, rank=rank, world_size=world) if _JOB_STORE is not None else {}
        with _stdout_to_stderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **store_kw)
            dist.barrier()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene

    W, H, K, Wm = args.width, args.height, args.steps, args.warmup
    n_frames = Wm + K
    # inputs resident in HBM before the timed region
    rgb_dev = [torch.from_numpy(f[0]).to(dev) for f in frames]
    dep_dev = [torch.from_numpy(f[1]).to(dev) for f in frames]
    poses = [f[2] for f in frames]
    torch.cuda.synchronize()

    # measured device-to-device copy bandwidth of this GPU with the library's float4 grid-stride copy (MI355X_MICROARCH.md: 6.29 TB/s
    # for that kernel) — the roofline's "measured" denominator: 4 GiB per direction, clocks warm, best launch of each of 5 rounds
    # (dsr_measure_copy_bandwidth_spread); max / median / min over the rounds are reported
    copy_gbs = None
    if rank == 0:
        import ctypes as C
        from dynslam_amd.engine import load_hip_api
        g3 = (C.c_double * 3)()
        if load_hip_api().measure_copy_bandwidth_spread(local_rank, 4 << 30, 5, g3) == 0:
            copy_gbs = {"max": round(g3[0], 1), "median": round(g3[1], 1), "min": round(g3[2], 1)}

    sc = StreetScene(W, H)
    kw = settings_kwargs(args.preset)
    if args.swap:
        kw["use_swapping"] = 1
    eng = EngineCore(default_settings(**kw, device=local_rank, sync_status=0), make_calib(*sc.intrinsics(), W, H))
    inst_kw = dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                   sdf_local_block_num=7142, hash_bucket_num=0x100000, excess_list_size=0x20000)
    inst_eng = [EngineCore(default_settings(**inst_kw, device=local_rank, sync_status=0), make_calib(*sc.intrinsics(), W, H))
                for _ in range(args.instances)]

    def step(i):
        if args.host_views:
            eng.update_view(frames[i][0], frames[i][1])
        else:
            eng.update_view_dev(rgb_dev[i].data_ptr(), dep_dev[i].data_ptr())
        for k, x0, y0, mask, rel in frames[i][3]:
            # ProcessSilhouette + RemoveSilhouette on the GPU, then FuseFrame of the instance
            # (InstanceReconstructor.cpp:238-263,569-700)
            eng.extract_silhouette(inst_eng[k], mask, x0, y0)
            eng.remove_silhouette(mask, x0, y0)
            inst_eng[k].set_pose_inv_m(rel)
            inst_eng[k].process_frame()
            inst_eng[k].prepare()
        eng.set_pose_inv_m(poses[i])
        eng.process_frame()
        eng.prepare()
        if args.decay:
            eng.decay(1, args.decay_min_age, False)

    def barrier():
        for ie in inst_eng:
            ie.sync()
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        eng.sync()
        torch.cuda.synchronize()

    for i in range(Wm):
        step(i)
    eng.sync()
    if not args.no_profile:
        eng.profile_enable(1 if args.profile_all else 2)
        eng.profile_reset()
    barrier()
    enq = []  # DSR_BENCH_STEP_TIMES=1: when the host had enqueued each step (ms since t0) — a diagnostic, not part of the contract
    with _no_gc():
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step(i)
            enq.append(round(1e3 * (time.perf_counter() - t0), 3))
        barrier()
        elapsed = time.perf_counter() - t0
    prof = eng.profile_get() if not args.no_profile else []
    eng.profile_enable(False)
    stats = eng.get_stats()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = K * world
        roofline, kernels = roofline_from_profile(prof, args, copy_gbs)
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is an N = 1 item
            try:
                cpu = cpu_baseline(frames, W, H, args.preset, args.cpu_budget_s)
            except Exception as ex:  # the baseline must never take the bench line down
                cpu = {"value": None, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"failed: {ex}"}
        out = {
            "metric": "frames/sec TSDF integrate+raycast (KITTI 1242x375, 5mm voxels); HBM GB/s vs peak",
            "value": round(total_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "backend": "hip",
            "config": {"workload": f"{'configs[2]: static map + ' + str(args.instances) + ' instance volumes' if args.instances else 'configs[1]: static map only'}"
                                   f"{' + voxel GC' if args.decay else ''}{' + host swapping' if args.swap else ''}, synthetic KITTI-like street {W}x{H}, "
                                   f"preset {args.preset} (voxel {kw['voxel_size']} m, mu {kw['mu']} m), "
                                   f"frames {Wm}..{Wm + K - 1} of a {n_frames}-frame sequence, one volume per GPU"
                                   f"{', views handed over as pageable host buffers (PCIe inclusive)' if args.host_views else ''}",
                       "visible_blocks_last_frame": stats.no_visible_blocks,
                       "allocated_blocks": kw["sdf_local_block_num"] - 1 - stats.last_free_block_id,
                       "status": stats.sticky_status, "decay": bool(args.decay), "swap": bool(args.swap), "instances": args.instances},
            "roofline": roofline, "cpu_baseline": cpu, "through_shim": None, "kernels": kernels,
        }
        if os.environ.get("DSR_BENCH_STEP_TIMES"):
            out["step_enqueued_ms"] = enq
    for ie in inst_eng:
        ie.close()
    eng.close()
    if rank == 0:
        out["through_shim"] = shim  # measured before this process opened the device (above)
        if scaling_leg:
            try:
                calib = make_calib(*sc.intrinsics(), W, H)
                kinds = volume_settings(args.preset)
                a = argparse.Namespace(**vars(args))
                a.volumes, a.cpu_budget_s = SCALING_VOLUMES, args.cpu_budget_s / 2
                line = run_volumes(a, frames8, lambda kind: EngineCore(default_settings(**kinds[kind], device=local_rank, sync_status=0), calib),
                                   dev, 1, 0, False, has_static=False)
                out["instance_volumes8_1gpu"] = {k: line[k] for k in ("value", "unit", "ms_per_step", "scaling", "config", "cpu_baseline")}
            except Exception as ex:
                out["instance_volumes8_1gpu"] = {"value": None, "note": f"failed: {ex}"}
            c = (out.get("instance_volumes8_1gpu") or {}).get("config") or {}
            if roofline is not None and c.get("composite_us"):
                # third kernel with a roofline of its own: compulsory bytes of an L-layer composite = 4 B of depth per
                # layer and pixel + the target's depth and colour read and written (16 B) — the winners' colour reads are a few %
                L, P = SCALING_VOLUMES, W * H
                comp = P * (4.0 * L + 16.0)
                gbs = comp / (c["composite_us"] * 1e-6) / 1e9
                roofline["composite"] = {"bound": "hbm", "kernel": "k_composite", "layers": L, "bytes_per_launch": comp,
                                         "avg_launch_us": c["composite_us"], "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                         "note": "HIP events on the exchange's stream around k_composite<true> in the instance_volumes8_1gpu leg"}
                roofline["composite_frac"] = roofline["composite"]["frac"]
        if nested_legs:
            calib = make_calib(*sc.intrinsics(), W, H)
            for key, fn in (("configs2", lambda: leg_configs2(args, dev, local_rank, calib)),
                            ("configs3_1gpu", lambda: leg_configs3_1gpu(args, dev, local_rank, calib)),
                            ("configs4_short", lambda: leg_configs4_short(args, dev, local_rank, calib))):
                try:
                    out[key] = dict(fn(), status="ok")
                except Exception as ex:  # a nested leg must never take the line down; its status says what happened
                    out[key] = {"value": None, "status": f"failed: {type(ex).__name__}: {ex}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
