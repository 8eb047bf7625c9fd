"""The reference's COMPLETE per-frame pipeline, from its own unmodified sources, running on the engine library.

tests/refhost/ref_dynslam_host.cpp builds what BuildDynSlamKittiOdometry builds (DynSLAMGUI.cpp:1153-1268) —
Input + PrecomputedDepthProvider, PrecomputedSegmentationProvider, VisoSparseSFProvider (over a scripted libviso2
stand-in), Evaluation, InfiniTamDriver, DynSlam — and calls DynSlam::ProcessFrame on a synthetic dataset written
in the reference's on-disk layout (tests/refhost/make_dataset.py).  Every DynSLAM class in the run is compiled
from /root/reference; the engine underneath is reached only through shim/ITMLib.h -> include/dsr.h.

CPU (here): the executable is built against the CPU ORACLE (dsr_* renamed to orc_* at compile time, test-only)
and its output is checked against the synthetic ground truth:
  * the tracker classifies the three moving boxes + the parked car as designed (uncertain / dynamic / dynamic / static);
  * the static map holds the street WITHOUT the moving cars (cut out by ProcessSilhouette / RemoveSilhouette through
    the view's UpdateHostFromDevice / UpdateDeviceFromHost round trip, InstanceReconstructor.cpp:180-197);
  * every instance volume (SetView + SetPose + Integrate of a per-track InfiniTamDriver, :569-700), raycast from the
    pose of its last fused frame, reproduces the depth that was cut out for it;
  * CompositeInstances changes the preview where the dynamic objects are; OBJ meshes and the memory CSV are written.
GPU: the same host linked against libdsr_hip.so produces the same digests, bit for bit, as the oracle-backed one.
"""
import os
import subprocess

import numpy as np
import pytest

from tests.refhost import build_pipeline as bp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "refhost", "_build")
N_FRAMES = 8
W, H = 1242, 375


def run_host(exe, root, out_bin, decay=0, evaluate=0, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS=str(min(16, os.cpu_count() or 1)), **(extra_env or {}))
    r = subprocess.run([exe, root, str(N_FRAMES), out_bin, "0.05", str(decay), str(evaluate)], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = r.stdout.strip().splitlines()[-1]
    kv = dict(tok.split("=", 1) for tok in line.split())
    return kv, r.stdout


def read_dump(path, n_objects):
    b = open(path, "rb").read()
    P = W * H
    off = [0]

    def take(dt, n, shape):
        a = np.frombuffer(b, dt, n, off[0])
        off[0] += a.nbytes
        return a.reshape(shape)
    d = {"colour_static": take(np.uint8, P * 4, (H, W, 4)), "depth_static": take(np.float32, P, (H, W)),
         "colour_fused": take(np.uint8, P * 4, (H, W, 4)), "depth_fused": take(np.float32, P, (H, W)),
         "view_rgb": take(np.uint8, P * 3, (H, W, 3)), "view_depth_mm": take(np.int16, P, (H, W)), "objects": []}
    for _ in range(n_objects):
        d["objects"].append({"raycast": take(np.uint8, P * 4, (H, W, 4)), "raycast_depth": take(np.float32, P, (H, W)),
                             "view_depth": take(np.float32, P, (H, W))})
    assert off[0] == len(b)
    return d


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from tests.refhost.make_dataset import write_dataset
    root = str(tmp_path_factory.mktemp("kitti_like"))
    write_dataset(root, N_FRAMES, W, H)
    return root


@pytest.fixture(scope="module")
def oracle_host(tmp_path_factory):
    if not bp.have_reference():
        pre = os.path.join(BUILD, "ref_dynslam_host_orc")
        if os.path.exists(pre):
            return pre
        pytest.skip("/root/reference is not on this machine and no prebuilt host")
    work = str(tmp_path_factory.mktemp("build_orc"))
    return bp.build("oracle", os.path.join(work, "ref_dynslam_host_orc"), work)


@pytest.fixture(scope="module")
def oracle_run(dataset, oracle_host, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("out") / "orc.bin")
    kv, log = run_host(oracle_host, dataset, out)
    recon = [int(k[5:]) for k in kv if k.startswith("track") and k[5:].isdigit() and ":recon1:" in kv[k]]
    return kv, log, read_dump(out, len(recon)), sorted(recon), np.load(os.path.join(dataset, "truth.npz"))


def test_tracker_classifies_the_objects_and_starts_their_reconstructions(oracle_run):
    kv, log, _, recon, _ = oracle_run
    assert kv["frames"] == str(N_FRAMES) and kv["tracks"] == "4"
    state = {i: kv[f"track{i}"].split(":")[0] for i in range(4)}
    # detection order per frame: boxes at 0.30 / 0.70 / 1.00 m per frame, then the parked car (Track.h:96-98 thresholds)
    assert state == {0: "Uncertain", 1: "Dynamic", 2: "Dynamic", 3: "Static"}
    assert recon == [1, 2, 3]  # the uncertain one is cut away and never fused (InstanceReconstructor.cpp:238-246)
    for i in recon:
        assert int(kv[f"track{i}"].split(":used")[1].split(":")[0]) > 100  # blocks in its own volume
    # an expiring track takes its engine with it; the views the host keeps are detached, not left with a dead handle
    assert kv["pruned_track"] == "3" and kv["pruned_view_detached"] == "1" and kv["pruned_has_reconstruction"] == "0"
    assert "Unknown motion for possibly dynamic object of class car; cutting away!" in log
    assert "Reaping track with max weight" in log  # ForceDynamicObjectCleanup -> Track::ReapReconstruction -> Decay(force)
    # GetUsedMemoryBytes = (allocated - lastFreeBlockId) blocks: one more than the blocks in use (InfiniTamDriver.h:241-244)
    assert int(kv["static_used_blocks"]) > 2000 and int(kv["static_memory_bytes"]) == (int(kv["static_used_blocks"]) + 1) * 4096


def test_static_map_holds_the_street_without_the_moving_cars(oracle_run):
    _, _, d, _, truth = oracle_run
    ds, zs, inst = d["depth_static"], truth["static_z_last"], truth["inst_last"]
    dm = truth["depth_mm"][-1].astype(np.float32) / 1000.0
    ok = (ds > 0) & (zs > 0) & (zs < 20)
    err = np.abs(ds - zs)[ok]
    assert ok.sum() > 0.4 * W * H
    assert np.median(err) < 0.04 and (err < 0.15).mean() > 0.93  # 5 cm voxels, 0.25 px disparity noise
    for k in range(3):  # nothing of a moving box — uncertain or dynamic — was fused into the static map
        m = inst == k
        assert m.sum() > 1000
        shows_car = ((np.abs(ds - dm) < 0.3) & (ds > 0) & m).sum() / m.sum()
        assert shows_car < 0.01, (k, shows_car)
        # ... because it was removed from the view the static map fuses (depth 0 inside the mask)
        assert (d["view_depth_mm"][m] == 0).mean() > 0.99
        assert (d["view_rgb"][m] == 0).all(axis=1).mean() > 0.99
    assert (d["view_depth_mm"][inst < 0] > 0).mean() > 0.5  # the rest of the view is intact


def test_instance_volumes_reproduce_the_views_cut_out_for_them(oracle_run):
    _, _, d, recon, _ = oracle_run
    for slot, track in enumerate(recon):
        o = d["objects"][slot]
        both = (o["raycast_depth"] > 0) & (o["view_depth"] > 0)
        assert both.sum() > 0.9 * (o["view_depth"] > 0).sum() > 3000, track
        err = np.abs(o["raycast_depth"] - o["view_depth"])[both]
        # 3.5 cm voxels; track 2 is ~16 m away where 0.25 px of disparity noise is ~0.17 m in the single view
        assert np.median(err) < (0.15 if track == 2 else 0.06), (track, np.median(err))
        assert (o["raycast"][..., :3].sum(axis=2) > 0)[both].mean() > 0.95  # the colour render covers the same pixels


def test_composite_preview_meshes_and_memory_log(oracle_run, dataset):
    kv, _, d, recon, truth = oracle_run
    inst = truth["inst_last"]
    changed = (d["depth_fused"] != d["depth_static"])
    for k in (1, 2):  # the dynamic objects are composited over the static raycast (positions: the reference's own pose logic)
        assert changed[inst == k].mean() > 0.5
        assert (d["colour_fused"][inst == k] != d["colour_static"][inst == k]).any(axis=1).mean() > 0.5
    far = ~changed
    # background dimmed by 10 % (InstanceReconstructor.cpp:945-954)
    cs, cf = d["colour_static"][far][:, :3].astype(np.int32), d["colour_fused"][far][:, :3].astype(np.int32)
    assert np.array_equal(cf, (cs * (1.0 - float(np.float32(0.10)))).astype(np.int32))  # `1.0 - dim_factor` with a float 0.10f
    # every preview type of the GUI renders something (kDepth through GetImage has no RGBA output: stays empty)
    for name in ("gray", "normal", "weight", "latest_raycast"):
        assert int(kv[f"preview_{name}"].split(":lit")[1]) > 0.3 * W * H, name
    assert len({kv[f"preview_{n}"].split(":")[0] for n in ("gray", "normal", "weight", "latest_raycast")}) == 4
    mesh_dir = os.path.join(dataset, "mesh_out", "synthetic")
    objs = [os.path.join(dp, f) for dp, _, fs in os.walk(mesh_dir) for f in fs if f.endswith(".obj")]
    names = sorted(os.path.basename(p) for p in objs)
    assert names == [f"instance-precomputed-dispnet-{i:06d}-mesh.obj" for i in recon] + [f"static-precomputed-dispnet-mesh-{N_FRAMES:06d}-frames.obj"]
    for p in objs:
        with open(p) as f:
            head = [next(f) for _ in range(4)]
        assert head[0].startswith("v ") and os.path.getsize(p) > 100000
    csvs = [f for f in os.listdir(os.path.join(dataset, "csv")) if f.endswith("-memory.csv")]
    assert len(csvs) == 1
    rows = open(os.path.join(dataset, "csv", csvs[0])).read().strip().splitlines()
    assert len(rows) == 1 + N_FRAMES  # header + Evaluation::LogMemoryUse once per frame (DynSlam.cpp:161)


def test_pipeline_is_deterministic_and_decay_runs(oracle_run, oracle_host, dataset, tmp_path):
    kv, _, _, _, _ = oracle_run
    kv2, _ = run_host(oracle_host, dataset, str(tmp_path / "again.bin"))
    assert kv2 == kv
    kv3, _ = run_host(oracle_host, dataset, str(tmp_path / "decay.bin"), decay=1)  # VoxelDecayParams(enabled, min age 3, max weight 1)
    assert int(kv3["static_decayed"]) > 0 and int(kv3["static_used_blocks"]) < int(kv["static_used_blocks"])
    assert kv3["tracks"] == "4"
    # DecayCatchup (InfiniTamDriver.h:210-225) drains the visible lists still queued; without decay it does nothing
    assert int(kv3["static_decayed_after_catchup"]) > int(kv3["static_decayed"])
    assert int(kv3["static_saved_decay_bytes"]) == int(kv3["static_decayed_after_catchup"]) * 4096
    assert kv["static_decayed_after_catchup"] == "0"


def read_depth_csv(root, which):
    """-> {frame: {(leg, delta): {"total", "error", "missing", "correct"}}} from the reference's <...>-<which>-depth-result.csv."""
    import csv
    path = [f for f in os.listdir(os.path.join(root, "csv")) if f.endswith(f"-{which}-depth-result.csv")]
    assert len(path) == 1
    out = {}
    with open(os.path.join(root, "csv", path[0])) as f:
        for row in csv.DictReader(f):
            rec = {}
            for k, v in row.items():
                parts = (k or "").split("-")
                if parts[0] in ("fusion", "input") and parts[1] in ("total", "error", "missing", "correct") and len(parts) == 3:
                    rec.setdefault((parts[0], parts[2]), {})[parts[1]] = int(v)
            out[int(row["frame"])] = rec
    return out


def test_reference_evaluation_scores_the_engine_renders_against_lidar(oracle_host, dataset, tmp_path):
    """FLAGS_enable_evaluation: after every frame Evaluation::EvaluateFrame (DynSlam.cpp:153-159, Evaluation.cpp:35-150) raycasts
    the map from the frame's pose (GetStaticMapRaycastDepthPreview: the fork's FREECAMERA_DEPTH render + CompositeInstanceDepthMaps),
    projects the LIDAR returns and scores fused and input depth in disparity space — the accuracy metric of the reference's
    experiments, computed by its own code on this engine's renders.  The synthetic LIDAR is the exact surface."""
    kv, log = run_host(oracle_host, dataset, str(tmp_path / "eval.bin"), evaluate=1)
    assert kv["tracks"] == "4" and log.count("Starting evaluation of frame") == N_FRAMES - 1
    static, dynamic = read_depth_csv(dataset, "static"), read_depth_csv(dataset, "dynamic")
    assert sorted(static) == list(range(1, N_FRAMES))
    for frame, rec in static.items():
        for delta, floor in (("1.00", 0.965), ("2.00", 0.985), ("3.00", 0.99)):
            fu, inp = rec[("fusion", delta)], rec[("input", delta)]
            assert fu["total"] == inp["total"] > 30000
            acc = fu["correct"] / (fu["correct"] + fu["error"])
            assert acc > floor, (frame, delta, acc)
            assert inp["correct"] / (inp["correct"] + inp["error"]) > 0.999  # the input is the truth + 0.25 px of noise
        # what the map has seen it reproduces: few LIDAR points without a rendered depth once a few frames are fused
        if frame >= 3:
            assert rec[("fusion", "1.00")]["missing"] < 0.15 * rec[("fusion", "1.00")]["total"]
    # fusing frames beats a single frame at the half-pixel level from the third frame on (the point of the paper's figure)
    for frame in range(3, N_FRAMES):
        fu, inp = static[frame][("fusion", "0.50")], static[frame][("input", "0.50")]
        assert fu["correct"] / (fu["correct"] + fu["error"]) > inp["correct"] / (inp["correct"] + inp["error"]) - 0.01
    # LIDAR points on reconstructed dynamic objects are scored against the composited instance renders
    last = dynamic[N_FRAMES - 1]
    assert last[("fusion", "12.00")]["correct"] > 3000
    assert last[("fusion", "12.00")]["correct"] / (last[("fusion", "12.00")]["correct"] + last[("fusion", "12.00")]["error"]) > 0.95


# --- GPU ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_reference_pipeline_on_the_hip_engine_matches_the_oracle(tmp_path):
    hip, orc = os.path.join(BUILD, "ref_dynslam_host"), os.path.join(BUILD, "ref_dynslam_host_orc")
    if not (os.path.exists(hip) and os.path.exists(orc)):
        pytest.fail("tests/refhost/_build/ref_dynslam_host{,_orc} are missing: __graft_entry__.build() makes them where "
                    "/root/reference exists and they travel to the GPU box with the snapshot")
    from tests.refhost.make_dataset import write_dataset
    root = str(tmp_path / "kitti_like")
    os.makedirs(root)
    write_dataset(root, N_FRAMES, W, H)
    for decay in (0, 1):
        got, _ = run_host(hip, root, str(tmp_path / f"hip{decay}.bin"), decay, evaluate=decay)
        want, _ = run_host(orc, root, str(tmp_path / f"orc{decay}.bin"), decay, evaluate=decay)
        assert got == want, {k: (got.get(k), want[k]) for k in want if got.get(k) != want[k]}
        assert open(tmp_path / f"hip{decay}.bin", "rb").read() == open(tmp_path / f"orc{decay}.bin", "rb").read()
    # The SAME unmodified host with its volumes placed per GPU by the shim's policy (INTEGRATION.md "multi-GPU without a source
    # change": DSR_DEVICES lists the GPUs, every track's InfiniTamDriver lands on the next one; here both entries name the one GPU
    # of the box, instance volumes on all of them) and with the cross-GPU forms forced: every digest of the default run.
    want, _ = run_host(orc, root, str(tmp_path / "orc_p.bin"))
    for k, extra in enumerate((dict(DSR_DEVICES="0,0"), dict(DSR_DEVICES="0,0,0", DSR_INSTANCES_ON_ALL_DEVICES="1", DSR_FORCE_PEER_PATH="1"),
                               dict(DSR_DEVICES="0,0", DSR_PIPELINED_VIEW="2"))):
        got, _ = run_host(hip, root, str(tmp_path / f"hip_p{k}.bin"), extra_env=extra)
        assert got == want, (extra, {k2: (got.get(k2), want[k2]) for k2 in want if got.get(k2) != want[k2]})
        assert open(tmp_path / f"hip_p{k}.bin", "rb").read() == open(tmp_path / "orc_p.bin", "rb").read(), extra
