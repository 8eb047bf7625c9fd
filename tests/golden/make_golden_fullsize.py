#!/usr/bin/env python3
"""Generates tests/golden/fullsize_digests.json: per-frame SHA-256 digests of the engine state
of the BENCHED configurations at BASELINE.json's full size (1242x375), computed offline with the
CPU oracle (oracle/dsr_oracle.cpp, OpenMP) — the `-m gpu` suite (tests/test_gpu_fullsize_golden.py)
replays the same call sequences on the HIP engine and matches the digests WITHOUT running the
oracle on the GPU box (VERDICT r1 "the benched state is never compared with the oracle").

Cases
  bench_5mm      bench.py's default workload, its exact table sizes (2^23 blocks / 2^23 buckets /
                 2^21 excess), frames 0..24 = the driver's `--steps 20 --warmup 5` run: type-3
                 re-tests over ~600 k entries, weights > 1, colour running means.
  cfg5_4mm_gc_swap   the `4mm` preset (mu = 0.016: its own short_division_exact outcome,
                 BASELINE configs[4]) with voxel GC (max_weight 1, min_age 3) + host swapping.
  seq06_5cm_50frames  configs[0]/[1] at the reference's own settings: 1226x370 (seq 06), 50 frames, 5 cm voxels,
                 upstream's default table sizes, voxel GC (1, 20).
  gc_defaults_5cm_230frames  voxel GC at the reference's defaults (max_weight 1, min_age 200) over 230 frames: the FIFO of
                 visible lists fills, wraps around and pops 30 lists.
  cfg2_instances BASELINE configs[2]: static 5 mm map + 4 instance volumes (0.035 m, mu 1.0,
                 7142 blocks: InstanceReconstructor.cpp:372-379), masks split on the device
                 (ProcessSilhouette / RemoveSilhouette), 3 frames.
  cfg3_static_plus_7_instances  BASELINE configs[3]: static 5 mm map + 7 instance volumes, 15 frames (the far boxes of
                 the synthetic street come inside the 20 m depth clip from frame ~5 on: all seven are fused and composited by the
                 end), and EVERY frame the fused preview of InstanceReconstructor::CompositeInstances (:911-990): the map and
                 every visible instance raycast (colour + float depth) from the frame's camera, z-composited in ascending track
                 id — digests of the composited colour and depth next to every volume's state.

Like state_digests.json these pin the ORACLE's outputs (the reference's engines are an empty
submodule: parity with upstream itself stays unpinned).

Run from the repo root (needs ~45 GB of RAM and a few minutes):
    python tests/golden/make_golden_fullsize.py [case ...]
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H = 1242, 375
COMMON = dict(max_w=100, view_frustum_min=0.2, view_frustum_max=30.0)
INSTANCE = dict(voxel_size=0.035, mu=1.0, sdf_local_block_num=7142, hash_bucket_num=0x100000,
                excess_list_size=0x20000, **COMMON)

CASES = {
    "bench_5mm": dict(frames=25, instances=0, decay=None, render_every=5,
                      settings=dict(voxel_size=0.005, mu=0.02, sdf_local_block_num=1 << 23, hash_bucket_num=1 << 23,
                                    excess_list_size=1 << 21, **COMMON)),
    "cfg5_4mm_gc_swap": dict(frames=12, instances=0, decay=(1, 3), render_every=4,
                             settings=dict(voxel_size=0.004, mu=0.016, sdf_local_block_num=1 << 22, hash_bucket_num=1 << 24,
                                           excess_list_size=1 << 22, use_swapping=1, **COMMON)),
    # BASELINE configs[0] / configs[1] at the reference's own settings: KITTI-odometry seq 06's image size
    # (1226x370), 50 frames, static map only, 5 cm voxels (the reference's experiments),
    # upstream's default table sizes, voxel GC as the GUI runs it scaled to the sequence (max_weight 1, min_age 20)
    "seq06_5cm_50frames": dict(frames=50, instances=0, decay=(1, 20), render_every=10, width=1226, height=370,
                               settings=dict(voxel_size=0.05, mu=0.2, sdf_local_block_num=0x40000, hash_bucket_num=0x100000,
                                             excess_list_size=0x20000, **COMMON)),
    # voxel GC at the reference's DEFAULTS (DynSLAMGUI.cpp:36-42: max_weight 1, min_age 200) long enough for the FIFO of
    # visible lists to fill (201 slots), wrap around and pop 30 lists: the steady-state GC path the 4541-frame run of
    # configs[4] exercises, here compared state for state.  5 cm voxels keep the oracle cheap.
    "gc_defaults_5cm_230frames": dict(frames=230, instances=0, decay=(1, 200), render_every=23,
                                      settings=dict(voxel_size=0.05, mu=0.2, sdf_local_block_num=0x40000, hash_bucket_num=0x100000,
                                                    excess_list_size=0x20000, **COMMON)),
    "cfg2_instances": dict(frames=3, instances=4, decay=None, render_every=1,
                           settings=dict(voxel_size=0.005, mu=0.02, sdf_local_block_num=1 << 21, hash_bucket_num=1 << 22,
                                         excess_list_size=1 << 20, **COMMON)),
    "cfg3_static_plus_7_instances": dict(frames=15, instances=7, decay=None, render_every=5, composite=True,
                                         settings=dict(voxel_size=0.005, mu=0.02, sdf_local_block_num=1 << 21, hash_bucket_num=1 << 22,
                                                       excess_list_size=1 << 20, **COMMON)),
}


def _h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def voxel_digest(e, ptrs, chunk=1 << 15):
    """SHA-256 over the AoS voxels of the blocks `ptrs` (ascending block index), streamed in
    chunks of `chunk` blocks so that neither side needs the whole array in host memory."""
    ptrs = np.sort(np.asarray(ptrs, np.int64))
    sha = hashlib.sha256()
    if len(ptrs) == 0:
        return sha.hexdigest()
    lo = int(ptrs[0])
    hi = int(ptrs[-1]) + 1
    for a in range(lo, hi, chunk):
        b = min(hi, a + chunk)
        sel = ptrs[(ptrs >= a) & (ptrs < b)] - a
        if len(sel) == 0:
            continue
        vox = e.dump_voxel_blocks(a, b - a)
        sha.update(np.ascontiguousarray(vox[sel]).tobytes())
    return sha.hexdigest()


def scene_digest(e, voxels, render):
    st = e.get_stats()
    ht = e.dump_hash_table()
    vis = e.dump_visible_list()
    out = {"last_free_block_id": int(st.last_free_block_id),
           "last_free_excess_list_id": int(st.last_free_excess_list_id),
           "no_visible_blocks": int(st.no_visible_blocks),
           "decayed_block_count": int(st.decayed_block_count),
           "hash_table": _h(ht), "visible_list": _h(vis)}
    if voxels == "visible":
        p = ht["ptr"][vis]
        out["voxels_visible"] = voxel_digest(e, p[p >= 0])
    elif voxels == "all":
        out["voxels_in_use"] = voxel_digest(e, ht["ptr"][ht["ptr"] >= 0])
    if render:
        rs = e.dump_render_state()
        for k in ("minmax", "raycast_result", "points", "normals", "raycast_image"):
            out[k] = _h(rs[k])
    return out


def fused_preview(e, inst, masks, T):
    """InstanceReconstructor::CompositeInstances (InstanceReconstructor.cpp:911-990) for one frame: the static map and every
    instance with a detection in this frame raycast (colour + float depth) from the frame's camera — the instance's pose is the
    model view composed with its object pose (:923,968) —, z-composited over the dimmed map in ascending track id (= 1 + k).
    -> (rgba [H, W, 4], depth [H, W]) through the engine's own C ABI (dsr_composite_instances / the oracle's restatement)."""
    import ctypes as C
    from dynslam_amd import _capi
    M = np.linalg.inv(np.asarray(T, np.float64)).astype(np.float32)
    rgba, depth = e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
    layers = sorted((1 + k, k, rel) for k, _, _, _, rel in masks)
    if layers:
        lc, ld = [], []
        for _, k, rel in layers:
            c, d = inst[k].get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=np.linalg.inv(np.asarray(rel, np.float64)).astype(np.float32),
                                     want_rgba=True, want_depth=True)
            lc.append(c); ld.append(d)
        lc, ld = np.ascontiguousarray(np.stack(lc)), np.ascontiguousarray(np.stack(ld))
        tids = np.array([t for t, _, _ in layers], np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        st = e.api.composite_instances(vp(rgba), vp(depth), vp(lc), vp(ld), vp(tids), len(layers), depth.size, 1.0, 1)
        assert st == 0, st
    return rgba, depth


def run_case(make_engine, case, frames, log=None):
    """make_engine(settings_kwargs, calib_args) -> EngineCore-like; frames = bench.make_frames(...)."""
    from dynslam_amd import _capi
    from dynslam_amd.engine import OutOfBlocksError
    from dynslam_amd.synth import StreetScene
    cw, ch = case.get("width", W), case.get("height", H)
    sc = StreetScene(cw, ch, n_instances=case["instances"])
    calib_args = (*sc.intrinsics(), cw, ch)
    e = make_engine(case["settings"], calib_args)
    inst = [make_engine(INSTANCE, calib_args) for _ in range(case["instances"])]
    swapping = bool(case["settings"].get("use_swapping"))
    out = {"frames": []}
    n = case["frames"]
    for i in range(n):
        rgba, d, T, masks = frames[i]
        e.update_view(rgba, d)
        for k, x0, y0, mask, rel in masks:  # InstanceReconstructor.cpp:238-263,569-700
            e.extract_silhouette(inst[k], mask, x0, y0)
            e.remove_silhouette(mask, x0, y0)
            inst[k].set_pose_inv_m(rel)
            inst[k].process_frame()
            inst[k].prepare()
        e.set_pose_inv_m(T)
        try:
            e.process_frame()
        except OutOfBlocksError:
            pass
        e.prepare()
        if case["decay"]:
            e.decay(case["decay"][0], case["decay"][1], False)
        last = i == n - 1
        render = last or (i + 1) % case["render_every"] == 0
        rec = scene_digest(e, "all" if last else ("visible" if render else None), render)
        if swapping:
            state, stored = e.dump_swap_state()
            rec["swap_state"] = _h(state)
            rec["swap_stored"] = _h(stored)
            rec["swap_stored_count"] = int((stored != 0).sum())
            ids = np.nonzero(stored)[0][:64]
            sha = hashlib.sha256()
            for t in ids.tolist():
                sha.update(np.ascontiguousarray(e.dump_stored_block(t)).tobytes())
            rec["stored_blocks_first64"] = sha.hexdigest()
        if inst and (render or last):
            rec["instances"] = [scene_digest(ie, "all", True) for ie in inst]
        if case.get("composite"):
            c_rgba, c_depth = fused_preview(e, inst, masks, T)
            rec["composite_rgba"], rec["composite_depth"] = _h(c_rgba), _h(c_depth)
            rec["composite_layers"] = len(masks)
            rec["composite_hit_fraction"] = round(float((c_depth > 0).mean()), 6)
        out["frames"].append(rec)
        if log:
            log(f"frame {i}: visible {rec['no_visible_blocks']}, free head {rec['last_free_block_id']}")
    # one free-view colour + depth render of the final map from an earlier pose
    pose = np.linalg.inv(np.asarray(frames[max(0, n - 3)][2], np.float64)).astype(np.float32)
    out["render_colour"] = _h(e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=pose)[0])
    out["render_depth"] = _h(e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=pose, want_rgba=False, want_depth=True)[1])
    for ie in inst:
        ie.close()
    e.close()
    return out


def case_frames(case):
    sys.path.insert(0, ROOT)
    from bench import make_frames
    return make_frames(case.get("width", W), case.get("height", H), case["frames"], case["instances"])


def oracle_factory(settings, calib_args):
    from dynslam_amd.engine import make_calib
    from oracle.oracle import OracleEngine, oracle_settings
    return OracleEngine(oracle_settings(**settings), make_calib(*calib_args), threads=os.cpu_count() or 1)


if __name__ == "__main__":
    import time
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize_digests.json")
    doc = json.load(open(path)) if os.path.exists(path) else {"cases": {}}
    doc["generator"] = "tests/golden/make_golden_fullsize.py (oracle/dsr_oracle.cpp)"
    for name in (sys.argv[1:] or sorted(CASES)):
        t0 = time.time()
        fr = case_frames(CASES[name])
        doc["cases"][name] = run_case(oracle_factory, CASES[name], fr, log=lambda s: print(name, s, flush=True))
        print(f"{name}: {time.time() - t0:.0f} s", flush=True)
        with open(path, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", path)
