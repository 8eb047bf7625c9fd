#!/usr/bin/env python3
"""Generates tests/golden/state_digests.json: SHA-256 digests of complete engine state after
a seeded synthetic sequence, computed with the CPU oracle (oracle/dsr_oracle.cpp).

The reference ships no golden vectors for this path and its engines cannot
be built or imported here (empty submodule), so these fixtures pin the ORACLE's outputs
(which tests/test_oracle_kat.py ties to independent closed-form derivations); the -m gpu
suite checks the HIP engine against the same digests without running the oracle.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CASES = {
    "street_5cm": dict(W=320, H=96, frames=5, decay=None,
                       settings=dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                                     sdf_local_block_num=40000, hash_bucket_num=0x10000, excess_list_size=0x4000)),
    "street_collisions_decay": dict(W=256, H=80, frames=7, decay=(1, 2),
                                    settings=dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2,
                                                  view_frustum_max=30.0, sdf_local_block_num=30000,
                                                  hash_bucket_num=1024, excess_list_size=0x8000)),
    "instance_volume": dict(W=256, H=80, frames=3, decay=None,
                            settings=dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2,
                                          view_frustum_max=12.0, sdf_local_block_num=7142,
                                          hash_bucket_num=0x10000, excess_list_size=0x4000)),
}


def _h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(make_engine, case):
    """make_engine(settings_kwargs, calib_args) -> EngineCore-like.  Returns the digest dict."""
    from dynslam_amd import _capi
    from dynslam_amd.engine import OutOfBlocksError
    from dynslam_amd.synth import StreetScene
    sc = StreetScene(case["W"], case["H"])
    e = make_engine(case["settings"], (*sc.intrinsics(), case["W"], case["H"]))
    for i in range(case["frames"]):
        rgba, d, T, _ = sc.frame(i)
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        try:
            e.process_frame()
        except OutOfBlocksError:
            pass
        e.prepare()
        if case["decay"]:
            e.decay(case["decay"][0], case["decay"][1], False)
    st = e.get_stats()
    ht = e.dump_hash_table()
    vox = e.dump_voxel_blocks()
    used = np.sort(ht["ptr"][ht["ptr"] >= 0])
    rs = e.dump_render_state()
    pose = np.linalg.inv(sc.pose(1).astype(np.float64)).astype(np.float32)
    out = {
        "last_free_block_id": int(st.last_free_block_id),
        "last_free_excess_list_id": int(st.last_free_excess_list_id),
        "no_visible_blocks": int(st.no_visible_blocks),
        "decayed_block_count": int(st.decayed_block_count),
        "hash_table": _h(ht),
        "visible_list": _h(e.dump_visible_list()),
        "voxels_in_use": _h(vox[used]),
        "minmax": _h(rs["minmax"]),
        "raycast_result": _h(rs["raycast_result"]),
        "points": _h(rs["points"]),
        "normals": _h(rs["normals"]),
        "raycast_image": _h(rs["raycast_image"]),
    }
    for name, t in (("shaded", _capi.IMAGE_FREECAMERA_SHADED), ("colour", _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME),
                    ("normal", _capi.IMAGE_FREECAMERA_COLOUR_FROM_NORMAL),
                    ("weight", _capi.IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT)):
        out["render_" + name] = _h(e.get_image(t, pose_m=pose)[0])
    out["render_depth"] = _h(e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=pose, want_rgba=False, want_depth=True)[1])
    mesh = e.mesh_scene()  # ITMMeshingEngine::MeshScene: triangle order is part of the digest
    out["mesh_triangles"] = int(len(mesh))
    out["mesh"] = _h(mesh)
    e.close()
    return out


def oracle_factory(settings, calib_args):
    from dynslam_amd.engine import make_calib
    from oracle.oracle import OracleEngine, oracle_settings
    return OracleEngine(oracle_settings(**settings), make_calib(*calib_args))


if __name__ == "__main__":
    digests = {name: run_case(oracle_factory, case) for name, case in CASES.items()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_digests.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (oracle/dsr_oracle.cpp)", "cases": digests}, f, indent=1, sort_keys=True)
    print("wrote", path)
