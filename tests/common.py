"""Shared helpers of the parity tests: run the same call sequence on the HIP engine and
on the CPU oracle and compare complete engine state."""
import numpy as np

from dynslam_amd import _capi
from dynslam_amd.engine import EngineCore, default_settings, make_calib
from dynslam_amd.synth import StreetScene

SMALL = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
             sdf_local_block_num=40000, hash_bucket_num=0x10000, excess_list_size=0x4000)

RENDER_TYPES = [
    _capi.IMAGE_FREECAMERA_SHADED, _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME,
    _capi.IMAGE_FREECAMERA_COLOUR_FROM_NORMAL, _capi.IMAGE_FREECAMERA_COLOUR_FROM_DEPTH_WEIGHT,
    _capi.IMAGE_FREECAMERA_DEPTH,
]


def make_pair(W=320, H=96, scene_kw=None, **settings_kw):
    """-> (scene, hip engine, oracle engine) with identical settings."""
    from oracle.oracle import OracleEngine, oracle_settings
    kw = dict(SMALL)
    kw.update(settings_kw)
    sc = StreetScene(W, H, **(scene_kw or {}))
    calib = make_calib(*sc.intrinsics(), W, H)
    g = EngineCore(default_settings(**kw), calib)
    o = OracleEngine(oracle_settings(**kw), calib)
    return sc, g, o


def feed(engines, sc, i, prepare=True, ignore_oob=False):
    from dynslam_amd.engine import OutOfBlocksError
    rgba, d, T, _ = sc.frame(i)
    raised = []
    for e in engines:
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        try:
            e.process_frame()
            raised.append(False)
        except OutOfBlocksError:
            if not ignore_oob:
                raise
            raised.append(True)
        if prepare:
            e.prepare()
    return raised


def assert_scene_equal(g, o, voxels=True):
    sg, so = g.get_stats(), o.get_stats()
    for k in ("last_free_block_id", "last_free_excess_list_id", "no_visible_blocks", "decayed_block_count"):
        assert getattr(sg, k) == getattr(so, k), f"{k}: {getattr(sg, k)} vs {getattr(so, k)}"
    hg, ho = g.dump_hash_table(), o.dump_hash_table()
    assert np.array_equal(hg, ho), f"hash table differs in {(hg != ho).sum()} entries"
    lg, lo = g.dump_visible_list(), o.dump_visible_list()
    assert np.array_equal(lg, lo), f"visible list differs: {len(lg)} vs {len(lo)} entries, only here {np.setdiff1d(lg, lo)[:8]}, only there {np.setdiff1d(lo, lg)[:8]}"
    tg, to = g.dump_visible_types(), o.dump_visible_types()
    assert np.array_equal(tg, to), f"visible types differ at {np.nonzero(tg != to)[0][:8]}: {tg[tg != to][:8]} vs {to[tg != to][:8]}"
    vg, vo = g.dump_allocation_lists(), o.dump_allocation_lists()
    n = so.last_free_block_id + 1  # only the live part of the free list is defined
    assert np.array_equal(vg[0][:n], vo[0][:n]), "the live part of the block free list differs"
    m = so.last_free_excess_list_id + 1
    assert np.array_equal(vg[1][:m], vo[1][:m]), "the live part of the excess free list differs"
    if voxels:
        bg, bo = g.dump_voxel_blocks(), o.dump_voxel_blocks()
        if not np.array_equal(bg, bo):
            bad = np.argwhere(bg != bo)
            raise AssertionError(f"voxel blocks differ at {len(bad)} voxels, first {bad[0]}: "
                                 f"{bg[tuple(bad[0])]} vs {bo[tuple(bad[0])]}")


def assert_render_equal(g, o, freeview=False, skip=()):
    rg, ro = g.dump_render_state(freeview), o.dump_render_state(freeview)
    keys = [k for k in ["minmax", "raycast_result", "raycast_image"] + ([] if freeview else ["points", "normals"]) if k not in skip]
    for k in keys:
        a, b = rg[k], ro[k]
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            where = ""
            if a.ndim == 3:  # an image: rows x columns x components
                where = (f"; rows {bad[:, 0].min()}..{bad[:, 0].max()}, columns {bad[:, 1].min()}..{bad[:, 1].max()}, per component "
                         f"{[int((bad[:, 2] == c).sum()) for c in range(a.shape[2])]}")
            raise AssertionError(f"render state '{k}' differs at {len(bad)} elements, first {bad[0]}: "
                                 f"{a[tuple(bad[0])]} vs {b[tuple(bad[0])]}{where}")
