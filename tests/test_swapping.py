"""Host swap-in/out (settings.use_swapping).  CPU: invariants of the
oracle's restatement of ITMSwappingEngine_CPU (swap round trip preserves voxels bit-exactly).
-m gpu: HIP == oracle on a forward-then-backward drive, including swap states, the host store
and the combination with voxel GC."""
import numpy as np
import pytest

from dynslam_amd.engine import make_calib
from dynslam_amd.synth import StreetScene

KW = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
          sdf_local_block_num=40000, hash_bucket_num=0x10000, excess_list_size=0x4000, use_swapping=1)
W, H = 320, 96
SEQ = list(range(0, 24, 4)) + list(range(20, -1, -4))


def oracle_engine(**kw):
    from oracle.oracle import OracleEngine, oracle_settings
    k = dict(KW); k.update(kw)
    sc = StreetScene(W, H)
    return sc, OracleEngine(oracle_settings(**k), make_calib(*sc.intrinsics(), W, H))


def hip_engine(**kw):
    from dynslam_amd.engine import EngineCore, default_settings
    k = dict(KW); k.update(kw)
    sc = StreetScene(W, H)
    return sc, EngineCore(default_settings(**k), make_calib(*sc.intrinsics(), W, H))


def step(e, sc, i, decay=None):
    rgba, d, T, _ = sc.frame(i)
    e.update_view(rgba, d); e.set_pose_inv_m(T); e.process_frame(); e.prepare()
    if decay:
        e.decay(*decay, False)


def test_oracle_swap_invariants(oracle_lib):
    sc, e = oracle_engine()
    seen_out = False
    for k, i in enumerate(SEQ):
        before_ht = e.dump_hash_table()
        step(e, sc, i)
        ht = e.dump_hash_table()
        st, hs = e.dump_swap_state()
        vt = e.dump_visible_types()
        stats = e.get_stats()
        out = ht["ptr"] == -1
        seen_out |= bool(out.any())
        # swapped-out entries: not visible any more, host copy present, state 0
        assert (hs[out] == 1).all() and (st[out] == 0).all() and (vt[out] == 0).all()
        # resident entries that were visible are in state 2 after the frame
        vis = e.dump_visible_list()
        res_vis = vis[ht["ptr"][vis] >= 0]
        assert (st[res_vis] == 2).all()
        # accounting: blocks in use == resident entries
        assert 40000 - 1 - stats.last_free_block_id == int((ht["ptr"] >= 0).sum())
        # at most SDF_TRANSFER_BLOCK_NUM blocks leave per frame
        assert int((out & (before_ht["ptr"] >= 0)).sum()) <= 4096
    assert seen_out
    # everything that came back has been merged: no entry is left in state 1 with a host copy
    st, hs = e.dump_swap_state()
    assert not ((st == 1) & (hs == 1)).any()


def test_oracle_swap_round_trip_preserves_voxels(oracle_lib):
    """A block swapped out and merged back into a fresh block equals its old content
    (combine(src, default) == src) — checked on a run WITHOUT new measurements in between."""
    sc, e = oracle_engine()
    for i in (0, 4, 8):
        step(e, sc, i)
    ht = e.dump_hash_table()
    out_ids = np.nonzero(ht["ptr"] == -1)[0]
    assert len(out_ids) > 100
    stored = {int(t): e.dump_stored_block(int(t)).copy() for t in out_ids[:50]}
    # look back at the old pose with an EMPTY depth image: nothing integrates, blocks only swap in
    rgba, d, T, _ = sc.frame(0)
    e.update_view(rgba, np.zeros_like(d)); e.set_pose_inv_m(T)
    # two frames: the first re-allocates (visible type 3 needs a previous visible list) ...
    e.process_frame()
    ht2 = e.dump_hash_table()
    vox = e.dump_voxel_blocks()
    back = [t for t in stored if ht2["ptr"][t] >= 0]
    # entries still invisible from there stay out; the ones that came back are bit-identical
    for t in back:
        got = vox[ht2["ptr"][t]]
        want = stored[t]
        has_w = want["w_depth"] > 0
        assert np.array_equal(got["sdf"][has_w], want["sdf"][has_w])
        assert np.array_equal(got["w_depth"], want["w_depth"])
        has_c = want["w_color"] > 0
        assert np.array_equal(got["clr"][has_c], want["clr"][has_c])


@pytest.mark.gpu
@pytest.mark.parametrize("decay", [None, (1, 2)])
def test_gpu_swapping_parity(hip_api, oracle_lib, decay):
    from tests.common import assert_render_equal, assert_scene_equal
    sc, g = hip_engine()
    sc, o = oracle_engine()
    for k, i in enumerate(SEQ):
        for e in (g, o):
            step(e, sc, i, decay)
        assert_scene_equal(g, o, voxels=(k in (5, len(SEQ) - 1)))
        sg, so = g.dump_swap_state(), o.dump_swap_state()
        assert np.array_equal(sg[0], so[0]) and np.array_equal(sg[1], so[1])
    assert_render_equal(g, o)
    ht = o.dump_hash_table()
    st, hs = o.dump_swap_state()
    assert hs.sum() > 1000
    # ITMGlobalCache: one slot per entry for the life of the scene — the backward half of the drive
    # swaps the same entries out again and must reuse their slots (no growth per frame)
    assert g.get_stats().host_store_slots == o.get_stats().host_store_slots
    for t in np.nonzero(hs)[0][::37].tolist():
        assert np.array_equal(g.dump_stored_block(t), o.dump_stored_block(t))
    assert g.dump_stored_block(int(np.nonzero(hs == 0)[0][0])) is None


@pytest.mark.gpu
def test_gpu_swapping_transfer_cap(hip_api, oracle_lib):
    """More than 4096 blocks become invisible at once: they leave over several frames, in
    ascending entry order, identically on both sides."""
    from tests.common import assert_scene_equal
    sc, g = hip_engine(sdf_local_block_num=60000)
    sc, o = oracle_engine(sdf_local_block_num=60000)
    for i in range(0, 12, 2):
        for e in (g, o):
            step(e, sc, i)
    far = sc.pose(0).copy(); far[2, 3] = 500.0  # jump far away: the whole map is invisible
    rgba, d, _, _ = sc.frame(0)
    counts = [int((o.dump_hash_table()["ptr"] == -1).sum())]
    for _ in range(4):
        for e in (g, o):
            e.update_view(rgba, np.zeros_like(d)); e.set_pose_inv_m(far); e.process_frame()
        assert_scene_equal(g, o, voxels=False)
        counts.append(int((o.dump_hash_table()["ptr"] == -1).sum()))
    inc = np.diff(counts)
    assert inc[0] == 4096 and (inc <= 4096).all() and inc[-1] == 0, counts
    assert_scene_equal(g, o)


@pytest.mark.gpu
def test_gpu_host_store_grows_slab_by_slab(hip_api, oracle_lib, monkeypatch):
    """The host store is a pool of pinned slabs the kernels address directly; with tiny slabs
    (64 blocks) the batches straddle slab boundaries, the pool has to grow ahead of the device's
    slot counter without ever reading it synchronously, and every stored block must still come
    back identical — after a reset as well (the slot counter restarts, the slabs are reused)."""
    from tests.common import assert_scene_equal
    monkeypatch.setenv("DSR_SLAB_BLOCKS", "64")
    sc, g = hip_engine()
    sc, o = oracle_engine()
    for rnd in range(2):
        for i in SEQ:
            for e in (g, o):
                step(e, sc, i)
        assert_scene_equal(g, o)
        st, hs = o.dump_swap_state()
        sg = g.dump_swap_state()
        assert np.array_equal(sg[0], st) and np.array_equal(sg[1], hs)
        assert hs.sum() > 1000  # > 15 slabs of 64 blocks
        for t in np.nonzero(hs)[0][::53].tolist():
            assert np.array_equal(g.dump_stored_block(t), o.dump_stored_block(t))
        for e in (g, o):
            e.reset_scene()
    g.close(); o.close()


def test_oracle_host_store_is_bounded_by_entries(oracle_lib):
    """Driving back and forth swaps the same entries out again and again; the store holds one
    slot per entry that was ever out, not one per transfer."""
    sc, e = oracle_engine()
    ever = np.zeros(0x10000 + 0x4000, dtype=bool)
    slots = []
    for rnd in range(2):
        for i in SEQ:
            step(e, sc, i)
            ever |= e.dump_swap_state()[1] == 1
        slots.append(e.get_stats().host_store_slots)
        assert slots[-1] == int(ever.sum()) > 1000
    # the second lap swaps ~6000 blocks out again and needs only the slots of the few entries it
    # sees for the first time
    assert slots[1] - slots[0] < 0.05 * slots[0]


@pytest.mark.gpu
def test_gpu_host_store_slots_are_reused(hip_api, oracle_lib):
    """every swap-out used to take fresh slots (up to 16 MiB of pinned memory per frame,
    for ever).  Two laps over the same street: the second lap re-uses the slots of the first, with
    voxel GC dropping host copies in between as well."""
    for decay in (None, (1, 2)):
        sc, g = hip_engine()
        sc, o = oracle_engine()
        laps = []
        for rnd in range(2):
            for i in SEQ:
                for e in (g, o):
                    step(e, sc, i, decay)
            laps.append((g.get_stats().host_store_slots, o.get_stats().host_store_slots))
        assert laps[0][0] == laps[0][1] > 1000 and laps[1][0] == laps[1][1], laps
        assert laps[1][0] - laps[0][0] < 0.10 * laps[0][0]  # the second lap re-uses the first lap's slots
        st = g.get_stats()
        assert st.host_store_capacity_slots >= st.host_store_slots
        hs = o.dump_swap_state()[1]
        for t in np.nonzero(hs)[0][::41].tolist():
            assert np.array_equal(g.dump_stored_block(t), o.dump_stored_block(t))
        g.close(); o.close()
