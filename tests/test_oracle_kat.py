"""Known-answer tests that pin the CPU oracle (the reference holds no tests or
golden vectors for this path, so these are authored here).  Independent numpy float32
re-derivations for cases small/simple enough to state in closed form:

 (1) hash function + chain structure, negative block coordinates;
 (2) single fronto-parallel plane: every voxel of every allocated block after 1 and k frames;
 (3) allocation order: an independent tiny model of mark + commit (last raster writer wins,
     losers go to the excess list next frame, free lists popped in ascending entry order);
 (4) free-list accounting (InfiniTamDriver.h:241-244 formula);
 (5) raycast of the fused plane hits within half a voxel, (6) float depth is metres, 0 = miss;
 (7) voxel GC invariants (SURVEY.md A.6 i-vi).
"""
import numpy as np
import pytest

from dynslam_amd import _capi
from dynslam_amd.engine import make_calib

f32 = np.float32


def py_hash(b, mask):
    bx, by, bz = (int(v) & 0xFFFFFFFF for v in b)
    return (((bx * 73856093) & 0xFFFFFFFF) ^ ((by * 19349669) & 0xFFFFFFFF) ^ ((bz * 83492791) & 0xFFFFFFFF)) & mask


def make_oracle(W, H, fx, fy, cx, cy, **kw):
    from oracle.oracle import OracleEngine, oracle_settings
    base = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                sdf_local_block_num=4096, hash_bucket_num=0x1000, excess_list_size=0x400)
    base.update(kw)
    return OracleEngine(oracle_settings(**base), make_calib(fx, fy, cx, cy, W, H))


def plane_frame(W, H, depth_mm, colour=(200, 100, 50)):
    rgba = np.empty((H, W, 4), np.uint8)
    rgba[..., 0], rgba[..., 1], rgba[..., 2], rgba[..., 3] = colour[0], colour[1], colour[2], 255
    return rgba, np.full((H, W), depth_mm, np.int16)


def trans(x, y, z):
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [x, y, z]
    return T


# --------------------------------------------------------------------------- (1)
@pytest.mark.parametrize("cam", [(0, 0, 0), (-7.3, -2.1, -9.4), (3.3, -0.2, 11.0)])
def test_hash_function_and_chains(oracle_lib, cam):
    W, H = 64, 48
    e = make_oracle(W, H, 60.0, 60.0, 31.5, 23.5, hash_bucket_num=16, excess_list_size=0x800)
    rgba, d = plane_frame(W, H, 2000)
    for _ in range(6):  # several frames so that chains grow
        e.update_view(rgba, d)
        e.set_pose_inv_m(trans(*cam))
        e.process_frame()
    ht = e.dump_hash_table()
    nb = 16
    used = np.nonzero(ht["ptr"] >= 0)[0]
    assert len(used) > nb  # chains exist
    if any(c < 0 for c in cam):
        assert (ht["pos"][used] < 0).any(), "test must cover negative block coordinates"
    seen_pos = set()
    for idx in used:
        pos = tuple(int(v) for v in ht["pos"][idx])
        assert pos not in seen_pos, "a block position is stored twice"
        seen_pos.add(pos)
        h = py_hash(pos, nb - 1)
        # walk the chain from the bucket: the entry must be reachable
        cur, ok = h, False
        for _ in range(10000):
            if cur == idx:
                ok = True
                break
            off = int(ht["offset"][cur])
            if off < 1:
                break
            cur = nb + off - 1
        assert ok, f"entry {idx} pos {pos} not reachable from bucket {h}"
    # (4) free-list accounting
    st = e.get_stats()
    assert st.num_allocated_voxel_blocks - 1 - st.last_free_block_id == len(used)
    ptrs = ht["ptr"][used]
    assert len(set(ptrs.tolist())) == len(ptrs)
    val, _ = e.dump_allocation_lists()
    free = set(val[: st.last_free_block_id + 1].tolist())
    assert free.isdisjoint(set(ptrs.tolist())) and len(free) + len(ptrs) == st.num_allocated_voxel_blocks


# --------------------------------------------------------------------------- (2)
def expected_plane_voxels(ht, W, H, fx, fy, cx, cy, vs, mu, dm):
    """float32 numpy re-derivation of computeUpdatedVoxelDepthInfo for camera = identity."""
    out = {}
    gx, gy, gz = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")  # [x,y,z]
    for idx in np.nonzero(ht["ptr"] >= 0)[0]:
        pos = ht["pos"][idx].astype(np.int32)
        X = ((pos[0] * 8 + gx).astype(f32) * f32(vs)).astype(f32)
        Y = ((pos[1] * 8 + gy).astype(f32) * f32(vs)).astype(f32)
        Z = ((pos[2] * 8 + gz).astype(f32) * f32(vs)).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            u = (f32(fx) * X / Z + f32(cx)).astype(f32)
            v = (f32(fy) * Y / Z + f32(cy)).astype(f32)
            eta = (f32(dm) - Z).astype(f32)
            q = (eta / f32(mu)).astype(f32)
        upd = (Z > 0) & ~((u < 1) | (u > W - 2) | (v < 1) | (v > H - 2)) & ~(eta < -f32(mu))
        newF = np.where(f32(1.0) < q, f32(1.0), q).astype(f32)
        # oldW = 0: newF = (0*oldF + 1*newF) / 1
        sdf = np.where(upd, (newF * f32(32767.0)).astype(f32).astype(np.int32), 32767).astype(np.int16)
        colour = upd & ~((eta > f32(mu)) | (np.abs(q) > f32(0.25)))
        lin = gx + gy * 8 + gz * 64
        blk = np.zeros(512, dtype=[("sdf", "i2"), ("w", "u1"), ("col", "?")])
        blk["sdf"][lin.ravel()] = sdf.ravel()
        blk["w"][lin.ravel()] = upd.ravel().astype(np.uint8)
        blk["col"][lin.ravel()] = colour.ravel()
        out[int(idx)] = blk
    return out


def test_single_plane_analytic(oracle_lib):
    W, H, fx, fy, cx, cy = 80, 60, 70.0, 70.0, 39.5, 29.5
    vs, mu = 0.05, 0.2
    e = make_oracle(W, H, fx, fy, cx, cy)
    rgba, d = plane_frame(W, H, 2000)
    dm = f32(2000) * f32(0.001) + f32(0.0)
    e.update_view(rgba, d)
    e.set_pose_inv_m(np.eye(4, dtype=np.float32))
    e.process_frame()
    ht = e.dump_hash_table()
    vox = e.dump_voxel_blocks()
    exp = expected_plane_voxels(ht, W, H, fx, fy, cx, cy, vs, mu, dm)
    assert len(exp) > 50
    n_upd = 0
    for idx, blk in exp.items():
        got = vox[ht["ptr"][idx]]
        assert np.array_equal(got["sdf"], blk["sdf"]), f"sdf differs in block {idx}"
        assert np.array_equal(got["w_depth"], blk["w"])
        assert np.array_equal(got["w_color"] > 0, blk["col"])
        # constant image => bilinear sample is the constant; first sample: c = (0*0 + c/255*1)/1*255
        c = got["clr"][blk["col"]]
        if len(c):
            for ch, val in enumerate((200, 100, 50)):
                want = int(f32(f32(val) / f32(255.0)) * f32(255.0))
                assert (np.abs(c[:, ch].astype(int) - want) <= 1).all()  # bilinear weights sum to 1 +- 1 ulp
        n_upd += int(blk["w"].sum())
    assert n_upd > 5000
    # untouched blocks stay default
    free_mask = np.ones(len(vox), bool)
    free_mask[ht["ptr"][ht["ptr"] >= 0]] = False
    assert (vox["sdf"][free_mask] == 32767).all() and (vox["w_depth"][free_mask] == 0).all()

    # k more frames of the same view: w = min(k, maxW); running mean of identical samples stays within 1 LSB
    for _ in range(4):
        e.update_view(rgba, d)
        e.process_frame()
    vox5 = e.dump_voxel_blocks()
    for idx, blk in exp.items():
        got = vox5[ht["ptr"][idx]]
        assert np.array_equal(got["w_depth"], blk["w"].astype(int) * 5)
        assert (np.abs(got["sdf"].astype(int) - blk["sdf"].astype(int)) <= 2).all()


def test_max_w_caps_weight(oracle_lib):
    W, H = 40, 30
    e = make_oracle(W, H, 35.0, 35.0, 19.5, 14.5, max_w=3)
    rgba, d = plane_frame(W, H, 1500)
    for _ in range(6):
        e.update_view(rgba, d)
        e.set_pose_inv_m(np.eye(4, dtype=np.float32))
        e.process_frame()
    vox = e.dump_voxel_blocks()
    assert vox["w_depth"].max() == 3 and vox["w_color"].max() == 3


# --------------------------------------------------------------------------- (3)
def model_alloc_frame(table, heads, val, eal, depth, W, H, fx, fy, cx, cy, vs, mu, vfmin, vfmax, invM, nb):
    """Independent model of buildHashAllocAndVisibleTypePP + the commit loop on a python dict
    table {idx: [pos, offset, ptr]} (float32 numpy scalars, upstream operation order)."""
    oo = f32(1.0) / (f32(vs) * f32(8))
    ifx, ify = f32(1.0) / f32(fx), f32(1.0) / f32(fy)
    alloc = {}

    def entry(i):
        return table.get(i, [(0, 0, 0), 0, -2])

    def mulM(v):
        return [f32(f32(f32(invM[r][0] * v[0]) + f32(invM[r][1] * v[1])) + f32(invM[r][2] * v[2])) + f32(invM[r][3] * f32(1.0))
                for r in range(3)]

    for y in range(H):
        for x in range(W):
            dmv = f32(depth[y, x])
            if dmv <= 0 or (dmv - f32(mu)) < 0 or (dmv - f32(mu)) < f32(vfmin) or (dmv + f32(mu)) > f32(vfmax):
                continue
            pz = dmv
            px = f32(pz * f32(f32(f32(x) - f32(cx)) * ifx))
            py = f32(pz * f32(f32(f32(y) - f32(cy)) * ify))
            norm = f32(np.sqrt(f32(f32(f32(px * px) + f32(py * py)) + f32(pz * pz))))
            f1 = f32(f32(1.0) - f32(f32(mu) / norm))
            s = [f32(c * oo) for c in mulM([f32(px * f1), f32(py * f1), f32(pz * f1)])]
            f2 = f32(f32(1.0) + f32(f32(mu) / norm))
            en = [f32(c * oo) for c in mulM([f32(px * f2), f32(py * f2), f32(pz * f2)])]
            dr = [f32(en[i] - s[i]) for i in range(3)]
            nrm = f32(np.sqrt(f32(f32(f32(dr[0] * dr[0]) + f32(dr[1] * dr[1])) + f32(dr[2] * dr[2]))))
            nsteps = int(np.ceil(f32(f32(2.0) * nrm)))
            with np.errstate(divide="ignore", invalid="ignore"):
                dr = [f32(c / f32(nsteps - 1)) for c in dr]
            pt = list(s)
            for _ in range(nsteps):
                b = tuple(int(np.floor(c)) for c in pt)
                h = py_hash(b, nb - 1)
                he = entry(h)
                found = he[0] == b and he[2] >= -1
                first_free = h if (not found and he[2] < -1) else -1
                cur = h
                if not found:
                    while he[1] >= 1:
                        cur = nb + he[1] - 1
                        he = entry(cur)
                        if he[0] == b and he[2] >= -1:
                            found = True
                            break
                        if he[2] < -1 and first_free < 0:
                            first_free = cur
                if not found:
                    if first_free >= 0:
                        alloc[first_free] = (1, b)
                    else:
                        alloc[cur] = (2, b)
                pt = [f32(pt[i] + dr[i]) for i in range(3)]
    for t in sorted(alloc):
        typ, b = alloc[t]
        if typ == 1:
            v = heads[0]; heads[0] -= 1
            if v >= 0:
                table[t] = [b, entry(t)[1], val[v]]
        else:
            v = heads[0]; heads[0] -= 1
            x = heads[1]; heads[1] -= 1
            if v >= 0 and x >= 0:
                off = eal[x]
                cur = entry(t)
                table[t] = [cur[0], off + 1, cur[2]]
                table[nb + off] = [b, 0, val[v]]
    heads[0] = max(heads[0], -1)
    heads[1] = max(heads[1], -1)


def test_allocation_order_against_independent_model(oracle_lib):
    W, H, fx, fy, cx, cy = 14, 10, 12.0, 12.0, 6.5, 4.5
    nb, nx, nblocks = 8, 64, 40   # 8 buckets: heavy collisions; 40 blocks: exhaustion on later frames
    vs, mu = 0.05, 0.2
    e = make_oracle(W, H, fx, fy, cx, cy, hash_bucket_num=nb, excess_list_size=nx, sdf_local_block_num=nblocks)
    rng = np.random.default_rng(7)
    table, heads = {}, [nblocks - 1, nx - 1]
    val, eal = list(range(nblocks)), list(range(nx))
    from dynslam_amd.engine import OutOfBlocksError
    for frame in range(7):
        d = rng.integers(900, 2500, size=(H, W)).astype(np.int16)
        d[rng.random((H, W)) < 0.1] = 0
        rgba = np.full((H, W, 4), 128, np.uint8)
        T = trans(0.3 * frame, -0.2 * frame, 0.1 * frame)
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        try:
            e.allocate_scene_from_depth()
        except OutOfBlocksError:
            pass
        depth_f = np.where((d <= 0) | (d > 32000), f32(-1.0), d.astype(f32) * f32(0.001) + f32(0.0)).astype(f32)
        invM = e.get_pose()[1]
        model_alloc_frame(table, heads, val, eal, depth_f, W, H, fx, fy, cx, cy, vs, mu, 0.2, 30.0, invM, nb)
        ht = e.dump_hash_table()
        st = e.get_stats()
        assert [st.last_free_block_id, st.last_free_excess_list_id] == heads, f"frame {frame}"
        for idx in range(nb + nx):
            pos, off, ptr = table.get(idx, [(0, 0, 0), 0, -2])
            assert int(ht["ptr"][idx]) == ptr and int(ht["offset"][idx]) == off, f"frame {frame} entry {idx}"
            if ptr >= 0:
                assert tuple(int(v) for v in ht["pos"][idx]) == pos
        vis = e.dump_visible_list()
        assert (np.diff(vis) > 0).all(), "visibleEntryIDs must be ascending"
    assert heads[0] == -1, "the scenario must exhaust the voxel block array"


def test_same_frame_collision_loser_goes_to_excess_next_frame(oracle_lib):
    W, H = 32, 24
    e = make_oracle(W, H, 30.0, 30.0, 15.5, 11.5, hash_bucket_num=4, excess_list_size=0x400)
    rgba, d = plane_frame(W, H, 2000)
    e.update_view(rgba, d)
    e.set_pose_inv_m(np.eye(4, dtype=np.float32))
    e.allocate_scene_from_depth()
    st = e.get_stats()
    # all buckets were free: every target was an ordered-list entry, nothing in the excess list yet
    assert st.last_free_excess_list_id == 0x400 - 1
    assert 4096 - 1 - st.last_free_block_id == 4
    e.allocate_scene_from_depth()
    st2 = e.get_stats()
    # second pass: losers append to their bucket's chain tail, one per bucket per frame
    assert 0x400 - 1 - st2.last_free_excess_list_id == 4
    ht = e.dump_hash_table()
    assert (ht["offset"][:4] >= 1).all()


# ------------------------------------------------------------------- (5) and (6)
def test_raycast_plane_and_float_depth(oracle_lib):
    W, H, fx, fy, cx, cy = 96, 72, 80.0, 80.0, 47.5, 35.5
    e = make_oracle(W, H, fx, fy, cx, cy)
    rgba, d = plane_frame(W, H, 2000)
    for _ in range(3):
        e.update_view(rgba, d)
        e.set_pose_inv_m(np.eye(4, dtype=np.float32))
        e.process_frame()
        e.prepare()
    rs = e.dump_render_state()
    rr = rs["raycast_result"]
    hit = rr[..., 3] > 0
    inner = np.zeros_like(hit); inner[6:-6, 6:-6] = True
    assert hit[inner].mean() > 0.99
    z_m = rr[..., 2] * 0.05
    assert np.abs(z_m[hit & inner] - 2.0).max() < 0.025  # within half a voxel
    # ICP maps: points in metres, normals facing the camera (0,0,-1), w conventions
    pts, nrm = rs["points"], rs["normals"]
    ok = pts[..., 3] > 0
    assert ok[inner].mean() > 0.95
    assert np.allclose(pts[ok][:, :3], rr[ok][:, :3] * 0.05, atol=1e-6)
    assert (nrm[ok & inner][:, 2] < -0.9).all()
    assert (pts[~ok] == np.array([0, 0, 0, -1], np.float32)).all()
    assert (rs["raycast_image"][ok][:, 0] > 200).all()  # (0.8*angle+0.2)*255 with angle ~ 1
    # free-view float depth: metres, 0 on miss (InstanceReconstructor.cpp:861-867)
    _, dep = e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, want_rgba=False, want_depth=True)
    assert np.abs(dep[inner] - 2.0).max() < 0.025
    side = trans(0, 0, 0); side[:3, :3] = [[-1, 0, 0], [0, 1, 0], [0, 0, -1]]  # looking away from the plane
    _, dep2 = e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=side, want_rgba=False, want_depth=True)
    assert (dep2 == 0).all()
    # colour from volume reproduces the constant colour (<= 1 LSB)
    img, _ = e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME)
    c = img[inner & (img[..., 3] == 255)]
    assert len(c) > 1000 and (np.abs(c[:, :3].astype(int) - np.array([200, 100, 50])) <= 2).mean() > 0.9


def test_view_conversion(oracle_lib):
    W, H = 16, 8
    e = make_oracle(W, H, 10.0, 10.0, 7.5, 3.5)
    d = np.array([[0, -5, 1, 500, 32000, 32001, 20000, 32767] * 2] * H, np.int16)
    e.update_view(np.zeros((H, W, 4), np.uint8), d)
    got = e.get_view()[1]
    want = np.where((d <= 0) | (d > 32000), f32(-1.0), d.astype(f32) * f32(0.001)).astype(f32)
    assert np.array_equal(got, want)


# --------------------------------------------------------------------------- (7)
def test_decay_invariants(oracle_lib):
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    W, H = 192, 64
    sc = StreetScene(W, H, noise_px=0.6)
    N = 60000
    e = OracleEngine(oracle_settings(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                                     sdf_local_block_num=N, hash_bucket_num=0x10000, excess_list_size=0x4000),
                     make_calib(*sc.intrinsics(), W, H))
    max_w = 1
    processed = set()
    for i in range(8):
        rgba, d, T, _ = sc.frame(i)
        e.update_view(rgba, d); e.set_pose_inv_m(T); e.process_frame()
        vis_before = e.dump_visible_list()
        dec_before = e.get_stats().decayed_block_count
        ht_before = e.dump_hash_table()
        e.decay(max_w, 2, False)
        st = e.get_stats()
        ht = e.dump_hash_table()
        freed = np.nonzero((ht_before["ptr"] >= 0) & (ht["ptr"] < 0))[0]
        assert len(freed) == st.decayed_block_count - dec_before
        # (ii) freed blocks are all-default and appear exactly once on the free list
        val, _ = e.dump_allocation_lists()
        live_free = val[: st.last_free_block_id + 1]
        assert len(set(live_free.tolist())) == len(live_free)
        if len(freed):
            vox = e.dump_voxel_blocks()
            fp = ht_before["ptr"][freed]
            assert set(fp.tolist()) <= set(live_free.tolist())
            assert (vox["sdf"][fp] == 32767).all() and (vox["w_depth"][fp] == 0).all() and (vox["clr"][fp] == 0).all()
            # freed entries are tombstones: chain link kept, left the visible list
            assert np.array_equal(ht["offset"][freed], ht_before["offset"][freed])
            assert not set(freed.tolist()) & set(e.dump_visible_list().tolist())
        # (iii) accounting
        assert N - 1 - st.last_free_block_id == int((ht["ptr"] >= 0).sum())
        # (v) saved-memory formula of InfiniTamDriver.h:246-250
        assert st.decayed_block_count * st.voxel_bytes * st.block_voxels == st.decayed_block_count * 4096
        assert len(e.dump_visible_list()) == len(vis_before) - len(set(freed.tolist()) & set(vis_before.tolist()))
    assert e.get_stats().decayed_block_count > 0
    # (i) Reap: no voxel with 0 < w <= maxW remains anywhere
    e.decay(2, 0, True)
    ht = e.dump_hash_table()
    vox = e.dump_voxel_blocks()
    alive = vox[ht["ptr"][ht["ptr"] >= 0]]
    assert not ((alive["w_depth"] > 0) & (alive["w_depth"] <= 2)).any()
    assert (alive["w_depth"].max(axis=1) > 0).all(), "fully empty blocks must have been freed"
    # (iv) every remaining block is reachable by lookup; (vi) freed positions can be re-allocated
    nb = 0x10000
    for idx in np.nonzero(ht["ptr"] >= 0)[0][:500]:
        pos = tuple(int(v) for v in ht["pos"][idx])
        cur, ok = py_hash(pos, nb - 1), False
        for _ in range(1000):
            if cur == idx: ok = True; break
            off = int(ht["offset"][cur])
            if off < 1: break
            cur = nb + off - 1
        assert ok
    before = e.get_stats().last_free_block_id
    rgba, d, T, _ = sc.frame(7)
    e.update_view(rgba, d); e.set_pose_inv_m(T); e.process_frame()
    assert e.get_stats().last_free_block_id < before
    ht2 = e.dump_hash_table()
    used = ht2[ht2["ptr"] >= 0]
    assert len({tuple(p) for p in used["pos"].tolist()}) == len(used), "re-allocation must not duplicate a position"
