"""Randomised differential test: one seeded sequence of engine calls — frames from poses that jump, voxel GC passes, resets,
free-view renders from perturbed cameras, with or without a tracking render in between — on the HIP engine and on the CPU oracle,
the complete engine state compared after every call.  Settings are drawn per seed so that the sequences cross the corners the
hand-written parity tests reach one at a time: a table of a few hundred buckets (long excess chains, the excess list running out),
a block array that runs out, instance-sized volumes (the one-workgroup kernels with their sorted list, its merge and its
fall-back), map-sized ones, host swapping, odd image sizes.

The suite runs a few seeds; `DSR_FUZZ_SEEDS=a:b` runs seeds a..b-1 (a soak on the GPU box: `profiles/r06x_fuzz_*.log`)."""
import os

import numpy as np
import pytest

from dynslam_amd import _capi
from tests.common import RENDER_TYPES, assert_render_equal, assert_scene_equal, make_pair


def _seeds():
    spec = os.environ.get("DSR_FUZZ_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    # one of every kind of settings: map-sized, instance-sized (300 / 7142 blocks), tiny tables, swapping; 188, 398: the sequences
    # that found round 6's list path skipping entries which are visible without owning a block (k_small.h `ghost`); 835: a frame
    # without visible blocks after frames without a Prepare
    return [1, 4, 5, 12, 13, 24, 188, 398, 835]


def _draw_settings(rng):
    W, H = [(256, 80), (320, 96), (251, 83), (192, 64)][rng.integers(4)]
    if os.environ.get("DSR_FUZZ_SIZE"):   # a soak at another image size, "640x192"
        W, H = [int(v) for v in os.environ["DSR_FUZZ_SIZE"].split("x")]
    kind = rng.integers(4)
    if os.environ.get("DSR_FUZZ_KIND"):   # a soak of one kind of settings
        kind = int(os.environ["DSR_FUZZ_KIND"])
    if kind == 0:      # instance-sized volume (k_small.h), upstream's table
        kw = dict(voxel_size=0.035, mu=1.0, sdf_local_block_num=int(rng.choice([300, 2000, 7142])), hash_bucket_num=0x100000,
                  excess_list_size=0x20000, view_frustum_max=float(rng.choice([8.0, 12.0, 30.0])))
    elif kind == 1:    # instance-sized volume behind a tiny table: chains, the excess list runs out
        kw = dict(voxel_size=0.05, mu=float(rng.choice([0.2, 0.4])), sdf_local_block_num=int(rng.choice([1500, 9000, 16000])),
                  hash_bucket_num=int(rng.choice([0x100, 0x400, 0x1000])), excess_list_size=int(rng.choice([0x40, 0x400, 0x4000])))
    elif kind == 2:    # map-sized volume (the multi-workgroup kernels)
        kw = dict(voxel_size=float(rng.choice([0.05, 0.08])), mu=float(rng.choice([0.2, 0.32])), sdf_local_block_num=int(rng.choice([17000, 40000])),
                  hash_bucket_num=int(rng.choice([0x2000, 0x10000])), excess_list_size=int(rng.choice([0x200, 0x4000])))
    else:              # host swapping
        kw = dict(voxel_size=0.05, mu=0.2, sdf_local_block_num=int(rng.choice([12000, 40000])), hash_bucket_num=0x10000,
                  excess_list_size=0x4000, use_swapping=1)
    kw["max_w"] = int(rng.choice([3, 100]))
    return W, H, kw, kind


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _seeds())
def test_random_call_sequences_equal_the_oracle(hip_api, seed):
    from dynslam_amd.engine import OutOfBlocksError
    rng = np.random.default_rng(1000 + seed)
    W, H, kw, kind = _draw_settings(rng)
    sc, g, o = make_pair(W=W, H=H, scene_kw=dict(noise_px=float(rng.choice([0.0, 0.4, 0.8]))), **kw)
    swapping = bool(kw.get("use_swapping"))
    frame, fed, log = int(rng.integers(0, 4)), 0, []
    try:
        for step in range(int(rng.integers(10, 18))):
            op = rng.choice(["frame", "frame", "frame", "decay", "render", "reset"], p=[0.3, 0.2, 0.15, 0.15, 0.15, 0.05])
            if fed == 0:
                op = "frame"
            if op == "frame":
                frame = max(0, frame + int(rng.choice([1, 1, 1, 2, 5, -3])))   # mostly forward, jumps both ways
                prepare = bool(rng.random() < 0.8)
                log.append(("frame", frame, prepare))
                rgba, d, T, _ = sc.frame(frame)
                if rng.random() < 0.15:
                    d = d.copy(); d[:, : W // 3] = 0                            # a third of the image without depth
                raised = []
                for e in (g, o):
                    e.update_view(rgba, d)
                    e.set_pose_inv_m(T)
                    try:
                        e.process_frame(); raised.append(False)
                    except OutOfBlocksError:
                        raised.append(True)
                    if prepare:
                        e.prepare()
                assert raised[0] == raised[1], log
                fed += 1
                assert_scene_equal(g, o, voxels=False)
                if prepare:
                    # Prepare() is skipped without visible blocks and the range image "keeps its previous contents": on the
                    # instance path that is the image of the last ProcessFrame (k_small.h builds it inside the allocation's
                    # kernel), in the serial engine the image of the last Prepare — they differ after frames without a Prepare
                    # (seed 835).  Nothing reads that image before the next frame with visible blocks rebuilds it.
                    empty = o.get_stats().no_visible_blocks == 0
                    assert_render_equal(g, o, skip=("minmax",) if empty else ())
            elif op == "decay":   # (also on a swapping volume: blocks leave through the GC and through the host store in one sequence)
                args = (int(rng.choice([1, 2, 5, 100])), int(rng.choice([0, 0, 1, 3])), bool(rng.random() < 0.25))
                log.append(("decay",) + args)
                for e in (g, o):
                    e.decay(*args)
                assert_scene_equal(g, o, voxels=False)
            elif op == "render":
                T = sc.pose(max(0, frame + int(rng.integers(-2, 3)))).astype(np.float64)
                T[:3, 3] += rng.normal(0, 0.05, 3)
                M = np.linalg.inv(T).astype(np.float32)
                t = RENDER_TYPES[rng.integers(len(RENDER_TYPES))]
                log.append(("render", int(t)))
                cg, dg = g.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                co, do = o.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                assert np.array_equal(dg, do) and np.array_equal(cg, co), log
                assert_render_equal(g, o, freeview=True)
                assert np.array_equal(g.dump_visible_list(True), o.dump_visible_list(True)), log
            elif op == "reset":
                log.append(("reset",))
                for e in (g, o):
                    e.reset_scene()
                fed = 0
        assert_scene_equal(g, o)   # every voxel
        if swapping:
            sg, so = g.dump_swap_state(), o.dump_swap_state()
            assert np.array_equal(sg[0], so[0]) and np.array_equal(sg[1], so[1]), log
        # SaveSceneToMesh of whatever the sequence left (tombstones, exhausted arrays, swapped-out blocks): triangle for triangle
        tg, to = g.mesh_scene(), o.mesh_scene()
        assert tg.shape == to.shape and np.array_equal(tg.view(np.uint32), to.view(np.uint32)), (log, f"mesh differs: {tg.shape} vs {to.shape}")
    except AssertionError as ex:
        raise AssertionError(f"seed {seed} kind {kind} {W}x{H} {kw}\ncalls: {log}\n{ex}") from None
    finally:
        g.close(); o.close()


def _batch_seeds():
    spec = os.environ.get("DSR_FUZZ_BATCH_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    # 2, 219: a frame that only blanks (no volume of the batch has a detection) left the status words of the call undefined;
    # 412: a reset followed by a frame without visible blocks — the ray box forgot which pose the buffer's misses belong to
    return [1, 2, 3, 4, 219, 412]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _batch_seeds())
def test_random_batch_sequences_equal_the_oracle(hip_api, monkeypatch, seed):
    """The volume batch (dsr_batch_*: one launch of every kernel for all instance volumes of a GPU, the tracking render deferred and
    paired with the preview render) against the oracle running the reference's per-instance loop, over a random sequence: frames
    from jumping poses, detections that come and go, steps without a preview, previews of a subset from perturbed cameras, and
    per-volume calls in between (a voxel GC pass, a reset, a host-buffer image: each has to flush what the batch deferred)."""
    import torch
    monkeypatch.setenv("DSR_PIPELINED_VIEW", "0")
    from dynslam_amd.engine import Batch, EngineCore, OutOfBlocksError, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    rng = np.random.default_rng(7000 + seed)
    W, H = [(256, 80), (320, 96), (251, 83)][rng.integers(3)]
    n_inst = int(rng.integers(2, 7))
    nv = int(rng.integers(2, min(n_inst, 5) + 1))
    owned = {int(k): v for v, k in enumerate(sorted(rng.choice(n_inst, nv, replace=False)))}   # instance -> volume of the batch
    inst_kw = dict(voxel_size=0.035, mu=1.0, max_w=int(rng.choice([3, 100])), view_frustum_min=0.2, view_frustum_max=float(rng.choice([12.0, 30.0])),
                   sdf_local_block_num=int(rng.choice([300, 2000, 7142])), hash_bucket_num=0x100000, excess_list_size=0x20000)
    view_kw = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0, sdf_local_block_num=64,
                   hash_bucket_num=64, excess_list_size=64)
    sc = StreetScene(W, H, n_instances=n_inst, noise_px=float(rng.choice([0.0, 0.4])))
    calib = make_calib(*sc.intrinsics(), W, H)
    sync_status = int(rng.integers(2))
    bs = EngineCore(default_settings(**view_kw, sync_status=sync_status), calib)
    bi = [EngineCore(default_settings(**inst_kw, sync_status=sync_status), calib) for _ in range(nv)]
    os_ = OracleEngine(oracle_settings(**view_kw), calib, threads=8)
    oi = [OracleEngine(oracle_settings(**inst_kw), calib, threads=8) for _ in range(nv)]
    batch = Batch(bs, bi)
    dev = torch.device("cuda", 0)
    out = [(torch.zeros((H * W, 4), dtype=torch.uint8, device=dev), torch.zeros((H * W,), dtype=torch.float32, device=dev)) for _ in range(nv)]
    frame, log = int(rng.integers(0, 3)), []
    fused = [False] * nv
    try:
        for step in range(int(rng.integers(6, 12))):
            frame = max(0, frame + int(rng.choice([1, 1, 1, 2, 4, -2])))
            rgba, d, T, inst_id = sc.frame(frame)
            masks = []
            for k in range(n_inst):
                ys, xs = np.nonzero(inst_id == k)
                if len(ys) == 0 or rng.random() < 0.2:       # no detection of this instance in this frame
                    continue
                y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
                m = np.ascontiguousarray((inst_id[y0:y1, x0:x1] == k).astype(np.uint8))
                rel = (np.linalg.inv(sc.instance_pose(k, frame).astype(np.float64)) @ T.astype(np.float64)).astype(np.float32)
                masks.append((k, int(x0), int(y0), m, rel))
            log.append(("frame", frame, [k for k, *_ in masks]))
            for e in (bs, os_):
                e.update_view(rgba, d)
            if masks:
                mt = [torch.from_numpy(m).to(dev) for _, _, _, m, _ in masks]
                items = []
                for (k, x0, y0, m, rel), t in zip(masks, mt):
                    mk = (t.data_ptr(), m.shape[1], m.shape[0])
                    items.append((owned.get(k, -1), mk if k in owned else None, x0, y0, mk, x0, y0, rel if k in owned else None))
                status = batch.fuse(items, want_status=bool(sync_status))
                oob = set()
                for k, x0, y0, m, rel in masks:               # the reference's loop on the oracle
                    if k in owned:
                        os_.extract_silhouette(oi[owned[k]], m, x0, y0)
                    os_.remove_silhouette(m, x0, y0)
                    if k in owned:
                        e = oi[owned[k]]
                        e.set_pose_inv_m(rel)
                        try:
                            e.process_frame()
                        except OutOfBlocksError:
                            oob.add(owned[k])
                        e.prepare()
                        fused[owned[k]] = True
                if sync_status:
                    assert {items[i][0] for i, s in enumerate(status) if s != 0} == oob, (log, status, oob)
                del mt
            vb, vo = bs.get_view(), os_.get_view()
            assert np.array_equal(vb[0], vo[0]) and np.array_equal(vb[1], vo[1]), (log, "blanked main view differs")
            touched = [owned[k] for k, *_ in masks if k in owned]
            # a per-volume call between the batch's two calls: it has to see (and flush) what the batch deferred
            between = rng.choice(["none", "none", "decay", "reset", "image"], p=[0.35, 0.25, 0.15, 0.1, 0.15])
            v0 = int(rng.integers(nv))
            if between == "decay":
                args = (int(rng.choice([1, 5, 100])), int(rng.choice([0, 1, 2])), bool(rng.random() < 0.3))
                log.append(("decay", v0) + args)
                for e in (bi[v0], oi[v0]):
                    e.decay(*args)
            elif between == "reset":
                log.append(("reset", v0))
                for e in (bi[v0], oi[v0]):
                    e.reset_scene()
                fused[v0] = False
            elif between == "image" and fused[v0]:
                log.append(("image", v0))
                a = bi[v0].get_image(_capi.IMAGE_SCENERAYCAST)[0]
                b = oi[v0].get_image(_capi.IMAGE_SCENERAYCAST)[0]
                assert np.array_equal(a, b), (log, "tracking render image differs")
            for v in touched:
                gb, go = bi[v].get_view(), oi[v].get_view()
                assert np.array_equal(gb[0], go[0]) and np.array_equal(gb[1], go[1]), (log, f"cut-out of volume {v} differs")
            for v in range(nv):
                assert_scene_equal(bi[v], oi[v], voxels=False)
                if v in touched and not (between == "reset" and v == v0):
                    empty = oi[v].get_stats().no_visible_blocks == 0
                    try:
                        assert_render_equal(bi[v], oi[v], skip=("minmax",) if empty else ())
                    except AssertionError as ex:
                        raise AssertionError(f"volume {v}: {ex}") from None
            # the previews: a subset of the volumes that hold something, from perturbed cameras; some steps have none
            if rng.random() < 0.75:
                cand = [(k, rel) for k, _, _, _, rel in masks if k in owned and fused[owned[k]]]
                picks = [c for c in cand if rng.random() < 0.8]
                ritems = []
                for k, rel in picks:
                    M = np.linalg.inv(np.asarray(rel, np.float64))
                    M[:3, 3] += rng.normal(0, 0.03, 3)
                    ritems.append((owned[k], M.astype(np.float32)))
                if ritems:
                    log.append(("render", [v for v, _ in ritems]))
                    batch.render([(v, M, out[v][0].data_ptr(), out[v][1].data_ptr()) for v, M in ritems])
                    bs.sync()
                    for v, M in ritems:
                        oc, od = oi[v].get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
                        assert np.array_equal(out[v][1].cpu().numpy().reshape(H, W), od), (log, f"preview depth of volume {v}")
                        assert np.array_equal(out[v][0].cpu().numpy().reshape(H, W, 4), oc), (log, f"preview colour of volume {v}")
                        assert np.array_equal(bi[v].dump_visible_list(True), oi[v].dump_visible_list(True)), log
                        assert_render_equal(bi[v], oi[v], freeview=True)
        for v in range(nv):
            assert_scene_equal(bi[v], oi[v])   # every voxel
    except AssertionError as ex:
        raise AssertionError(f"batch seed {seed}: {W}x{H}, {n_inst} instances, volumes {owned}, {inst_kw}\ncalls: {log}\n{ex}") from None
    finally:
        batch.close()
        for e in [bs, os_] + bi + oi:
            e.close()


def _host_seeds():
    spec = os.environ.get("DSR_FUZZ_HOST_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return [1, 2, 3, 4, 5, 6]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _host_seeds())
def test_random_host_pipelines_equal_the_oracle(hip_api, monkeypatch, seed):
    """The reference host's frame (InfiniTamDriver + InstanceReconstructor: view upload, per detection ProcessSilhouette +
    RemoveSilhouette + the instance's ProcessFrame / Prepare, then the map's; previews in between) through the per-engine calls on
    engines a host WAITS on (sync_status), with the view pipeline in each of its forms (the views, their double buffers and the
    streams the forms share are the asynchronous part of the library): cut-outs as one call or two, masks on the host or in HBM,
    detections that come and go, previews and view read-backs at random places, a GC pass on the map now and then — views, scenes and
    render states against the oracle after every frame."""
    import torch
    from dynslam_amd.engine import EngineCore, OutOfBlocksError, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    rng = np.random.default_rng(9000 + seed)
    pv = int(rng.integers(0, 4))
    monkeypatch.setenv("DSR_PIPELINED_VIEW", str(pv))
    if rng.random() < 0.25:
        monkeypatch.setenv("DSR_FORCE_PEER_PATH", "1")
    W, H = [(256, 80), (320, 96), (251, 83)][rng.integers(3)]
    n_inst = int(rng.integers(1, 4))
    map_kw = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                  sdf_local_block_num=int(rng.choice([9000, 20000])), hash_bucket_num=0x8000, excess_list_size=0x2000)
    inst_kw = dict(voxel_size=0.035, mu=1.0, max_w=int(rng.choice([3, 100])), view_frustum_min=0.2, view_frustum_max=30.0,
                   sdf_local_block_num=int(rng.choice([600, 7142])), hash_bucket_num=0x100000, excess_list_size=0x20000)
    sc = StreetScene(W, H, n_instances=n_inst, noise_px=float(rng.choice([0.0, 0.4])))
    calib = make_calib(*sc.intrinsics(), W, H)
    gm = EngineCore(default_settings(**map_kw, sync_status=1), calib)
    gi = [EngineCore(default_settings(**inst_kw, sync_status=1), calib) for _ in range(n_inst)]
    om = OracleEngine(oracle_settings(**map_kw), calib, threads=8)
    oi = [OracleEngine(oracle_settings(**inst_kw), calib, threads=8) for _ in range(n_inst)]
    shared = pv == 0 and rng.random() < 0.5
    if shared:
        for e in gi:
            e.share_stream(gm)
    dev = torch.device("cuda", 0)
    frame, log = int(rng.integers(0, 3)), [("pipelined view", pv, "shared" if shared else "")]
    fused = [False] * n_inst
    keep = []
    try:
        for step in range(int(rng.integers(5, 10))):
            frame = max(0, frame + int(rng.choice([1, 1, 1, 2, 3, -2])))
            rgba, d, T, inst_id = sc.frame(frame)
            log.append(("frame", frame))
            for main in (gm, om):
                main.update_view(rgba, d)
            for k in range(n_inst):
                ys, xs = np.nonzero(inst_id == k)
                if len(ys) == 0 or rng.random() < 0.2:
                    continue
                y0, y1, x0, x1 = int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1
                mask = np.ascontiguousarray((inst_id[y0:y1, x0:x1] == k).astype(np.uint8))
                how = rng.choice(["two", "split", "two_dev", "split_dev"])
                log.append((how, k))
                if how == "two":
                    gm.extract_silhouette(gi[k], mask, x0, y0); gm.remove_silhouette(mask, x0, y0)
                elif how == "split":
                    gm.split_silhouette(gi[k], mask, x0, y0)
                else:
                    mt = torch.from_numpy(mask).to(dev)
                    keep.append(mt)
                    if how == "two_dev":
                        gm.extract_silhouette_dev(gi[k], mt.data_ptr(), x0, y0, mask.shape[1], mask.shape[0])
                        gm.remove_silhouette_dev(mt.data_ptr(), x0, y0, mask.shape[1], mask.shape[0])
                    else:
                        gm.split_silhouette_dev(gi[k], mt.data_ptr(), x0, y0, mask.shape[1], mask.shape[0])
                om.extract_silhouette(oi[k], mask, x0, y0); om.remove_silhouette(mask, x0, y0)
                rel = (np.linalg.inv(sc.instance_pose(k, frame).astype(np.float64)) @ T.astype(np.float64)).astype(np.float32)
                raised = []
                for e in (gi[k], oi[k]):
                    e.set_pose_inv_m(rel)
                    try:
                        e.process_frame(); raised.append(False)
                    except OutOfBlocksError:
                        raised.append(True)
                    e.prepare()
                assert raised[0] == raised[1], (log, "out-of-blocks status differs")
                fused[k] = True
                if rng.random() < 0.4:      # the instance's preview right away (PrepareNextStep of the reference host)
                    M = np.linalg.inv(np.asarray(rel, np.float64)); M[:3, 3] += rng.normal(0, 0.03, 3); M = M.astype(np.float32)
                    t = RENDER_TYPES[rng.integers(len(RENDER_TYPES))]
                    log.append(("preview", k, int(t)))
                    cg, dg = gi[k].get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                    co, do = oi[k].get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                    assert np.array_equal(dg, do) and np.array_equal(cg, co), (log, "instance preview differs")
            raised = []
            for main in (gm, om):
                main.set_pose_inv_m(T)
                try:
                    main.process_frame(); raised.append(False)
                except OutOfBlocksError:
                    raised.append(True)
                main.prepare()
            assert raised[0] == raised[1], (log, "out-of-blocks status of the map differs")
            if rng.random() < 0.2:
                args = (int(rng.choice([1, 3])), int(rng.choice([0, 2])), False)
                log.append(("decay",) + args)
                for main in (gm, om):
                    main.decay(*args)
            if rng.random() < 0.5:
                M = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
                t = RENDER_TYPES[rng.integers(len(RENDER_TYPES))]
                log.append(("map preview", int(t)))
                cg, dg = gm.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                co, do = om.get_image(t, pose_m=M, want_rgba=True, want_depth=True)
                assert np.array_equal(dg, do) and np.array_equal(cg, co), (log, "map preview differs")
            vg, vo = gm.get_view(), om.get_view()
            assert np.array_equal(vg[0], vo[0]) and np.array_equal(vg[1], vo[1]), (log, "blanked main view differs")
            assert_scene_equal(gm, om, voxels=False)
            assert_render_equal(gm, om, skip=("minmax",) if om.get_stats().no_visible_blocks == 0 else ())
            for k in range(n_inst):
                if not fused[k]:
                    continue
                vg, vo = gi[k].get_view(), oi[k].get_view()
                assert np.array_equal(vg[0], vo[0]) and np.array_equal(vg[1], vo[1]), (log, f"view of instance {k} differs")
                assert_scene_equal(gi[k], oi[k], voxels=False)
                assert_render_equal(gi[k], oi[k], skip=("minmax",) if oi[k].get_stats().no_visible_blocks == 0 else ())
            keep.clear()
        assert_scene_equal(gm, om)
        for k in range(n_inst):
            assert_scene_equal(gi[k], oi[k])
    except AssertionError as ex:
        raise AssertionError(f"host seed {seed}: {W}x{H}, {n_inst} instances, map {map_kw['sdf_local_block_num']} blocks, "
                             f"instances {inst_kw['sdf_local_block_num']} blocks max_w {inst_kw['max_w']}\ncalls: {log}\n{ex}") from None
    finally:
        for e in [gm, om] + gi + oi:
            e.close()


def _scene_seeds():
    spec = os.environ.get("DSR_FUZZ_SCENE_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return [1, 2, 3, 4]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _scene_seeds())
def test_random_sharded_scenes_equal_the_oracle(hip_api, seed):
    """`ShardedScene` — what `bench.py` runs: the volumes of one GPU as a batch (or one by one), renders written straight into the
    exchange's slots, the composite over the exchange's own target — on the HIP engines against the same class driving the oracle
    (CPU tensors, the oracle's composite), over a random sequence: jumping poses, detections that come and go (empty layers), steps
    without a preview, track ids in any order, tint / dimming drawn per step.  The composited preview of every step, bit for bit."""
    import torch
    from bench import _gen_frame
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.multigpu import ShardedScene
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, load_api, oracle_settings
    rng = np.random.default_rng(11000 + seed)
    W, H = [(256, 80), (320, 96)][rng.integers(2)]
    has_static = bool(rng.random() < 0.5)
    n_inst = int(rng.integers(1, 7))
    n_volumes = n_inst + (1 if has_static else 0)
    static_kw = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                     sdf_local_block_num=40000, hash_bucket_num=0x10000, excess_list_size=0x4000)
    inst_kw = dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                   sdf_local_block_num=7142, hash_bucket_num=int(rng.choice([0x10000, 0x100000])), excess_list_size=0x4000)
    # (the reference host throws when a volume runs out of blocks, and so does the oracle: exhaustion is the other tests' subject)
    kinds = {"static": static_kw, "instance": inst_kw, "view": dict(static_kw, sdf_local_block_num=64, hash_bucket_num=64, excess_list_size=64)}
    sc = StreetScene(W, H, n_instances=n_inst)
    calib = make_calib(*sc.intrinsics(), W, H)
    torch.cuda.set_device(0)
    use_batch = bool(rng.random() < 0.7)
    sg = ShardedScene(lambda kind: EngineCore(default_settings(**kinds[kind], device=0, sync_status=0), calib), W, H, n_volumes, 1, 0,
                      torch.device("cuda", 0), None, has_static=has_static, use_batch=use_batch)
    so = ShardedScene(lambda kind: OracleEngine(oracle_settings(**kinds[kind]), calib, threads=8), W, H, n_volumes, 1, 0, torch.device("cpu"),
                      None, has_static=has_static)
    so.exchange.host_api = load_api()
    frame, log = int(rng.integers(0, 3)), [("static" if has_static else "no static", n_inst, "batch" if use_batch else "loop")]
    keep = []
    try:
        for step in range(int(rng.integers(4, 9))):
            frame = max(0, frame + int(rng.choice([1, 1, 1, 2, 3, -2])))
            rgba, d, T, masks = _gen_frame((W, H, frame, n_inst))
            masks = [m for m in masks if rng.random() < 0.8]        # detections come and go
            log.append(("frame", frame, [m[0] for m in masks]))
            keep = [torch.from_numpy(rgba).cuda(), torch.from_numpy(d).cuda(), [torch.from_numpy(np.ascontiguousarray(m[3])).cuda() for m in masks]]
            dev_masks = [(k, x0, y0, (t.data_ptr(), m.shape[1], m.shape[0]), rel) for (k, x0, y0, m, rel), t in zip(masks, keep[2])]
            sg.step(keep[0].data_ptr(), keep[1].data_ptr(), T, dev_masks)
            so.step(rgba, d, T, masks)
            if rng.random() < 0.25:
                sg.sync()
                continue                                             # a step without a preview
            M = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
            inst_m = {k: np.linalg.inv(rel.astype(np.float64)).astype(np.float32) for k, _, _, _, rel in masks}
            ids = rng.permutation(n_inst) * 3 + 1
            track_ids = {k: int(ids[k]) for k in range(n_inst)}
            tint, dim = float(rng.choice([1.0, 0.35])), bool(rng.random() < 0.5)
            log.append(("preview", tint, dim))
            og = sg.preview(M, inst_m, track_ids, tint_strength=tint, dim_background=dim)
            oo = so.preview(M, inst_m, track_ids, tint_strength=tint, dim_background=dim)
            sg.sync(); torch.cuda.synchronize()
            dg, do = og[1].cpu().numpy(), oo[1].cpu().numpy()
            cg, co = og[0].cpu().numpy(), oo[0].cpu().numpy()
            assert np.array_equal(dg, do), (log, f"composited depth differs at {(dg != do).sum()} pixels")
            assert np.array_equal(cg, co), (log, f"composited colour differs at {(cg != co).any(axis=-1).sum()} pixels")
    except AssertionError as ex:
        raise AssertionError(f"scene seed {seed}: {W}x{H}\ncalls: {log}\n{ex}") from None
    finally:
        sg.close(); so.close()
