"""-m gpu: the volume batch (dsr_batch_*, k_batch.h) — every kernel of an instance frame launched once for all instance volumes of
a GPU — against the per-volume calls on the HIP engine and against the oracle running the reference's loop
(InstanceReconstructor.cpp:238-263,569-700,956-986): views, hash tables, lists, voxels, range images, raycasts, ICP maps, preview
renders, bit for bit."""
import numpy as np
import pytest

from dynslam_amd import _capi
from dynslam_amd.engine import make_calib
from dynslam_amd.synth import StreetScene

pytestmark = pytest.mark.gpu

INSTANCE = dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                sdf_local_block_num=7142, hash_bucket_num=0x100000, excess_list_size=0x20000)
VIEW = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
            sdf_local_block_num=64, hash_bucket_num=64, excess_list_size=64)


def _engines(factory, calib, n, **kw):
    return factory(dict(VIEW, **kw), calib), [factory(dict(INSTANCE, **kw), calib) for _ in range(n)]


def _hip(settings, calib):
    from dynslam_amd.engine import EngineCore, default_settings
    return EngineCore(default_settings(**settings), calib)


def _orc(settings, calib):
    from oracle.oracle import OracleEngine, oracle_settings
    return OracleEngine(oracle_settings(**settings), calib, threads=8)


@pytest.mark.parametrize("size,sync_status", [((320, 96), 1), ((320, 96), 0), ((1242, 375), 0)])
def test_batch_equals_the_per_volume_calls_and_the_oracle(hip_api, monkeypatch, size, sync_status):
    import torch
    monkeypatch.setenv("DSR_PIPELINED_VIEW", "0")  # (engines a host waits on get a view pipeline by default; a batch has one stream)
    from dynslam_amd.engine import Batch
    from tests.common import assert_render_equal, assert_scene_equal
    W, H = size
    n_inst = 4
    sc = StreetScene(W, H, n_instances=n_inst)
    calib = make_calib(*sc.intrinsics(), W, H)
    bs, bi = _engines(_hip, calib, 3, sync_status=sync_status)   # batch-driven: volumes for instances 0, 1, 3 (2 lives "elsewhere")
    ps, pi = _engines(_hip, calib, 3, sync_status=sync_status)   # the same through the per-volume calls
    os_, oi = _engines(_orc, calib, 3)
    owned = {0: 0, 1: 1, 3: 2}
    batch = Batch(bs, bi)
    dev = torch.device("cuda", 0)
    out = [(torch.zeros((H * W, 4), dtype=torch.uint8, device=dev), torch.zeros((H * W,), dtype=torch.float32, device=dev)) for _ in range(3)]
    frames = 6 if W < 1000 else 4
    for i in range(frames):
        rgba, d, T, inst_id = sc.frame(i)
        masks = []
        for k in range(n_inst):
            ys, xs = np.nonzero(inst_id == k)
            if len(ys) == 0 or (k == 1 and i == 2):  # instance 1 has no detection in frame 2
                continue
            y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
            m = np.ascontiguousarray((inst_id[y0:y1, x0:x1] == k).astype(np.uint8))
            rel = (np.linalg.inv(sc.instance_pose(k, i).astype(np.float64)) @ T.astype(np.float64)).astype(np.float32)
            masks.append((k, int(x0), int(y0), m, rel))
        assert masks, "the scene must show instances"
        for e in (bs, ps, os_):
            e.update_view(rgba, d)
        # --- batch
        mt = [torch.from_numpy(m).to(dev) for _, _, _, m, _ in masks]
        items = []
        for (k, x0, y0, m, rel), t in zip(masks, mt):
            mk = (t.data_ptr(), m.shape[1], m.shape[0])
            items.append((owned.get(k, -1), mk if k in owned else None, x0, y0, mk, x0, y0, rel if k in owned else None))
        status = batch.fuse(items, want_status=bool(sync_status))
        if sync_status:
            assert all(s == 0 for s in status)
        # --- per volume (HIP) and the reference's loop (oracle)
        for main, inst in ((ps, pi), (os_, oi)):
            for k, x0, y0, m, rel in masks:
                if k in owned:
                    main.extract_silhouette(inst[owned[k]], m, x0, y0)
                main.remove_silhouette(m, x0, y0)
                if k in owned:
                    e = inst[owned[k]]
                    e.set_pose_inv_m(rel)
                    e.process_frame()
                    e.prepare()
        bs.sync()
        vb, vo = bs.get_view(), os_.get_view()
        assert np.array_equal(vb[0], vo[0]) and np.array_equal(vb[1], vo[1]), f"frame {i}: blanked main view differs"
        for v in range(3):
            gb, go = bi[v].get_view(), oi[v].get_view()
            assert np.array_equal(gb[0], go[0]) and np.array_equal(gb[1], go[1]), f"frame {i}: cut-out of volume {v} differs"
            assert_scene_equal(bi[v], oi[v], voxels=(i == frames - 1))
            assert_scene_equal(bi[v], pi[v], voxels=False)
            assert_render_equal(bi[v], oi[v])
        # --- the preview renders of the volumes with a detection, into HBM buffers
        visible = [(owned[k], np.linalg.inv(np.asarray(rel, np.float64)).astype(np.float32)) for k, _, _, _, rel in masks if k in owned]
        batch.render([(v, M, out[v][0].data_ptr(), out[v][1].data_ptr()) for v, M in visible])
        bs.sync()
        for v, M in visible:
            oc, od = oi[v].get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, want_rgba=True, want_depth=True)
            assert np.array_equal(out[v][0].cpu().numpy().reshape(H, W, 4), oc), f"frame {i}: preview colour of volume {v}"
            assert np.array_equal(out[v][1].cpu().numpy().reshape(H, W), od), f"frame {i}: preview depth of volume {v}"
            assert np.array_equal(bi[v].dump_visible_list(True), oi[v].dump_visible_list(True))
            assert_render_equal(bi[v], oi[v], freeview=True)
        # per-volume calls in between stay valid: the voxel GC swaps list buffers the batch must pick up
        if i == 3:
            for e in (bi[0], pi[0], oi[0]):
                e.decay(1, 0, False)
            assert_scene_equal(bi[0], oi[0], voxels=False)
    assert oi[0].get_stats().no_visible_blocks > 0
    batch.close()
    for e in [bs, ps, os_] + bi + pi + oi:
        e.close()


def test_batch_argument_errors(hip_api, monkeypatch):
    from dynslam_amd.engine import Batch, DsrError
    monkeypatch.setenv("DSR_PIPELINED_VIEW", "0")
    W, H = 320, 96
    sc = StreetScene(W, H)
    calib = make_calib(*sc.intrinsics(), W, H)
    src, vols = _engines(_hip, calib, 2)
    big = _hip(dict(VIEW, sdf_local_block_num=40000, hash_bucket_num=0x10000, excess_list_size=0x4000), calib)
    with pytest.raises(DsrError):
        Batch(src, [vols[0], big])          # not an instance-sized volume
    with pytest.raises(DsrError):
        Batch(src, [vols[0], vols[0]])      # listed twice
    b = Batch(src, vols)
    with pytest.raises(DsrError):
        b.fuse([(0, (1, 4, 4), 0, 0, None, 0, 0, np.eye(4, dtype=np.float32))])  # no view yet
    b.close()
    for e in [src, big] + vols:
        e.close()
