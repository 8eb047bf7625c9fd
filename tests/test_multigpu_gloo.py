"""world_size-2 (and 3) gloo tests of the one-volume-per-GPU sharding and of the all-gather
that feeds the fused preview: every rank must end up with every instance layer, indexed so
that compositing in ascending track id reproduces the serial host loop.  The composite
arithmetic itself is checked against the oracle's restatement (CPU here; the HIP kernel is
checked in test_gpu_composite.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dynslam_amd.multigpu import PreviewExchange, max_local_instances, volume_owner, volumes_of_rank


def test_volume_assignment():
    assert [volume_owner(v, 1) for v in range(5)] == [0, 0, 0, 0, 0]
    # config 4: static map on GPU0 + 7 instance volumes on GPUs 1-7
    assert [volume_owner(v, 8) for v in range(8)] == [0, 1, 2, 3, 4, 5, 6, 7]
    assert [volume_owner(v, 3) for v in range(6)] == [0, 1, 2, 1, 2, 1]
    for world in (1, 2, 3, 8):
        for n in (1, 5, 8, 12):
            owned = sorted(v for r in range(world) for v in volumes_of_rank(r, n, world))
            assert owned == list(range(n))
    assert max_local_instances(8, 8) == 1 and max_local_instances(5, 2) == 4 and max_local_instances(6, 3) == 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _layer(k, P):
    rng = np.random.default_rng(100 + k)
    d = rng.uniform(2.0, 9.0, P).astype(np.float32)
    d[rng.random(P) < 0.5] = 0.0
    c = rng.integers(0, 256, (P, 4)).astype(np.uint8)
    return c, d


def _worker(rank, world, port, n_volumes, P, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = PreviewExchange(P, n_volumes, world, rank, torch.device("cpu"))
    for slot, k in enumerate(ex.local_instances):
        c, d = _layer(k, P)
        ex.local_rgba[slot] = torch.from_numpy(c)
        ex.local_depth[slot] = torch.from_numpy(d)
    ex.gather()
    track_ids = {k: 40 - 3 * k for k in range(n_volumes - 1)}  # descending: order must follow ids, not ranks
    layers, tids = ex.ordered_layers(track_ids)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), depth=ex.all_depth.numpy(), rgba=ex.all_rgba.numpy(),
             layers=np.array(layers), tids=np.array(tids))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_volumes", [(2, 4), (3, 6)])
def test_all_gather_and_composite_order(tmp_path, oracle_lib, world, n_volumes):
    P = 257
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_volumes, P, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # every rank holds identical gathered buffers and ordering
    for r in res[1:]:
        for k in ("depth", "rgba", "layers", "tids"):
            assert np.array_equal(r[k], res[0][k])
    layers, tids = res[0]["layers"], res[0]["tids"]
    assert list(tids) == sorted(tids) and len(layers) == n_volumes - 1
    # layer l really is the instance whose track id is tids[l]
    inst_of_tid = {40 - 3 * k: k for k in range(n_volumes - 1)}
    for l, t in zip(layers, tids):
        c, d = _layer(inst_of_tid[int(t)], P)
        assert np.array_equal(res[0]["depth"][l], d) and np.array_equal(res[0]["rgba"][l], c)
    # compositing the gathered layers == the serial host loop over tracks in ascending id
    import ctypes as C
    bg_c, bg_d = _layer(99, P)
    lr = np.ascontiguousarray(res[0]["rgba"][layers]); ld = np.ascontiguousarray(res[0]["depth"][layers])
    t_c, t_d = bg_c.copy(), bg_d.copy()
    ids = np.ascontiguousarray(tids, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert oracle_lib.composite_instances(vp(t_c), vp(t_d), vp(lr), vp(ld), vp(ids), len(layers), P, 1.0, 1) == 0
    # independent numpy statement of CompositeColor / the dim pass
    e_c = bg_c.copy(); e_d = bg_d.copy()
    e_c[:, :3] = (e_c[:, :3].astype(np.float64) * (1.0 - float(np.float32(0.10)))).astype(np.uint8)
    pal = [(0x1f, 0x77, 0xb4), (0xff, 0x7f, 0x0e), (0x2c, 0xa0, 0x2c), (0xd6, 0x27, 0x28), (0x94, 0x67, 0xbd),
           (0x8c, 0x56, 0x4b), (0xe3, 0x77, 0xc2), (0x71, 0x71, 0x71), (0xbc, 0xbd, 0x22), (0x17, 0xbe, 0xcf)]
    for l, t in zip(layers, tids):
        s_d, s_c = res[0]["depth"][l], res[0]["rgba"][l]
        on_top = (s_d != 0) & ((e_d == 0) | (e_d > s_d))
        e_d[on_top] = s_d[on_top]
        tint = np.array(pal[int(t) % 10], dtype=np.float64)
        e_c[on_top, :3] = np.minimum(255.0, s_c[on_top, :3].astype(np.float64) * 0.5 + tint * 1.0).astype(np.uint8)
    assert np.array_equal(t_d, e_d) and np.array_equal(t_c, e_c)


# ---------------------------------------------------------------------------------------------
# configs[3] end to end on CPU: ShardedScene (the class bench.py --gpus N runs on RCCL) driven with
# the CPU oracle as the engine and gloo as the backend — the static map on rank 0, the instance
# volumes on the other ranks, fusion + render + all-gather + composite per frame — must give the
# same composited preview as all volumes in one process.

SH_W, SH_H, SH_FRAMES = 256, 80, 3
SH_STATIC = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
                 sdf_local_block_num=40000, hash_bucket_num=0x10000, excess_list_size=0x4000)
SH_INST = dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
               sdf_local_block_num=7142, hash_bucket_num=0x10000, excess_list_size=0x4000)
SH_VIEW = dict(SH_STATIC, sdf_local_block_num=64, hash_bucket_num=64, excess_list_size=64)


def _sharded_run(world, rank, n_volumes, group=None, has_static=True, hip=False):
    from bench import _gen_frame
    from dynslam_amd.engine import make_calib
    from dynslam_amd.multigpu import ShardedScene
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, load_api, oracle_settings
    n_inst = n_volumes - 1 if has_static else n_volumes
    sc = StreetScene(SH_W, SH_H, n_instances=n_inst)
    calib = make_calib(*sc.intrinsics(), SH_W, SH_H)
    kinds = {"static": SH_STATIC, "instance": SH_INST, "view": SH_VIEW}
    if hip:  # the real engines, every rank on cuda:0 (a one-GPU box), collectives over gloo staged through host memory
        from dynslam_amd.engine import EngineCore, default_settings
        torch.cuda.set_device(0)
        scene = ShardedScene(lambda kind: EngineCore(default_settings(**kinds[kind], device=0, sync_status=0), calib), SH_W, SH_H,
                             n_volumes, world, rank, torch.device("cuda", 0), group, has_static=has_static)
    else:
        scene = ShardedScene(lambda kind: OracleEngine(oracle_settings(**kinds[kind]), calib), SH_W, SH_H, n_volumes, world, rank,
                             torch.device("cpu"), group, has_static=has_static)
        scene.exchange.host_api = load_api()  # CPU composite = the oracle's restatement (tests only)
    track_ids = {k: 7 + 2 * k for k in range(n_inst)}
    out = None
    for i in range(SH_FRAMES):
        rgba, d, T, masks = _gen_frame((SH_W, SH_H, i, n_inst))
        if hip:  # frames and masks resident in HBM, as bench.py hands them over
            keep = (torch.from_numpy(rgba).cuda(), torch.from_numpy(d).cuda(), [torch.from_numpy(np.ascontiguousarray(m[3])).cuda() for m in masks])
            dev_masks = [(k, x0, y0, (t.data_ptr(), m.shape[1], m.shape[0]), rel) for (k, x0, y0, m, rel), t in zip(masks, keep[2])]
            scene.step(keep[0].data_ptr(), keep[1].data_ptr(), T, dev_masks)
            scene.sync()
        else:
            scene.step(rgba, d, T, masks)
        M = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
        inst_m = {k: np.linalg.inv(rel.astype(np.float64)).astype(np.float32) for k, _, _, _, rel in masks}
        out = scene.preview(M, inst_m, track_ids)
    res = None
    if hip:
        scene.sync()
        torch.cuda.synchronize()
    if rank == 0:
        res = (out[0].cpu().numpy().copy(), out[1].cpu().numpy().copy(), scene.exchange.all_depth.cpu().numpy().copy())
    scene.close()
    return res


def _sharded_worker(rank, world, port, n_volumes, out_dir, has_static=True, hip=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _sharded_run(world, rank, n_volumes, has_static=has_static, hip=hip)
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), rgba=res[0], depth=res[1], layers=res[2])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_volumes", [(2, 3), (3, 4), (8, 8)])  # (8, 8): configs[3] as the driver's 8-GPU run shards it
def test_sharded_scene_equals_single_process(tmp_path, oracle_lib, world, n_volumes):
    mp.spawn(_sharded_worker, args=(world, _free_port(), n_volumes, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    rgba1, depth1, layers1 = _sharded_run(1, 0, n_volumes)
    assert (depth1 > 0).mean() > 0.3
    # the instance layers really arrived (non-empty) and the composite used them
    # (boxes 4.. of the synthetic street start beyond the 20 m depth clip: at most 4 instances are in range in 3 frames)
    assert sum(int((got["layers"][l] > 0).any()) for l in range(got["layers"].shape[0])) >= min(n_volumes - 2, 4)
    assert np.array_equal(got["depth"], depth1)
    assert np.array_equal(got["rgba"], rgba1)


@pytest.mark.parametrize("world,n_volumes", [(2, 2), (3, 4), (8, 8)])  # (8, 8): north_star's "8 concurrent instance volumes"
def test_instance_volumes_sharded_equals_single_process(tmp_path, oracle_lib, world, n_volumes):
    """The scaling workload of `bench.py --gpus N`: N instance volumes and NO static map, instance k on rank k mod world,
    composited over an empty frame — same preview as all volumes in one process, bit for bit."""
    mp.spawn(_sharded_worker, args=(world, _free_port(), n_volumes, str(tmp_path), False), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    rgba1, depth1, layers1 = _sharded_run(1, 0, n_volumes, has_static=False)
    assert (depth1 > 0).any() and (depth1 == 0).any()  # instances over an EMPTY frame: most pixels stay empty
    assert sum(int((got["layers"][l] > 0).any()) for l in range(got["layers"].shape[0])) >= min(n_volumes, 4) - 1
    assert np.array_equal(got["depth"], depth1)
    assert np.array_equal(got["rgba"], rgba1)
    # where a layer hit, the composite holds the nearest hit of all layers
    hit = layers1 > 0
    nearest = np.where(hit, layers1, np.inf).min(axis=0)
    assert np.array_equal(np.where(np.isfinite(nearest), nearest, 0.0).astype(np.float32), depth1)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_volumes,has_static", [(2, 3, True), (3, 3, False)])
def test_sharded_hip_engines_equal_single_process_and_oracle(tmp_path, hip_api, oracle_lib, world, n_volumes, has_static):
    """The N > 1 path with the REAL engines: `world` processes, each with its own HIP engines on cuda:0 (a one-GPU box),
    device-resident frames and masks, renders written into the exchange slots, gloo staging for the collective, the HIP
    composite on rank 0 — the composited preview equals the one-process HIP run AND the one-process oracle run, bit for bit."""
    mp.spawn(_sharded_worker, args=(world, _free_port(), n_volumes, str(tmp_path), has_static, True), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    rgba1, depth1, layers1 = _sharded_run(1, 0, n_volumes, has_static=has_static, hip=True)
    rgba0, depth0, layers0 = _sharded_run(1, 0, n_volumes, has_static=has_static, hip=False)
    assert (depth0 > 0).any()
    assert np.array_equal(depth1, depth0) and np.array_equal(rgba1, rgba0)  # one process: HIP == oracle
    assert np.array_equal(got["depth"], depth0) and np.array_equal(got["rgba"], rgba0)  # sharded HIP == oracle
