"""One-volume-per-GPU sharding and the fused-preview exchange.

The reference drives every volume (static map + one InfiniTamDriver per tracked car,
InstanceReconstructor.cpp:382-389) sequentially on one GPU.  Volumes share no data during
allocate / integrate / raycast / decay, so they shard by volume: the static map lives on
rank 0, instance k on rank 1 + (k mod (world-1)).  The only exchange is the fused preview
(CompositeInstances, InstanceReconstructor.cpp:933-990): every rank renders its volumes from
the same free camera, the per-volume depth (f32) and colour (RGBA) buffers are ALL-GATHERED
(RCCL over xGMI; `torch.distributed` backend "nccl" on ROCm, "gloo" in the CPU tests) and
each rank z-composites them on its GPU with `dsr_composite_instances_dev`.

torch is plumbing here (process group, device buffers); the arithmetic is the HIP library.
"""
import ctypes as C

import numpy as np


def volume_owner(volume_index, world_size):
    """volume 0 = static map, volume 1+k = instance k  ->  owning rank."""
    if world_size <= 1 or volume_index == 0:
        return 0
    return 1 + ((volume_index - 1) % (world_size - 1))


def volumes_of_rank(rank, n_volumes, world_size):
    return [v for v in range(n_volumes) if volume_owner(v, world_size) == rank]


def max_local_instances(n_volumes, world_size):
    """Largest number of INSTANCE volumes any rank owns (the all-gather slot count)."""
    counts = [sum(1 for v in volumes_of_rank(r, n_volumes, world_size) if v > 0) for r in range(max(1, world_size))]
    return max(counts) if counts else 0


class PreviewExchange:
    """All-gather of per-instance raycast buffers + ordering for the composite.

    Every rank contributes `slots` layers (its instance renders, padded with empty layers:
    depth 0 = miss everywhere, which never wins a pixel).  After `gather()` every rank holds
    all layers; `ordered_layers()` lists them in ascending track id, the order the host's
    `for track in GetActiveTracks()` loop composites in.
    """

    def __init__(self, n_pixels, n_volumes, world_size, rank, device, group=None, local_only=False):
        import torch
        self.torch = torch
        self.local_only = bool(local_only)  # never touch the process group (a one-rank scene inside a larger job)
        self.P = int(n_pixels)
        self.world, self.rank = int(world_size), int(rank)
        self.group = group
        self.slots = max(1, max_local_instances(n_volumes, world_size))
        self.local_instances = [v - 1 for v in volumes_of_rank(rank, n_volumes, world_size) if v > 0]
        self.device = device
        self.local_depth = torch.zeros((self.slots, self.P), dtype=torch.float32, device=device)
        self.local_rgba = torch.zeros((self.slots, self.P, 4), dtype=torch.uint8, device=device)
        self.all_depth = torch.zeros((self.world * self.slots, self.P), dtype=torch.float32, device=device)
        self.all_rgba = torch.zeros((self.world * self.slots, self.P, 4), dtype=torch.uint8, device=device)
        # layer index in the gathered buffers of every instance
        self.layer_of_instance = {}
        for r in range(self.world):
            inst = [v - 1 for v in volumes_of_rank(r, n_volumes, world_size) if v > 0]
            for s, k in enumerate(inst):
                self.layer_of_instance[k] = r * self.slots + s

    def slot_ptrs(self, local_slot):
        """Device pointers (rgba, depth) of a local slot: pass them to dsr_get_image_dev."""
        return (self.local_rgba[local_slot].data_ptr(), self.local_depth[local_slot].data_ptr())

    def gather(self):
        import torch.distributed as dist
        if self.world == 1 and (self.local_only or not (dist.is_available() and dist.is_initialized())):
            self.all_depth.copy_(self.local_depth)
            self.all_rgba.copy_(self.local_rgba)
            return
        # (with a process group the collective runs for a single rank too: `torchrun --nproc-per-node 1` exercises
        #  the RCCL path of the multi-GPU layout on a one-GPU box)
        dist.all_gather_into_tensor(self.all_depth, self.local_depth, group=self.group)
        dist.all_gather_into_tensor(self.all_rgba.view(self.world * self.slots, self.P * 4),
                                    self.local_rgba.view(self.slots, self.P * 4), group=self.group)

    def ordered_layers(self, track_id_of_instance):
        """-> (layer indices, track ids) sorted by ascending track id."""
        items = sorted((track_id_of_instance[k], self.layer_of_instance[k]) for k in self.layer_of_instance
                       if k in track_id_of_instance)
        return [l for _, l in items], [t for t, _ in items]

    def composite_into(self, target_rgba, target_depth, track_id_of_instance, tint_strength=1.0, dim_background=True,
                       host_api=None):
        """composite() for tensors on either side: GPU tensors -> dsr_composite_instances_dev on torch's
        current stream; CPU tensors (the gloo tests) -> `host_api.composite_instances` (the tests pass the
        oracle's restatement; the product has no CPU composite)."""
        if getattr(self.device, "type", "cpu") == "cuda":
            return self.composite(target_rgba, target_depth, track_id_of_instance, tint_strength, dim_background)
        api = host_api or getattr(self, "host_api", None)
        if api is None:
            raise RuntimeError("CPU tensors need host_api (tests only): there is no CPU fallback for the composite")
        layers, tids = self.ordered_layers(track_id_of_instance)
        if not layers:
            return
        lr = np.ascontiguousarray(self.all_rgba.numpy()[layers])
        ld = np.ascontiguousarray(self.all_depth.numpy()[layers])
        ids = np.asarray(tids, dtype=np.int32)
        t_c, t_d = target_rgba.numpy(), target_depth.numpy()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        st = api.composite_instances(vp(t_c), vp(t_d), vp(lr), vp(ld), vp(ids), len(layers), self.P, float(tint_strength),
                                     int(bool(dim_background)))
        if st != 0:
            raise RuntimeError("composite_instances failed")

    def composite(self, target_rgba, target_depth, track_id_of_instance, tint_strength=1.0, dim_background=True,
                  stream_ptr=None):
        """z-composite the gathered layers into target_* (torch tensors on this rank's GPU)
        with the HIP kernel behind dsr_composite_instances_dev.  No CPU fallback."""
        from .engine import load_hip_api
        api = load_hip_api()
        torch = self.torch
        layers, tids = self.ordered_layers(track_id_of_instance)
        if not layers:
            return
        idx = torch.tensor(layers, dtype=torch.long, device=self.device)
        lr = self.all_rgba.index_select(0, idx).contiguous()
        ld = self.all_depth.index_select(0, idx).contiguous()
        ids = np.asarray(tids, dtype=np.int32)
        dev_index = self.device.index if hasattr(self.device, "index") and self.device.index is not None else 0
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream(self.device).cuda_stream
        st = api.composite_instances_dev(
            dev_index, C.c_void_p(stream_ptr),
            C.c_void_p(target_rgba.data_ptr()) if target_rgba is not None else None, C.c_void_p(target_depth.data_ptr()),
            C.c_void_p(lr.data_ptr()), C.c_void_p(ld.data_ptr()), ids.ctypes.data_as(C.c_void_p), len(layers), self.P,
            float(tint_strength), int(bool(dim_background)))
        if st != 0:
            raise RuntimeError(f"dsr_composite_instances_dev failed: {api.last_error().decode()}")
        self._keepalive = (lr, ld, ids)


class ShardedScene:
    """BASELINE configs[3] as one object per rank: the static map on rank 0, instance volume k on
    rank 1 + (k mod (world-1)) (`volume_owner`), fused per frame by `step()`; `preview()` renders every
    volume from the shared camera, ALL-GATHERS the instance layers (depth f32 + RGBA) and z-composites
    them over the static map's render on rank 0 — DynSlam::GetStaticMapRaycastPreview +
    InstanceReconstructor::CompositeInstances (DynSlam.h:96-132, InstanceReconstructor.cpp:911-990)
    with the volumes on different GPUs.

    `make_engine(kind)` returns an EngineCore-like object for kind in {"static", "instance", "view"}:
    "view" is a volume-less holder of the full input frame on ranks that own instances but not the
    static map (ProcessSilhouette reads the full frame: InstanceReconstructor.cpp:59-133).  Engines
    on a GPU are driven through the "_dev" entry points and never synchronise with the host; with CPU
    tensors (the gloo tests run the CPU oracle through this same class) the host-buffer entry points
    are used.
    """

    def __init__(self, make_engine, width, height, n_volumes, world_size, rank, device, group=None, local_only=False):
        import torch
        self.torch = torch
        self.W, self.H, self.P = int(width), int(height), int(width) * int(height)
        self.world, self.rank, self.device = int(world_size), int(rank), device
        self.n_volumes = int(n_volumes)
        self.on_gpu = getattr(device, "type", "cpu") == "cuda"
        mine = volumes_of_rank(rank, n_volumes, world_size)
        self.owns_static = 0 in mine
        self.static = make_engine("static") if self.owns_static else None
        self.instances = {v - 1: make_engine("instance") for v in mine if v > 0}
        # the full frame the instance views are cut from
        self.source = self.static if self.owns_static else (make_engine("view") if self.instances else None)
        self.exchange = PreviewExchange(self.P, n_volumes, world_size, rank, device, group, local_only)
        self.target_rgba = torch.zeros((self.P, 4), dtype=torch.uint8, device=device)
        self.target_depth = torch.zeros((self.P,), dtype=torch.float32, device=device)

    def engines(self):
        out = ([self.static] if self.static is not None else []) + list(self.instances.values())
        if self.source is not None and self.source is not self.static:
            out.append(self.source)
        return out

    def close(self):
        for e in self.engines():
            e.close()

    def sync(self):
        for e in self.engines():
            e.sync()

    # -- fusion -----------------------------------------------------------------------------
    def step(self, rgba, depth_mm, static_pose, masks):
        """One frame.  rgba / depth_mm: numpy arrays (host path) or device pointers (ints) of the full
        frame, resident on this rank's GPU; static_pose: camera->world of the static map;
        masks: [(instance k, x0, y0, bbox-local uint8 mask, camera->object pose of k)] for EVERY
        instance (each rank picks its own).  Order per rank as on one GPU: cut the instance views out
        of the full frame, blank them in the static view, fuse (InstanceReconstructor.cpp:238-263,569-700)."""
        if self.source is not None:
            if isinstance(rgba, int):
                self.source.update_view_dev(rgba, depth_mm)
            else:
                self.source.update_view(rgba, depth_mm)
        for k, x0, y0, mask, rel in masks:
            ie = self.instances.get(k)
            if ie is not None:
                self.source.extract_silhouette(ie, mask, x0, y0)
            if self.source is not None:
                # every rank that holds the frame blanks every silhouette, in order, like the one main view of the
                # reference: a later instance's cut-out must not see pixels an earlier (overlapping) mask removed
                self.source.remove_silhouette(mask, x0, y0)
            if ie is not None:
                ie.set_pose_inv_m(rel)
                ie.process_frame()
                ie.prepare()
        if self.owns_static:
            self.static.set_pose_inv_m(static_pose)
            self.static.process_frame()
            self.static.prepare()

    # -- fused preview ----------------------------------------------------------------------
    def _render(self, eng, pose_m, rgba_t, depth_t):
        from . import _capi
        if self.on_gpu:
            eng.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m, None, rgba_t.data_ptr(), None)
            eng.get_image_dev(_capi.IMAGE_FREECAMERA_DEPTH, pose_m, None, None, depth_t.data_ptr())
        else:
            c, _ = eng.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=pose_m)
            _, d = eng.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=pose_m, want_rgba=False, want_depth=True)
            rgba_t.copy_(self.torch.from_numpy(c.reshape(self.P, 4)))
            depth_t.copy_(self.torch.from_numpy(d.reshape(self.P)))

    def preview(self, static_pose_m, instance_pose_m, track_id_of_instance, tint_strength=1.0, dim_background=True):
        """static_pose_m: world->camera of the preview camera; instance_pose_m[k]: object k -> camera
        (the model view composed with the instance pose, InstanceReconstructor.cpp:923,968).
        Returns (rgba, depth) tensors of the composited preview on rank 0, (None, None) elsewhere.
        Nothing here waits on the host: the engine streams, the collective and the composite are
        ordered with events (dsr_stream_wait_for_engine)."""
        torch = self.torch
        ex = self.exchange
        cur = torch.cuda.current_stream(self.device).cuda_stream if self.on_gpu else None
        for slot, k in enumerate(ex.local_instances):
            ie = self.instances[k]
            if self.on_gpu:
                ie.wait_for_stream(cur)  # the previous frame's all-gather has to be done with this slot
            if k in instance_pose_m:
                self._render(ie, instance_pose_m[k], ex.local_rgba[slot], ex.local_depth[slot])
            else:  # not visible in this frame: an empty layer never wins a pixel
                ex.local_depth[slot].zero_()
            if self.on_gpu:
                ie.stream_wait_for_engine(cur)  # the all-gather (torch's stream) reads what the engine stream writes
        if self.owns_static:
            if self.on_gpu:
                self.static.wait_for_stream(cur)  # ... and the previous composite with the target
            self._render(self.static, static_pose_m, self.target_rgba, self.target_depth)
            if self.on_gpu:
                self.static.stream_wait_for_engine(cur)
        ex.gather()
        if self.rank != 0:
            return None, None
        ex.composite_into(self.target_rgba, self.target_depth, track_id_of_instance, tint_strength, dim_background)
        return self.target_rgba, self.target_depth
