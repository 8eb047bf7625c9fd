"""One-volume-per-GPU sharding and the fused-preview exchange.

The reference drives every volume (static map + one InfiniTamDriver per tracked car,
InstanceReconstructor.cpp:382-389) sequentially on one GPU.  Volumes share no data during
allocate / integrate / raycast / decay, so they shard by volume: the static map lives on
rank 0, instance k on rank 1 + (k mod (world-1)).  The only exchange is the fused preview
(CompositeInstances, InstanceReconstructor.cpp:933-990): every rank renders its volumes from
the same free camera, the per-volume depth (f32) and colour (RGBA) buffers are ALL-GATHERED
and rank 0 z-composites them on its GPU.

On GPUs the exchange IS the C ABI's (`dsr_exchange_*`, include/dsr.h — the same entry points a C++ host calls): the library
owns the layer buffers, calls RCCL itself (ncclCommInitRank from a unique id this module only carries between the ranks) and
composites where the all-gather left the layers: `NativeLayers` below is a thin caller.  `PreviewExchange` (torch tensors +
`torch.distributed`) remains for what RCCL cannot host: the CPU tests (gloo, the oracle as the engine) and several ranks
sharing ONE GPU (tests on a one-GPU box).

torch is plumbing here (process group, device buffers); the arithmetic is the HIP library.
"""
import ctypes as C

import sys

import numpy as np

_logged = set()


def _log_once(key, msg):
    """A silent fall-back hides a performance regression: say it, once per process, on stderr."""
    if key not in _logged:
        _logged.add(key)
        print(msg, file=sys.stderr)


def torch_empty_like_cpu(t):
    import torch
    return torch.empty(t.shape, dtype=t.dtype, device="cpu")


def volume_owner(volume_index, world_size, has_static=True):
    """Owning rank of a volume.  With a static map (configs[3]): volume 0 = static map on rank 0, volume 1+k =
    instance k on rank 1 + (k mod (world-1)).  Without (north_star's "N concurrent instance volumes": every volume is an
    instance volume): volume k = instance k on rank k mod world."""
    if world_size <= 1:
        return 0
    if not has_static:
        return volume_index % world_size
    if volume_index == 0:
        return 0
    return 1 + ((volume_index - 1) % (world_size - 1))


def volumes_of_rank(rank, n_volumes, world_size, has_static=True):
    return [v for v in range(n_volumes) if volume_owner(v, world_size, has_static) == rank]


def instances_of_rank(rank, n_volumes, world_size, has_static=True):
    """Instance numbers (0-based) a rank owns."""
    off = 1 if has_static else 0
    return [v - off for v in volumes_of_rank(rank, n_volumes, world_size, has_static) if v >= off]


def max_local_instances(n_volumes, world_size, has_static=True):
    """Largest number of INSTANCE volumes any rank owns (the all-gather slot count)."""
    counts = [len(instances_of_rank(r, n_volumes, world_size, has_static)) for r in range(max(1, world_size))]
    return max(counts) if counts else 0


class PreviewExchange:
    """All-gather of per-instance raycast buffers + ordering for the composite.

    Every rank contributes `slots` layers (its instance renders, padded with empty layers: depth 0 = miss everywhere,
    which never wins a pixel).  A layer is 8 bytes per pixel — the float depth plane followed by the RGBA plane — and a
    rank's layers are ONE contiguous buffer, so the exchange is ONE `all_gather_into_tensor` per frame (round 2 issued
    two: on a launch-bound step the collective's fixed cost counts twice).  After `gather()` every rank holds all
    layers where the collective left them; the composite reads them in place through per-layer pointers
    (`dsr_composite_layer_ptrs_dev`) in ascending track id, the order the host's `for track in GetActiveTracks()` loop
    composites in.
    """

    def __init__(self, n_pixels, n_volumes, world_size, rank, device, group=None, local_only=False, has_static=True):
        import torch
        self.torch = torch
        self.local_only = bool(local_only)  # never touch the process group (a one-rank scene inside a larger job)
        self.P = int(n_pixels)
        self.world, self.rank = int(world_size), int(rank)
        self.group = group
        self.slots = max(1, max_local_instances(n_volumes, world_size, has_static))
        self.local_instances = instances_of_rank(rank, n_volumes, world_size, has_static)
        self.device = device
        pb = self.P * 4  # bytes of one plane
        self.local = torch.zeros((self.slots, 2, pb), dtype=torch.uint8, device=device)
        self.all = torch.zeros((self.world * self.slots, 2, pb), dtype=torch.uint8, device=device)
        # typed views of the two planes (strided: a layer's planes are adjacent, layers 8 B x P apart)
        self.local_depth = self.local[:, 0].view(torch.float32)
        self.local_rgba = self.local[:, 1].view(self.slots, self.P, 4)
        self.all_depth = self.all[:, 0].view(torch.float32)
        self.all_rgba = self.all[:, 1].view(self.world * self.slots, self.P, 4)
        # layer index in the gathered buffer of every instance
        self.layer_of_instance = {}
        for r in range(self.world):
            for s, k in enumerate(instances_of_rank(r, n_volumes, world_size, has_static)):
                self.layer_of_instance[k] = r * self.slots + s
        self._ptr_cache = {}

    def slot_ptrs(self, local_slot):
        """Device pointers (rgba, depth) of a local slot: pass them to dsr_get_image_dev."""
        base = self.local.data_ptr() + local_slot * 8 * self.P
        return (base + 4 * self.P, base)

    def gather(self):
        import torch.distributed as dist
        if self.world == 1 and (self.local_only or not (dist.is_available() and dist.is_initialized())):
            self.all.copy_(self.local)
            return
        # (with a process group the collective runs for a single rank too: `torchrun --nproc-per-node 1` exercises
        #  the RCCL path of the multi-GPU layout on a one-GPU box)
        if self.all.is_cuda and dist.get_backend(self.group) != "nccl":
            # a process group without device collectives (gloo: ranks on different hosts over TCP, or several ranks sharing
            # one GPU in the tests): the layers are staged through host memory
            host = torch_empty_like_cpu(self.all)
            dist.all_gather_into_tensor(host, self.local.cpu(), group=self.group)
            self.all.copy_(host)
            return
        dist.all_gather_into_tensor(self.all, self.local, group=self.group)

    def ordered_layers(self, track_id_of_instance):
        """-> (layer indices, track ids) sorted by ascending track id."""
        items = sorted((track_id_of_instance[k], self.layer_of_instance[k]) for k in self.layer_of_instance
                       if k in track_id_of_instance)
        return [l for _, l in items], [t for t, _ in items]

    def _layer_ptr_arrays(self, track_id_of_instance):
        key = tuple(sorted(track_id_of_instance.items()))
        hit = self._ptr_cache.get(key)
        if hit is None:
            layers, tids = self.ordered_layers(track_id_of_instance)
            base = self.all.data_ptr()
            n = len(layers)
            dp = (C.c_void_p * max(1, n))(*[base + l * 8 * self.P for l in layers])
            rp = (C.c_void_p * max(1, n))(*[base + l * 8 * self.P + 4 * self.P for l in layers])
            ids = (C.c_int32 * max(1, n))(*tids)
            if len(self._ptr_cache) >= 4:  # track ids change all the time in a long run: keep the few most recent sets
                self._ptr_cache.pop(next(iter(self._ptr_cache)))
            hit = self._ptr_cache[key] = (n, rp, dp, ids)
        return hit

    def composite_into(self, target_rgba, target_depth, track_id_of_instance, tint_strength=1.0, dim_background=True,
                       host_api=None):
        """z-composite the gathered layers, where they lie, into target_* (torch tensors on this rank's device) with
        `dsr_composite_layer_ptrs_dev` on torch's current stream.  With CPU tensors (the gloo tests) the same entry point
        of `host_api` is called — the tests pass the oracle's restatement; the product has no CPU composite."""
        on_gpu = getattr(self.device, "type", "cpu") == "cuda"
        if on_gpu:
            from .engine import load_hip_api
            api = load_hip_api()
            dev_index = self.device.index if self.device.index is not None else 0
            stream = C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)
        else:
            api = host_api or getattr(self, "host_api", None)
            if api is None:
                raise RuntimeError("CPU tensors need host_api (tests only): there is no CPU fallback for the composite")
            dev_index, stream = -1, None
        n, rp, dp, ids = self._layer_ptr_arrays(track_id_of_instance)
        if n == 0:
            return
        st = api.composite_layer_ptrs_dev(
            dev_index, stream, C.c_void_p(target_rgba.data_ptr()) if target_rgba is not None else None,
            C.c_void_p(target_depth.data_ptr()), rp if target_rgba is not None else None, dp, ids, n, self.P,
            float(tint_strength), int(bool(dim_background)))
        if st != 0:
            msg = api.last_error()
            raise RuntimeError(f"composite_layer_ptrs_dev failed: {msg.decode() if msg else st}")

    def composite(self, target_rgba, target_depth, track_id_of_instance, tint_strength=1.0, dim_background=True):
        return self.composite_into(target_rgba, target_depth, track_id_of_instance, tint_strength, dim_background)


class _DeviceArray:
    """A torch-importable view (`__cuda_array_interface__`) of HBM the library owns."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class NativeLayers:
    """The fused-preview exchange through the C ABI (`dynslam_amd.engine.Exchange` = `dsr_exchange_*`): what `ShardedScene` uses on
    GPUs whenever RCCL can host the ranks (one GPU per rank, or a single rank).  Same surface as `PreviewExchange` as far as
    `ShardedScene` and the tests need it."""

    def __init__(self, n_pixels, n_volumes, world_size, rank, device, group=None, local_only=False, has_static=True):
        import torch
        import torch.distributed as dist
        from .engine import Exchange
        self.torch = torch
        self.P, self.world, self.rank, self.device = int(n_pixels), int(world_size), int(rank), device
        self.slots = max(1, max_local_instances(n_volumes, world_size, has_static))
        self.local_instances = instances_of_rank(rank, n_volumes, world_size, has_static)
        dev_index = device.index if device.index is not None else 0
        if self.world == 1 and (local_only or not (dist.is_available() and dist.is_initialized())):
            self.x = Exchange(self.P, self.slots, devices=[dev_index])
        else:
            # rank 0 makes the communicator's id, torch.distributed only CARRIES it (a 128-byte broadcast); the collective of the
            # data path is the library's own RCCL call
            uid = torch.zeros(128, dtype=torch.uint8, device=device)
            if self.rank == 0:
                uid.copy_(torch.frombuffer(bytearray(Exchange.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0, group=group)
            self.x = Exchange(self.P, self.slots, unique_id=bytes(uid.cpu().numpy().tobytes()), world_size=self.world, rank=self.rank,
                              device=dev_index)
        self.slot_of_instance = {}  # instance -> (rank, slot)
        for r in range(self.world):
            for s, k in enumerate(instances_of_rank(r, n_volumes, world_size, has_static)):
                self.slot_of_instance[k] = (r, s)

    def close(self):
        self.x.close()

    def stream(self):
        return self.x.stream(self.rank)

    def ordered_layers(self, track_id_of_instance):
        key = tuple(track_id_of_instance.items())
        hit = getattr(self, "_ordered", None)
        if hit is None or hit[0] != key:
            items = sorted((track_id_of_instance[k], rs) for k, rs in self.slot_of_instance.items() if k in track_id_of_instance)
            hit = self._ordered = (key, [(r, s, t) for t, (r, s) in items])
        return hit[1]

    @property
    def all_depth(self):
        """The gathered depth layers as this rank holds them, [world * slots, P] float32 (tests)."""
        self.x.sync()
        rows = []
        for r in range(self.world):
            for s in range(self.slots):
                _, d = self.x.layer_ptrs(self.rank, r, s)
                rows.append(self.torch.as_tensor(_DeviceArray(d, (self.P,), "<f4"), device=self.device).clone())
        return self.torch.stack(rows)


class ShardedScene:
    """The volumes of one DynSLAM scene sharded one-volume-per-GPU, as one object per rank.

    `has_static=True` is BASELINE configs[3]: the static map on rank 0, instance volume k on rank 1 + (k mod (world-1)).
    `has_static=False` is north_star's scaling workload: `n_volumes` concurrent INSTANCE volumes, instance k on rank
    k mod world.  `step()` fuses one frame into every owned volume; `preview()` renders every volume from the shared
    camera, ALL-GATHERS the instance layers (depth f32 + RGBA, one collective) and z-composites them on rank 0 over the
    static map's render (or over an empty frame without one) — DynSlam::GetStaticMapRaycastPreview +
    InstanceReconstructor::CompositeInstances (DynSlam.h:96-132, InstanceReconstructor.cpp:911-990) with the volumes on
    different GPUs.

    `make_engine(kind)` returns an EngineCore-like object for kind in {"static", "instance", "view"}: "view" is a
    volume-less holder of the full input frame on ranks that own instances but not the static map (ProcessSilhouette
    reads the full frame: InstanceReconstructor.cpp:59-133).  Engines on a GPU are driven through the "_dev" entry
    points and never synchronise with the host; with CPU tensors (the gloo tests run the CPU oracle through this same
    class) the host-buffer entry points are used.
    """

    def __init__(self, make_engine, width, height, n_volumes, world_size, rank, device, group=None, local_only=False,
                 has_static=True, share_streams=True, use_batch=True, maps=False):
        import torch
        self.torch = torch
        self.W, self.H, self.P = int(width), int(height), int(width) * int(height)
        self.world, self.rank, self.device = int(world_size), int(rank), device
        self.n_volumes = int(n_volumes)
        self.has_static = bool(has_static)
        self.on_gpu = getattr(device, "type", "cpu") == "cuda"
        # maps=True (bench.py's map_volumes leg): every volume is a MAP-sized volume fed with the whole frame — no view split, no
        # instance masks; volume k on rank k mod world, the fused preview as for instance volumes.  The case sharding by volume is
        # made for: a volume's frame is a millisecond of a full GPU.
        self.maps = bool(maps)
        assert not (self.maps and self.has_static), "map volumes: has_static=False (every volume is one)"
        self.owns_static = self.has_static and 0 in volumes_of_rank(rank, n_volumes, world_size, True)
        self.static = make_engine("static") if self.owns_static else None
        self.instances = {k: make_engine("map" if self.maps else "instance") for k in instances_of_rank(rank, n_volumes, world_size, self.has_static)}
        # the full frame the instance views are cut from
        self.source = self.static if self.owns_static else (make_engine("view") if (self.instances and not self.maps) else None)
        # one instance volume next to its view engine on a GPU of their own (north_star's layout at 8 GPUs): ONE stream for the pair,
        # no cross-stream event in the frame (dsr_engine_share_stream)
        from .engine import DsrError
        if self.on_gpu and share_streams and not self.owns_static and len(self.instances) == 1 and not self.maps:
            try:
                next(iter(self.instances.values())).share_stream(self.source)
            except DsrError as ex:  # engines with a view pipeline of their own (created for a host that waits on them): as they are
                _log_once("share_stream", f"ShardedScene: the instance volume keeps a stream of its own ({ex}); create the engines "
                                          "with view_pipeline=VIEW_PIPELINE_OFF to queue the pair on one stream")
        # several instance volumes on this GPU: driven as ONE batch — every kernel of an instance frame launched once for all of
        # them (dsr_batch_*; results identical to the per-volume calls).  Up to 8 per batch; a rank with more keeps the loop.
        self.batch, self.batch_index = None, {}
        if self.on_gpu and use_batch and not self.maps and 2 <= len(self.instances) <= 8 and hasattr(self.source.api, "batch_create"):
            from .engine import Batch
            order = sorted(self.instances)
            try:
                self.batch = Batch(self.source, [self.instances[k] for k in order])
                self.batch_index = {k: i for i, k in enumerate(order)}
            except DsrError as ex:  # not batchable (pipelined views, other table sizes): the per-volume loop
                self.batch = None
                _log_once("batch", f"ShardedScene: {len(self.instances)} instance volumes are driven one by one, not as a batch ({ex}); "
                                   "create the engines with view_pipeline=VIEW_PIPELINE_OFF")
        # GPUs: the C ABI's exchange (RCCL called by the library) whenever RCCL can host the ranks — a process group on "nccl", or a
        # single rank; several ranks on ONE GPU (gloo, tests) and CPU tensors (the oracle, tests) go through torch.distributed
        import torch.distributed as dist
        grouped = dist.is_available() and dist.is_initialized() and not local_only
        self.native = self.on_gpu and (not grouped or dist.get_backend(group) == "nccl" or int(world_size) == 1)
        if self.native:
            self.exchange = NativeLayers(self.P, n_volumes, world_size, rank, device, group, local_only, self.has_static)
            r, d = self.exchange.x.target_ptrs(self.rank)  # the exchange's own composite target, seen as torch tensors
            self.target_rgba = torch.as_tensor(_DeviceArray(r, (self.P, 4), "|u1"), device=device)
            self.target_depth = torch.as_tensor(_DeviceArray(d, (self.P,), "<f4"), device=device)
        else:
            self.exchange = PreviewExchange(self.P, n_volumes, world_size, rank, device, group, local_only, self.has_static)
            self.target_rgba = torch.zeros((self.P, 4), dtype=torch.uint8, device=device)
            self.target_depth = torch.zeros((self.P,), dtype=torch.float32, device=device)

    def engines(self):
        out = ([self.static] if self.static is not None else []) + list(self.instances.values())
        if self.source is not None and self.source is not self.static:
            out.append(self.source)
        return out

    def close(self):
        self.sync()
        if self.batch is not None:
            self.batch.close()
            self.batch = None
        if self.native:
            self.exchange.close()
        for e in self.engines():
            e.close()

    def reset(self):
        """ResetScene of every volume this rank holds (bench.py: a diagnostic leg starts from empty volumes again)."""
        for e in self.engines():
            e.reset_scene()

    def sync(self):
        for e in self.engines():
            e.sync()
        if self.native:
            self.exchange.x.sync()

    # -- fusion -----------------------------------------------------------------------------
    def step(self, rgba, depth_mm, static_pose, masks):
        """One frame.  rgba / depth_mm: numpy arrays (host path) or device pointers (ints) of the full
        frame, resident on this rank's GPU; static_pose: camera->world of the static map;
        masks: [(instance k, x0, y0, mask, camera->object pose of k)] for EVERY instance (each rank picks its own);
        `mask` is a bbox-local uint8 numpy array (host: staged and synchronised per call, what the reference's host
        does) or a tuple (device pointer, box_w, box_h) of a mask already in HBM (no copy, no synchronisation).
        Order per rank as on one GPU: cut the instance views out of the full frame, blank them in the static view,
        fuse (InstanceReconstructor.cpp:238-263,569-700)."""
        if self.maps:  # every owned volume fuses the whole frame from the static pose (configs[1]'s step, per volume)
            for e in self.instances.values():
                if isinstance(rgba, int):
                    e.update_view_dev(rgba, depth_mm)
                else:
                    e.update_view(rgba, depth_mm)
                e.set_pose_inv_m(static_pose)
                e.process_frame()
                e.prepare()
            return
        if self.source is not None:
            if isinstance(rgba, int):
                self.source.update_view_dev(rgba, depth_mm)
            else:
                self.source.update_view(rgba, depth_mm)
        if self.batch is not None and masks and all(isinstance(m[3], tuple) for m in masks):
            # the whole instance side of the frame as one batch call: every cut-out and blanking in the host's order, then pose,
            # fusion and tracking render of every owned instance
            self.batch.fuse([(self.batch_index.get(k, -1), mask if k in self.batch_index else None, x0, y0, mask, x0, y0,
                              rel if k in self.batch_index else None) for k, x0, y0, mask, rel in masks])
            masks = ()
        for k, x0, y0, mask, rel in masks:
            ie = self.instances.get(k)
            dev_mask = isinstance(mask, tuple)
            # every rank that holds the frame blanks every silhouette, in order, like the one main view of the
            # reference: a later instance's cut-out must not see pixels an earlier (overlapping) mask removed
            if ie is not None:  # cut-out + blanking of an instance this rank owns: one launch (dsr_view_split_silhouette)
                if dev_mask:
                    self.source.split_silhouette_dev(ie, mask[0], x0, y0, mask[1], mask[2])
                else:
                    self.source.split_silhouette(ie, mask, x0, y0)
            elif self.source is not None:
                if dev_mask:
                    self.source.remove_silhouette_dev(mask[0], x0, y0, mask[1], mask[2])
                else:
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
    def _render(self, eng, pose_m, rgba_ptr, depth_ptr, rgba_t=None, depth_t=None):
        """Colour + float depth of one volume from the preview camera: what the host's GetImage(kColor) + GetFloatImage
        (kDepth) pair returns (InfiniTamDriver.cpp:165-209), from ONE raycast and one shading pass."""
        from . import _capi
        if self.on_gpu:
            eng.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m, None, rgba_ptr, depth_ptr)
        else:
            c, d = eng.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=pose_m, want_rgba=True, want_depth=True)
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
        if self.native:
            return self._preview_native(static_pose_m, instance_pose_m, track_id_of_instance, tint_strength, dim_background)
        cur = torch.cuda.current_stream(self.device).cuda_stream if self.on_gpu else None
        for slot, k in enumerate(ex.local_instances):
            ie = self.instances[k]
            if self.on_gpu:
                ie.wait_for_stream(cur)  # the previous frame's all-gather has to be done with this slot
            if k in instance_pose_m:
                rp, dp = ex.slot_ptrs(slot)
                self._render(ie, instance_pose_m[k], rp, dp, ex.local_rgba[slot], ex.local_depth[slot])
            else:  # not visible in this frame: an empty layer never wins a pixel
                ex.local_depth[slot].zero_()
            if self.on_gpu:
                ie.stream_wait_for_engine(cur)  # the all-gather (torch's stream) reads what the engine stream writes
        if self.owns_static:
            if self.on_gpu:
                self.static.wait_for_stream(cur)  # ... and the previous composite with the target
            self._render(self.static, static_pose_m, self.target_rgba.data_ptr(), self.target_depth.data_ptr(),
                         self.target_rgba, self.target_depth)
            if self.on_gpu:
                self.static.stream_wait_for_engine(cur)
        elif self.rank == 0 and not self.has_static:  # no static map: the instances are composited over an empty frame
            self.target_rgba.zero_()
            self.target_depth.zero_()
        ex.gather()
        if self.rank != 0:
            return None, None
        ex.composite_into(self.target_rgba, self.target_depth, track_id_of_instance, tint_strength, dim_background)
        return self.target_rgba, self.target_depth

    def _preview_native(self, static_pose_m, instance_pose_m, track_id_of_instance, tint_strength, dim_background):
        """The same preview through `dsr_exchange_*`: renders straight into the exchange's slots, ONE RCCL all-gather issued by the
        library, the composite over the exchange's own target on rank 0 — every ordering (engine streams, exchange stream) is
        the library's."""
        from . import _capi
        x = self.exchange.x
        if self.batch is not None:
            if not hasattr(self, "_slot_ptrs"):
                self._slot_ptrs = [x.slot_ptrs(self.rank, slot) for slot in range(len(self.exchange.local_instances))]
            ritems = []
            for slot, k in enumerate(self.exchange.local_instances):
                if k in instance_pose_m:
                    ritems.append((self.batch_index[k], instance_pose_m[k], self._slot_ptrs[slot][0], self._slot_ptrs[slot][1]))
                else:
                    x.render_slot(self.rank, slot, None)  # not visible in this frame: an empty layer
            if ritems:
                xs = x.stream(self.rank)
                self.source.wait_for_stream(xs)  # the previous gather / composite is done with the slots
                self.batch.render(ritems, _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME)
                self.source.stream_wait_for_engine(xs)
        else:
            for slot, k in enumerate(self.exchange.local_instances):
                x.render_slot(self.rank, slot, self.instances[k] if k in instance_pose_m else None, _capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME,
                              instance_pose_m.get(k))
        if self.owns_static:
            self.static.wait_for_stream(x.stream(self.rank))  # the previous composite is done with the target
            self.static.get_image_dev(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, static_pose_m, None, self.target_rgba.data_ptr(),
                                      self.target_depth.data_ptr())
        elif self.rank == 0 and not self.has_static:
            x.clear_target(0)
        x.gather_and_composite(0, self.exchange.ordered_layers(track_id_of_instance), target_engine=self.static if self.owns_static else None,
                               tint_strength=tint_strength, dim_background=dim_background)
        if self.rank != 0:
            return None, None
        return self.target_rgba, self.target_depth
