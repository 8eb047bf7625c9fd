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

    def __init__(self, n_pixels, n_volumes, world_size, rank, device, group=None):
        import torch
        self.torch = torch
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
        if self.world == 1:
            self.all_depth.copy_(self.local_depth)
            self.all_rgba.copy_(self.local_rgba)
            return
        dist.all_gather_into_tensor(self.all_depth, self.local_depth, group=self.group)
        dist.all_gather_into_tensor(self.all_rgba.view(self.world * self.slots, self.P * 4),
                                    self.local_rgba.view(self.slots, self.P * 4), group=self.group)

    def ordered_layers(self, track_id_of_instance):
        """-> (layer indices, track ids) sorted by ascending track id."""
        items = sorted((track_id_of_instance[k], self.layer_of_instance[k]) for k in self.layer_of_instance
                       if k in track_id_of_instance)
        return [l for _, l in items], [t for t, _ in items]

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
