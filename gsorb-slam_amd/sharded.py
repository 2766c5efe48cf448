"""Scene (Gaussian) sharding across the GPUs of one node — the only place the path exchanges data.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in
the CPU tests). Every rank owns a shard of the map (its Gaussians and their optimiser state)
and rasterizes ONLY its shard, with background 0, into a layer made of the op's own outputs:

    rgb  [3,H,W]  colour render             (colors_precomp = rgb,      src/Render.cc:927-946)
    ds   [2,H,W]  depth / silhouette render (colors_precomp = [z,1,0],  src/Render.cc:949-981)
                  ds[0] = alpha-blended depth, ds[1] = accumulated opacity S = 1 - T_final

The layers are exchanged with ONE all-gather (5 floats per pixel per rank: 16 MB at 1200x680)
and composited front to back with the "over" operator: out = sum_g (prod_{h<g} (1 - S_h)) *
layer_g. This is exact when the shards are depth-separable for the view (convex cells in
camera order) — SURVEY.md §8e scheme B — and otherwise an approximation whose PSNR against the
single-GPU render must be reported, not assumed.

Backward needs no per-splat exchange: every rank evaluates the same loss on the same
composite, and autograd reaches only its OWN layer (the gathered copies of the other layers
are constants), hence only its own Gaussians. The single gradient collective is the all-reduce
of the camera-pose gradient (each rank holds the contribution of its shard), 16 floats.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class LayerCompositor:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def composite(self, rgb: torch.Tensor, ds: torch.Tensor, order_key: float):
        """rgb [3,H,W], ds [2,H,W] = this rank's layer (may require grad); order_key = any
        scalar that sorts the shards front to back for this camera (e.g. the shard's nearest
        camera-space depth). Returns (rgb, depth, silhouette) of the whole scene."""
        layer = torch.cat([rgb, ds], 0)
        if self.world == 1:
            return rgb, ds[0:1], ds[1:2]
        with torch.no_grad():
            gathered = [torch.empty_like(layer) for _ in range(self.world)]
            dist.all_gather(gathered, layer.detach().contiguous(), group=self.group)
            key = torch.tensor([float(order_key)], dtype=torch.float64, device=layer.device)
            keys = [torch.empty_like(key) for _ in range(self.world)]
            dist.all_gather(keys, key, group=self.group)
            order = sorted(range(self.world), key=lambda g: (float(keys[g]), g))
        T = torch.ones_like(layer[0:1])
        out = torch.zeros_like(layer[0:4])
        for g in order:
            L = layer if g == self.rank else gathered[g]   # own layer keeps its autograd history
            out = out + T * L[0:4]
            T = T * (1.0 - L[4:5])
        return out[0:3], out[3:4], 1.0 - T

    def all_reduce_pose_grad(self, grad: torch.Tensor) -> torch.Tensor:
        """Sum of the per-shard pose gradients (dL/dTcw 4x4, or quaternion+translation)."""
        if self.world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)
        return grad


def shard_by_depth_slabs(depths: torch.Tensor, world: int):
    """Depth-separable partition for one view: rank g gets the g-th quantile slab of camera-space
    depth. Returns a list of index tensors (front to back)."""
    order = torch.argsort(depths)
    return list(torch.tensor_split(order, world))
