"""Scene (Gaussian) sharding across the GPUs of one node — the only place the path exchanges data.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in
the CPU tests). Every rank owns a shard of the map (its Gaussians and their optimiser state)
and rasterizes ONLY its shard, with background 0, into a layer made of the op's own outputs:

    rgb  [3,H,W]  colour render             (colors_precomp = rgb,      src/Render.cc:927-946)
    ds   [2,H,W]  depth / silhouette render (colors_precomp = [z,1,0],  src/Render.cc:949-981)
                  ds[0] = alpha-blended depth, ds[1] = accumulated opacity S = 1 - T_final

The layers are composited front to back with the "over" operator: out = sum_g (prod_{h<g} (1 - S_h)) *
layer_g. Only the silhouettes (and surface depths) are all-gathered — 2 floats per pixel per rank; every rank
premultiplies its own layer with its prefix transmittance and ONE all-reduce sums the four channels
(_CompositeFn below). This is exact when the shards are depth-separable for the view (convex cells in
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


def _all_gather(tensor: torch.Tensor, world: int, group=None):
    """all_gather of equally shaped tensors. RCCL ("nccl") gathers device tensors directly; the gloo backend
    (CPU tests, and the one-GPU two-process test) has no device all_gather, so device tensors are staged
    through host memory there."""
    if dist.get_backend(group) == "gloo" and tensor.is_cuda:
        host = tensor.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return [p.to(tensor.device) for p in parts]
    parts = [torch.empty_like(tensor) for _ in range(world)]
    dist.all_gather(parts, tensor, group=group)
    return parts


def _all_reduce_sum(t: torch.Tensor, group=None):
    """In-place sum over the ranks (gloo has no device collectives: device tensors are staged through host memory there)."""
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _CompositeFn(torch.autograd.Function):
    """Front-to-back "over" compositing of the ranks' layers with two small collectives instead of an all-gather of the
    whole layers. Per pixel, with the layers ordered front to back, out = sum_k P_k L_k, P_k = prod_{h before k} (1 - S_h):
      forward : all-gather of (silhouette S, surface depth) — 2 floats / pixel / rank — from which every rank forms ITS
                prefix transmittance P_own; ONE all-reduce (sum) of the 4 premultiplied channels P_own * (rgb, depth);
      backward: dL/dL_own = P_own * g needs nothing from the others; the layer's occlusion of what lies BEHIND it,
                dL/dS_own = - sum_{k behind own} (prod_{h before k, h != own} (1 - S_h)) (g . L_k) + g_sil * prod_{h != own} (1 - S_h),
                needs the others' g . L_k: one all-gather of 1 float / pixel / rank.
    At 1200x680 (3.26 MB per plane): 6.5 MB x world gathered + 13 MB reduced forward, 3.3 MB x world gathered backward;
    the all-gather of full 7-row layers it replaces moved 22.8 MB x world.
    GPU layers: what sits between the collectives is three elementwise HIP kernels behind the C ABI (csrc/gsr_shard.h:
    gsr_composite_forward / _backward_local / _backward_occlusion; round 4 — ~15 tensor launches per direction and a Python loop
    over the ranks before). CPU layers (the world-2 gloo tests, which drive the oracle): the tensor expressions below.
    SPMD: forward AND backward contain collectives — every rank must call composite() and must backpropagate through its
    result in the same iteration (LayerCompositor.composite makes the node part of the graph whenever grad mode is on)."""

    @staticmethod
    def forward(ctx, layer4, sil, sur, key, comp):
        world, rank, group = comp.world, comp.rank, comp.group
        dev = layer4.device
        with torch.no_grad():
            pad = torch.zeros((1,) + tuple(sil.shape[1:]), dtype=sil.dtype, device=dev)
            pad[0, 0, 0] = key                                   # the order key travels in a padding row (may be a device scalar: no host sync)
            mine = torch.cat([sil, sur if sur is not None else torch.zeros_like(sil), pad], 0).contiguous()
            g = torch.stack(_all_gather(mine, world, group)).contiguous()     # [world, 3, H, W], rank order
            order = torch.argsort(g[:, 2, 0, 0].double(), stable=True)   # front to back: by key, ties by rank (stays on the device)
            if layer4.is_cuda:
                from . import capi
                l4 = layer4.detach().to(torch.float32).contiguous()
                contrib, sil_tot, surf = capi.composite_forward(world, rank, order, g, l4, sur is not None)
                out4 = _all_reduce_sum(contrib, group)
                ctx.fused = True
                ctx.save_for_backward(l4, g, order)
            else:
                S = g[:, 0].index_select(0, order)                   # [world, H, W] silhouettes, front to back
                SU = g[:, 1].index_select(0, order)
                pos = torch.argmax((order == rank).to(torch.int8))   # this rank's slot in the order (no host sync)
                one_m = 1.0 - S
                P = torch.cumprod(torch.cat([torch.ones_like(S[:1]), one_m[:-1]], 0), 0)    # exclusive prefix transmittance per slot
                P_own = P.index_select(0, pos.reshape(1))[0:1]       # [1,H,W]
                out4 = _all_reduce_sum((P_own * layer4).contiguous().clone(), group)
                T_all = P[-1:] * one_m[-1:]                          # transmittance behind the last layer
                sil_tot = 1.0 - T_all
                # surface depth: of the first layer, front to back, behind which the accumulated transmittance is <= 0.5
                # (else of the last layer that has one)
                T_after = P * one_m
                surf = torch.zeros_like(sil)
                found = torch.zeros_like(sil, dtype=torch.bool)
                if sur is not None:
                    for k in range(world):
                        has = SU[k:k + 1] > 0
                        surf = torch.where(~found & has, SU[k:k + 1], surf)
                        found = found | (has & (T_after[k:k + 1] <= 0.5))
                ctx.fused = False
                ctx.order = order
                ctx.save_for_backward(layer4, S, P, pos)
        ctx.comp = comp
        ctx.mark_non_differentiable(surf)
        return out4, sil_tot, surf

    @staticmethod
    def backward(ctx, g4, gsil, _gsurf):
        comp = ctx.comp
        world, rank, group = comp.world, comp.rank, comp.group
        if ctx.fused:
            from . import capi
            l4, g, order = ctx.saved_tensors
            c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
            d_layer, c_own = capi.composite_backward_local(world, rank, order, g, l4, c(g4))
            c_all = torch.stack(_all_gather(c_own, world, group)).contiguous()       # [world, 1, H, W] in RANK order
            dS = capi.composite_backward_occlusion(world, rank, order, g, c_all, c(gsil))
            return d_layer, dS, None, None, None
        layer4, S, P, pos = ctx.saved_tensors
        if g4 is None:
            g4 = torch.zeros_like(layer4)
        P_own = P.index_select(0, pos.reshape(1))[0:1]
        d_layer = P_own * g4
        c_own = (g4 * layer4).sum(0, keepdim=True).contiguous()  # g . L_own: what the layers in FRONT of this one need
        c = torch.stack(_all_gather(c_own, world, group))[:, 0]  # [world, H, W] in RANK order
        # back to slot order: slot k holds rank order[k]; the ranks know the order from the forward's gather (saved S is
        # already in slot order, c is in rank order) — recover it from the saved permutation
        c = c.index_select(0, ctx.order)
        one_m = 1.0 - S
        # prefix products that leave this layer out: for the slots behind it, prod_{h before k, h != own} (1 - S_h)
        idx = torch.arange(world, device=S.device)
        behind = (idx > pos).reshape(world, 1, 1).to(S.dtype)
        excl = torch.where((idx == pos).reshape(world, 1, 1), torch.ones_like(one_m), one_m)     # own factor replaced by 1
        P_excl = torch.cumprod(torch.cat([torch.ones_like(S[:1]), excl[:-1]], 0), 0)
        dS = -(behind * P_excl * c).sum(0, keepdim=True)
        if gsil is not None:
            dS = dS + gsil * torch.prod(excl, 0, keepdim=True)   # d(1 - prod (1 - S_h)) / dS_own
        return d_layer, dS, None, None, None


class LayerCompositor:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def composite(self, rgb: torch.Tensor, ds: torch.Tensor, order_key, sur: torch.Tensor | None = None):
        """rgb [3,H,W], ds [2,H,W] = this rank's layer (may require grad); order_key = any
        scalar that sorts the shards front to back for this camera (e.g. the shard's nearest
        camera-space depth; a python float or a 0-d tensor). Returns (rgb, depth, silhouette) of the whole scene.

        With `sur` [1,H,W] (the layer's own surface / median depth, which carries no gradient) a fourth value is
        returned: the surface depth of the first layer, front to back, behind which the accumulated transmittance
        is <= 0.5 (else of the last layer that has one) — exact when nothing translucent lies in front of that
        layer, an approximation otherwise (each layer only knows where ITS OWN transmittance crosses 0.5).

        Exchange: _CompositeFn (an all-gather of 2 floats/pixel/rank and one all-reduce of 4 channels forward, an
        all-gather of 1 float/pixel/rank backward)."""
        if self.world == 1:
            return (rgb, ds[0:1], ds[1:2]) if sur is None else (rgb, ds[0:1], ds[1:2], sur)
        layer4, own_sil = torch.cat([rgb, ds[0:1]], 0), ds[1:2]
        if torch.is_grad_enabled() and not (layer4.requires_grad or own_sil.requires_grad):
            # the backward of the node contains collectives: a rank whose layer happens to be detached must still take part
            layer4 = layer4.detach().requires_grad_(True)
        out4, sil, surf = _CompositeFn.apply(layer4, own_sil, None if sur is None else sur.detach(), order_key, self)
        res = (out4[0:3], out4[3:4], sil)
        return res if sur is None else res + (surf,)

    def all_reduce_pose_grad(self, grad: torch.Tensor) -> torch.Tensor:
        """Sum of the per-shard pose gradients (dL/dTcw 4x4, or quaternion+translation)."""
        if self.world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)
        return grad

    def all_reduce_vector(self, t: torch.Tensor) -> torch.Tensor:
        """Sum of a small tensor over the ranks, staying where it is (RCCL reduces device tensors in place; gloo, which
        has no device collectives, goes through the host). No value reaches Python: no sync on the RCCL path."""
        if self.world == 1:
            return t
        if dist.get_backend(self.group) == "gloo" and t.is_cuda:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            return h.to(t.device)
        out = t.detach().clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out


def shard_by_depth_slabs(depths: torch.Tensor, world: int):
    """Depth-separable partition for ONE view: rank g gets the g-th quantile slab of camera-space
    depth. Returns a list of index tensors (front to back). The mapping loop of the reference draws a different keyframe every
    iteration (src/Render.cc:406-425): use KdPartition there."""
    order = torch.argsort(depths)
    return list(torch.tensor_split(order, world))


class KdPartition:
    """A partition of the map that holds while the view changes: a k-d split of the Gaussians' centres into `world` convex cells
    (axis-aligned boxes), balanced by count, one cell per rank. Convex cells of a BSP can be ordered front to back EXACTLY for any
    camera (visit, at every split, the side that holds the camera centre first), which a partition by the depth of one view cannot.
    What stays approximate is the extent of the splats: a Gaussian centred in one cell reaches into its neighbours, and the
    compositor treats a cell's layer as one slab — report PSNR against the one-GPU render (tests/test_kd_partition_gloo.py).

    nodes [world - 1, 4] float32 = {axis, split, left, right}: a child >= 0 is a node index, a child < 0 the leaf (rank) -1 - child;
    node 0 is the root, parents are numbered before their children; a point goes LEFT when x[axis] < split. The same array drives
    gsr_shard_order on the device (include/gsr.h) and SlamLoop::SetShard (torch_ext/SlamLoop.h).
    Owner rule for map growth (src/Render.cc:557-594, src/Gaussian.cc:40-95): a new Gaussian belongs to the cell that holds its centre
    (assign). After pruning (src/Gaussian.cc:180-258) the cells drift out of balance: rebalance_* below re-split and move the rows."""

    def __init__(self, nodes: torch.Tensor, world: int):
        self.world = int(world)
        self.nodes = torch.as_tensor(nodes, dtype=torch.float32).reshape(max(self.world - 1, 0), 4).cpu().contiguous()

    @staticmethod
    def view_weights(Tcws, floor: float = 0.05) -> torch.Tensor:
        """Axis weights for build() from the poses the map is looked at from (world -> camera matrices [k,4,4] or a list): weight of a world axis = floor + the
        mean |component| of the cameras' viewing direction (the third row of R_cw) along it. Cells cut ALONG the viewing direction lie in front of each other:
        their layers are slabs in depth for those views — what the compositor's "one layer = one slab" assumes (measured: 117-152 dB against the one-GPU
        render, where cells side by side across the image sit at 44-65 dB: tests/test_kd_partition_gloo.py)."""
        T = torch.stack([torch.as_tensor(t, dtype=torch.float32).reshape(4, 4) for t in Tcws]) if not torch.is_tensor(Tcws) else Tcws.reshape(-1, 4, 4).float()
        return floor + T[:, 2, :3].abs().mean(0).cpu()

    @classmethod
    def build(cls, xyz: torch.Tensor, world: int, weights=None) -> "KdPartition":
        """weights [3] (None: ones): a split takes the axis of the largest WEIGHTED extent — view_weights(keyframe poses) stacks the cells along the direction the
        cameras look in wherever the map is deep enough (the mitigation for side-by-side cells' boundary splats: VERDICT r5 weak 14)."""
        pts = torch.as_tensor(xyz, dtype=torch.float32).detach().cpu().reshape(-1, 3)
        wts = torch.ones(3) if weights is None else torch.as_tensor(weights, dtype=torch.float32).reshape(3).cpu()
        nodes: list = []

        def rec(idx, lo, hi):
            if hi - lo == 1:
                return -1 - lo
            me = len(nodes)
            nodes.append(None)                                   # parents before children
            k = (hi - lo) // 2
            if idx.numel() == 0:
                axis, split = 0, 0.0
                left = right = idx
            else:
                sub = pts[idx]
                axis = int(torch.argmax((sub.max(0).values - sub.min(0).values) * wts))
                vals, perm = torch.sort(sub[:, axis], stable=True)
                m = min(max(int(round(idx.numel() * k / (hi - lo))), 0), idx.numel())
                if m == 0:
                    split = float(vals[0])                        # everything goes right
                elif m == idx.numel():
                    split = float(torch.nextafter(vals[-1], torch.tensor(float("inf"))))
                else:
                    split = float(vals[m])
                    m = int(torch.searchsorted(vals, vals[m]))    # ties with the split value go right together (x < split is the rule)
                left, right = idx[perm[:m]], idx[perm[m:]]
            l = rec(left, lo, lo + k)
            r = rec(right, lo + k, hi)
            nodes[me] = (float(axis), split, float(l), float(r))
            return me

        if world > 1:
            rec(torch.arange(pts.shape[0]), 0, world)
        return cls(torch.tensor(nodes, dtype=torch.float32).reshape(-1, 4) if nodes else torch.zeros((0, 4)), world)

    def assign(self, xyz: torch.Tensor) -> torch.Tensor:
        """[n] int64: the rank whose cell holds each point."""
        pts = torch.as_tensor(xyz).detach()
        node = torch.zeros((pts.shape[0],), dtype=torch.int64, device=pts.device)
        for i in range(self.nodes.shape[0]):
            axis, split, l, r = int(self.nodes[i, 0]), float(self.nodes[i, 1]), int(self.nodes[i, 2]), int(self.nodes[i, 3])
            here, go_left = node == i, pts[:, axis] < split
            node = torch.where(here & go_left, torch.full_like(node, l), torch.where(here & ~go_left, torch.full_like(node, r), node))
        return -1 - node if self.world > 1 else node

    def order(self, Tcw: torch.Tensor) -> list:
        """The ranks front to back for the camera of Tcw (world -> camera): the CPU twin of gsr_shard_order."""
        if self.world == 1:
            return [0]
        T = torch.as_tensor(Tcw, dtype=torch.float64).detach().cpu().reshape(4, 4)
        c = -(T[:3, :3].t() @ T[:3, 3])
        out, stack = [], [0]
        while stack:
            n = stack.pop()
            if n < 0:
                out.append(-1 - n)
                continue
            axis, split, l, r = int(self.nodes[n, 0]), float(self.nodes[n, 1]), int(self.nodes[n, 2]), int(self.nodes[n, 3])
            near_left = float(c[axis]) < split
            stack.append(r if near_left else l)                  # far side: popped second
            stack.append(l if near_left else r)
        return out

    def key(self, Tcw: torch.Tensor, rank: int) -> float:
        """An order key for LayerCompositor.composite: this rank's position in the front-to-back order."""
        return float(self.order(Tcw).index(rank))


def _exchange_device(group=None) -> torch.device:
    """Where a group's collectives take their tensors: an RCCL ("nccl") group moves device memory only (a CPU tensor raises "No backend type
    associated with device type cpu": ADVICE r5), every other backend used here (gloo) host memory."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _gather_rows(rows: torch.Tensor, world: int, group=None):
    """all-gather of row blocks of different lengths: returns the list of every rank's rows (on the host)"""
    dev = _exchange_device(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c) for c in counts]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=dev)
    pad[:rows.shape[0]] = rows.detach().to(dev)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return [p[:c].cpu() for p, c in zip(parts, counts)]


def rebalance_rows(xyz: torch.Tensor, payload: torch.Tensor, rank: int, world: int, group=None, tolerance: float = 1.25):
    """Re-split of a sharded map whose cells have drifted out of balance (pruning removes where the scene is over-explained, growth adds
    where it is not). Collective. xyz [n,3] the rank's centres, payload [n,k] everything that travels with a Gaussian (its raw
    parameters and Adam moments). Returns (partition, keep, arrivals): the new KdPartition, the indices of the own rows that stay,
    and the payload rows that arrive from the other ranks — or None when max(count) <= tolerance * mean(count) (nothing moves).
    Exchange: one all-gather of the centres (12 B per Gaussian of the whole map) and one of the rows that change owner."""
    mine = xyz.detach().cpu().to(torch.float32)
    counts = [p.shape[0] for p in _gather_rows(torch.zeros((mine.shape[0], 0)), world, group)]
    total = sum(counts)
    if total == 0 or max(counts) <= tolerance * total / world:
        return None
    everyone = torch.cat(_gather_rows(mine, world, group), 0)
    part = KdPartition.build(everyone, world)
    owner = part.assign(mine)
    stay = owner == rank
    leaving = torch.cat([payload.detach().cpu().to(torch.float32)[~stay], owner[~stay].to(torch.float32).unsqueeze(1)], 1)
    arrivals = torch.cat([p[p[:, -1] == rank][:, :-1] for r, p in enumerate(_gather_rows(leaving, world, group)) if r != rank] or
                         [torch.zeros((0, payload.shape[1]))], 0)
    return part, torch.nonzero(stay).squeeze(-1), arrivals


def rebalance_loop(loop, rank: int, world: int, group=None, tolerance: float = 1.25):
    """Re-balance of a sharded C++ loop (_C.SlamLoop with set_shard): the same exchange as ShardedMapper.rebalance on the loop's exported rows
    (parameters and Adam moments travel together), then the loop is told its new cell. Collective; returns the new KdPartition or None."""
    rows = loop.export_rows()
    res = rebalance_rows(rows[:, 0:3], rows, rank, world, group, tolerance)
    if res is None:
        return None
    part, keep, arrivals = res
    loop.replace_rows(keep, arrivals)
    loop.set_shard(group, rank, world, part.nodes)
    return part


def make_sharded_mapper(harness_mod):
    """Returns the ShardedMapper class bound to the harness module (gsorb-slam_amd/harness.py): the reference's
    mapping / tracking loops (src/Render.cc:420-483, :1054-1126) with the MAP sharded over the ranks — scheme B."""

    class ShardedMapper(harness_mod.SlamRenderer):
        """Every rank owns a shard of the Gaussians in its own GaussianMap (parameters AND Adam state stay local:
        Adam is element-wise, so the sharded optimiser steps exactly like the unsharded one, src/Gaussian.cc:152-175)
        and rasterizes only that shard. Per iteration:
          * ONE all-gather of the layers (rgb, depth, silhouette, surface depth: 6 floats/pixel) + compositing;
            the loss is then evaluated identically on every rank, and autograd reaches only the rank's own layer,
            i.e. its own Gaussians — no per-splat gradient exchange;
          * mapping: one all-reduce of three scalars (the scale regularisers are a sum / a mean over the WHOLE map);
          * tracking: one all-reduce of the pose gradient (7 floats: d/dquat, d/dtrans) before the pose Adam step —
            every rank holds the contribution of its shard and an identical copy of the pose optimiser.
        Exact for shards that are depth-separable for the view (up to the residual transmittance of pixels that
        stop early, forward.cu:360-364); otherwise report PSNR against the one-GPU render."""

        def __init__(self, gmap, width, height, group=None, partition=None, **kw):
            super().__init__(gmap, width, height, **kw)
            self.comp = LayerCompositor(group)
            self.partition = partition                           # KdPartition: the cells' order holds for every view; None: nearest depth of the shard

        def render_pair(self, Tcw, tracking=False):
            rimage, rsur, rdepth = super().render_pair(Tcw, tracking)
            with torch.no_grad():                                # the shard's nearest camera depth, as a device scalar
                xyz = self.map.xyz
                if self.partition is not None:                   # exact front-to-back order of the k-d cells for this camera
                    key = self.partition.key(Tcw, self.comp.rank)
                elif len(self.map):
                    z = (xyz * Tcw[2, :3]).sum(1) + Tcw[2, 3]      # (as a matrix-vector product rocBLAS takes 0.7 ms at 1 M rows)
                    key = torch.where(z > 0.2, z, torch.full_like(z, float("inf"))).min()   # what the rasterizer keeps (auxiliary.h:154)
                else:
                    key = float("inf")
            rgb, depth, sil, sur = self.comp.composite(rimage, rdepth[0:2], key, sur=rsur)
            return rgb, sur, torch.cat([depth, sil], 0)

        def _reduce_regularisers(self, sum_over, sum_spread, count):
            mine = torch.stack([sum_over.detach(), sum_spread.detach(), torch.as_tensor(count, dtype=sum_over.dtype, device=sum_over.device)])
            tot = self.comp.all_reduce_vector(mine)
            # value = whole-map total, gradient = this shard's part
            over = sum_over + (tot[0] - mine[0])
            spread = sum_spread + (tot[1] - mine[1])
            return over, spread, tot[2]

        def _replicated_term_grad_scale(self):
            # the pose gradients of the ranks are SUMMED (_sync_pose_grads): a term that does not depend on the shard
            # (harness.track's feature reprojection error) would otherwise be counted `world` times
            return 1.0 / self.comp.world

        def _densify_renders(self, Tcw):
            # the mask of Render::AddGaussian is taken on the composite of all ranks' layers: the same on every rank
            rim, _, rds = self.render_pair(Tcw, tracking=True)
            return rim, rds

        def _owned(self, pw):
            # owner rule: a new Gaussian belongs to the rank whose cell holds the back-projected point
            if self.comp.world == 1:
                return None
            if self.partition is None:
                raise RuntimeError("map growth on a sharded map needs a KdPartition")
            return self.partition.assign(pw) == self.comp.rank

        def rebalance(self, tolerance: float = 1.25):
            """After pruning / growth: re-split the map when the cells are out of balance (collective). Returns True when rows moved."""
            g = self.map
            if self.comp.world == 1 or self.partition is None:
                return False
            state = [g.opt.state.get(getattr(g, n), {}) for n in g.NAMES]
            rows = lambda ts: torch.cat([t.detach().reshape(len(g), -1) for t in ts], 1)
            par = rows([getattr(g, n) for n in g.NAMES])
            zeros = lambda n: torch.zeros_like(getattr(g, n))
            m = rows([st.get("exp_avg", zeros(n)) for st, n in zip(state, g.NAMES)])
            v = rows([st.get("exp_avg_sq", zeros(n)) for st, n in zip(state, g.NAMES)])
            res = rebalance_rows(g.xyz, torch.cat([par, m, v], 1), self.comp.rank, self.comp.world, self.comp.group, tolerance)
            if res is None:
                return False
            self.partition, keep, arr = res
            mask = torch.ones(len(g), dtype=torch.bool, device=g.device)
            mask[keep.to(g.device)] = False
            g.prune(mask)
            k = par.shape[1]
            g.append_rows(arr[:, :k].to(g.device), arr[:, k:2 * k].to(g.device), arr[:, 2 * k:].to(g.device))
            return True

        def _sync_pose_grads(self):
            g = self.map
            if self.comp.world > 1:
                flat = torch.cat([g.cam_quat.grad.reshape(-1), g.cam_trans.grad.reshape(-1)])
                self.comp.all_reduce_pose_grad(flat)
                g.cam_quat.grad.copy_(flat[:4].reshape(g.cam_quat.shape))
                g.cam_trans.grad.copy_(flat[4:7].reshape(g.cam_trans.shape))

    return ShardedMapper


# ======================================================================================
# Scheme A — tile-band sharding (replicated map, sharded image). SURVEY.md §8e, DESIGN.md §7.
#
# Every rank holds the whole map and renders only tile rows [y0, y1) of the frame (C ABI:
# gsr_forward_args.band_y0/band_y1). A band render is bit-identical to the same rows of the
# one-GPU render, so the gathered image IS the one-GPU image. Exchanges per iteration:
#   forward : one all-gather of the band pixels (4 floats/pixel in total over all ranks: 13 MB at 1200x680)
#   backward: one all-reduce of the packed per-splat accumulators (16 floats/splat: 64 MB at 1 M)
#             between the blend stage and the per-splat stage (gsr_backward_args.stages)
# after which every rank holds the full, identical gradients and steps its replica of the
# optimiser — no parameter broadcast. Use it when the map fits one GPU (always, with 288 GB)
# and the per-frame latency is what matters (tracking / mapping inner loops).
# ======================================================================================
def band_rows(grid_y: int, world: int, row_cost=None):
    """Split tile rows 0..grid_y into `world` contiguous bands [(y0,y1)...]. With `row_cost`
    (e.g. the rendered-pair count per tile row of the previous frame) the bands are balanced by
    cumulative cost, otherwise by row count. Bands may be empty when world > grid_y."""
    if row_cost is None:
        cuts = [(grid_y * r) // world for r in range(world + 1)]
    else:
        cost = torch.as_tensor(row_cost, dtype=torch.float64).flatten()
        assert cost.numel() == grid_y
        cum = torch.cumsum(cost + 1e-9, 0)
        cuts = [0]
        for r in range(1, world):
            target = float(cum[-1]) * r / world
            y = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64)).item()) + 1
            cuts.append(min(max(y, cuts[-1]), grid_y))
        cuts.append(grid_y)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class HipBandBackend:
    """The product backend: C ABI through gsorb-slam_amd/capi.py (no fallback)."""

    def __init__(self, capi):
        self.capi = capi

    def forward(self, settings, band, out, **splats):
        return self.capi.forward(settings, band=band, out=out, **splats)

    def backward_partial(self, st, dL_dpix):
        """blend backward of this rank's band (the accumulators are zero after a forward); returns the flat
        buffer to sum over ranks"""
        self._grads = self.capi.alloc_grads(st.P, st.M, st.geom.device)
        self.capi.backward(st, dL_dpix, grads=self._grads, stages=2)
        return self.capi.acc_view(st)

    def backward_finish(self, st, dL_dpix, summed):
        # `summed` is the accumulator view itself (all-reduced in place)
        return self.capi.backward(st, dL_dpix, grads=self._grads, stages=4)


class TileBandRenderer:
    def __init__(self, backend, group=None, tile: int = 16):
        self.backend, self.group, self.tile = backend, group, tile
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def bands(self, height: int, row_cost=None):
        return band_rows((height + self.tile - 1) // self.tile, self.world, row_cost)

    def forward(self, settings, height: int, width: int, device, row_cost=None, **splats):
        """Returns (color [3,H,W], depth [1,H,W], state). The images are complete on every rank."""
        bands = self.bands(height, row_cost)
        img = torch.zeros((4, height, width), dtype=torch.float32, device=device)
        st = self.backend.forward(settings, bands[self.rank], (img[0:3], img[3:4]), **splats)
        if self.world > 1:
            rows = [(min(height, y0 * self.tile), min(height, y1 * self.tile)) for y0, y1 in bands]
            hmax = max(b - a for a, b in rows)
            mine = torch.zeros((4, hmax, width), dtype=torch.float32, device=device)
            a, b = rows[self.rank]
            mine[:, :b - a] = img[:, a:b]
            parts = _all_gather(mine, self.world, self.group)
            for r, (a, b) in enumerate(rows):
                if r != self.rank:
                    img[:, a:b] = parts[r][:, :b - a]
        return img[0:3], img[3:4], st

    def backward(self, st, dL_dpix):
        """dL_dpix [3,H,W] must be the same on every rank (the loss is evaluated on the gathered
        image). Returns the full gradients, identical on every rank."""
        part = self.backend.backward_partial(st, dL_dpix)
        if self.world > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        return self.backend.backward_finish(st, dL_dpix, part)
