#!/usr/bin/env python3
"""bench.py — forward+backward rasterization throughput on MI355X.

Metric (BASELINE.json): splats·pixels/s for fwd+bwd at 1 M Gaussians, 1200x680
= P * W * H / t(fwd+bwd). One "step" = one forward + one backward of the hot path on one
synthetic scene whose inputs are already resident in HBM, through the C ABI
(gsr_forward_ws + gsr_backward: the sync-free workspace entry points).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1, headline line: STRONG scaling of the rasterize path with its exchange (shard_render): the same --splats scene
cut into k-d cells over the ranks, every rank renders its cell, the layers travel by pixel-row band (the band exchange,
DESIGN.md section 7), the composite's gradient is taken back to every rank's layer and the pose gradient all-reduced inside
the timed region. The collective-free replica figure (every rank its own scene) is kept as `replica_rasterize`.

`shard_step` (same JSON line, --mode all | shard-step): the step that DOES exchange data — one sharded mapping
iteration and one sharded tracking iteration of the C++ loop (torch_ext/DirectLoop.cpp: ORB_SLAM2::SlamLoop with SetShard;
src/Render.cc:420-483, :1054-1126 with the map cut into k-d cells over the ranks; BASELINE.json config 4: --splats Gaussians in
TOTAL, strong scaling). Its timed region contains the fused rasterizer pair of the rank's cell forwards and backwards, the two
grouped point-to-point exchanges of the band exchange around the band's compositor and loss kernels, tracking's all-reduce of
the pose rows and the Adam / pose step, over RCCL ("nccl"; a one-rank RCCL group when N = 1).

`other_workloads`: the other frame shapes of BASELINE.json's configs, each with its kernels one by one and a roofline block —
among them what ONE RANK of configs 4 and 5 runs (`rank-250k-*`: a k-d cell of the map over the whole frame).

The JSON line also carries
  roofline     : the dominant kernel (backward blend) against the HBM roofline, timed live
                 with HIP events recorded by the library on the launching stream
  cpu_baseline : the CPU oracle ("port", OpenMP over all host cores) on a bounded sample; beside it, only with the opt-in reference build (GSR_REFERENCE_BUILD=1, off by default: oracle/build_ref.sh),
                 `reference_kernels_on_this_gpu`: the reference's own CUDA kernels translated by hipify-perl and compiled by hipcc, timed on this GPU
  parity       : the timed scene against the oracle and (`against_reference_kernels`) against those kernels — the metric's "PSNR vs ref"
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def _hip():
    h = C.CDLL("libamdhip64.so.7")  # already loaded by torch: same runtime instance
    h.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    h.hipEventSynchronize.argtypes = [C.c_void_p]
    h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    h.hipEventDestroy.argtypes = [C.c_void_p]
    return h


def cpu_baseline(sc, P, W, H, budget_s=20.0):
    """Oracle (OpenMP build) fwd+bwd on the same scene, bounded to ~budget_s of CPU work; plus the single-thread figure on a
    bounded sample (every 8th splat of the same scene: a 1 M-splat fwd+bwd takes minutes on one core)."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    o = oracle.Oracle(omp=True)
    kw = dict(means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, scales=sc.scales,
              rotations=sc.rotations)
    t0 = time.perf_counter()
    o.forward(copy_stages=False, **kw)
    o.backward(sc.dL_dpix, accum_double=False)
    first = time.perf_counter() - t0
    reps = int(max(1, min(10, (budget_s - first) // max(first, 1e-3))))   # median of 10 after a warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        o.forward(copy_stages=False, **kw)
        o.backward(sc.dL_dpix, accum_double=False)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts)) if ts else first
    blended, walked = o.census()   # (pixel, splat) pairs the forward blended / walked: for useful_lane_frac
    out = {"value": P * W * H / t, "unit": "splats*pixels/s", "cores": cores, "kind": "port",
           "sample": f"{len(ts) or 1} fwd+bwd of the same {P}-splat {W}x{H} scene after one warm-up (median), oracle/libgsr_oracle_omp.so, "
                     f"{t * 1e3:.0f} ms each", "blended_pairs": blended, "walked_pairs": walked}
    # single thread: the deterministic single-thread build on every 8th splat of the scene
    sub = slice(0, None, 8)
    o1 = oracle.Oracle(omp=False)
    kw1 = dict(means3D=sc.means3D[sub], opacities=sc.opacities[sub], cam=sc.cam, colors=sc.colors[sub], scales=sc.scales[sub],
               rotations=sc.rotations[sub])
    t0 = time.perf_counter()
    o1.forward(copy_stages=False, **kw1)
    o1.backward(sc.dL_dpix, accum_double=False)
    t1 = time.perf_counter() - t0
    P1 = int(len(sc.means3D[sub]))
    out["single_thread"] = {"value": P1 * W * H / t1, "unit": "splats*pixels/s", "cores": 1,
                            "sample": f"one fwd+bwd of every 8th splat of the scene ({P1} splats, {W}x{H}), oracle/libgsr_oracle.so, {t1 * 1e3:.0f} ms"}
    # Beside the CPU figure, only with the OPT-IN reference build (GSR_REFERENCE_BUILD=1 and oracle/_ref built; off by default): the REFERENCE's own kernels on this GPU (oracle/build_ref.sh: the reference's .cu files translated by
    # hipify-perl and compiled by hipcc with its default contraction, i.e. what a user of the reference gets on this hardware without this library) —
    # a baseline like the CPU one, never the product.
    try:
        from oracle import ref
        if ref.available():
            rt = ref.time_scene(sc, iters=5)
            tt = (rt["forward_ms"] + rt["backward_ms"]) * 1e-3
            out["reference_kernels_on_this_gpu"] = {
                "value": P * W * H / tt, "unit": "splats*pixels/s", "forward_ms": rt["forward_ms"], "backward_ms": rt["backward_ms"], "kind": "reference",
                "sample": f"5 fwd+bwd of the same scene after one warm-up, inputs resident, HIP events; CudaRasterizer::Rasterizer::forward / backward "
                          f"(hipified at build time, oracle/_ref/libgsr_ref_fma.so): hipcub radix sort + scan, 16x16-thread tiles, per-pixel atomics — as the reference wrote them"}
    except Exception as e:                                   # (a baseline: its absence never fails the bench)
        out["reference_kernels_on_this_gpu"] = {"error": repr(e)}
    return out


def parity_block(gsr, sc, s, ws, ins, dev):
    """The timed scene checked against the CPU oracle, outside the timed region (BASELINE.json metric: '...; PSNR vs ref',
    bit-exact indices, gradients within 1e-4): same protocol as tests/test_gpu_parity.py — pixels within 1e-5 of one of the
    blend's branch thresholds (exp() is not bit-reproducible between libm and the GPU) are left out of the image comparison
    and their upstream gradient is zeroed for both sides."""
    from oracle import oracle
    o, f = oracle.forward_scene(sc, omp=True)
    mc, _ = o.margins(f)
    ok = mc >= 1e-5
    if ws is None:                       # (the other workloads: through the allocating entry point)
        st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
        torch.cuda.synchronize()
        R = st.num_rendered
        d = gsr.debug_export(st)
    else:
        st = gsr.forward_ws(s, ws, ins, None)
        torch.cuda.synchronize()
        R = st.num_rendered = ws.status()[0]     # (the sync-free entry point leaves R on the device)
        d = gsr.debug_export(st)
        st.num_rendered = -1
    idx = int((st.radii.cpu().numpy() != f.radii).sum()) + int((d["point_list"] != f.stages["point_list"]).sum()) \
        + int((d["ranges"] != f.stages["ranges"]).sum()) + int(R != f.num_rendered)
    col = st.color.cpu().numpy()
    mse = float((((col - f.color) ** 2)[:, ok]).mean())
    mse_all = float(((col - f.color) ** 2).mean())
    psnr = lambda m: float("inf") if m == 0 else 20.0 * np.log10(1.0 / np.sqrt(m))     # src/Utils.cc:33-37
    g_in = (sc.dL_dpix * ok[None]).astype(np.float32)
    b = o.backward(g_in)
    grads = gsr.capi.alloc_grads(sc.P, 0, dev)
    gr = gsr.backward(st, torch.as_tensor(g_in, device=dev), grads=grads)
    torch.cuda.synchronize()
    worst = {}
    for n in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        a, r = getattr(gr, n).cpu().numpy().astype(np.float64), np.asarray(getattr(b, n), np.float64)
        worst[n] = float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-30))
    vs_ref = None
    try:     # the metric's "PSNR vs ref" against the REFERENCE's own kernels on this GPU — only with the opt-in reference build (GSR_REFERENCE_BUILD=1; off by default)
        from oracle import ref
        if ref.available():
            rr, fr = ref.forward_scene(sc)
            br = rr.backward(g_in)
            vis = fr.radii > 0
            geo = sum(int((d[k].reshape(len(vis), -1)[vis].view(np.uint32) != fr.stages[k].reshape(len(vis), -1)[vis].view(np.uint32)).sum()) for k in ("means2D", "depths", "conic_opacity"))
            idx_r = int((st.radii.cpu().numpy() != fr.radii).sum()) + int((d["point_list"] != fr.stages["point_list"]).sum()) + int((d["ranges"] != fr.stages["ranges"]).sum()) \
                + int((d["point_list_keys"] != fr.stages["keys_sorted"]).sum()) + int(R != fr.num_rendered)
            wr = {n: float(np.abs(getattr(gr, n).cpu().numpy().astype(np.float64) - np.asarray(getattr(br, n), np.float64)).max() / max(np.abs(getattr(br, n)).max(), 1e-30))
                  for n in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations")}
            vs_ref = {"against": "oracle/_ref/libgsr_ref.so: CudaRasterizer::Rasterizer::forward / backward themselves (the reference's .cu files translated by hipify-perl at "
                                 "build time, hipcc -ffp-contract=off) on the timed scene, on this GPU",
                      "index_mismatches": idx_r, "indices_compared": "radii, ranges, sorted keys, sorted point_list, num_rendered",
                      "projected_geometry_values_that_differ": geo, "geometry_compared": "means2D, depths, conic + opacity of every visible splat, bit for bit",
                      "psnr_vs_ref_db": psnr(float((((col - fr.color) ** 2)[:, ok]).mean())), "psnr_vs_ref_db_all_pixels": psnr(float(((col - fr.color) ** 2).mean())),
                      "max_abs_colour_err": float(np.abs(col - fr.color)[:, ok].max()), "max_grad_rel_err": max(wr.values()), "grad_rel_err": wr,
                      "oracle_vs_ref_index_mismatches": int((f.radii != fr.radii).sum()) + int((f.stages["point_list"] != fr.stages["point_list"]).sum())}
    except Exception as e:
        vs_ref = {"error": repr(e)}
    return {"against_reference_kernels": vs_ref,
            "against": "oracle/libgsr_oracle_omp.so (CPU restatement of the reference path) on the timed scene",
            "bar": "the plain one: every gradient tensor within 1e-4 of the fp32 oracle (tensor scale), colours within 1e-4, integer stages bit-exact — "
                   "no appeal to the exact-state evaluation the randomised sweep allows its one ill-conditioned case (tests/test_gpu_fuzz.py, DESIGN.md section 2)",
            "passes_plain_1e-4_bar": bool(idx == 0 and max(worst.values()) <= 1e-4 and float(np.abs(col - f.color)[:, ok].max()) <= 1e-4),
            "psnr_vs_oracle_db": psnr(mse), "psnr_vs_oracle_db_all_pixels": psnr(mse_all),
            "max_abs_colour_err": float(np.abs(col - f.color)[:, ok].max()),
            "knife_edge_pixel_frac": float((~ok).mean()),
            "max_grad_rel_err": max(worst.values()), "grad_rel_err": worst, "index_mismatches": idx,
            "indices_compared": "radii, ranges, sorted point_list, num_rendered"}


def quick_raster(gsr, dev, cam, arrays, grad_in, steps=50, prewarm=30, dual=False, grad_ds=None):
    """fwd+bwd of one scene through the sync-free C-ABI entry points, like the headline step (same buffers, same stages):
    ms per step and the two blend kernels' live HIP-event averages. `arrays`: dict of numpy / torch inputs (means3D,
    opacities, colors, scales, rotations, all activated and camera-frame). dual: the fused colour + depth / silhouette pass."""
    s = gsr.capi.Settings.from_camera(cam, device=dev)
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    ins = dict(means3D=t(arrays["means3D"]), opacities=t(arrays["opacities"]), colors=t(arrays["colors"]), shs=None,
               scales=t(arrays["scales"]), rotations=t(arrays["rotations"]), cov3D=None)
    P, W, H = int(ins["means3D"].shape[0]), cam.width, cam.height
    g_in = t(grad_in)
    g_ds = t(grad_ds) if dual else None
    st0 = gsr.forward(s, ins["means3D"], ins["opacities"], colors=ins["colors"], scales=ins["scales"], rotations=ins["rotations"])
    R, V = st0.num_rendered, int((st0.radii > 0).sum())
    del st0
    ws = gsr.capi.Workspace(P, W, H, max_rendered=int(R * 1.25) + 1024, device=dev)
    grads = gsr.capi.alloc_grads(P, 0, dev, intermediates=False)
    hip = _hip()
    stream = torch.cuda.current_stream().cuda_stream
    evk = max(int(os.environ.get("GSR_BENCH_EVENT_EVERY", "4")), 1)   # (event pairs on every fourth step: rasterize() says why)
    ev = []
    for _ in range((steps + evk - 1) // evk):
        e = [C.c_void_p() for _ in range(4)]
        for x in e:
            hip.hipEventCreate(C.byref(x))
        f = (C.c_void_p * 10)()
        b = (C.c_void_p * 6)()
        f[8], f[9] = e[2], e[3]     # GSR_FWD_BLEND
        b[2], b[3] = e[0], e[1]     # GSR_BWD_BLEND
        ev.append((e, f, b))

    def step(fe=None, be=None):
        st = gsr.forward_ws(s, ws, ins, None, events=fe, dual=dual)
        gsr.backward(st, g_in, grads=grads, events=be, once=True, dL_dds=g_ds)
    for _ in range(prewarm):
        step()
    # (a small frame's 30 steps are 6 ms of GPU work: behind seconds of host-side scene generation the chip is still leaving its idle state — one run
    # timed the first cell of the rank regime at 1.2 ms per step, 0.20 on every other occasion: keep stepping until 40 ms have gone by)
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.04:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    n, ovf = ws.status()
    assert not ovf and n == R, (n, R, ovf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i % evk == 0:
            step(ev[i // evk][1], ev[i // evk][2])
        else:
            step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(steps, 1) * 1e3

    def avg(i0, i1):
        tot = 0.0
        for e, _, _ in ev:
            v = C.c_float(0)
            hip.hipEventElapsedTime(C.byref(v), e[i0], e[i1])
            tot += v.value
        return tot / max(len(ev), 1)
    bwd_ms, fwd_ms = avg(0, 1), avg(2, 3)
    # Outside the timed region: the library's per-stage event pairs on eight more steps (what the step's kernels take one by one: their sum against
    # ms_per_step is the launch gaps a small frame pays), and the launches per step from the library's own counter.
    L = gsr.capi.lib()
    n0 = int(L.gsr_debug_launch_count())
    step()
    launches = int(L.gsr_debug_launch_count()) - n0
    names = ["preprocess", "bin_count+colscan", "bin_fill", "tile_sort_cut", "blend_fwd", "bwd_clear", "blend_bwd", "splat_bwd"]
    stage = np.zeros(8)
    reps = 8
    for _ in range(reps):
        e = [C.c_void_p() for _ in range(16)]
        for x in e:
            hip.hipEventCreate(C.byref(x))
        f = (C.c_void_p * 10)(*e[:10])
        b = (C.c_void_p * 6)(*e[10:])
        step(f, b)
        torch.cuda.synchronize()
        for k in range(8):
            if k == 5:
                continue                                  # (the clear stage does not run: one backward per forward)
            v = C.c_float(0)
            hip.hipEventElapsedTime(C.byref(v), e[2 * k], e[2 * k + 1])
            stage[k] += v.value * 1e3 / reps
        for x in e:
            hip.hipEventDestroy(x)
    return {"splats": P, "width": W, "height": H, "visible": V, "tile_instances": R, "ms_per_step": ms,
            "bwd_blend_ms": bwd_ms, "fwd_blend_ms": fwd_ms, "steps": steps, "prewarm_steps": prewarm,
            "splats_pixels_per_s": P * W * H / (ms * 1e-3),
            "launches_per_step": launches,
            "stage_us": {n: round(float(stage[k]), 2) for k, n in enumerate(names) if k != 5},
            "sum_kernel_us": round(float(stage.sum()), 1),
            "roofline": step_roofline(P, V, R, W * H, ms, bwd_ms, dual)}


def step_roofline(P, V, R, N, ms_step, bwd_blend_ms, dual=False):
    """SURVEY.md section 8d's algorithmic bytes of one fwd+bwd with this run's R and V, against the step's time and the HBM peak; and the dominant
    kernel (the backward blend: 40R + 20N + 36V) against its live HIP-event average. BASELINE config 5 asks for exactly this on the 2 M shape."""
    total_ref = 152 * P + 340 * V + 128 * R + 44 * N
    total = total_ref - 40 * V           # the two intermediates the timed call does not store (config.omitted_stores)
    k_bytes = 40 * R + 20 * N + 36 * V
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "formula": "152 P + 340 V + 128 R + 44 N (SURVEY.md 8d)" + (" — the plain render's bytes: the fused pair's two extra planes and its tenth sum are not in the survey's formula" if dual else ""),
            "algorithmic_bytes": total, "algorithmic_bytes_with_reference_intermediates": total_ref,
            "achieved": total / (ms_step * 1e-3) / 1e9, "frac": total / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "achieved_with_reference_intermediates": total_ref / (ms_step * 1e-3) / 1e9, "frac_with_reference_intermediates": total_ref / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "dominant_kernel": {"kernel": "K_blend_bwd", "algorithmic_bytes": k_bytes, "avg_launch_ms": bwd_blend_ms,
                                "achieved": k_bytes / (bwd_blend_ms * 1e-3) / 1e9 if bwd_blend_ms > 0 else None,
                                "frac": k_bytes / (bwd_blend_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bwd_blend_ms > 0 else None}}


def post_mapping_map(gsr, dev, P=300_000, iters=100):
    """BASELINE config 2's stand-in for a TRAINED map (no dataset offline): a Replica-camera map of P SinglePixel-initialised
    Gaussians (src/Gaussian.cc:70-74) put through the harness's own mapping loop — one densification (Render.cc:557-594),
    `iters` mapping iterations with the scale regularisers (Render.cc:420-483), one opacity pruning (Render.cc:598-616) —
    against the render of a fatter 'true' scene (splats 3x the single-pixel size), so that the optimiser grows and reshapes
    the splats like a mapping session does. Returns the camera, the optimised map's activated parameters and what happened."""
    syn = gsr.synthetic
    hz = __import__("gsorb_slam_amd.harness", fromlist=["x"])
    camd = syn.CAMERAS["replica"]
    cam = syn.make_camera(**camd)
    W, H = cam.width, cam.height
    truth = syn.make_scene(P, cam, seed=77, scale_mult=3.0)
    def as_map(sc, noise):
        g = hz.GaussianMap(hz.Config(), camd["fx"], camd["fy"], device=dev)
        rng = np.random.default_rng(5)
        xyz = sc.means3D + noise * 0.003 * rng.standard_normal(sc.means3D.shape).astype(np.float32)
        col = np.clip(sc.colors + noise * 0.1 * rng.standard_normal(sc.colors.shape), 0, 1).astype(np.float32)
        g.add_points(torch.tensor(xyz), torch.tensor(col))
        return g
    gt = as_map(truth, 0.0)
    op = torch.tensor(truth.opacities)
    with torch.no_grad():
        gt.log_scales.copy_(torch.log(torch.tensor(truth.scales))); gt.unnorm_quat.copy_(torch.tensor(truth.rotations))
        gt.logit_opacities.copy_(torch.log(op / (1 - op)))
        T = torch.eye(4, device=dev)
        rgb, sur, _ = hz.SlamRenderer(gt, W, H).render_rgb(T, tracking=True)
    frame = hz.Frame(rgb.clone(), sur[0].clone(), T)
    # the map under optimisation: every 2nd true point, single-pixel sized (holes: the densification has something to add)
    sub = syn.Scene(cam, truth.means3D[::2], truth.scales[::2], truth.rotations[::2], truth.opacities[::2], truth.colors[::2])
    g = as_map(sub, 1.0)
    g.scene_radius = float(np.abs(truth.means3D).max())
    r = hz.SlamRenderer(g, W, H)
    n0 = len(g)
    added = r.densify(frame)
    losses = r.map_frames([frame], iters=iters)
    pruned = r.remove_low_opacity()
    with torch.no_grad():
        opac, scales, rots = r.activations(g.unnorm_quat, g.logit_opacities, g.log_scales)
        arrays = dict(means3D=g.xyz.detach().clone(), opacities=opac.clone(), colors=g.rgb.detach().clone(), scales=scales.clone(), rotations=rots.clone())
    rng = np.random.default_rng(11)
    info = {"initial_splats": n0, "densified": int(added), "pruned": int(pruned), "mapping_iterations": iters,
            "loss_first": float(losses[0]), "loss_last": float(losses[-1]),
            "mean_scale_over_single_pixel": float((scales.mean(1) / (g.xyz[:, 2].abs() / camd["fx"])).mean())}
    return cam, arrays, rng.standard_normal((3, H, W)).astype(np.float32), info


def other_workloads(a, gsr, dev):
    """The same fwd+bwd step on the other shapes BASELINE.json's configs describe, timed AFTER the headline (<= 50 steps each):
    the round-3 build bought the headline with these, so the driver's line carries them from now on."""
    syn = gsr.synthetic
    out = {}
    def synth(name, splats, camera, what, scale_mult=1.0, two_walls=False, parity=False):
        cam = syn.make_camera(**syn.CAMERAS[camera])
        sc = syn.make_scene(splats, cam, seed=0, scale_mult=scale_mult)
        if two_walls:
            rng = np.random.default_rng(3)
            z = np.where(rng.random(splats) < 0.5, 1.5, 4.0) + 0.02 * rng.random(splats)
            sc.means3D = (sc.means3D * (z / sc.means3D[:, 2])[:, None]).astype(np.float32)
        arrays = dict(means3D=sc.means3D, opacities=sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
        d = quick_raster(gsr, dev, cam, arrays, sc.dL_dpix, steps=a.other_steps)
        d["what"] = what
        if parity and not a.no_cpu:      # the same check as the headline's `parity` block, outside every timed region
            d["parity"] = parity_block(gsr, sc, gsr.capi.Settings.from_camera(cam, device=dev), None, None, dev)
        out[name] = d
    synth("scannet-2M", 2_000_000, "scannet", "2 M Gaussians, 640x480 ScanNet camera (BASELINE config 5's shape): tile lists of ~3 900 entries", parity=True)
    synth("fat-x4", 1_000_000, "replica", "the headline scene with every splat 4x the single-pixel size (beyond the 5x5-patch reach word)", scale_mult=4.0, parity=True)
    synth("scale-x2", 1_000_000, "replica", "the headline scene with every splat 2x the single-pixel size (lists just over 1024 entries)", scale_mult=2.0)
    synth("two-walls", 1_000_000, "replica", "the headline scene with every splat on one of two thin depth slabs (the tile sort's crowded-bin case)", two_walls=True)
    # the per-rank regime of BASELINE configs 4 and 5: a quarter of 1 M Gaussians at 1200x680, an eighth of 2 M at 640x480
    out["rank-250k-1200x680"] = rank_regime(a, gsr, dev, 1_000_000, "replica", 4)
    out["rank-250k-640x480"] = rank_regime(a, gsr, dev, 2_000_000, "scannet", 8)
    cam, arrays, g_in, info = post_mapping_map(gsr, dev)
    d = quick_raster(gsr, dev, cam, arrays, g_in, steps=a.other_steps)
    d["what"] = "a map AFTER mapping (BASELINE config 2's trained-map stand-in): see bench.py:post_mapping_map"
    d["map"] = info
    out["post-mapping-300k"] = d
    return out


def cpp_loop_ms(a, gsr, dev, P=1_000_000, track_iters=20, map_iters=20):
    """The C++ loop driver (torch_ext/SlamLoop.{h,cpp} in libgsr_torch.so) at the headline frame: ms per tracking iteration and
    per mapping iteration, through tests/cpp/slam_loop_main.bin on a scene file (the same front end tests/test_gpu_cpp_loop.py
    uses), and the fused rasterizer pair (colour + depth / silhouette in one pass, fwd+bwd) of the same scene through the C ABI."""
    import struct
    import subprocess
    import tempfile
    syn = gsr.synthetic
    hz = __import__("gsorb_slam_amd.harness", fromlist=["x"])
    exe = os.path.join(ROOT, "tests", "cpp", "slam_loop_main.bin")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/slam_loop_main.bin is missing: run __graft_entry__.build()"}
    camd = syn.CAMERAS[a.camera]
    cam = syn.make_camera(**camd)
    W, H = cam.width, cam.height
    sc = syn.make_scene(P, cam, seed=0)
    rng = np.random.default_rng(3)
    op = sc.opacities.reshape(-1, 1)
    logit = np.log(op / (1 - op)).astype(np.float32)
    g = hz.GaussianMap(hz.Config(), camd["fx"], camd["fy"], device=dev)
    g.add_points(torch.tensor(sc.means3D), torch.tensor(sc.colors))
    with torch.no_grad():
        g.log_scales.copy_(torch.log(torch.tensor(sc.scales))); g.unnorm_quat.copy_(torch.tensor(sc.rotations))
        g.logit_opacities.copy_(torch.tensor(logit))
        T_true = torch.eye(4, device=dev)
        rgb, sur, _ = hz.SlamRenderer(g, W, H).render_rgb(T_true, tracking=True)
    T_init = np.eye(4, dtype=np.float32)
    T_init[:3, 3] = (0.004, -0.003, 0.005)
    xyz = sc.means3D + 0.002 * rng.standard_normal(sc.means3D.shape).astype(np.float32)
    col = np.clip(sc.colors + 0.05 * rng.standard_normal(sc.colors.shape), 0, 1).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.environ.get("GSR_LOOP_SCENE_OUT") or os.path.join(tmp, "scene.bin")   # (scripts/loop_profile.sh keeps the file)
        with open(path, "wb") as f:
            f.write(struct.pack("<6i2f", P, W, H, track_iters, map_iters, 3, camd["fx"], camd["fy"]))
            for arr in (xyz, col, sc.rotations, logit, np.log(sc.scales), rgb.cpu().numpy(), sur[0].cpu().numpy(), T_true.cpu().numpy(), T_init):
                f.write(np.ascontiguousarray(arr, np.float32).tobytes())
        r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, GSR_LOOP_WARMUP=str(a.loop_warmup)))
    if r.returncode != 0:
        return {"error": r.stderr[-500:]}
    o = {ln.split()[0]: [float(x) for x in ln.split()[1:]] for ln in r.stdout.splitlines() if ln.strip()}
    arrays = dict(means3D=sc.means3D, opacities=sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    pair = quick_raster(gsr, dev, cam, arrays, sc.dL_dpix, steps=a.other_steps, dual=True,
                        grad_ds=np.random.default_rng(5).standard_normal((2, H, W)).astype(np.float32))
    return {"what": f"ORB_SLAM2::SlamLoop (C++, libgsr_torch.so) at {P} Gaussians, {W}x{H}, wall clock per iteration: the iterations as fixed "
                    "sequences of C-ABI launches on a persistent workspace (torch_ext/DirectLoop.cpp: fused pair, fused loss / SSIM kernels, "
                    "gsr_map_prepare / gsr_map_update / gsr_pose_update). mapping = SlamLoop::MapFrame (Render::RenderForFrame's loop: the losses "
                    "are read back once, after the last iteration, like the reference's loop, which never looks at one); mapping_per_iteration_readback "
                    "= MappingIteration in a loop (the caller looks at every loss: posted by the finish kernel into host-mapped memory, like tracking's — round 5 read it back "
                    "behind a stream synchronisation: 0.66 ms); tracking reads its loss every iteration "
                    "(Render.cc:1107); raster_pair = fwd+bwd of the fused colour + depth/silhouette pass alone",
            "mapping": o.get("mapframe_ms_per_iter", o["map_ms_per_iter"])[0], "mapping_per_iteration_readback": o["map_ms_per_iter"][0],
            "tracking": o["track_ms_per_iter"][0], "raster_pair": pair["ms_per_step"],
            "raster_pair_bwd_blend_ms": pair["bwd_blend_ms"], "raster_pair_fwd_blend_ms": pair["fwd_blend_ms"],
            "tracking_iterations_run": len(o.get("track", [])), "mapping_iterations_run": map_iters, "warmup_iterations": a.loop_warmup,
            "vs_shard_step_unsharded_same_scene": "loop_ms runs in a FRESH process (tests/cpp/slam_loop_main.bin) on the headline scene (seed 0, a damaged copy of the map against the "
                                                  "true map's render); shard_step.unsharded_same_scene is the same SlamLoop class inside this process on shard_step's scene (seed 1234, "
                                                  "the map against its own re-tinted render). Round 5's two figures (0.521 / 0.456 ms per tracking iteration) differed because the fresh "
                                                  "process timed its 20 iterations after ~7 warm-up iterations (4 ms of GPU work: clocks still ramping) while shard_step warms up with 20 "
                                                  "iterations behind the whole bench; with warmup_iterations more tracking + mapping iterations before its clock starts (GSR_LOOP_WARMUP) "
                                                  "loop_ms is a steady-clock figure too"}


def boundary_ms(gsr, sc, dev, steps=20):
    """fwd+bwd through the operator boundary GSORB-SLAM calls (the diff_gaussian_rasterization Python op over the libtorch
    host layer over the C ABI): what a maintainer's loop sees per call pair, host overhead and allocations included."""
    sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))
    import diff_gaussian_rasterization as dgr
    cam = sc.cam
    s = gsr.capi.Settings.from_camera(cam, device=dev)
    rs = dgr.GaussianRasterizationSettings(image_height=cam.height, image_width=cam.width, tanfovx=s.tanfovx, tanfovy=s.tanfovy,
                                           bg=s.bg, scale_modifier=1.0, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix,
                                           sh_degree=0, campos=s.campos, prefiltered=False)
    rast = dgr.GaussianRasterizer(raster_settings=rs)
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    means, op, col, sca, rot = (t(x).requires_grad_(True) for x in (sc.means3D, sc.opacities, sc.colors, sc.scales, sc.rotations))
    m2d = torch.zeros_like(means, requires_grad=True)
    g_in = t(sc.dL_dpix)

    def once():
        out = rast(means3D=means, means2D=m2d, opacities=op, colors_precomp=col, scales=sca, rotations=rot)
        out[0].backward(g_in)
        for x in (means, op, col, sca, rot, m2d):
            x.grad = None
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


_ONE_RANK = {"group": None, "td": None, "error": None}


def one_rank_group(dev):
    """A process group of ONE rank with backend "nccl" (N = 1 runs): RCCL itself then executes the sharded loop's collectives on the loop's stream.
    Created once per process, destroyed at the end of main()."""
    if _ONE_RANK["group"] is None and _ONE_RANK["error"] is None:
        try:
            import socket
            import torch.distributed as td1
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            td1.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            _ONE_RANK.update(group=td1.group.WORLD, td=td1)
        except Exception as e:                                   # (stated in the record, never silent)
            _ONE_RANK["error"] = type(e).__name__
    return _ONE_RANK["group"]


def rank_regime(a, gsr, dev, total, camera, cells):
    """What ONE rank of BASELINE configs 4-5 runs: `total` Gaussians cut into `cells` k-d cells (sharded.KdPartition), the rank holds one cell and
    rasterizes it over the WHOLE frame. Every cell is timed through the plain fwd+bwd step (the slowest one is what a synchronous N-rank step waits for: reported),
    and the slowest cell also through one sharded mapping / tracking iteration of the C++ loop at N = 1 with a one-rank RCCL group executing every collective."""
    syn = gsr.synthetic
    sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
    sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))
    from diff_gaussian_rasterization import _C
    camd = syn.CAMERAS[camera]
    cam = syn.make_camera(**camd)
    W, H = cam.width, cam.height
    sc = syn.make_scene(total, cam, seed=1234)
    t = lambda x: torch.tensor(x, dtype=torch.float32)
    part = sharded.KdPartition.build(t(sc.means3D), cells)
    owner = part.assign(t(sc.means3D)).numpy()
    per_cell = []
    for c in range(cells):
        idx = np.nonzero(owner == c)[0]
        arrays = dict(means3D=sc.means3D[idx], opacities=sc.opacities[idx], colors=sc.colors[idx], scales=sc.scales[idx], rotations=sc.rotations[idx])
        d = quick_raster(gsr, dev, cam, arrays, sc.dL_dpix, steps=a.other_steps)
        d["cell"] = c
        per_cell.append(d)
    worst = max(per_cell, key=lambda d: d["ms_per_step"])
    out = dict(worst)
    out["what"] = (f"one rank's share of {total} Gaussians cut into {cells} k-d cells, whole {W}x{H} frame: the plain fwd+bwd step of the SLOWEST cell "
                   f"(ms_per_step; every cell in cells_ms_per_step), its kernels one by one (stage_us, sum_kernel_us), launches_per_step")
    out["cells_ms_per_step"] = [round(d["ms_per_step"], 4) for d in per_cell]
    out["cells_tile_instances"] = [d["tile_instances"] for d in per_cell]
    out["splats_pixels_per_s_of_the_whole_map_if_every_rank_took_this_long"] = total * W * H / (worst["ms_per_step"] * 1e-3)
    # the sharded loop's iterations on that cell, one rank, RCCL executing the collectives
    group = one_rank_group(dev)
    idx = np.nonzero(owner == worst["cell"])[0]
    op = t(sc.opacities[idx]).reshape(-1, 1)
    raw = [t(sc.means3D[idx]), t(sc.colors[idx]), t(sc.rotations[idx]), torch.log(op / (1 - op)), torch.log(t(sc.scales[idx]))]
    L = gsr.capi.lib()
    res = {}
    for name, shard in (("sharded_one_rank_rccl", True), ("unsharded", False)):
        loop = _C.SlamLoop(W, H, camd["fx"], camd["fy"], dev)
        loop.set_map(*raw)
        if shard:
            loop.set_shard(group, 0, 1, torch.empty(0))
        T = torch.eye(4, device=dev)
        rgb, sur, _ = loop.render_composite(T)
        rgb, depth = (rgb * 0.9 + 0.05).contiguous(), sur[0].contiguous()
        T0 = T.clone(); T0[:3, 3] = torch.tensor([0.004, -0.003, 0.005], device=dev)
        k = max(a.shard_steps, 20)

        def timed(fn):
            fn(k)
            per = []
            for _ in range(3):
                torch.cuda.synchronize()
                n0 = int(L.gsr_debug_launch_count())
                t0 = time.perf_counter()
                ran = fn(k)
                torch.cuda.synchronize()
                per.append(((time.perf_counter() - t0) / max(ran, 1) * 1e3, (int(L.gsr_debug_launch_count()) - n0) / max(ran, 1)))
            return sorted(per)[1]
        m_ms, m_l = timed(lambda n: len(loop.map_frame(rgb, depth, T, n)))
        t_ms, t_l = timed(lambda n: len(loop.track(rgb, depth, T0, n)[0]))
        res[name] = {"mapping_ms_per_iter": m_ms, "tracking_ms_per_iter": t_ms, "library_launches_per_mapping_iter": round(m_l, 2), "library_launches_per_tracking_iter": round(t_l, 2),
                     "collectives_per_iter": ({"mapping": 2, "tracking": 3, "what": "the band exchange: two grouped point-to-point exchanges (+ the pose rows' all-reduce while tracking); at ONE rank the exchanges have no peer and launch nothing"} if shard and group is not None else 0), "transport": loop.shard_transport() if shard else None}
        del loop
    out["loop"] = res
    out["loop"]["what"] = ("ORB_SLAM2::SlamLoop on the same cell: wall clock per mapping (MapFrame: one read-back per batch) / tracking iteration, median of three batches; "
                           "library_launches = kernels the C ABI launched per iteration (gsr_debug_launch_count; includes the once-per-call preparation spread over the batch), "
                           "rccl_launches = the collectives RCCL executes on the loop's stream")
    if _ONE_RANK["error"]:
        out["loop"]["one_rank_group_error"] = _ONE_RANK["error"]
    return out


def band_exchange_bytes(W, H, world, rank):
    """Bytes one rank SENDS per iteration in the band exchange (DirectLoop.cpp: ensure_direct_): forward = the rows of its layer every peer's band needs (colour and
    silhouette planes with ten rows either side for the mapping loss's SSIM window, depth and surface depth on the band), backward = every peer's layer gradient
    (five planes; tracking on the surface depth: four) on its own band."""
    hb = -(-H // world)
    band = lambda k, halo: (max(0, min(H, k * hb) - halo), min(H, min(H, (k + 1) * hb) + halo))
    rows = lambda k, halo: max(0, band(k, halo)[1] - band(k, halo)[0])
    peers = [k for k in range(world) if k != rank]
    # (a tracking iteration on the surface depth — LoopConfig::use_sur_depth, the default — leaves the blended-depth plane at home, both ways; ShardRenderStep ships every plane)
    return {"map_fwd": sum(4 * rows(k, 10) + 2 * rows(k, 0) for k in peers) * W * 4, "track_fwd": sum(5 * rows(k, 0) for k in peers) * W * 4,
            "render_fwd": sum(6 * rows(k, 0) for k in peers) * W * 4, "bwd": 5 * rows(rank, 0) * W * 4 * len(peers), "track_bwd": 4 * rows(rank, 0) * W * 4 * len(peers)}


def shard_step(a, gsr, td, rank, world, dev):
    """One sharded MAPPING iteration and one sharded TRACKING iteration of the C++ loop (torch_ext/DirectLoop.cpp: SlamLoop::SetShard) with
    every collective inside the timed region. --splats Gaussians in TOTAL, cut into `world` k-d cells (sharded.KdPartition: the partition
    that holds while the view changes), one cell per rank; the loop's three collectives per iteration go through the process group on the
    loop's stream (RCCL; with one rank a one-rank RCCL group so that the collectives are RCCL's launches, not copies)."""
    syn = gsr.synthetic
    sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
    sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))
    from diff_gaussian_rasterization import _C
    camd = syn.CAMERAS[a.camera]
    cam = syn.make_camera(**camd)
    W, H = cam.width, cam.height
    sc = syn.make_scene(a.splats, cam, seed=1234, scale_mult=a.scale_mult)      # the SAME scene on every rank
    t = lambda x: torch.tensor(x, dtype=torch.float32)
    op = t(sc.opacities).reshape(-1, 1)
    raw = [t(sc.means3D), t(sc.colors), t(sc.rotations), torch.log(op / (1 - op)), torch.log(t(sc.scales))]
    part = sharded.KdPartition.build(raw[0], world)
    idx = torch.nonzero(part.assign(raw[0]) == rank).squeeze(-1)
    group, backend, own_group = None, "none (single process: the exchange is local copies)", False
    if world > 1:
        group, backend = td.group.WORLD, td.get_backend()
    else:
        group = one_rank_group(dev)                              # one rank: RCCL itself still runs the loop's collectives
        if group is not None:
            backend, own_group = "nccl (one rank)", True
        else:                                                    # (stated in the record, never silent)
            backend = f"none (single process: one-rank RCCL group failed: {_ONE_RANK['error']})"
    band = os.environ.get("GSR_BENCH_BAND", "1") != "0"     # (A/B hook: 0 = round 5's replicated composite)
    loop = _C.SlamLoop(W, H, camd["fx"], camd["fy"], dev, band_exchange=band)
    loop.set_map(*[x[idx] for x in raw])
    loop.set_shard(group, rank, world, part.nodes)
    transport = loop.shard_transport()   # "rccl": the loop's own communicator on its stream; "c10d": the group's collectives (fallback); "local": no group
    T = torch.eye(4, device=dev)
    rgb, sur, _ = loop.render_composite(T)
    rgb, depth = (rgb * 0.9 + 0.05).contiguous(), sur[0].contiguous()
    T0 = T.clone(); T0[:3, 3] = torch.tensor([0.004, -0.003, 0.005], device=dev)

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    batches = []

    def timed(fn):
        fn(max(a.shard_steps, 20))      # warm-up: clocks, and RCCL's first launches (measured: the first 20 iterations after 3 run 8 % slow)
        per = []
        for _ in range(3):              # three timed batches, the MEDIAN is reported (a batch now and then runs 10-15 % slow behind a one-rank RCCL group: all three are in `batches_ms`)
            barrier()
            t0 = time.perf_counter()
            ran = fn(max(a.shard_steps, 1))
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                td.all_reduce(tt, op=td.ReduceOp.MAX)
                dt = float(tt.item())
            per.append(dt / max(ran, 1) * 1e3)
        batches.append([round(x, 4) for x in per])
        return sorted(per)[1], ran

    map_ms, n = timed(lambda k: len(loop.map_frame(rgb, depth, T, k)))      # SlamLoop::MapFrame: the losses are read back once, like Render::RenderForFrame
    track_ms, ran = timed(lambda k: len(loop.track(rgb, depth, T0, k)[0]))  # (it stops early only if the loss stalls: what ran is what is counted)
    same = None
    if world == 1:      # the SAME loop class on the SAME scene without SetShard: what the compositing and its exchange cost at one rank
        plain = _C.SlamLoop(W, H, camd["fx"], camd["fy"], dev)
        plain.set_map(*raw)
        m_ms, _ = timed(lambda k: len(plain.map_frame(rgb, depth, T, k)))
        t_ms, _ = timed(lambda k: len(plain.track(rgb, depth, T0, k)[0]))
        same = {"mapping_ms_per_iter": m_ms, "tracking_ms_per_iter": t_ms}
        del plain
    plane = W * H * 4
    ex = band_exchange_bytes(W, H, world, rank)
    return {"what": "one sharded mapping iteration / one sharded tracking iteration of the C++ loop (ORB_SLAM2::SlamLoop with SetShard, torch_ext/DirectLoop.cpp) — round 6, the BAND "
                    "exchange: fused rasterizer pair on the rank's k-d cell, gsr_shard_order, one grouped point-to-point exchange (every rank's layer on this rank's band of pixel "
                    "rows), gsr_band_composite_forward, the fused loss kernels on the band, gsr_band_composite_backward (every rank's layer gradient on the band), the second "
                    "exchange (each rank's rows back; the mapping loss's sums ride in it), backward with the Adam step fused (mapping) or the pose sums out of the backward's "
                    "per-splat stage + all-reduce of 784 floats + gsr_pose_finish (tracking) — collectives INSIDE the timed region" if band else
                    "round 5's replicated composite (GSR_BENCH_BAND=0): layer all-gather, gsr_composite_forward, all-reduce, the fused loss kernels on the whole frame, "
                    "gsr_composite_backward_* around an all-gather, backward",
            "exchange": "band (LoopConfig::band_exchange)" if band else "replicated composite",
            "scaling": "strong", "total_splats": a.splats, "splats_per_rank": int(idx.numel()), "width": W, "height": H,
            "partition": f"k-d cells x{world} (sharded.KdPartition)", "backend": backend, "rccl_ranks": (td.get_world_size() if world > 1 else (1 if own_group else 0)),
            "mapping_ms_per_iter": map_ms, "tracking_ms_per_iter": track_ms, "tracking_iterations_run": ran,
            "unsharded_same_scene": same, "warmup_iters": max(a.shard_steps, 20), "transport": transport, "batches_ms": batches,
            "mapping_splats_pixels_per_s": 2 * a.splats * W * H / (map_ms * 1e-3),
            "collectives_per_mapping_iter": ({"grouped_p2p_exchanges": 2, "bytes_sent_per_rank": ex["map_fwd"] + ex["bwd"] + 64 * (world - 1), "all_reduce_floats": 0} if band else
                                             {"all_gather_bytes_sent_per_rank": 3 * plane * (world - 1), "all_reduce_bytes": 4 * plane, "all_reduce_floats": 3}),
            "collectives_per_tracking_iter": ({"grouped_p2p_exchanges": 2, "bytes_sent_per_rank": ex["track_fwd"] + ex["track_bwd"], "all_reduce_floats": 784} if band else
                                              {"all_gather_bytes_sent_per_rank": 3 * plane * (world - 1), "all_reduce_bytes": 4 * plane, "all_reduce_floats": 512 * 12}),
            "timed_iters": n}


def shard_render(a, gsr, td, rank, world, dev, weak=False):
    """The rasterize path WITH its exchange, through the C++ loop's sharded render (ORB_SLAM2::SlamLoop::ShardRenderStep, torch_ext/DirectLoop.cpp).
    Strong scaling (default): --splats Gaussians in total, cut into `world` k-d cells (sharded.KdPartition). Weak scaling (weak=True; the use case
    BASELINE configs 4-5 describe: a map that grows with the node): --splats Gaussians PER RANK — rank r owns the r-th of `world` equal depth slabs of the
    view's depth range, the map is world x --splats Gaussians. Per step every rank renders its cell (fused colour + depth / silhouette pass), the layers
    are composited (all-gather of 2 floats/pixel/rank + one all-reduce of 4 planes), a fixed upstream gradient is taken back through the composite
    (all-gather of 1 float/pixel/rank) and the rasterizer's backward to the Gaussians and the pose, and the pose rows are all-reduced (24 KB) —
    north_star: 'shards Gaussians across the GPUs with an RCCL all-reduce on pose/loss gradients only'. The collectives are RCCL calls on the loop's stream."""
    syn = gsr.synthetic
    sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
    sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))
    from diff_gaussian_rasterization import _C
    camd = syn.CAMERAS[a.camera]
    cam = syn.make_camera(**camd)
    W, H = cam.width, cam.height
    t = lambda x: torch.tensor(x, dtype=torch.float32)
    if weak:   # every rank generates only ITS slab: depth range [0.5, 6.0] cut into `world` equal parts, --splats Gaussians each; the slabs ARE the front-to-back order
        z0, z1 = 0.5 + 5.5 * rank / world, 0.5 + 5.5 * (rank + 1) / world
        sc = syn.make_scene(a.splats, cam, seed=1234 + rank, scale_mult=a.scale_mult, z_range=(z0, z1))
        idx, nodes, how = torch.arange(a.splats), torch.empty(0), "depth slabs"
    else:
        sc = syn.make_scene(a.splats, cam, seed=1234, scale_mult=a.scale_mult)      # the SAME scene on every rank
        part = sharded.KdPartition.build(t(sc.means3D), world)
        idx, nodes, how = torch.nonzero(part.assign(t(sc.means3D)) == rank).squeeze(-1), part.nodes, "k-d cells"
    total = a.splats * world if weak else a.splats
    op = t(sc.opacities).reshape(-1, 1)
    raw = [t(sc.means3D)[idx], t(sc.colors)[idx], t(sc.rotations)[idx], torch.log(op / (1 - op))[idx], torch.log(t(sc.scales))[idx]]
    loop = _C.SlamLoop(W, H, camd["fx"], camd["fy"], dev)
    loop.set_map(*raw)
    loop.set_shard(td.group.WORLD if world > 1 else None, rank, world, nodes)
    Tcw = torch.eye(4, device=dev)
    G = torch.randn((5, H, W), generator=torch.Generator().manual_seed(7)).to(dev).contiguous()

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup + 10):
        loop.shard_render_step(Tcw, G)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loop.shard_render_step(Tcw, G)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / max(a.steps, 1) * 1e3
    plane = W * H * 4
    ex = band_exchange_bytes(W, H, world, rank)
    return {"what": "sharded fwd+bwd rasterize with its exchange through the C++ loop (SlamLoop::ShardRenderStep): fused colour + depth/silhouette pass of the rank's cell, "
                    "the band exchange (every rank's layer on this rank's band of rows; gsr_band_composite_forward / _backward; each rank's gradient rows back), backward, "
                    "pose-row all-reduce; collectives (RCCL on the loop's stream) INSIDE the timed region",
            "scaling": "weak" if weak else "strong", "partition": how, "total_splats": total, "splats_per_rank": int(idx.numel()), "width": W, "height": H, "ms_per_step": ms,
            "value": total * W * H / (ms * 1e-3), "unit": "splats*pixels/s",
            "backend": (td.get_backend() if world > 1 else "none (single process)"), "ranks": (td.get_world_size() if world > 1 else 1),
            "transport": loop.shard_transport(),
            "collective_bytes_per_rank_per_step": {"p2p_forward_sent": ex["render_fwd"], "p2p_backward_sent": ex["bwd"], "all_reduce_pose": 512 * 12 * 4 if world > 1 else 0,
                                                   "round5_replicated_composite_sent": (3 * plane * (world - 1) + 2 * 4 * plane * (world - 1) // max(world, 1)) if world > 1 else 0}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm", type=int, default=150, help="untimed steps before the warm-up steps (clock ramp; see rasterize())")
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--camera", choices=["replica", "tum", "scannet"], default="replica",
                    help="replica = the headline 1200x680 frame; scannet + --splats 2000000 = the config-5 shape")
    ap.add_argument("--mode", choices=["all", "rasterize", "shard-step"], default="all",
                    help="rasterize: the headline fwd+bwd line only; shard-step: only the sharded mapping/tracking step; all: both")
    ap.add_argument("--shard-steps", type=int, default=20, help="timed iterations of each shard_step loop")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: which scene-shard mode gives the headline value (strong: --splats in total; weak: --splats per rank, the map grows "
                         "N-fold); the other mode is reported beside it (shard_render_weak / shard_render_strong)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-other", action="store_true", help="skip the other_workloads / loop_ms blocks (N = 1 only, after the headline)")
    ap.add_argument("--other-steps", type=int, default=50, help="timed steps of each entry of other_workloads")
    ap.add_argument("--loop-warmup", type=int, default=60, help="loop_ms: untimed tracking + mapping iterations of the C++ loop driver before its clock starts (0: round 5's cold figure)")
    ap.add_argument("--depth-layout", choices=["uniform", "two-walls"], default="uniform",
                    help="experiment: 'two-walls' moves every splat along its pixel ray onto one of two thin depth slabs "
                         "(1.5 m and 4 m, 2 cm thick): every tile list has two depth clusters, the tile sort's hard case")
    ap.add_argument("--splat-order", choices=["map", "tile"], default="map",
                    help="experiment: 'tile' hands the splats over sorted by the 16x16 tile of their projected centre "
                         "(what a spatially coherent map order would buy the gathers); 'map' = the headline workload")
    ap.add_argument("--scale-mult", type=float, default=1.0, help="splat size multiplier (1 = the headline workload; >1: fatter splats, for experiments)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = world > 1
    td = None
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the backend of every real run. GSR_BENCH_BACKEND=gloo exists so that the N > 1 code path can
        # be exercised on a box with fewer GPUs than ranks (ranks then share devices: `local_rank % device_count`);
        # its numbers mean nothing.
        backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend)
    ndev = max(torch.cuda.device_count(), 1)
    if world > 1 and os.environ.get("GSR_BENCH_BACKEND", "nccl") == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)

    gsr = entry.load_package()
    gsr.lib()  # fails loudly if the HIP extension is missing
    out = {}
    if a.mode in ("all", "rasterize"):
        out = rasterize(a, gsr, td, rank, world, dev)
        if world == 1 and not a.no_other and a.splats == 1_000_000 and a.camera == "replica" and a.scale_mult == 1.0 and a.depth_layout == "uniform":
            out["other_workloads"] = other_workloads(a, gsr, dev)
            out["loop_ms"] = cpp_loop_ms(a, gsr, dev)
        sr = shard_render(a, gsr, td, rank, world, dev, weak=(a.scaling == "weak"))
        sr_other = shard_render(a, gsr, td, rank, world, dev, weak=(a.scaling != "weak")) if world > 1 else None
        if world > 1:
            assert td.get_world_size() == a.gpus == world, (td.get_world_size(), a.gpus, world)   # one rank per GPU, all of them in the collectives
        if rank == 0:
            if world > 1:
                # N > 1: the headline value is the step that EXCHANGES data (strong scaling: the same --splats scene split over the
                # ranks); the collective-free replica figure (every rank its own scene: linear by construction) is kept beside it
                out["replica_rasterize"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "scaling": "weak",
                                            "what": "every rank renders its OWN --splats scene, no collective in the timed region"}
                out.update(value=sr["value"], ms_per_step=sr["ms_per_step"], scaling=sr["scaling"])
                what = (f"{a.splats} Gaussians PER RANK ({sr['total_splats']} in the map: rank r owns the r-th depth slab)" if a.scaling == "weak"
                        else f"{a.splats} Gaussians in TOTAL cut into k-d cells over {world} ranks")
                out["config"]["workload"] = (what + f", {sr['width']}x{sr['height']}, sharded fwd+bwd rasterize through the C++ loop (SlamLoop::ShardRenderStep) with layer "
                                             f"compositing (three HIP kernels, gsr_composite_*) and pose-row all-reduce inside the timed region; backend {sr['backend']} ({sr['ranks']} ranks)")
                out["config"]["parallelism"] = f"scene shards ({sr['partition']}) x{world}, {sr['scaling']} scaling, RCCL on the loop's stream (bootstrap: torch.distributed {sr['backend']})"
                out["config"]["splats_total"] = sr["total_splats"]
                out.pop("step_ms_percentiles", None)
                out["shard_render_" + sr_other["scaling"]] = sr_other
            out["shard_render"] = sr
    if a.mode in ("all", "shard-step"):
        ss = shard_step(a, gsr, td, rank, world, dev)
        if a.mode == "shard-step":
            out = {"metric": "ms per sharded mapping iteration (collectives in the timed region)", "value": ss["mapping_ms_per_iter"],
                   "unit": "ms", "n_gpus": world, "steps": ss["timed_iters"], "warmup": 3, "ms_per_step": ss["mapping_ms_per_iter"],
                   "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": f"{a.splats} Gaussians in total, k-d cells over {world} rank(s), {ss['width']}x{ss['height']}"}}
        out["shard_step"] = ss
    if _ONE_RANK["td"] is not None:
        _ONE_RANK["td"].destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)   # (RCCL prints a version banner through C stdio: out before the line, which must be the last one)
        print(json.dumps(out), flush=True)
    if dist:
        td.destroy_process_group()


def rasterize(a, gsr, td, rank, world, dev):
    dist = world > 1
    syn = gsr.synthetic
    P = a.splats
    cam = syn.make_camera(**syn.CAMERAS[a.camera])
    W, H = cam.width, cam.height
    sc = syn.make_scene(P, cam, seed=rank, scale_mult=a.scale_mult)  # each rank: its own scene shard
    if a.depth_layout == "two-walls":
        rng = np.random.default_rng(3)
        z = np.where(rng.random(P) < 0.5, 1.5, 4.0) + 0.02 * rng.random(P)
        sc.means3D = (sc.means3D * (z / sc.means3D[:, 2])[:, None]).astype(np.float32)
    if a.splat_order == "tile":
        u = sc.means3D[:, 0] / sc.means3D[:, 2] * cam.fx + cam.width / 2
        v = sc.means3D[:, 1] / sc.means3D[:, 2] * cam.fy + cam.height / 2
        key = (np.clip(v // 16, 0, 4095).astype(np.int64) << 12) | np.clip(u // 16, 0, 4095).astype(np.int64)
        perm = np.argsort(key, kind="stable")
        for n in ("means3D", "opacities", "colors", "scales", "rotations"):
            setattr(sc, n, np.ascontiguousarray(getattr(sc, n)[perm]))
    s = gsr.capi.Settings.from_camera(cam, device=dev)
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    ins = dict(means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), shs=None,
               scales=t(sc.scales), rotations=t(sc.rotations), cov3D=None)
    grad_in = t(sc.dL_dpix)

    # size the persistent workspace from one untimed callback-mode forward
    st0 = gsr.forward(s, ins["means3D"], ins["opacities"], colors=ins["colors"], scales=ins["scales"],
                      rotations=ins["rotations"])
    R = st0.num_rendered
    V = int((st0.radii > 0).sum())
    del st0
    ws = gsr.capi.Workspace(P, W, H, max_rendered=int(R * 1.25) + 1024, device=dev)
    # the gradient buffers the operator wrappers pass on the scales + rotations path (torch_ext/Rasterizer.cpp): the two
    # intermediates nobody consumes there, dL_dconic and dL_dcov3D, are NULL
    grads = gsr.capi.alloc_grads(P, 0, dev, intermediates=False)

    hip = _hip()
    n_ev = max(min(a.steps, 32), 1)   # HIP-event pairs on the first <= 32 timed steps (1 200 timing events in flight made a 200-step run 15 % slower on the wall clock than its own events said)
    ev = []  # per timed step: (start, stop) around the backward blend kernel, and around the forward blend
    for _ in range(n_ev):
        e = [C.c_void_p() for _ in range(6)]
        for x in e:
            hip.hipEventCreate(C.byref(x))
        ev.append(e)
    NB, NF = 3, 5
    def ev_arrays(e):
        f = (C.c_void_p * (2 * NF))()
        b = (C.c_void_p * (2 * NB))()
        f[2 * 4], f[2 * 4 + 1] = e[2], e[3]     # GSR_FWD_BLEND
        b[2 * 1], b[2 * 1 + 1] = e[0], e[1]     # GSR_BWD_BLEND
        return f, b

    stream = torch.cuda.current_stream().cuda_stream

    def step(fe=None, be=None, e=None):
        if e is not None:
            hip.hipEventRecord(e[4], C.c_void_p(stream))
        st = gsr.forward_ws(s, ws, ins, None, events=fe)
        gsr.backward(st, grad_in, grads=grads, events=be, once=True)  # one backward per forward: no accumulator re-zero
        if e is not None:
            hip.hipEventRecord(e[5], C.c_void_p(stream))

    # The driver times 20 steps after 5 warm-up steps: 14 ms of work, during which the clocks are still ramping (round 2:
    # 0.565 ms/step with --steps 20 --warmup 5 against 0.518 with 200 / 20). A fixed, untimed pre-warm brings the chip to
    # its steady state first, so that short and long runs report the same step.
    for _ in range(a.prewarm):
        step()
    for _ in range(a.warmup):
        step()
    n, ovf = ws.status()
    assert not ovf and n == R, (n, R, ovf)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    arrays = [ev_arrays(e) for e in ev]
    barrier()
    t0 = time.perf_counter()
    # HIP-event pairs (around the step and around its two blend kernels: six records) ride on every FOURTH timed step: each record is a marker
    # the stream serialises on, and with all six on every step the timed region ran 0.470 ms/step where the same launches without them run 0.444
    # (measured: every step 0.470, every 2nd 0.457, every 4th 0.455; GSR_BENCH_EVENT_EVERY=1 restores the old behaviour). The roofline's kernel
    # durations are the averages over the sampled launches — still live, still inside the timed region, on the launching stream.
    evk = max(int(os.environ.get("GSR_BENCH_EVENT_EVERY", "4")), 1)
    for i in range(a.steps):
        if i % evk == 0 and i // evk < n_ev:
            step(*arrays[i // evk], ev[i // evk])
        else:
            step()
    n_used = min((a.steps + evk - 1) // evk, n_ev)
    host_dt = time.perf_counter() - t0          # what the host needed to ENQUEUE the timed steps (it must stay below dt or the run is host-bound)
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())

    ms_step = dt / max(a.steps, 1) * 1e3
    value = world * P * W * H / (dt / max(a.steps, 1))

    def avg_ms(i0, i1):
        tot = 0.0
        for e in ev[:n_used]:
            ms = C.c_float(0)
            hip.hipEventSynchronize(e[i1])
            hip.hipEventElapsedTime(C.byref(ms), e[i0], e[i1])
            tot += ms.value
        return tot / max(n_used, 1)

    def step_percentiles():
        ts = []
        for e in ev[:n_used]:
            ms = C.c_float(0)
            hip.hipEventSynchronize(e[5])
            hip.hipEventElapsedTime(C.byref(ms), e[4], e[5])
            ts.append(ms.value)
        ts = np.sort(np.asarray(ts)) if ts else np.zeros(1)
        return {"p10": float(np.percentile(ts, 10)), "p50": float(np.percentile(ts, 50)), "p90": float(np.percentile(ts, 90)),
                "what": "GPU time of one fwd+bwd step between HIP events on the launching stream, per SAMPLED timed step (every %d-th: a sampled step carries six event records and runs ~15 us slower than an unsampled one)" % evk,
                "sampled_steps": n_used}

    if rank == 0:
        bwd_blend_ms = avg_ms(0, 1)
        fwd_blend_ms = avg_ms(2, 3)
        N = W * H
        # algorithmic bytes of the backward blend kernel per launch: 40R + 20N + 36V
        alg_bytes = 40 * R + 20 * N + 36 * V
        achieved = alg_bytes / (bwd_blend_ms * 1e-3) / 1e9
        total_alg_ref = 152 * P + 340 * V + 128 * R + 44 * N   # whole fwd+bwd as SURVEY.md §8d counts it (every store of the reference)
        total_alg = total_alg_ref - 40 * V                     # ... minus the two intermediates the timed call does not store (dL_dconic 16 B, dL_dcov3D 24 B)
        traffic = None   # HBM/fabric bytes per launch of the dominant kernel: PMC counters cannot be read live,
        pmc = {}         # so the committed rocprofv3 --pmc summary of this same command is quoted
        mix = {}
        quoted = {"traffic": None, "valu_mix": None}   # which committed profile each QUOTED figure comes from (the newest that exists)
        def newest(names):
            for n in names:
                if os.path.exists(os.path.join(ROOT, "profiles", n)):
                    return n
            return None
        try:
            quoted["traffic"], quoted["valu_mix"] = newest(["r06_traffic.json", "r05_traffic.json", "r04_traffic.json"]), newest(["r06_valu_mix.json", "r05_valu_mix.json", "r03_valu_mix.json"])
            tj = json.load(open(os.path.join(ROOT, "profiles", quoted["traffic"])))
            mix = json.load(open(os.path.join(ROOT, "profiles", quoted["valu_mix"])))["kernels"]
            if P == 1_000_000 and a.camera == "replica" and a.scale_mult == 1.0:
                pmc = tj["kernels"]
                traffic = pmc["K_blend_bwd"]["traffic_bytes"]
        except Exception:
            pmc = {}

        def valu_roofline(kernel, launch_ms, body_ops):
            """VALU pipe occupancy of the kernel: wave-level VALU instructions (PMC) x cycles per instruction (the kernel's own class
            mix: 2 cycles for v_add/mul/fma, 4 for compares / selects / min / max / shifts / DPP, 8 for exp / rcp —
            scripts/valu_mix.py, profiles/r03_valu_mix.json) / (1024 SIMDs x the clock the counters saw x launch time)."""
            k = pmc.get(kernel)
            if not k or "valu_instructions" not in k:
                return None
            vi = k["valu_instructions"]
            cpi = mix.get(kernel, {}).get("cycles_per_instruction", 4.0)
            clock = k.get("clock_ghz", 2.4)
            d = {"kernel": kernel, "valu_insts_per_launch": vi, "cycles_per_inst": cpi, "simds": 1024, "clock_ghz": clock,
                 "avg_launch_ms": launch_ms, "frac": vi * cpi / (1024 * clock * 1e9) / (launch_ms * 1e-3),
                 "lds_pipe_busy": k.get("lds_pipe_busy"), "waves_per_simd": k.get("waves_per_simd")}
            d["body_ops_per_pair"] = body_ops
            d["quoted_from"] = {"valu_insts_per_launch, clock_ghz, lds_pipe_busy, waves_per_simd": "profiles/" + str(quoted["traffic"]) + " (rocprofv3 --pmc of this command, committed)",
                                "cycles_per_inst": "profiles/" + str(quoted["valu_mix"]) + " (instruction-class mix of the kernel's assembly)",
                                "avg_launch_ms": "live: HIP events of this run"}
            return d
        rv = valu_roofline("K_blend_bwd", bwd_blend_ms, 44)   # 33 in the per-pixel loop + 11 per pixel in the reduce phase
        rv_f = valu_roofline("K_blend_fwd", fwd_blend_ms, 23)
        out = {
            "metric": "splats*pixels/s (fwd+bwd) @1M Gaussians 1200x680" if (P == 1_000_000 and a.camera == "replica")
                      else f"splats*pixels/s (fwd+bwd) @{P} Gaussians {W}x{H}",
            "value": value, "unit": "splats*pixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "host_enqueue_ms_per_step": host_dt / max(a.steps, 1) * 1e3, "step_ms_percentiles": step_percentiles(), "prewarm_steps": a.prewarm, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{P} random-init Gaussians per GPU (SinglePixel scale init, camera frame), "
                                   f"{W}x{H} {a.camera} camera, RGB colours, fwd+bwd rasterize through the C ABI "
                                   f"(gsr_forward_ws + gsr_backward), inputs resident in HBM",
                       "splats": P, "width": W, "height": H, "visible": V, "tile_instances": R,
                       "parallelism": f"scene-shard x{world}" if world > 1 else "single GPU",
                       "omitted_stores": "dL_dconic and dL_dcov3D are not materialised (the operator wrappers pass no buffer for them on the "
                                         "scales + rotations path: INTEGRATION.md section 3); the reference kernel stores both"},
            "roofline": {"bound": "hbm", "limiter": "instruction issue and the latency of a wave's own instruction chain at 12 waves per CU (VALU pipe 0.55 busy, "
                                                    "LDS pipe 0.66), not HBM: see roofline_valu and DESIGN.md section 4 — the contract's HBM figure is "
                                                    "reported as asked",
                         "kernel": "K_blend_bwd", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_quoted_from": None if traffic is None else "profiles/" + str(quoted["traffic"]) + " (PMC counters cannot be read live: the committed "
                                                                                "rocprofv3 --pmc summary of this same command; achieved / avg_launch_ms are live)",
                         "algorithmic_bytes": alg_bytes, "avg_launch_ms": bwd_blend_ms,
                         "fwd_blend_avg_launch_ms": fwd_blend_ms,
                         "whole_step": {"algorithmic_bytes": total_alg, "algorithmic_bytes_with_reference_intermediates": total_alg_ref,
                                        "note": "algorithmic_bytes counts the stores the timed call makes (no dL_dconic / dL_dcov3D: config.omitted_stores); "
                                                "one backward per forward: the accumulators are cleared by the forward blend's tail, not by a re-zero in the backward "
                                                "(the stages the operator wrappers pass: torch_ext/Rasterizer.cpp)",
                                        "achieved": total_alg / (ms_step * 1e-3) / 1e9,
                                        "frac": total_alg / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(sc, P, W, H)
            out["parity"] = parity_block(gsr, sc, s, ws, ins, dev)
            out["boundary_ms"] = {"value": boundary_ms(gsr, sc, dev),
                                  "what": "fwd+bwd through diff_gaussian_rasterization.GaussianRasterizer (Python op -> libtorch host layer -> C ABI), "
                                          "20 calls, wall clock per call pair including allocations and the one host read of num_rendered"}
        if rv is not None:
            # useful_lane_frac: lane-instructions the blend arithmetic of the pairs that were actually blended needs
            # (census of the same scene by the CPU oracle x VALU ops per pair in the loop body) / all lane slots issued
            bp = out.get("cpu_baseline", {}).get("blended_pairs")
            for d in (rv, rv_f):
                if d is not None:
                    d["blended_pairs"] = bp
                    d["useful_lane_frac"] = None if bp is None else bp * d["body_ops_per_pair"] / (d["valu_insts_per_launch"] * 64.0)
            out["roofline_valu"] = rv
            if rv_f is not None:
                out["roofline_valu"]["forward"] = rv_f
        return out
    return {}


if __name__ == "__main__":
    main()
