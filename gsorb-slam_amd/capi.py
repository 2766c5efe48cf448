"""ctypes binding of the C ABI (include/gsr.h) on torch device tensors.

torch is used only as the owner of device memory and of the HIP stream; every
computation happens in csrc/libgsr_hip.so. Missing library => ImportError at
first use (no fallback of any kind).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = ["gsr_forward", "gsr_forward_ws", "gsr_ws_status", "gsr_backward", "gsr_mark_visible",
           "gsr_visible_filter", "gsr_geom_bytes", "gsr_image_bytes", "gsr_binning_bytes",
           "gsr_debug_export", "gsr_acc_view", "gsr_knn_bytes", "gsr_dist2", "gsr_ssim_partials", "gsr_ssim_forward", "gsr_ssim_backward",
           "gsr_adam_step", "gsr_pose_grad", "gsr_to_camera", "gsr_pose_from_quat", "gsr_pose_from_quat_backward",
           "gsr_pixel_loss", "gsr_pixel_loss_backward", "gsr_pixel_loss_backward_add", "gsr_track_loss", "gsr_scale_reg", "gsr_scale_reg_backward",
           "gsr_map_prepare", "gsr_map_update", "gsr_map_loss_total", "gsr_map_loss_forward", "gsr_map_loss_finish", "gsr_map_loss_backward", "gsr_pose_update", "gsr_pose_step", "gsr_pose_finish", "gsr_composite_forward", "gsr_composite_backward_local",
           "gsr_composite_backward_occlusion", "gsr_shard_order", "gsr_reproj_loss", "gsr_error_string", "gsr_last_hip_error", "gsr_abi_version",
           "gsr_debug_launch_count", "gsr_band_composite_forward", "gsr_band_composite_backward", "gsr_shard_map_totals", "gsr_map_loss_partials_rows",
           "gsr_map_loss_forward_rows", "gsr_map_loss_finish_rows", "gsr_map_loss_backward_rows", "gsr_track_loss_rows", "gsr_transmittance_view"]


def library_path() -> str:
    return os.environ.get("GSR_LIB_OVERRIDE") or os.path.join(_HERE, "csrc", "libgsr_hip.so")


class GsrError(RuntimeError):
    pass


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class ForwardArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("background", C.c_void_p),
                ("width", C.c_int), ("height", C.c_int), ("means3D", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("scale_modifier", C.c_float), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("prefiltered", C.c_int),
                ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("radii", C.c_void_p),
                ("profile_events", C.POINTER(C.c_void_p)), ("band_y0", C.c_int), ("band_y1", C.c_int),
                ("out_ds", C.c_void_p), ("pre_Tcw", C.c_void_p), ("means_cam_out", C.c_void_p), ("raw", C.c_void_p), ("out_sil", C.c_void_p)]


class RawOutputs(C.Structure):
    _fields_ = [("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("reg_limit", C.c_float), ("reg_partial", C.c_void_p)]


class BackwardArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("R", C.c_int),
                ("background", C.c_void_p), ("width", C.c_int), ("height", C.c_int),
                ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("scales", C.c_void_p), ("scale_modifier", C.c_float), ("rotations", C.c_void_p),
                ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
                ("cam_pos", C.c_void_p), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("radii", C.c_void_p), ("geom_buffer", C.c_void_p), ("binning_buffer", C.c_void_p),
                ("image_buffer", C.c_void_p), ("binning_bytes", C.c_size_t), ("dL_dpix", C.c_void_p),
                ("dL_dmean2D", C.c_void_p), ("dL_dconic", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_dcolor", C.c_void_p), ("dL_dmean3D", C.c_void_p), ("dL_dcov3D", C.c_void_p),
                ("dL_dsh", C.c_void_p), ("dL_dscale", C.c_void_p), ("dL_drot", C.c_void_p),
                ("profile_events", C.POINTER(C.c_void_p)), ("band_y0", C.c_int), ("band_y1", C.c_int),
                ("stages", C.c_int), ("dL_dds", C.c_void_p), ("ds_detach_depth", C.c_int), ("fused_map_update", C.c_void_p), ("dds_depth_only", C.c_int),
                ("fused_pose_step", C.c_void_p)]


class DebugArrays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("means2D", "depths", "conic_opacity", "rgb", "tiles_touched",
                                          "point_list", "point_list_keys", "ranges", "final_T",
                                          "n_contrib")]


class MapUpdateArgs(C.Structure):
    """gsr_map_update_args (include/gsr.h)"""
    _fields_ = [("n", C.c_size_t), ("xyz", C.c_void_p), ("rgb", C.c_void_p), ("unnorm_quat", C.c_void_p), ("logit", C.c_void_p),
                ("log_scales", C.c_void_p), ("exp_avg", C.c_void_p * 5), ("exp_avg_sq", C.c_void_p * 5), ("dL_dmeans_cam", C.c_void_p),
                ("dL_dcolors", C.c_void_p), ("dL_drotations", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("opacities", C.c_void_p), ("scales", C.c_void_p), ("Tcw", C.c_void_p), ("reg_out", C.c_void_p), ("reg_limit", C.c_float),
                ("w_long", C.c_float), ("w_scalar", C.c_float), ("geom", C.c_void_p), ("lr", C.c_double * 5), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int * 5)]


class PoseStepArgs(C.Structure):
    _fields_ = [("means_world", C.c_void_p), ("update", C.c_void_p), ("sums_only", C.c_int), ("overflow_out", C.c_void_p)]


class PoseUpdateArgs(C.Structure):
    """gsr_pose_update_args (include/gsr.h)"""
    _fields_ = [("quat_trans", C.c_void_p), ("moments", C.c_void_p), ("best", C.c_void_p), ("history", C.c_void_p), ("Tcw", C.c_void_p),
                ("partial", C.c_void_p), ("loss", C.c_void_p), ("geom", C.c_void_p), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int), ("skip", C.c_void_p)]


def lib():
    """Load csrc/libgsr_hip.so (built by __graft_entry__.build()); raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "— there is no CPU fallback for the rasterizer")
    L = C.CDLL(path)
    L.gsr_forward.restype = C.c_int
    L.gsr_forward.argtypes = [C.POINTER(ForwardArgs), ALLOC_FN, C.c_void_p, ALLOC_FN, C.c_void_p, ALLOC_FN,
                              C.c_void_p, C.c_void_p]
    L.gsr_forward_ws.restype = C.c_int
    L.gsr_forward_ws.argtypes = [C.POINTER(ForwardArgs), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.gsr_ws_status.restype = C.c_int
    L.gsr_ws_status.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gsr_backward.restype = C.c_int
    L.gsr_backward.argtypes = [C.POINTER(BackwardArgs), C.c_void_p]
    L.gsr_mark_visible.restype = C.c_int
    L.gsr_mark_visible.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_visible_filter.restype = C.c_int
    L.gsr_visible_filter.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    for n in ("gsr_geom_bytes", "gsr_image_bytes", "gsr_binning_bytes"):
        getattr(L, n).restype = C.c_size_t
    L.gsr_geom_bytes.argtypes = [C.c_int]
    L.gsr_image_bytes.argtypes = [C.c_int, C.c_int]
    L.gsr_binning_bytes.argtypes = [C.c_size_t]
    L.gsr_debug_export.restype = C.c_int
    L.gsr_debug_export.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(DebugArrays), C.c_void_p]
    L.gsr_acc_view.restype = C.c_int
    L.gsr_acc_view.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.gsr_knn_bytes.restype = C.c_size_t
    L.gsr_knn_bytes.argtypes = [C.c_int]
    L.gsr_dist2.restype = C.c_int
    L.gsr_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gsr_ssim_partials.restype = C.c_size_t
    L.gsr_ssim_partials.argtypes = [C.c_int, C.c_int, C.c_int]
    L.gsr_ssim_forward.restype = C.c_int
    L.gsr_ssim_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_ssim_backward.restype = C.c_int
    L.gsr_ssim_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    L.gsr_pose_grad.restype = C.c_int
    L.gsr_pose_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_to_camera.restype = C.c_int
    L.gsr_to_camera.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_pose_from_quat.restype = C.c_int
    L.gsr_pose_from_quat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_pose_from_quat_backward.restype = C.c_int
    L.gsr_pose_from_quat_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_adam_step.restype = C.c_int
    L.gsr_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_int, C.c_void_p]
    L.gsr_pixel_loss.restype = C.c_int
    L.gsr_pixel_loss.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_pixel_loss_backward.restype = C.c_int
    L.gsr_pixel_loss_backward.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)] + [C.c_void_p] * 5
    L.gsr_scale_reg.restype = C.c_int
    L.gsr_scale_reg.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_scale_reg_backward.restype = C.c_int
    L.gsr_scale_reg_backward.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_pixel_loss_backward_add.restype = C.c_int
    L.gsr_pixel_loss_backward_add.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)] + [C.c_void_p] * 6
    L.gsr_pose_finish.restype = C.c_int
    L.gsr_pose_finish.argtypes = [C.POINTER(PoseUpdateArgs), C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_pose_step.restype = C.c_int
    L.gsr_pose_step.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(PoseUpdateArgs), C.c_void_p, C.c_void_p]
    L.gsr_track_loss.restype = C.c_int
    L.gsr_track_loss.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)] + [C.c_void_p] * 6
    L.gsr_map_prepare.restype = C.c_int
    L.gsr_map_prepare.argtypes = [C.c_size_t] + [C.c_void_p] * 9 + [C.c_float] * 3 + [C.c_void_p] * 3
    L.gsr_map_update.restype = C.c_int
    L.gsr_map_update.argtypes = [C.POINTER(MapUpdateArgs), C.c_void_p]
    L.gsr_map_loss_total.restype = C.c_int
    L.gsr_map_loss_total.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_composite_forward.restype = C.c_int
    L.gsr_composite_forward.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_composite_backward_local.restype = C.c_int
    L.gsr_composite_backward_local.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_composite_backward_occlusion.restype = C.c_int
    L.gsr_composite_backward_occlusion.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.gsr_reproj_loss.restype = C.c_int
    L.gsr_reproj_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_shard_order.restype = C.c_int
    L.gsr_shard_order.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_map_loss_forward.restype = C.c_int
    L.gsr_map_loss_forward.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gsr_map_loss_finish.restype = C.c_int
    L.gsr_map_loss_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 5
    L.gsr_map_loss_backward.restype = C.c_int
    L.gsr_map_loss_backward.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)] + [C.c_void_p] * 5
    L.gsr_pose_update.restype = C.c_int
    L.gsr_pose_update.argtypes = [C.POINTER(PoseUpdateArgs), C.c_void_p]
    # round 6: the band exchange's kernels and the loss kernels on a band of rows
    L.gsr_band_composite_forward.restype = C.c_int
    L.gsr_band_composite_forward.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4
    L.gsr_band_composite_backward.restype = C.c_int
    L.gsr_band_composite_backward.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3
    L.gsr_shard_map_totals.restype = C.c_int
    L.gsr_shard_map_totals.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 4
    L.gsr_map_loss_partials_rows.restype = C.c_size_t
    L.gsr_map_loss_partials_rows.argtypes = [C.c_int] * 4
    L.gsr_map_loss_forward_rows.restype = C.c_int
    L.gsr_map_loss_forward_rows.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.gsr_map_loss_finish_rows.restype = C.c_int
    L.gsr_map_loss_finish_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]
    L.gsr_map_loss_backward_rows.restype = C.c_int
    L.gsr_map_loss_backward_rows.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]
    L.gsr_track_loss_rows.restype = C.c_int
    L.gsr_track_loss_rows.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)] + [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.gsr_transmittance_view.restype = C.c_int
    L.gsr_transmittance_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.gsr_error_string.restype = C.c_char_p
    L.gsr_error_string.argtypes = [C.c_int]
    L.gsr_last_hip_error.restype = C.c_char_p
    L.gsr_abi_version.restype = C.c_int
    L.gsr_debug_launch_count.restype = C.c_ulonglong
    _LIB = L
    return L


def _check(rc: int) -> int:
    if rc < 0:
        L = lib()
        msg = L.gsr_error_string(rc).decode()
        if rc == -3:
            msg += ": " + L.gsr_last_hip_error().decode()
        raise GsrError(f"gsr error {rc}: {msg}")
    return rc


def _p(t):
    if t is None or t.numel() == 0:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def _f32(t, dev):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    return t.to(device=dev, dtype=torch.float32).contiguous()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class Settings:
    """The 11 fields of GaussianRasterizationSettings (include/Rasterizer.cuh:79-91)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False

    @staticmethod
    def from_camera(cam, device="cuda"):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device).contiguous()
        return Settings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, t(cam.bg), cam.scale_modifier,
                        t(cam.viewmatrix), t(cam.projmatrix), cam.sh_degree, t(cam.campos), False)


@dataclass
class ForwardState:
    """What crosses forward -> backward (the reference saves the same things, include/Rasterizer.cuh:190-203)."""
    settings: Settings
    P: int
    M: int
    num_rendered: int
    inputs: dict
    color: torch.Tensor
    depth: torch.Tensor
    radii: torch.Tensor
    geom: torch.Tensor
    binning: torch.Tensor
    image: torch.Tensor
    ws_binning_bytes: int = 0
    band: tuple = (0, 0)
    ds: torch.Tensor | None = None   # [2,H,W] fused depth / silhouette channels (forward(..., dual=True))
    dirty: bool = False   # a backward ran its per-splat stage without GSR_STAGE_REZERO: clear before the next one
    keep: list = field(default_factory=list)


def _fwd_args(s: Settings, means3D, opacities, colors, shs, scales, rotations, cov3D, color, depth, radii,
              events=None, band=(0, 0), ds=None):
    P = int(means3D.shape[0])
    M = 0 if shs is None or shs.numel() == 0 else int(shs.shape[1])
    a = ForwardArgs(P, s.sh_degree, M, _p(s.bg), s.image_width, s.image_height, _p(means3D), _p(shs), _p(colors),
                    _p(opacities), _p(scales), s.scale_modifier, _p(rotations), _p(cov3D), _p(s.viewmatrix),
                    _p(s.projmatrix), _p(s.campos), s.tanfovx, s.tanfovy, int(s.prefiltered), _p(color),
                    _p(depth), _p(radii), events, int(band[0]), int(band[1]), _p(ds))
    return a, P, M


def _prep(s: Settings, means3D, opacities, colors, shs, scales, rotations, cov3D):
    dev = s.viewmatrix.device
    ins = dict(means3D=_f32(means3D, dev), opacities=_f32(opacities, dev), colors=_f32(colors, dev),
               shs=_f32(shs, dev), scales=_f32(scales, dev), rotations=_f32(rotations, dev),
               cov3D=_f32(cov3D, dev))
    if ins["means3D"].dim() != 2 or ins["means3D"].shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")  # src/Rasterizer.cu:158-160
    return dev, ins


def forward(s: Settings, means3D, opacities, colors=None, shs=None, scales=None, rotations=None,
            cov3D_precomp=None, band=(0, 0), out=None, dual: bool = False, out_sil=None) -> ForwardState:
    """gsr_forward with torch-owned blobs (the reference's resizeFunctional, src/Rasterizer.cu:127-134).
    dual: also blend the depth / silhouette channels in the same pass (state.ds [2,H,W]; include/gsr.h: out_ds).
    out_sil [H,W]: the plain forward also stores the silhouette 1 - final T there (include/gsr.h: out_sil; not with dual)."""
    L = lib()
    dev, ins = _prep(s, means3D, opacities, colors, shs, scales, rotations, cov3D_precomp)
    H, W = s.image_height, s.image_width
    P = int(ins["means3D"].shape[0])
    color = out[0] if out is not None else torch.empty((3, H, W), dtype=torch.float32, device=dev)
    depth = out[1] if out is not None else torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((max(P, 1),), dtype=torch.int32, device=dev)
    ds = torch.empty((2, H, W), dtype=torch.float32, device=dev) if dual else None
    blobs = {}

    def mk(name):
        def cb(_user, nbytes):
            t = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)
            blobs[name] = t
            return t.data_ptr()
        return ALLOC_FN(cb)

    cbs = [mk("geom"), mk("binning"), mk("image")]
    a, P, M = _fwd_args(s, ins["means3D"], ins["opacities"], ins["colors"], ins["shs"], ins["scales"],
                        ins["rotations"], ins["cov3D"], color, depth, radii, None, band, ds)
    if out_sil is not None:
        a.out_sil = _p(out_sil)
    with torch.cuda.device(dev):
        R = _check(L.gsr_forward(C.byref(a), cbs[0], None, cbs[1], None, cbs[2], None, _stream()))
    return ForwardState(s, P, M, R, ins, color, depth, radii[:P], blobs["geom"], blobs["binning"],
                        blobs["image"], band=tuple(band), ds=ds)


class Workspace:
    """Persistent blobs for the sync-free gsr_forward_ws / gsr_backward loop."""

    def __init__(self, P: int, width: int, height: int, max_rendered: int, device="cuda"):
        L = lib()
        self.device = torch.device(device)
        self.binning_bytes = int(L.gsr_binning_bytes(max_rendered))
        mk = lambda n: torch.empty((int(n),), dtype=torch.uint8, device=self.device)
        self.geom = mk(L.gsr_geom_bytes(P))
        self.image = mk(L.gsr_image_bytes(width, height))
        self.binning = mk(self.binning_bytes)
        self.P, self.W, self.H = P, width, height
        self.color = torch.empty((3, height, width), dtype=torch.float32, device=self.device)
        self.depth = torch.empty((1, height, width), dtype=torch.float32, device=self.device)
        self.radii = torch.empty((max(P, 1),), dtype=torch.int32, device=self.device)
        self.ds = torch.empty((2, height, width), dtype=torch.float32, device=self.device)   # fused depth / silhouette channels (dual=True)
        self.grads = None

    def status(self):
        n, o = C.c_int(0), C.c_int(0)
        _check(lib().gsr_ws_status(_p(self.geom), _stream(), C.byref(n), C.byref(o)))
        return n.value, bool(o.value)


def forward_ws(s: Settings, ws: Workspace, means3D, opacities, colors=None, shs=None, scales=None,
               rotations=None, cov3D_precomp=None, events=None, dual: bool = False, pre_Tcw=None, means_cam_out=None, raw=None) -> ForwardState:
    """pre_Tcw [4,4] (device, world -> camera) + means_cam_out [P,3]: `means3D` are WORLD means, the projection kernel moves them into the camera
    frame itself and leaves them in means_cam_out (gsr_forward_args.pre_Tcw)."""
    L = lib()
    dev, ins = (s.viewmatrix.device, means3D) if isinstance(means3D, dict) else \
        _prep(s, means3D, opacities, colors, shs, scales, rotations, cov3D_precomp)
    a, P, M = _fwd_args(s, ins["means3D"], ins["opacities"], ins["colors"], ins["shs"], ins["scales"],
                        ins["rotations"], ins["cov3D"], ws.color, ws.depth, ws.radii, events, (0, 0), ws.ds if dual else None)
    assert P == ws.P and s.image_width == ws.W and s.image_height == ws.H
    if pre_Tcw is not None:
        a.pre_Tcw, a.means_cam_out = _p(pre_Tcw), _p(means_cam_out)
    if raw is not None:   # (opacities_out, scales_out, rotations_out, reg_limit, reg_partial or None): the three inputs are RAW parameters
        ro = RawOutputs(_p(raw[0]), _p(raw[1]), _p(raw[2]), float(raw[3]), _p(raw[4]))
        a.raw = C.cast(C.pointer(ro), C.c_void_p)
    _check(L.gsr_forward_ws(C.byref(a), _p(ws.geom), _p(ws.binning), ws.binning_bytes, _p(ws.image), _stream()))
    return ForwardState(s, P, M, -1, ins, ws.color, ws.depth, ws.radii[:P], ws.geom, ws.binning, ws.image,
                        ws_binning_bytes=ws.binning_bytes, ds=ws.ds if dual else None)


@dataclass
class Grads:
    dL_dmeans2D: torch.Tensor
    dL_dconic: torch.Tensor
    dL_dopacity: torch.Tensor
    dL_dcolors: torch.Tensor
    dL_dmeans3D: torch.Tensor
    dL_dcov3D: torch.Tensor
    dL_dsh: torch.Tensor
    dL_dscales: torch.Tensor
    dL_drotations: torch.Tensor


def alloc_grads(P: int, M: int, dev, intermediates: bool = True) -> Grads:
    """intermediates=False: no dL_dconic / dL_dcov3D buffers — what the operator wrappers pass on the scales + rotations
    path, where nothing consumes them (the kernel then skips 40 bytes of stores per splat)."""
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    return Grads(e(P, 3), e(P, 2, 2) if intermediates else None, e(P, 1), e(P, 3), e(P, 3), e(P, 6) if intermediates else None,
                 e(P, M, 3), e(P, 3), e(P, 4))


def backward(st: ForwardState, dL_dpix, grads: Grads | None = None, events=None, stages: int = 0, once: bool = False,
             dL_dds=None, detach_depth_color: bool = False, fused_map_update=None, dds_depth_only: bool = False, fused_pose_step=None) -> Grads:
    """gsr_backward; output shapes follow src/Rasterizer.cu:253-261. `once`: the caller runs one backward per
    forward (blend + per-splat without the re-zero of the accumulators); a later call on the same state
    is then started with a clear. dL_dds [2,H,W]: upstream gradient of the fused depth / silhouette channels."""
    if stages == 0:
        stages = (2 | 4) if once else (2 | 4 | 8)
        if st.dirty:
            stages |= 1
    L = lib()
    s = st.settings
    dev = s.viewmatrix.device
    g = _f32(dL_dpix, dev)
    gds = _f32(dL_dds, dev) if dL_dds is not None else None
    P, M = st.P, st.M
    out = grads if grads is not None else alloc_grads(P, M, dev)
    if P == 0:
        return out
    i = st.inputs
    has_sr = i["scales"] is not None and i["scales"].numel() > 0
    a = BackwardArgs(P, s.sh_degree, M, st.num_rendered, _p(s.bg), s.image_width, s.image_height, _p(i["means3D"]),
                     _p(i["shs"]), _p(i["colors"]), _p(i["scales"]), s.scale_modifier, _p(i["rotations"]),
                     _p(i["cov3D"]), _p(s.viewmatrix), _p(s.projmatrix), _p(s.campos), s.tanfovx, s.tanfovy,
                     _p(st.radii), _p(st.geom), _p(st.binning), _p(st.image), st.ws_binning_bytes, _p(g),
                     _p(out.dL_dmeans2D), _p(out.dL_dconic), _p(out.dL_dopacity), _p(out.dL_dcolors),
                     _p(out.dL_dmeans3D), _p(out.dL_dcov3D), _p(out.dL_dsh) if M > 0 else None,
                     _p(out.dL_dscales) if has_sr else None, _p(out.dL_drotations) if has_sr else None, events,
                     int(st.band[0]), int(st.band[1]), int(stages), _p(gds), int(bool(detach_depth_color)),
                     C.c_void_p(C.addressof(fused_map_update)) if fused_map_update is not None else None,   # (a MapUpdateArgs: map_update_args())
                     int(dds_depth_only),                                                                     # (False / True / 2: include/gsr.h)
                     C.c_void_p(C.addressof(fused_pose_step)) if fused_pose_step is not None else None)       # (a PoseStepArgs)
    if stages & 4:
        st.dirty = not (stages & 8)
    with torch.cuda.device(dev):
        _check(L.gsr_backward(C.byref(a), _stream()))
    if not has_sr:
        out.dL_dscales.zero_()
        out.dL_drotations.zero_()
    return out


def mark_visible(positions, viewmatrix, projmatrix):
    dev = viewmatrix.device
    p = _f32(positions, dev)
    P = int(p.shape[0])
    out = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P:
        _check(lib().gsr_mark_visible(P, _p(p), _p(_f32(viewmatrix, dev)), _p(_f32(projmatrix, dev)), _p(out), _stream()))
    return out


def visible_filter(s: Settings, means3D, scales, rotations, width=None, height=None):
    dev = s.viewmatrix.device
    m, sc, r = _f32(means3D, dev), _f32(scales, dev), _f32(rotations, dev)
    P = int(m.shape[0])
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    if P:
        _check(lib().gsr_visible_filter(P, width or s.image_width, height or s.image_height, _p(m), _p(sc),
                                        s.scale_modifier, _p(r), _p(s.viewmatrix), _p(s.projmatrix), s.tanfovx,
                                        s.tanfovy, int(s.prefiltered), _p(radii), _stream()))
    return radii


def debug_export(st: ForwardState) -> dict:
    """Stage arrays in the reference's layout (tests only)."""
    s = st.settings
    dev = s.viewmatrix.device
    P, R = st.P, max(st.num_rendered, 0)
    W, H = s.image_width, s.image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    o = dict(means2D=z((max(P, 1), 2), torch.float32), depths=z((max(P, 1),), torch.float32),
             conic_opacity=z((max(P, 1), 4), torch.float32), rgb=z((max(P, 1), 3), torch.float32),
             tiles_touched=z((max(P, 1),), torch.int32), point_list=z((max(R, 1),), torch.int32),
             point_list_keys=z((max(R, 1),), torch.int64), ranges=z((T, 2), torch.int32),
             final_T=z((H * W,), torch.float32), n_contrib=z((H * W,), torch.int32))
    d = DebugArrays(*[_p(o[n]) for n, _ in DebugArrays._fields_])
    _check(lib().gsr_debug_export(P, W, H, R, _p(st.geom), _p(st.binning), _p(st.image), C.byref(d), _stream()))
    out = {k: v.cpu().numpy() for k, v in o.items()}
    for k in ("means2D", "depths", "conic_opacity", "rgb", "tiles_touched"):
        out[k] = out[k][:P]
    out["point_list"] = out["point_list"][:R].astype("uint32")
    out["point_list_keys"] = out["point_list_keys"][:R].astype("uint64")
    out["tiles_touched"] = out["tiles_touched"].astype("uint32")
    out["ranges"] = out["ranges"].astype("uint32")
    out["n_contrib"] = out["n_contrib"].astype("uint32")
    return out


def dist2(points, workspace=None):
    """distCUDA2 (reference include/spatial.h:13): mean squared distance to the 3 nearest neighbours, [P]."""
    L = lib()
    p = points if isinstance(points, torch.Tensor) else torch.as_tensor(points)
    dev = p.device if p.is_cuda else torch.device("cuda")
    p = _f32(p, dev)
    P = int(p.shape[0])
    out = torch.empty((P,), dtype=torch.float32, device=dev)
    if P:
        nbytes = int(L.gsr_knn_bytes(P))
        ws = workspace if workspace is not None and workspace.numel() >= nbytes else \
            torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _check(L.gsr_dist2(P, _p(p), _p(out), _p(ws), ws.numel(), _stream()))
    return out


# ---- fused SSIM and Adam (include/gsr.h) ---------------------------------------------------------
class _FusedSSIM(torch.autograd.Function):
    """mean SSIM of img1 vs img2 ([C,H,W] float32 on the GPU) with an 11-tap separable window; gradient w.r.t. img1."""

    @staticmethod
    def forward(ctx, img1, img2, taps):
        L = lib()
        # (inside forward() grad mode is off: a .contiguous() COPY of a sliced / permuted render has requires_grad False,
        # so the need for a gradient is read from the context, not from the tensor)
        need = bool(ctx.needs_input_grad[0])
        img1, img2 = img1.contiguous(), img2.contiguous()
        Cc, H, W = (int(x) for x in img1.shape)
        tp = (C.c_float * 11)(*[float(x) for x in taps])
        partial = torch.empty((int(L.gsr_ssim_partials(Cc, H, W)),), dtype=torch.float32, device=img1.device)
        dmaps = torch.empty((3, Cc, H, W), dtype=torch.float32, device=img1.device) if need else None
        with torch.cuda.device(img1.device):
            _check(L.gsr_ssim_forward(_p(img1), _p(img2), Cc, H, W, tp, _p(partial), _p(dmaps) if need else None, _stream()))
        ctx.save_for_backward(img1, img2, dmaps if need else img1)
        ctx.taps, ctx.need = tp, need
        return partial.sum() / float(Cc * H * W)

    @staticmethod
    def backward(ctx, grad):
        img1, img2, dmaps = ctx.saved_tensors
        if not ctx.need:
            return None, None, None
        Cc, H, W = (int(x) for x in img1.shape)
        g = grad.to(torch.float32).contiguous()
        out = torch.empty_like(img1)
        with torch.cuda.device(img1.device):
            _check(lib().gsr_ssim_backward(_p(img1), _p(img2), _p(dmaps), Cc, H, W, ctx.taps, _p(g), _p(out), _stream()))
        return out, None, None


def ssim_mean(img1, img2, taps):
    """Fused mean-SSIM (reference Utils.cc:77-100) of two [C,H,W] GPU images; differentiable w.r.t. img1 only
    (the mapping and tracking losses compare a render with a fixed frame)."""
    if img2.requires_grad:
        raise GsrError("ssim_mean: only the first image may require a gradient")
    return _FusedSSIM.apply(_f32(img1, img1.device) if img1.dtype != torch.float32 else img1, img2.to(torch.float32), tuple(taps))


LOSS_PARTIALS = 1024  # GSR_LOSS_PARTIALS


class _PixelLoss(torch.autograd.Function):
    """The pixel terms of the tracking (mode 0) / mapping (mode 1) loss as one reduction pass + one gradient pass
    (include/gsr.h: gsr_pixel_loss). Differentiable in `image` and `depth`; `sur` (median depth) has no gradient."""

    @staticmethod
    def forward(ctx, image, depth, sur, sil, frame_rgb, frame_depth, mode, sil_thr, w):
        L = lib()
        dev = image.device
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        image, depth, sur, sil, frame_rgb, frame_depth = c(image), c(depth), c(sur), c(sil), c(frame_rgb), c(frame_depth)
        H, W = int(image.shape[-2]), int(image.shape[-1])
        # the kernels take raw pointers: every plane must live on the image's device and hold H*W (3*H*W) floats
        if not image.is_cuda or image.numel() != 3 * H * W:
            raise GsrError("fused pixel loss: image must be a [3,H,W] tensor on the GPU")
        for name, t, n in (("depth", depth, H * W), ("sur", sur, H * W), ("sil", sil, H * W), ("frame_rgb", frame_rgb, 3 * H * W),
                           ("frame_depth", frame_depth, H * W)):
            if t is not None and (t.device != dev or t.numel() != n):
                raise GsrError(f"fused pixel loss: {name} must hold {n} elements on {dev} (got {tuple(t.shape)} on {t.device})")
        w3 = (C.c_float * 3)(*[float(x) for x in w])
        partial = torch.empty((LOSS_PARTIALS * 5,), dtype=torch.float32, device=dev)
        sums = torch.empty((8,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _check(L.gsr_pixel_loss(_p(image), _p(depth) if depth is not None else None, _p(sur) if sur is not None else None,
                                    _p(sil) if sil is not None else None, _p(frame_rgb), _p(frame_depth), H, W, int(mode), float(sil_thr), w3,
                                    _p(partial), _p(sums), _stream()))
        ctx.saved = (image, depth, sil, frame_rgb, frame_depth, sums)
        ctx.cfg = (H, W, int(mode), float(sil_thr), w3)
        ctx.mark_non_differentiable(sums)
        return sums[5].clone(), sums   # (the loss is not a view of the buffer saved for backward)

    @staticmethod
    def backward(ctx, go, _):
        image, depth, sil, frame_rgb, frame_depth, sums = ctx.saved
        H, W, mode, thr, w3 = ctx.cfg
        g = go.detach().to(torch.float32).contiguous()
        dimage = torch.empty_like(image)
        ddepth = torch.empty_like(depth) if depth is not None and ctx.needs_input_grad[1] else None
        with torch.cuda.device(image.device):
            _check(lib().gsr_pixel_loss_backward(_p(image), _p(depth) if depth is not None else None, _p(sil) if sil is not None else None,
                                                 _p(frame_rgb), _p(frame_depth), H, W, mode, thr, w3, _p(sums), _p(g), _p(dimage),
                                                 _p(ddepth) if ddepth is not None else None, _stream()))
        return dimage, ddepth, None, None, None, None, None, None, None


def tracking_pixel_loss(image, depth, sil, frame_rgb, frame_depth, w_image, w_depth, sil_thr=0.99, depth_is_surface=False):
    """w_image * sum_M |image - rgb| + w_depth * sum_M |depth - frame depth|, M = sil > thr & ~isnan(frame depth)
    (Render.cc:1088-1105). depth_is_surface: `depth` is the median-depth plane (no gradient flows into it)."""
    d, s = (None, depth) if depth_is_surface else (depth, None)
    return _PixelLoss.apply(image, d, s, sil, frame_rgb, frame_depth, 0, sil_thr, (w_image, w_depth, 0.0))[0]


def mapping_pixel_loss(image, depth, sur, sil, frame_rgb, frame_depth, w_l1, w_depth, w_sur, sil_thr=0.99):
    """w_l1 * mean |image - rgb| + w_depth * masked mean |depth - fd| (fd > 0) + w_sur * masked mean |sur - fd| (fd > 0 & sil > thr)
    (Render.cc:436-471). Returns (loss, sums [8]: see include/gsr.h)."""
    return _PixelLoss.apply(image, depth, sur, sil, frame_rgb, frame_depth, 1, sil_thr, (w_l1, w_depth, w_sur))


class _ScaleReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_scales, limit, w_long, w_scalar):
        ls = log_scales.detach().to(torch.float32).contiguous()
        if not ls.is_cuda or ls.dim() != 2 or ls.shape[1] != 3:
            raise GsrError("fused scale regularisers: log_scales must be an [n,3] tensor on the GPU")
        partial = torch.empty((LOSS_PARTIALS * 3,), dtype=torch.float32, device=ls.device)
        out = torch.empty((4,), dtype=torch.float32, device=ls.device)
        with torch.cuda.device(ls.device):
            _check(lib().gsr_scale_reg(_p(ls), ls.shape[0], float(limit), float(w_long), float(w_scalar), _p(partial), _p(out), _stream()))
        ctx.saved = (ls, out)
        ctx.cfg = (float(limit), float(w_long), float(w_scalar))
        ctx.mark_non_differentiable(out)
        return out[3].clone(), out

    @staticmethod
    def backward(ctx, go, _):
        ls, out = ctx.saved
        limit, wl, wsc = ctx.cfg
        g = go.detach().to(torch.float32).contiguous()
        d = torch.empty_like(ls)
        with torch.cuda.device(ls.device):
            _check(lib().gsr_scale_reg_backward(_p(ls), ls.shape[0], limit, wl, wsc, _p(out), _p(g), _p(d), _stream()))
        return d, None, None, None


def scale_regularisers(log_scales, limit, w_long, w_scalar):
    """w_long * reg_long + w_scalar * reg_scalar of Render.cc:449-462 for log_scales [n,3]; returns (value, out [4] =
    {number of oversized components, reg_scalar, sum of max - min over them, value})."""
    return _ScaleReg.apply(log_scales, limit, w_long, w_scalar)


POSE_PARTIALS = 512  # GSR_POSE_PARTIALS


def composite_forward(world, rank, order, gathered, layer4, has_sur):
    """gsr_composite_forward on device tensors: (contrib [4,H,W], sil_total [1,H,W], surf [1,H,W])."""
    H, W = int(layer4.shape[-2]), int(layer4.shape[-1])
    contrib = torch.empty_like(layer4)
    sil = torch.empty((1, H, W), dtype=torch.float32, device=layer4.device)
    surf = torch.empty_like(sil)
    with torch.cuda.device(layer4.device):
        _check(lib().gsr_composite_forward(int(world), int(rank), _p(order), _p(gathered), int(gathered.shape[1]), _p(layer4), H, W, int(bool(has_sur)),
                                           _p(contrib), _p(sil), _p(surf), _stream()))
    return contrib, sil, surf


def composite_backward_local(world, rank, order, gathered, layer4, g4):
    H, W = int(layer4.shape[-2]), int(layer4.shape[-1])
    d_layer = torch.empty_like(layer4)
    c_own = torch.empty((1, H, W), dtype=torch.float32, device=layer4.device)
    with torch.cuda.device(layer4.device):
        _check(lib().gsr_composite_backward_local(int(world), int(rank), _p(order), _p(gathered), int(gathered.shape[1]), _p(layer4),
                                                  _p(g4) if g4 is not None else None, H, W, _p(d_layer), _p(c_own), _stream()))
    return d_layer, c_own


def composite_backward_occlusion(world, rank, order, gathered, c_all, g_sil):
    H, W = int(gathered.shape[-2]), int(gathered.shape[-1])
    dS = torch.empty((1, H, W), dtype=torch.float32, device=gathered.device)
    with torch.cuda.device(gathered.device):
        _check(lib().gsr_composite_backward_occlusion(int(world), int(rank), _p(order), _p(gathered), int(gathered.shape[1]), _p(c_all),
                                                      _p(g_sil) if g_sil is not None else None, H, W, _p(dS), _stream()))
    return dS


def band_composite_forward(world, rank, order, layers_all, own_layer, row_begin, row_end, halo, out_rgbd, out_sil, out_sur):
    """gsr_band_composite_forward on device tensors: layers_all [world,6,H,W] (None with world 1), own_layer [6,H,W]; writes the band's rows of the outputs."""
    H, W = int(own_layer.shape[-2]), int(own_layer.shape[-1])
    with torch.cuda.device(own_layer.device):
        _check(lib().gsr_band_composite_forward(int(world), int(rank), _p(order), _p(layers_all) if layers_all is not None else None, _p(own_layer), H, W,
                                                int(row_begin), int(row_end), int(halo), _p(out_rgbd), _p(out_sil), _p(out_sur), _stream()))


def band_composite_backward(world, rank, order, layers_all, own_layer, g4, row_begin, row_end, d_all, d_own, g_sil=None):
    H, W = int(own_layer.shape[-2]), int(own_layer.shape[-1])
    with torch.cuda.device(own_layer.device):
        _check(lib().gsr_band_composite_backward(int(world), int(rank), _p(order), _p(layers_all) if layers_all is not None else None, _p(own_layer), _p(g4),
                                                 _p(g_sil) if g_sil is not None else None, H, W,
                                                 int(row_begin), int(row_end), _p(d_all) if d_all is not None else None, _p(d_own), _stream()))


def reproj_loss(obs, Xw, inv_sigma2, Tcw, fx, fy, cx, cy, weight, pose_row, loss, inliers=None, refresh=2, grad_scale=1.0):
    """gsr_reproj_loss: adds the reprojection term's twelve pose sums to pose_row [12] and weight * Lrpj to loss [1] (device tensors)."""
    M = int(obs.shape[0])
    with torch.cuda.device(Tcw.device):
        _check(lib().gsr_reproj_loss(_p(obs), _p(Xw), _p(inv_sigma2), M, _p(Tcw), float(fx), float(fy), float(cx), float(cy), float(weight), float(grad_scale),
                                     int(refresh), _p(inliers) if inliers is not None else None, _p(pose_row), _p(loss), _stream()))


def shard_order(kd_nodes, Tcw):
    """gsr_shard_order: the ranks of a k-d partition front to back for the camera of Tcw (device tensors; sharded.KdPartition.nodes)."""
    world = int(kd_nodes.shape[0]) + 1
    order = torch.empty((world,), dtype=torch.int64, device=Tcw.device)
    with torch.cuda.device(Tcw.device):
        _check(lib().gsr_shard_order(world, _p(_f32(kd_nodes, Tcw.device)), _p(_f32(Tcw, Tcw.device)), _p(order), _stream()))
    return order


def map_prepare(xyz, logit, log_scales, unnorm_quat, Tcw, reg=None):
    """gsr_map_prepare: (means_cam, opacities [n], scales, rotations[, reg_out [4]]) of n raw Gaussians under the device pose Tcw;
    reg = (limit, w_long, w_scalar) also evaluates the scale regularisers in the same pass."""
    n = int(xyz.shape[0])
    mc, sc, rot = torch.empty_like(xyz), torch.empty_like(log_scales), torch.empty_like(unnorm_quat)
    op = torch.empty((n,), dtype=torch.float32, device=xyz.device)
    part = torch.empty((3 * ((n + 255) // 256),), dtype=torch.float32, device=xyz.device) if reg else None
    out = torch.empty((4,), dtype=torch.float32, device=xyz.device) if reg else None
    lim, wl, ws = reg if reg else (0.0, 0.0, 0.0)
    with torch.cuda.device(xyz.device):
        _check(lib().gsr_map_prepare(n, _p(xyz), _p(logit), _p(log_scales), _p(unnorm_quat), _p(Tcw), _p(mc), _p(op), _p(sc), _p(rot), float(lim),
                                     float(wl), float(ws), _p(part) if reg else None, _p(out) if reg else None, _stream()))
    return (mc, op, sc, rot, out) if reg else (mc, op, sc, rot)


def map_update(params, moments, grads, acts, Tcw, lrs, steps, reg=None, geom=None, betas=(0.9, 0.999), eps=1e-15):
    """gsr_map_update, in place: params = (xyz, rgb, unnorm_quat, logit, log_scales), moments = ((exp_avg,) * 5, (exp_avg_sq,) * 5),
    grads = (dL_dmeans_cam, dL_dcolors, dL_drotations, dL_dopacities, dL_dscales), acts = (opacities, scales) of gsr_map_prepare;
    reg = (reg_out, limit, w_long, w_scalar)."""
    a = map_update_args(params, moments, grads, acts, Tcw, lrs, steps, reg, geom, betas, eps)
    with torch.cuda.device(params[0].device):
        _check(lib().gsr_map_update(C.byref(a), _stream()))


def map_update_args(params, moments, grads, acts, Tcw, lrs, steps, reg=None, geom=None, betas=(0.9, 0.999), eps=1e-15):
    """The gsr_map_update_args of map_update's arguments (grads may be None: gsr_backward_args.fused_map_update ignores them). The returned
    struct keeps the tensors alive through `_keep`."""
    a = MapUpdateArgs()
    a._keep = (params, moments, grads, acts, Tcw, reg, geom)
    a.n = int(params[0].shape[0])
    a.xyz, a.rgb, a.unnorm_quat, a.logit, a.log_scales = (_p(t) for t in params)
    for g in range(5):
        a.exp_avg[g] = _p(moments[0][g]); a.exp_avg_sq[g] = _p(moments[1][g]); a.lr[g] = float(lrs[g]); a.step[g] = int(steps[g])
    if grads is not None:
        a.dL_dmeans_cam, a.dL_dcolors, a.dL_drotations, a.dL_dopacities, a.dL_dscales = (_p(t) for t in grads)
    a.opacities, a.scales = _p(acts[0]), _p(acts[1])
    a.Tcw = _p(Tcw)
    if reg:
        a.reg_out = _p(reg[0]); a.reg_limit, a.w_long, a.w_scalar = float(reg[1]), float(reg[2]), float(reg[3])
    a.geom = _p(geom) if geom is not None else None
    a.beta1, a.beta2, a.eps = float(betas[0]), float(betas[1]), float(eps)
    return a


def pose_update(quat_trans, moments, best, history_slot, Tcw, partial, loss, lr, step, geom=None, betas=(0.9, 0.999), eps=1e-15):
    """gsr_pose_update, in place (see include/gsr.h)."""
    a = PoseUpdateArgs(_p(quat_trans), _p(moments), _p(best), _p(history_slot), _p(Tcw), _p(partial), _p(loss),
                       _p(geom) if geom is not None else None, float(lr), float(betas[0]), float(betas[1]), float(eps), int(step))
    with torch.cuda.device(quat_trans.device):
        _check(lib().gsr_pose_update(C.byref(a), _stream()))


class _ToCamera(torch.autograd.Function):
    """mc = X R^T + t (Render.cc:750-752) as one elementwise kernel; backward: dL/dX = dmc R and the pose gradient from one
    reduction kernel instead of a 3 x N x 3 GEMM. The pose stays on the device."""

    @staticmethod
    def forward(ctx, Tcw, X):
        Xc = X.contiguous()
        Td = Tcw.detach().to(device=Xc.device, dtype=torch.float32).contiguous()
        mc = torch.empty_like(Xc)
        with torch.cuda.device(Xc.device):
            _check(lib().gsr_to_camera(_p(Xc), int(Xc.shape[0]), _p(Td), _p(mc), _stream()))
        ctx.save_for_backward(Xc, Td)
        return mc

    @staticmethod
    def backward(ctx, dmc):
        X, Td = ctx.saved_tensors
        dT = dX = None
        dmc = dmc.contiguous()
        want_T, want_X = ctx.needs_input_grad
        if not (want_T or want_X):
            return None, None
        part = torch.empty((POSE_PARTIALS, 12), dtype=torch.float32, device=X.device) if want_T else None
        dX = torch.empty_like(X) if want_X else None
        with torch.cuda.device(X.device):
            _check(lib().gsr_pose_grad(_p(X), _p(dmc), int(X.shape[0]), _p(Td), _p(part), _p(dX), _stream()))
        if want_T:
            s = part.sum(0)
            dT = torch.zeros((4, 4), dtype=torch.float32, device=X.device)
            dT[:3, :3] = s[:9].reshape(3, 3)
            dT[:3, 3] = s[9:]
        return dT, dX


def to_camera(Tcw, X):
    """Camera-frame means [n,3] of world-frame means X under the pose Tcw [4,4] (float32, GPU); differentiable in both."""
    return _ToCamera.apply(Tcw, X)


class _Rt2T(torch.autograd.Function):
    """Tcw [4,4] from an un-normalised quaternion (r,x,y,z) [4,1] and a translation [3,1] (include/Utils.h:56-77)."""

    @staticmethod
    def forward(ctx, quat, trans):
        q, t = quat.contiguous(), trans.contiguous()
        T = torch.empty((4, 4), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            _check(lib().gsr_pose_from_quat(_p(q), _p(t), _p(T), _stream()))
        ctx.save_for_backward(q)
        ctx.shapes = (quat.shape, trans.shape)
        return T

    @staticmethod
    def backward(ctx, dT):
        (q,) = ctx.saved_tensors
        dq = torch.empty((4,), dtype=torch.float32, device=q.device)
        dt = torch.empty((3,), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            _check(lib().gsr_pose_from_quat_backward(_p(q), _p(dT.contiguous()), _p(dq), _p(dt), _stream()))
        return dq.reshape(ctx.shapes[0]), dt.reshape(ctx.shapes[1])


def rt2T(quat, trans):
    return _Rt2T.apply(quat, trans)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps) — no weight decay, no amsgrad, as src/Gaussian.cc:144-175 configures the reference's
    optimisers — with the per-tensor update in one HIP kernel. Same param_groups / state layout (step, exp_avg,
    exp_avg_sq), so the reference's optimiser-state surgery on pruning and concatenation works unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        L = lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise GsrError("FusedAdam: parameters must be contiguous float32 GPU tensors")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                if "step" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                st["step"] += 1
                g = p.grad.contiguous()
                with torch.cuda.device(p.device):
                    _check(L.gsr_adam_step(_p(p), _p(g), _p(st["exp_avg"]), _p(st["exp_avg_sq"]), p.numel(), float(group["lr"]), float(b1),
                                           float(b2), float(group["eps"]), int(st["step"]), _stream()))


def acc_view(st: ForwardState) -> torch.Tensor:
    """The packed per-splat backward accumulators of a forward state, as a float32 view [P*16] into its
    geometry blob (what tile-band sharding all-reduces between the blend and per-splat stages)."""
    ptr, n = C.c_void_p(), C.c_size_t()
    _check(lib().gsr_acc_view(_p(st.geom), st.P, C.byref(ptr), C.byref(n)))
    off = ptr.value - st.geom.data_ptr()
    return st.geom[off:off + n.value * 4].view(torch.float32)


def transmittance_view(st: ForwardState) -> torch.Tensor:
    """The final transmittance [H,W] the forward left in its image blob (forward.cu's final_T; include/gsr.h:
    gsr_transmittance_view): 1 - T is the silhouette the plain forward does not render."""
    ptr = C.c_void_p()
    H, W = st.settings.image_height, st.settings.image_width
    _check(lib().gsr_transmittance_view(_p(st.image), W, H, C.byref(ptr)))
    off = ptr.value - st.image.data_ptr()
    return st.image[off:off + H * W * 4].view(torch.float32).view(H, W)
