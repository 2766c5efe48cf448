"""CPU: the C-ABI shared library loads and exports every symbol include/gsr.h declares
(no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gsr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", src)) - {"gsr_alloc_fn"})


def test_header_declares_the_reference_entry_points():
    d = _declared()
    for name in ("gsr_forward", "gsr_backward", "gsr_mark_visible", "gsr_visible_filter"):
        assert name in d  # rasterizer.h:24-99 forward / backward / markVisible / visible_filter


def test_library_exports_every_declared_symbol(gsr):
    path = gsr.library_path()
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/gsr.h but not exported"
    assert set(gsr.capi.EXPORTS) == set(_declared())
    assert gsr.lib().gsr_abi_version() == 10


def test_workspace_sizes_are_sane(gsr):
    L = gsr.lib()
    assert L.gsr_geom_bytes(0) > 0 and L.gsr_geom_bytes(1000) >= 1000 * 112
    assert L.gsr_geom_bytes(1_000_000) < 1_000_000 * 248 + 200_000       # 48 B record + 8 B reach entry + 8 x 16 B bucketed bin records + 64 B backward accumulators
    assert L.gsr_binning_bytes(1_000_000) < 1_000_000 * 44 + 4096        # 12 B/instance + 32 B of quad-hit log (reference: 24 + sort temp)
    assert L.gsr_image_bytes(1200, 680) >= 1200 * 680 * 8
    assert L.gsr_error_string(-1).decode() == "invalid argument"


def test_ssim_partials_follow_the_streaming_kernels_grid(gsr):
    """gsr_ssim_partials = the waves of an SSIM launch (csrc/gsr_train.h: ssim_grid): strips of 54 output columns, segments of 11 m - 10 rows,
    about one wave per SIMD (1024) where the image is large enough, segments of at least 16 rows where it is not. Host arithmetic: no GPU."""
    L = gsr.lib()
    for C_, H, W in [(3, 680, 1200), (3, 480, 640), (1, 17, 23), (3, 1, 1), (3, 2160, 3840), (4, 64, 5000)]:
        nsx = -(-W // 54)
        want = max(1, min(1024 // (C_ * nsx), -(-H // 16)))
        rows = -(-H // want)
        rows = (rows + 20) // 11 * 11 - 10
        assert (rows + 10) % 11 == 0 and rows >= 1
        assert int(L.gsr_ssim_partials(C_, H, W)) == C_ * nsx * (-(-H // rows)), (C_, H, W)
    assert int(L.gsr_ssim_partials(3, 680, 1200)) == 897
    assert int(L.gsr_ssim_partials(0, 5, 5)) == 0


def test_struct_layout_matches_header(gsr):
    # field order of the ctypes mirrors must follow include/gsr.h
    src = open(os.path.join(ROOT, "include", "gsr.h")).read()
    def fields(struct):
        body = src[src.index("typedef struct " + struct):]
        body = body[body.index("{") + 1:body.index("} " + struct)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())[0])
        return names
    assert fields("gsr_forward_args") == [n for n, _ in gsr.capi.ForwardArgs._fields_]
    assert fields("gsr_backward_args") == [n for n, _ in gsr.capi.BackwardArgs._fields_]
    assert fields("gsr_debug_arrays") == [n for n, _ in gsr.capi.DebugArrays._fields_]
    # (ABI 9 appended one field to each of these)
    assert fields("gsr_pose_update_args") == [n for n, _ in gsr.capi.PoseUpdateArgs._fields_]
    assert fields("gsr_pose_step_args") == [n for n, _ in gsr.capi.PoseStepArgs._fields_]


def test_no_cpu_fallback_when_library_is_missing(gsr, monkeypatch):
    monkeypatch.setenv("GSR_LIB_OVERRIDE", "/nonexistent/libgsr_hip.so")
    monkeypatch.setattr(gsr.capi, "_LIB", None)
    with pytest.raises(ImportError):
        gsr.capi.lib()
    monkeypatch.delenv("GSR_LIB_OVERRIDE")
    monkeypatch.setattr(gsr.capi, "_LIB", None)
    gsr.capi.lib()


def test_argument_validation_rejects_inconsistent_sh_degree_before_any_gpu_work(gsr):
    """gsr_forward checks 0 <= D <= 3 and (D+1)^2 <= M (csrc/gsr_api.hip:check_forward); gsr_backward must repeat it —
    K_splat_bwd writes dL_dsh[0..(D+1)^2) per splat. Validation precedes every HIP call, so this runs without a GPU
    (the non-NULL pointers below are never dereferenced)."""
    import ctypes as C
    capi = gsr.capi
    L = capi.lib()
    dummy = C.c_void_p(0x1000)
    def bwd(D, M):
        a = capi.BackwardArgs()
        a.P, a.D, a.M, a.R, a.width, a.height = 10, D, M, 0, 64, 48
        for n in ("background", "means3D", "shs", "scales", "rotations", "viewmatrix", "projmatrix", "cam_pos", "geom_buffer",
                  "binning_buffer", "image_buffer", "dL_dpix", "dL_dsh"):
            setattr(a, n, dummy)
        L.gsr_backward.restype = C.c_int
        return L.gsr_backward(C.byref(a), None)
    EINVAL = -1
    assert L.gsr_error_string(EINVAL).decode() == "invalid argument"
    assert bwd(3, 9) == EINVAL        # degree 3 needs 16 coefficients
    assert bwd(4, 25) == EINVAL and bwd(-1, 16) == EINVAL
    assert bwd(2, 0) == EINVAL
    def fwd(D, M):
        a = capi.ForwardArgs()
        a.P, a.D, a.M, a.width, a.height = 10, D, M, 64, 48
        for n in ("background", "means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "cam_pos",
                  "out_color", "out_depth"):
            setattr(a, n, dummy)
        L.gsr_forward_ws.restype = C.c_int
        return L.gsr_forward_ws(C.byref(a), dummy, dummy, C.c_size_t(1 << 20), dummy, None)
    assert fwd(3, 9) == EINVAL and fwd(1, 3) == EINVAL


def test_reference_build_loads_and_exports_its_entry_points():
    """oracle/_ref (test infrastructure: the reference's own rasterizer for this GPU, oracle/build_ref.sh) — no compute here; the -m gpu tests call it"""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import ref
    if not ref.available():
        pytest.skip("opt-in: oracle/_ref exists only after GSR_REFERENCE_BUILD=1 python -c 'import __graft_entry__ as g; g.build()'")
    for fma in (False, True):
        lib = ref.lib(fma)
        for name in ("gsref_state_new", "gsref_state_free", "gsref_forward", "gsref_backward", "gsref_stage", "gsref_time", "gsref_mark_visible", "gsref_visible_filter", "gsref_dist2"):
            assert hasattr(lib, name), name
    # the recipe leaves no text of the reference behind
    assert sorted(os.listdir(os.path.join(ROOT, "oracle", "_ref"))) == ["dropin_ref.bin", "gsr_ref_C.so", "libgsr_ref.so", "libgsr_ref_fma.so"]
