"""TEST INFRASTRUCTURE ONLY: the REFERENCE's own rasterizer on the GPU (oracle/_ref/libgsr_ref*.so, built by oracle/build_ref.sh from the reference's
sources where they lie: hipify-perl + hipcc, see the script's header). Same interface as oracle.Oracle — forward(**scene) -> oracle.Forward with the reference's
state arrays under the same names, backward(dL_dpix) -> oracle.Backward — so a test can put the CPU restatement, the reference and the HIP library side by side.

    Reference()            -ffp-contract=off: the source's arithmetic as written
    Reference(fma=True)    hipcc's default contraction: what a default nvcc build (--fmad=true) is like — NOT a checker, a census (cf. libgsr_oracle_fma.so)

OPT-IN: used only when a human has set GSR_REFERENCE_BUILD=1 (build and run); by default available() is False and every caller skips.
Needs a GPU. Only tests/ (and, opted in, bench.py's baseline leg and smoke()) may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import oracle as _o

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def enabled() -> bool:
    """OPT-IN: a human sets GSR_REFERENCE_BUILD=1 to build and to use oracle/_ref (oracle/build_ref.sh's header says why this is not the default)."""
    return os.environ.get("GSR_REFERENCE_BUILD", "0") == "1"


def available() -> bool:
    return enabled() and os.path.exists(os.path.join(_HERE, "_ref", "libgsr_ref.so"))


def build(force: bool = False) -> None:
    """(re)build when opted in and /root/reference is present; a no-op otherwise"""
    if enabled():
        subprocess.run(["bash", os.path.join(_HERE, "build_ref.sh")] + (["--force"] if force else []), check=True)


def lib(fma: bool = False):
    if fma not in _LIBS:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libgsr_ref_fma.so" if fma else "libgsr_ref.so"))
        L.gsref_state_new.restype = C.c_void_p
        L.gsref_state_free.argtypes = [C.c_void_p]
        L.gsref_forward.restype = C.c_int
        L.gsref_forward.argtypes = [C.c_void_p, C.POINTER(_o._Scene), C.c_void_p, C.c_void_p, C.c_void_p]
        L.gsref_backward.restype = C.c_int
        L.gsref_backward.argtypes = [C.c_void_p, C.POINTER(_o._Scene)] + [C.c_void_p] * 10
        L.gsref_stage.restype = C.c_void_p
        L.gsref_stage.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.gsref_time.restype = C.c_int
        L.gsref_time.argtypes = [C.c_void_p, C.POINTER(_o._Scene), C.c_void_p, C.c_int, C.c_void_p]
        L.gsref_visible_filter.restype = C.c_int
        L.gsref_visible_filter.argtypes = [C.POINTER(_o._Scene), C.c_int, C.c_int, C.c_void_p]
        L.gsref_dist2.restype = C.c_int
        L.gsref_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.gsref_mark_visible.restype = C.c_int
        L.gsref_mark_visible.argtypes = [C.c_int] + [C.c_void_p] * 4
        _LIBS[fma] = L
    return _LIBS[fma]


class Reference:
    """One forward (+ optional backward) of CudaRasterizer::Rasterizer (rasterizer.h:30-85) on the GPU."""

    def __init__(self, fma: bool = False):
        self.lib = lib(fma)
        self.state = C.c_void_p(self.lib.gsref_state_new())
        self._keep = None

    def __del__(self):
        try:
            self.lib.gsref_state_free(self.state)
        except Exception:
            pass

    _scene = _o.Oracle._scene

    def forward(self, copy_stages: bool = True, **kw) -> _o.Forward:
        s, P, M = self._scene(**kw)
        self._s, self._P, self._M = s, P, M
        W, H = s.W, s.H
        color = np.zeros((3, H, W), np.float32)
        depth = np.zeros((1, H, W), np.float32)
        radii = np.zeros((max(P, 1),), np.int32)
        R = self.lib.gsref_forward(self.state, C.byref(s), _o._ptr(color), _o._ptr(depth), _o._ptr(radii))
        if R < 0:
            raise RuntimeError("the reference's forward failed (see stderr)")
        radii = radii[:P]
        stages = {}
        if copy_stages and P > 0:
            for name, (idx, dt) in _o._STAGES.items():
                n = C.c_size_t(0)
                p = self.lib.gsref_stage(self.state, idx, C.byref(n))
                if n.value == 0 or not p:
                    stages[name] = np.zeros((0,), dt)
                    continue
                buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p)
                stages[name] = np.frombuffer(buf, dtype=dt).copy()
            stages["means2D"] = stages["means2D"].reshape(P, 2)
            stages["conic_opacity"] = stages["conic_opacity"].reshape(P, 4)
            stages["ranges"] = stages["ranges"].reshape(-1, 2)
        return _o.Forward(color, depth, radii, int(R), stages)

    def backward(self, dL_dpix) -> _o.Backward:
        P, M = self._P, self._M
        g = _o._f32(dL_dpix)
        z = lambda *shape: np.zeros(shape, np.float32)
        out = _o.Backward(z(P, 3), z(P, 2, 2), z(P, 1), z(P, 3), z(P, 3), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4))
        rc = self.lib.gsref_backward(self.state, C.byref(self._s), _o._ptr(g), _o._ptr(out.dL_dmeans2D), _o._ptr(out.dL_dconic), _o._ptr(out.dL_dopacity),
                                     _o._ptr(out.dL_dcolors), _o._ptr(out.dL_dmeans3D), _o._ptr(out.dL_dcov3D), _o._ptr(out.dL_dsh), _o._ptr(out.dL_dscales),
                                     _o._ptr(out.dL_drotations))
        if rc != 0:
            raise RuntimeError("the reference's backward failed (see stderr)")
        return out


def time_scene(scene, iters: int = 5, fma: bool = True) -> dict:
    """ms per forward / backward of the reference's kernels on this GPU, inputs resident (bench.py's baseline leg; fma=True: hipcc's default contraction, the
    build a user of the reference would get)"""
    r = Reference(fma)
    s, P, M = r._scene(means3D=scene.means3D, opacities=scene.opacities, cam=scene.cam, colors=scene.colors, shs=scene.shs, scales=scene.scales, rotations=scene.rotations)
    g = _o._f32(scene.dL_dpix)
    ms = np.zeros(2, np.float32)
    R = r.lib.gsref_time(r.state, C.byref(s), _o._ptr(g), int(iters), _o._ptr(ms))
    if R < 0:
        raise RuntimeError("the reference's timing pass failed (see stderr)")
    return {"forward_ms": float(ms[0]), "backward_ms": float(ms[1]), "num_rendered": int(R), "iters": int(iters)}


def forward_scene(scene, fma: bool = False, copy_stages: bool = True):
    r = Reference(fma)
    f = r.forward(copy_stages=copy_stages, means3D=scene.means3D, opacities=scene.opacities, cam=scene.cam, colors=scene.colors, shs=scene.shs,
                  scales=scene.scales, rotations=scene.rotations)
    return r, f


def mark_visible(means3D, cam) -> np.ndarray:
    m = _o._f32(means3D)
    out = np.zeros(m.shape[0], np.uint8)
    v, p = _o._f32(cam.viewmatrix), _o._f32(cam.projmatrix)
    if lib().gsref_mark_visible(m.shape[0], _o._ptr(m), _o._ptr(v), _o._ptr(p), _o._ptr(out)) != 0:
        raise RuntimeError("the reference's markVisible failed")
    return out.astype(bool)


def dist2(points, fma: bool = False) -> np.ndarray:
    """SimpleKNN::knn (src/simple_knn.cu:185-220; distCUDA2): mean squared distance of every point to its three nearest neighbours"""
    p = _o._f32(points)
    out = np.zeros(p.shape[0], np.float32)
    if lib(fma).gsref_dist2(p.shape[0], _o._ptr(p), _o._ptr(out)) != 0:
        raise RuntimeError("the reference's k-NN failed")
    return out


def filter_radii(means3D, scales, rotations, cam, width=None, height=None) -> np.ndarray:
    """Rasterizer::visible_filter (rasterizer_impl.cu:348-403; Render.cc:784-831 calls it on an enlarged image): the radii alone"""
    r = Reference()
    z = np.zeros((len(means3D), 1), np.float32)
    s, P, M = r._scene(means3D=means3D, opacities=z, cam=cam, colors=np.zeros((len(means3D), 3), np.float32), scales=scales, rotations=rotations)
    out = np.zeros(max(P, 1), np.int32)
    if r.lib.gsref_visible_filter(C.byref(s), width or cam.width, height or cam.height, _o._ptr(out)) != 0:
        raise RuntimeError("the reference's visible_filter failed")
    return out[:P]
