"""Builds the libtorch host layer, in-tree:

  torch_ext/libgsr_torch.so            Rasterizer.cpp + FusedOps.cpp + SlamLoop.cpp + DirectLoop.cpp — what a GSORB-SLAM checkout links instead of its
                                       diff_gaussian_rasterization target: the drop-in operator, the fused loop
                                       operations and the tracking / mapping / map-growth loops; linked against csrc/libgsr_hip.so
  diff_gaussian_rasterization/_C.so    ext.cpp (pybind11) on top of libgsr_torch.so: the Python operator's host layer

Host-only C++ (g++): no device code lives here — the kernels are in csrc/ behind the C ABI.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libgsr_torch.so")
OUT_DIR = os.path.join(PKG, "diff_gaussian_rasterization")
OUT = os.path.join(OUT_DIR, "_C.so")
LIB_SRCS = [os.path.join(HERE, f) for f in ("Rasterizer.cpp", "FusedOps.cpp", "SlamLoop.cpp", "DirectLoop.cpp")]
HEADERS = [os.path.join(HERE, f) for f in ("Rasterizer.h", "FusedOps.h", "SlamLoop.h")] + [os.path.join(PKG, "..", "include", "gsr.h")]
EXT_SRC = os.path.join(HERE, "ext.cpp")


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def _flags():
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = ["-std=c++17", "-O2", "-fPIC", f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-w"] + [f"-I{i}" for i in inc]
    return tlib, cflags


def link_flags(tlib):
    """What a C++ program needs to link against the host layer (used by tests/cpp/build.py too)."""
    csrc = os.path.join(PKG, "csrc")
    return [f"-L{HERE}", "-lgsr_torch", f"-L{tlib}", f"-L{csrc}", "-lgsr_hip", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
            "-Wl,--no-as-needed", "-ltorch_hip", "-Wl,--as-needed", f"-Wl,-rpath,{tlib}", f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{HERE}"]


def build_lib(force: bool = False) -> str:
    if not force and not _stale(LIB, LIB_SRCS + HEADERS):
        return LIB
    tlib, cflags = _flags()
    objs = []
    for src in LIB_SRCS:
        obj = os.path.join(HERE, os.path.basename(src).replace(".cpp", ".o"))
        subprocess.run(["g++", *cflags, "-c", src, "-o", obj], check=True)
        objs.append(obj)
    csrc = os.path.join(PKG, "csrc")
    subprocess.run(["g++", "-shared", "-o", LIB] + objs + [
        f"-L{tlib}", f"-L{csrc}", "-lgsr_hip", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ldl",
        f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN/../csrc"], check=True)
    for o in objs:
        os.remove(o)
    return LIB


def build(force: bool = False) -> str:
    build_lib(force)
    if not force and not _stale(OUT, [EXT_SRC, LIB] + HEADERS):
        return OUT
    tlib, cflags = _flags()
    obj = os.path.join(HERE, "ext.o")
    subprocess.run(["g++", *cflags, "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H", "-c", EXT_SRC, "-o", obj], check=True)
    csrc = os.path.join(PKG, "csrc")
    subprocess.run(["g++", "-shared", "-o", OUT, obj, f"-L{HERE}", "-lgsr_torch", f"-L{tlib}", f"-L{csrc}", "-lgsr_hip", "-lc10", "-lc10_hip",
                    "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN/../csrc",
                    "-Wl,-rpath,$ORIGIN/../torch_ext"], check=True)
    os.remove(obj)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
