"""Builds the libtorch host layer (Rasterizer.cpp + ext.cpp) into
gsorb-slam_amd/diff_gaussian_rasterization/_C.so, in-tree, linked against csrc/libgsr_hip.so.

Host-only C++ (g++): no device code lives here — the kernels are in csrc/ behind the C ABI.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT_DIR = os.path.join(PKG, "diff_gaussian_rasterization")
OUT = os.path.join(OUT_DIR, "_C.so")
SRCS = [os.path.join(HERE, f) for f in ("Rasterizer.cpp", "ext.cpp")]
DEPS = SRCS + [os.path.join(HERE, "Rasterizer.h"), os.path.join(PKG, "..", "include", "gsr.h")]


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    objs = []
    for src in SRCS:
        obj = os.path.join(HERE, os.path.basename(src).replace(".cpp", ".o"))
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-c", src, "-o", obj, "-DTORCH_EXTENSION_NAME=_C",
               "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-D__HIP_PLATFORM_AMD__=1",
               "-DUSE_ROCM=1", "-w"] + [f"-I{i}" for i in inc]
        subprocess.run(cmd, check=True)
        objs.append(obj)
    csrc = os.path.join(PKG, "csrc")
    cmd = ["g++", "-shared", "-o", OUT] + objs + [
        f"-L{tlib}", f"-L{csrc}", "-lgsr_hip", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
        "-ltorch_python", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN/../csrc"]
    subprocess.run(cmd, check=True)
    for o in objs:
        os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
