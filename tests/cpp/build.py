"""Builds tests/cpp/render_api_test against libtorch + the in-tree host layer (test infrastructure)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "render_api_test.bin")


def build(force=False):
    src = os.path.join(HERE, "render_api_test.cpp")
    ext = os.path.join(ROOT, "gsorb-slam_amd", "torch_ext")
    deps = [src, os.path.join(ext, "Rasterizer.cpp"), os.path.join(ext, "Rasterizer.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    csrc = os.path.join(ROOT, "gsorb-slam_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-w", src, os.path.join(ext, "Rasterizer.cpp"), "-o", OUT,
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"] + [f"-I{i}" for i in inc] + [
           f"-L{tlib}", f"-L{csrc}", "-lgsr_hip", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
           "-Wl,--no-as-needed", "-ltorch_hip", "-Wl,--as-needed", f"-Wl,-rpath,{tlib}", f"-Wl,-rpath,{csrc}"]
    subprocess.run(cmd, check=True)
    return OUT


LOOP_OUT = os.path.join(HERE, "slam_loop_main.bin")


def build_loop(force=False):
    """The libtorch tracking / mapping loop driver (gsorb-slam_amd/torch_ext/SlamLoop.{h,cpp}) + its scene-file front end."""
    src = os.path.join(HERE, "slam_loop_main.cpp")
    ext = os.path.join(ROOT, "gsorb-slam_amd", "torch_ext")
    srcs = [src, os.path.join(ext, "SlamLoop.cpp"), os.path.join(ext, "FusedOps.cpp"), os.path.join(ext, "Rasterizer.cpp")]
    deps = srcs + [os.path.join(ext, "SlamLoop.h"), os.path.join(ext, "FusedOps.h"), os.path.join(ext, "Rasterizer.h")]
    if not force and os.path.exists(LOOP_OUT) and all(os.path.getmtime(LOOP_OUT) >= os.path.getmtime(d) for d in deps):
        return LOOP_OUT
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    csrc = os.path.join(ROOT, "gsorb-slam_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-w"] + srcs + ["-o", LOOP_OUT,
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"] + [f"-I{i}" for i in inc] + [
           f"-L{tlib}", f"-L{csrc}", "-lgsr_hip", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
           "-Wl,--no-as-needed", "-ltorch_hip", "-Wl,--as-needed", f"-Wl,-rpath,{tlib}", f"-Wl,-rpath,{csrc}"]
    subprocess.run(cmd, check=True)
    return LOOP_OUT


if __name__ == "__main__":
    print(build("--force" in sys.argv))
    print(build_loop("--force" in sys.argv))
