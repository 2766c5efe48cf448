"""Builds the C++ test drivers against libtorch + the in-tree host layer torch_ext/libgsr_torch.so (test infrastructure):
render_api_test (the drop-in operator through the reference's C++ API) and slam_loop_main (the SlamLoop driver's scene-file
front end, also what bench.py times as `loop_ms`)."""
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXT = os.path.join(ROOT, "gsorb-slam_amd", "torch_ext")
OUT = os.path.join(HERE, "render_api_test.bin")
LOOP_OUT = os.path.join(HERE, "slam_loop_main.bin")
DROPIN_OUT = os.path.join(HERE, "dropin_hip.bin")


def _ext():
    spec = importlib.util.spec_from_file_location("gsr_build_ext", os.path.join(EXT, "build_ext.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _build(src, out, force, extra=()):
    be = _ext()
    lib = be.build_lib()
    deps = [src, lib] + be.HEADERS
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    tlib, cflags = be._flags()
    cflags = [f for f in cflags if f not in ("-O2", "-fPIC")] + ["-O1"]
    subprocess.run(["g++", *cflags, *extra, src, "-o", out] + be.link_flags(tlib), check=True)
    return out


def build(force=False):
    return _build(os.path.join(HERE, "render_api_test.cpp"), OUT, force)


def build_loop(force=False):
    return _build(os.path.join(HERE, "slam_loop_main.cpp"), LOOP_OUT, force)


def build_dropin(force=False):
    """the caller written against the reference's C++ API (dropin_main.cpp), compiled against THIS repository's host layer; oracle/build_ref.sh compiles
    the same file against the reference's"""
    return _build(os.path.join(HERE, "dropin_main.cpp"), DROPIN_OUT, force, ['-DDROPIN_HEADER="Rasterizer.h"', f"-I{EXT}"])


if __name__ == "__main__":
    print(build("--force" in sys.argv))
    print(build_loop("--force" in sys.argv))
    print(build_dropin("--force" in sys.argv))
