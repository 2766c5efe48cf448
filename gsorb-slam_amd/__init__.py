"""gsorb-slam_amd — MI355X-native differentiable Gaussian-splat rasterizer.

Drop-in for the one hot path of GSORB-SLAM: `CudaRasterizer::Rasterizer`
(reference Thirdparty/diff_gaussian_rasterization/cuda_rasterizer/rasterizer.h:24-99)
behind the C ABI of include/gsr.h, plus the libtorch / Python operator layers
that mirror include/Rasterizer.cuh and diff_gaussian_rasterization/__init__.py.

The directory name contains a hyphen (it is fixed by the project layout), so
import it with `importlib` — see `load_package()` in tests/conftest.py — under
the module name `gsorb_slam_amd`.

There is NO CPU fallback: every compute entry point goes to the HIP library
and raises if it is missing.
"""
from . import capi, synthetic  # noqa: F401
from .capi import (GsrError, backward, debug_export, dist2, forward, forward_ws, lib, library_path,  # noqa: F401
                   mark_visible, visible_filter)
