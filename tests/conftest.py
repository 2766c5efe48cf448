"""pytest configuration: registers the `gpu` marker and loads the hyphen-named package."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_package():
    """Import /gsorb-slam_amd (hyphen in the directory name) as module `gsorb_slam_amd`."""
    if "gsorb_slam_amd" in sys.modules:
        return sys.modules["gsorb_slam_amd"]
    pkg_dir = os.path.join(ROOT, "gsorb-slam_amd")
    spec = importlib.util.spec_from_file_location("gsorb_slam_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gsorb_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gsr():
    return load_package()


@pytest.fixture(scope="session")
def syn(gsr):
    return gsr.synthetic
