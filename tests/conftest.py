import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- both trace kernels in ONE pytest run --------------------------------------------------------
# Small scenes (<= 8 objects, default flags) have two kernels: the general trace_static_kernel and the same
# kernel compiled for the uploaded scene (pt_scene_specialise) -- the one bench.py times.  Every image / segment
# test of the modules below therefore runs twice: "general", and "scene_kernel" with PT_SPECIALISE=1 in the
# environment, which makes pt_scene_upload specialise every scene it applies to (include/pt.h).  Tests that
# never reach a by-value trace kernel (device-function hooks, primary hits, the quantiser, indexed scenes) run once.
BOTH_KERNELS_MODULES = {"test_gpu_parity", "test_gpu_launch_shapes", "test_gpu_pipeline", "test_gpu_fullsize",
                        "test_gpu_multi", "test_zz_no_alloc"}
ONE_KERNEL_NAMES = ("philox", "primary_hits", "dev_", "rgb8", "error_behaviour", "spatial_index", "indexed_kernel",
                    "config4", "config5", "refuses", "needs_a", "limit", "exactly_what_the_bench_runs")


def pytest_generate_tests(metafunc):
    if "kernel_mode" not in metafunc.fixturenames:
        return
    mod = metafunc.module.__name__.split(".")[-1]
    name = metafunc.function.__name__
    both = mod in BOTH_KERNELS_MODULES and not any(k in name for k in ONE_KERNEL_NAMES)
    metafunc.parametrize("kernel_mode", ["general", "scene_kernel"] if both else ["general"], indirect=True)


@pytest.fixture(autouse=True)
def kernel_mode(request, monkeypatch):
    mode = getattr(request, "param", "general")
    if mode == "scene_kernel":
        monkeypatch.setenv("PT_SPECIALISE", "1")
    else:
        monkeypatch.delenv("PT_SPECIALISE", raising=False)
    return mode


def _have_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
