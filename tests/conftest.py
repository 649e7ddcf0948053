import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "scikit-downscale_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

try:  # xarray is not installed here nor on the GPU box: the stand-in lets the xarray branches of core.py execute at all
    import xarray  # noqa: F401
except ImportError:
    sys.path.append(os.path.join(ROOT, "tests", "xarray_stub"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd") and os.path.exists("/dev/dri")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd) in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev_ctx():
    """Context on the development build of the engine (`make dev`: the SD_* environment switches that select alternative
    code paths exist only there; the production library never reads the environment)."""
    from skdownscale_amd import _lib
    from skdownscale_amd.engine import Context

    if not os.path.exists(_lib.DEV_LIB_PATH):
        pytest.skip("development library not built (make -C scikit-downscale_amd dev)")
    return Context(0, lib_path=_lib.DEV_LIB_PATH)
