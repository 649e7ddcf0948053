"""CPU: the C-ABI library loads and exports every symbol include/sd_downscale.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sd_downscale.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for required in ("sd_bcsd_fit", "sd_bcsd_predict", "sd_bcsd_fit_predict_dev", "sd_analog_fit", "sd_analog_predict",
                     "sd_analogreg_predict", "sd_ctx_create", "sd_last_error"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from skdownscale_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsd_downscale.so not built yet (python __graft_entry__.py)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"symbols declared in the header but not exported: {missing}"
    assert lib.sd_version() >= 100
    # the ctypes signature table covers the whole header (sd_last_error is bound separately)
    assert sorted(set(_lib.SIGNATURES) | {"sd_last_error"}) == declared_functions()


def test_missing_library_fails_loudly(monkeypatch):
    from skdownscale_amd import _lib

    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsd_downscale.so")
    with pytest.raises(_lib.EngineError, match="no CPU fallback"):
        _lib.load()


def test_no_gpu_is_an_error_not_a_fallback():
    from skdownscale_amd import _lib
    from skdownscale_amd.engine import Context

    if not os.path.exists(_lib.LIB_PATH) or os.path.exists("/dev/kfd"):
        pytest.skip("needs the built library and no GPU")
    with pytest.raises((_lib.EngineError, ValueError)):
        Context(0)
