"""TEST INFRASTRUCTURE ONLY -- import shim for the *reference* scikit-downscale.

Only usable in the build container, where ``/root/reference`` exists.  It is used by
``tests/golden/make_golden.py`` to generate golden vectors and by the (CPU-only,
auto-skipped when the reference is absent) cross-check tests.  Nothing under
``scikit-downscale_amd/`` may import this module.

``import skdownscale.pointwise_models`` fails here because ``core.py``/``zscore.py``
import xarray (not installed).  We register empty stub packages whose ``__path__``
points into the reference tree so the numeric modules (bcsd, quantile, gard, groupers,
base, utils, trend) import and run unmodified.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("SKDOWNSCALE_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "skdownscale", "pointwise_models"))


def load():
    """Return a namespace with the reference's hot-path classes."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    if "skdownscale" not in sys.modules or not getattr(sys.modules["skdownscale"], "_sd_shim", False):
        pkg = types.ModuleType("skdownscale")
        pkg.__path__ = [os.path.join(REF_ROOT, "skdownscale")]
        pkg._sd_shim = True
        sub = types.ModuleType("skdownscale.pointwise_models")
        sub.__path__ = [os.path.join(REF_ROOT, "skdownscale", "pointwise_models")]
        sys.modules["skdownscale"] = pkg
        sys.modules["skdownscale.pointwise_models"] = sub
    ns = types.SimpleNamespace()
    for name in ("utils", "trend", "quantile", "groupers", "base", "bcsd", "gard"):
        setattr(ns, name, importlib.import_module(f"skdownscale.pointwise_models.{name}"))
    ns.BcsdTemperature = ns.bcsd.BcsdTemperature
    ns.BcsdPrecipitation = ns.bcsd.BcsdPrecipitation
    ns.PureAnalog = ns.gard.PureAnalog
    ns.AnalogRegression = ns.gard.AnalogRegression
    ns.QuantileMapper = ns.quantile.QuantileMapper
    ns.CunnaneTransformer = ns.quantile.CunnaneTransformer
    return ns
