"""skdownscale_amd -- MI355X-native engine for scikit-downscale's per-grid-cell hot path.

Public names mirror ``skdownscale.pointwise_models`` for the hot path only
(``skdownscale/pointwise_models/__init__.py:17-36`` of the reference).
"""
from .bcsd import BcsdGridModel, BcsdPrecipitation, BcsdTemperature
from .core import GridArray, GridDataset, PointWiseDownscaler
from .gard import AnalogGridModel, AnalogRegression, PureAnalog, PureRegression, RegressionGridModel
from .groupers import DAY_GROUPER, MONTH_GROUPER, PaddedDOYGrouper
from .quantile import (CunnaneGridModel, CunnaneTransformer, EquidistantCdfMatcher, QmGridModel, QuantileMapper,
                       QuantileMapperGridModel, QuantileMappingReressor, TrendAwareQuantileMappingRegressor)
from .trend import LinearTrendTransformer

__all__ = [
    "AnalogRegression",
    "BcsdPrecipitation",
    "BcsdTemperature",
    "PointWiseDownscaler",
    "PureAnalog",
    "MONTH_GROUPER",
    "DAY_GROUPER",
    "PaddedDOYGrouper",
    "GridArray",
    "GridDataset",
    "BcsdGridModel",
    "AnalogGridModel",
    "QuantileMappingReressor", "TrendAwareQuantileMappingRegressor",
    "QuantileMapper",
    "EquidistantCdfMatcher",
    "QmGridModel",
    "CunnaneTransformer",
    "CunnaneGridModel",
    "QuantileMapperGridModel",
    "PureRegression",
    "LinearTrendTransformer",
    "RegressionGridModel",
]
__version__ = "0.1.0"
