"""GPU: randomised parity sweep (random sizes, group layouts, tie-heavy dyadic data, kinds) of the BCSD, analog and
quantile-mapping paths (incl. detrended BCSD, thresholded regressions, every extrapolate mode) against the oracles; the generator lives in tools/dev/fuzz_gpu.py."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dev"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations(seed):
    import fuzz_gpu

    fuzz_gpu.main(45, seed)


def test_random_configurations_of_the_round3_paths():
    """tools/dev/fuzz_r3.py: tile-shaped analog fit stage vs the two-transpose path (bit for bit) and the oracle, detrended
    mapping of 2 113 .. 19 456 sample segments, TrendAwareQuantileMappingRegressor, float32 transport."""
    import warnings

    import fuzz_r3

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # (the oracle's squared distances overflow for the 1e154-scaled cells)
        assert fuzz_r3.main(120.0, 7000, None, 80) == 0


def test_random_grids_through_the_pointwise_downscaler():
    """tools/dev/fuzz_pointwise.py: PointWiseDownscaler around BcsdTemperature / BcsdPrecipitation / PureAnalog / AnalogRegression
    / QuantileMappingReressor /
    EquidistantCdfMatcher on random grids (1 or 2 spatial dims, with / without a feature dim, time leading or not, float32, masked cells, random spatial
    blocks) against the oracles' per-cell loops (core.py:69-143).  5 444 cases (seeds 9000..14443) passed at the end of round 3."""
    import fuzz_pointwise

    assert fuzz_pointwise.main(120.0, 9000, 200) == 0


@pytest.mark.parametrize("seed", [41, 42])
def test_random_configurations_of_the_fused_bcsd_kernels(seed):
    """tools/dev/fuzz_fx.py: bcsd_fx_kernel / bcsd_fxp_kernel against the NumPy oracle -- every sort width, equal / longer / shorter
    predict series, partly filled lanes, exact ties, near-tied observations, zero-inflated series, masked and non-finite cells, one
    call and via a fitted state, and the state again with random tails (`extrapolate`, `n_endpoints`: sd_bcsd_state_set_tails)."""
    import fuzz_fx

    fuzz_fx.main(250, seed)


def test_random_configurations_of_the_fused_analog_call():
    """tools/dev/fuzz_analog_fused.py: sd_analog_fit_predict* (analog_f1_fused_kernel, its hand-back of tied cells, its internal
    fall-back to the two calls) against analog_fit -> analog_predict, bit for bit; 6 140 cases passed at the end of round 4
    (profiles/r04/fuzz_analog_fused_summary.txt)."""
    import fuzz_analog_fused

    fuzz_analog_fused.main(300, 5)


def test_random_configurations_of_the_candidate_list_search():
    """tools/dev/fuzz_topk.py: analog_slab_topk_kernel (F = 2 .. 6, k <= 30: matrix-core and v_readlane pre-filters, candidate lists
    pruned by the register network, batches handed back on ties) against the heap kernel of the same slab scan, bit for bit, and the
    oracle's brute force; 2 150 cases (seeds 1 .. 5) passed when the kernel was written (profiles/r06/fuzz_summary.txt)."""
    import fuzz_topk

    fuzz_topk.main(120, 77)
