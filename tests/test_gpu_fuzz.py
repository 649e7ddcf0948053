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
