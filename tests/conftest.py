import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def small_panel():
    from quilt_amd.synth import make_synthetic_panel
    # the reference's own unit-test scale (test-drivers.R:324-334): K = 1000, 500 SNPs, two stress grids
    return make_synthetic_panel(K=1000, nSNPs=500, seed=4916, ref_error=0.01, nGen=10, expRate=100,
                                region_bp=5000)


@pytest.fixture(scope="session")
def ragged_panel():
    from quilt_amd.synth import make_synthetic_panel
    # K not a multiple of 16, nSNPs not a multiple of 32, a single special in one grid is possible
    return make_synthetic_panel(K=1237, nSNPs=1003, seed=77, ref_error=1e-3, nGen=100, expRate=1.0,
                                region_bp=60000, nMaxDH=40, stress_grids=(1, 5, 30))


@pytest.fixture(scope="session")
def medium_panel():
    from quilt_amd.synth import make_synthetic_panel
    return make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
