"""quilt_amd/driver.py against a second, literal statement of the R loop nest (tests/r_driver_twin.py:
get_and_impute_one_sample one sample, one Gibbs sample, one seek iteration at a time, restated from the R text and sharing
no code with the driver).  Both run on the CPU oracle, so what is compared is the driver's own logic: starting labels and
hand-over, which rounds accumulate, the selection of the next small panel (complete lists in the twin, truncated lists /
exhausted branch in the driver), read confidence, consensus labels, the phasing iterations, recast_haps, the final division
-- and that lock-step batching and the pipelining of batches change nothing."""
import numpy as np
import pytest

from tests.oracle_backend import OracleBackend
from tests.r_driver_twin import get_and_impute_one_sample


@pytest.fixture(scope="module")
def twin_panel():
    from quilt_amd.synth import make_synthetic_panel
    return make_synthetic_panel(K=400, nSNPs=3200, seed=77, ref_error=1e-3)


def _compare(res, tw):
    assert res.nDosage == tw["nDosage"]
    assert np.array_equal(res.read_labels, tw["read_labels"]), "consensus read labels"
    np.testing.assert_allclose(res.dosage, tw["dosage"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(res.gp_t, tw["gp_t"], rtol=0, atol=1e-13)
    # recast_haps leaves a site's haploid dosages alone unless the argmax genotype disagrees (then exact 0 / 1 values)
    np.testing.assert_allclose(res.phasing_haps, tw["phasing_haps"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kw", [
    dict(nGibbsSamples=7, n_seek_its=3, Ksubset=64, Knew=64),                              # the defaults' shape
    dict(nGibbsSamples=3, n_seek_its=3, Ksubset=64, Knew=24),                              # part of the small panel is kept
    dict(nGibbsSamples=4, n_seek_its=2, n_burn_in_seek_its=0, Ksubset=48, Knew=48),        # every seek iteration accumulates
    dict(nGibbsSamples=2, n_seek_its=3, Ksubset=200, Knew=200, K_top_matches=1, heuristic_match_thin=0.03),   # ranks run out
], ids=["defaults", "Knew<Ksubset", "no-burn-in", "exhausted-ranks"])
def test_driver_equals_the_literal_r_loop(twin_panel, kw):
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests import r_driver_twin
    r_driver_twin.N_EXHAUSTED[0] = 0
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=300 + i, n_reads=260) for i in range(3)]
    common = dict(small_ref_panel_gibbs_iterations=6, small_ref_panel_block_gibbs_iterations=(1, 3), seed=11)
    drv = Driver(panel, OracleBackend(panel), DriverParams(**common, **kw))
    got = drv.run(samples, sample_offset=5)
    for i, s in enumerate(samples):
        tw = get_and_impute_one_sample(panel, s, 5 + i, **common, **kw)
        _compare(got[i], tw)
    # the last case is there for the selection's exhausted branch (every entry of every list, then a draw from the rest of
    # the panel: functions.R:2276-2302)
    if kw.get("K_top_matches") == 1:
        assert r_driver_twin.N_EXHAUSTED[0] >= 18


def test_streamed_batches_equal_the_literal_r_loop(twin_panel):
    """Batches pipelined through run_stream (phasing rounds of one batch fused with the main rounds of the next)."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=400 + i, n_reads=200) for i in range(5)]
    kw = dict(nGibbsSamples=3, n_seek_its=2, Ksubset=64, Knew=40, small_ref_panel_gibbs_iterations=5,
              small_ref_panel_block_gibbs_iterations=(2,), seed=3)
    drv = Driver(panel, OracleBackend(panel), DriverParams(**kw))
    out = list(drv.run_stream([(samples[:2], 0), (samples[2:4], 2), (samples[4:], 4)]))
    flat = [r for batch in out for r in batch]
    for i, s in enumerate(samples):
        _compare(flat[i], get_and_impute_one_sample(panel, s, i, **kw))
