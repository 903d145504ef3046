"""GPU parity against the committed fixtures of tests/golden, through the C ABI (no oracle call in these tests)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _problem():
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.golden.make_golden import PANEL, SAMPLE
    panel = make_synthetic_panel(**PANEL)
    return panel, make_synthetic_sample(panel, **SAMPLE)


def test_fullpass_matches_fixture():
    from quilt_amd.native import DevicePanel
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    panel, _ = _problem()
    z = np.load(os.path.join(GOLD, "fullpass_small.npz"))
    dev = DevicePanel(panel)
    cols = z["cols"]
    n_thin = int((cols >= 0).sum())
    for label in (1, 2):
        out = dict(alphaHat_t=np.zeros((panel.K, panel.nGrids), order="F"), c=np.ones(panel.nGrids),
                   dosage=np.zeros(panel.nSNPs), best_haps_stuff_list=[None] * n_thin)
        Rcpp_haploid_dosage_versus_refs(dev, np.asfortranarray(z[f"gl{label}"]), gammaSmall_cols_to_get=cols,
                                        return_dosage=True, get_best_haps_from_thinned_sites=True,
                                        return_gamma_t=False, return_betaHat_t=False, **out)
        assert np.abs(out["dosage"] - z[f"dosage{label}"]).max() <= 2e-4      # fp32-state dosage pass
        np.testing.assert_allclose(out["c"], z[f"c{label}"], rtol=1e-4)
        for j, e in enumerate(out["best_haps_stuff_list"]):                    # fp64-state ranking pass: identical lists
            assert np.array_equal(e["top_matches"], z[f"best_idx{label}_{j}"])
            np.testing.assert_allclose(e["top_matches_values"], z[f"best_val{label}_{j}"], rtol=1e-9)
    dev.close()


def test_gibbs_matches_fixture():
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    panel, sample = _problem()
    z = np.load(os.path.join(GOLD, "gibbs_small.npz"))
    dev = DevicePanel(panel)
    out = forwardBackwardGibbsNIPT_batch(dev, [sample], [z["which"]], [z["H0"]], None, [int(z["first_read"])], None,
                                         seed_reads=[int(z["seed_reads"])], seed_shard=[int(z["seed_shard"])],
                                         gibbs_initialize_iteratively=True)[0]
    assert np.array_equal(out["H"], z["H"])                                     # integer output: bit-exact
    np.testing.assert_allclose(out["hapProbs_t"], z["hapProbs_t"], rtol=0, atol=1e-9)
    dev.close()


def test_nipt_block_gibbs_matches_fixture():
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from tests.golden.make_golden import SAMPLE
    panel, _ = _problem()
    z = np.load(os.path.join(GOLD, "nipt_small.npz"))
    ff = float(z["ff"])
    s3 = make_synthetic_sample(panel, seed=SAMPLE["seed"] + 1, n_reads=60, ff=ff)
    dev = DevicePanel(panel)
    out = forwardBackwardGibbsNIPT_batch(dev, [s3], [z["which"]], [z["H0"]], None, [int(z["first_read"])], None, ff=ff,
                                         seed_reads=[int(z["seed_reads"])], seed_shard=[int(z["seed_shard"])],
                                         gibbs_initialize_iteratively=True)[0]
    assert np.array_equal(out["H"], z["H"]) and np.array_equal(out["H_class"], z["H_class"])
    np.testing.assert_allclose(out["hapProbs_t"], z["hapProbs_t"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["genProbsF_t"], z["genProbsF_t"], rtol=0, atol=1e-9)
    dev.close()


def test_rare_common_gibbs_matches_fixture():
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    from tests.golden.make_golden import RC_SEED, SAMPLE
    panel, _ = _problem()
    z = np.load(os.path.join(GOLD, "rare_common_small.npz"))
    rc = make_rare_common(panel, RC_SEED)
    _, s_all = make_synthetic_sample_rare_common(panel, rc, SAMPLE["seed"] + 2, n_reads=60)
    dev = DevicePanel(panel)
    drc = DeviceRareCommon(dev, rc)
    out = forwardBackwardGibbsNIPT_batch(dev, [s_all], [z["which"]], [z["H0"]], None, [0], None,
                                         seed_reads=[int(z["seed_reads"])], seed_shard=[int(z["seed_shard"])],
                                         disable_read_category_usage=True, rare_common=drc)[0]
    assert np.array_equal(out["H"], z["H"])
    np.testing.assert_allclose(out["hapProbs_t"][:2], z["hapProbs_t"][:2], rtol=0, atol=1e-9)
    drc.close()
    dev.close()
