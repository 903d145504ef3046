"""The HIP path (through the C ABI) against the R-twin fixtures (tests/golden/make_golden_rtwin.py): the same checks as
tests/test_rtwin_cpu.py makes of the C oracle, so that the product is pinned to the second, independent restatement too."""
import os

import numpy as np
import pytest

from tests.util import panel_from_rhb, sample_from_arrays

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fullpass_hip_matches_r_twin():
    from quilt_amd.native import DevicePanel
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    z = np.load(os.path.join(GOLD, "rtwin_fullpass.npz"))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], z["nMaxDH"], z["ref_error"])
    gl, cols = np.asfortranarray(z["gl"]), z["cols"]
    n_thin = int((cols >= 0).sum())
    for symbols in (False, True):
        dev = DevicePanel(panel, use_eMatDH_special_symbols=symbols)
        for bits in (32, 64):
            dev.set_dosage_precision(bits)
            dosage, gamma = np.zeros(panel.nSNPs), np.zeros((panel.K, panel.nGrids), order="F")
            best = [None] * n_thin
            Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, dosage=dosage, gamma_t=gamma,
                                            best_haps_stuff_list=best, return_betaHat_t=False, return_gamma_t=True,
                                            get_best_haps_from_thinned_sites=True)
            assert np.abs(dosage - z["dosage"]).max() < (2e-6 if bits == 32 else 1e-12)
            np.testing.assert_allclose(gamma, z["gamma_t"], rtol=2e-3 if bits == 32 else 1e-9, atol=2e-6 if bits == 32 else 1e-300)
            idx = np.concatenate([b["top_matches"] for b in best])
            ptr = np.cumsum([0] + [len(b["top_matches"]) for b in best])
            assert np.array_equal(idx, z["best_idx"]) and np.array_equal(ptr, z["best_ptr"])
        dev.close()


@pytest.mark.parametrize("name", ["rtwin_gibbs_labels.npz", "rtwin_gibbs_init.npz"])
def test_gibbs_hip_matches_r_twin(name):
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    z = np.load(os.path.join(GOLD, name))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], 255, z["ref_error"])
    s = sample_from_arrays(z["read_ptr"], z["u"], z["bq"], z["wif"])
    dev = DevicePanel(panel)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]), None,
                                        perform_block_gibbs=False, gibbs_initialize_iteratively=bool(z["init_iter"]),
                                        return_state=True)
    assert np.array_equal(got["H"], z["H"]) and np.array_equal(got["H_class"], z["H_class"])
    np.testing.assert_allclose(got["alphaHat_t1"], z["alphaHat_t1"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(got["betaHat_t2"], z["betaHat_t2"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(got["c1"], z["c1"], rtol=1e-8)
    np.testing.assert_allclose(got["hapProbs_t"][:2], z["hapProbs_t"][:2], rtol=1e-8, atol=1e-14)
    dev.close()


@pytest.mark.parametrize("name", ["rtwin_shard_0.npz", "rtwin_shard_1.npz"])
def test_shard_passes_hip_match_r_twin(name):
    """a15 on the device against the R-twin side: labels and classes identical, state to 1e-8."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    z = np.load(os.path.join(GOLD, name))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], 255, z["ref_error"])
    s = sample_from_arrays(z["read_ptr"], z["u"], z["bq"], z["wif"])
    dev = DevicePanel(panel)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]), z["runif_shard"],
                                        return_state=True)
    assert np.array_equal(got["H"], z["H"]) and np.array_equal(got["H_class"], z["H_class"])
    np.testing.assert_allclose(got["alphaHat_t1"], z["alphaHat_t1"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(got["betaHat_t2"], z["betaHat_t2"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(got["eMatGrid_t1"], z["eMatGrid_t1"], rtol=1e-8)
    np.testing.assert_allclose(got["c1"], z["c1"], rtol=1e-8)
    np.testing.assert_allclose(got["hapProbs_t"][:2], z["hapProbs_t"][:2], rtol=1e-8, atol=1e-14)
    dev.close()


@pytest.mark.parametrize("name", ["rtwin_block_0.npz", "rtwin_block_1.npz"])
def test_block_passes_hip_match_r_twin(name):
    """a13 / a14 on the device against the R-twin side (NIPT: block definition, six relabellings, label re-draw)."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    z = np.load(os.path.join(GOLD, name))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], 255, z["ref_error"])
    s = sample_from_arrays(z["read_ptr"], z["u"], z["bq"], z["wif"])
    dev = DevicePanel(panel)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]), None,
                                        ff=float(z["ff"]), runif_block=z["runif_block"], runif_resample=z["runif_resample"],
                                        L_grid=z["L_grid"], block_gibbs_quantile_prob=float(z["quantile_prob"]),
                                        shuffle_bin_radius=int(z["shuffle_bin_radius"]), return_state=True)
    assert np.array_equal(got["H"], z["H"]) and np.array_equal(got["H_class"], z["H_class"])
    np.testing.assert_allclose(got["alphaHat_t1"], z["alphaHat_t1"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(got["c1"], z["c1"], rtol=1e-8)
    np.testing.assert_allclose(got["hapProbs_t"], z["hapProbs_t"], rtol=1e-8, atol=1e-14)
    dev.close()
