"""The C oracle against fixtures produced by an INDEPENDENT restatement of the reference's R twins (oracle/rtwin.py;
tests/golden/make_golden_rtwin.py): oracle/*.c reads QUILT/src/*.cpp, the fixtures come from a reading of
QUILT/R/reference-single.R and QUILT/R/gibbs-nipt.R in another parameterisation.  What the two share must agree:
dosage, gamma, best-haplotype lists, alpha / c under always_normalize without emission rescaling; Gibbs read labels and
H_class under the same uniforms, the per-label state, hapProbs.  rtwin_shard_* / rtwin_block_*: whole Gibbs calls WITH
their shard passes (diploid) resp. block definition + block passes (NIPT), from the restatement of
QUILT/R/gibbs-nipt-block.R (R_shard_block_gibbs_resampler, R_block_gibbs_resampler and their parts)."""
import os

import numpy as np
import pytest

from tests.util import panel_from_rhb, sample_from_arrays

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def test_fullpass_oracle_matches_r_twin(oracle):
    z = np.load(os.path.join(GOLD, "rtwin_fullpass.npz"))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], z["nMaxDH"], z["ref_error"])
    assert (panel.hapMatcherR == 0).any(), "the fixture exercises special haplotypes"
    gl, cols = np.asfortranarray(z["gl"]), z["cols"]
    for always in (True, False):
        for norm_e in (True, False):
            r = oracle.haploid_dosage_versus_refs(panel, gl, cols, return_gamma_t=True, always_normalize=always,
                                                  normalize_emissions=norm_e, get_best_haps_from_thinned_sites=True)
            assert np.abs(r["dosage"] - z["dosage"]).max() < 1e-12
            np.testing.assert_allclose(r["gamma_t"], z["gamma_t"], rtol=1e-9, atol=1e-300)
            np.testing.assert_allclose(r["gamma_t"].sum(axis=0), 1.0, rtol=1e-12)
            idx = np.concatenate([b[0] for b in r["best_haps"]])
            ptr = np.cumsum([0] + [len(b[0]) for b in r["best_haps"]])
            assert np.array_equal(idx, z["best_idx"]) and np.array_equal(ptr, z["best_ptr"])
            if always and not norm_e:
                np.testing.assert_allclose(r["c"], z["c_R"], rtol=1e-10)
                np.testing.assert_allclose(r["alphaHat_t"], z["alphaHat_t"], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("name", ["rtwin_gibbs_labels.npz", "rtwin_gibbs_init.npz"])
def test_gibbs_oracle_matches_r_twin(oracle, name):
    z = np.load(os.path.join(GOLD, name))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], 255, z["ref_error"])
    s = sample_from_arrays(z["read_ptr"], z["u"], z["bq"], z["wif"])
    r = oracle.forwardBackwardGibbsNIPT(panel, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]),
                                        np.zeros(3 * (panel.nGrids - 1)), perform_block_gibbs=False,
                                        gibbs_initialize_iteratively=bool(z["init_iter"]))
    assert np.array_equal(r["H"], z["H"]) and np.array_equal(r["H_class"], z["H_class"])
    assert (r["H"] != z["H0"]).sum() > 10
    np.testing.assert_allclose(r["alphaHat_t"][0], z["alphaHat_t1"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(r["betaHat_t"][1], z["betaHat_t2"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(r["c"][0], z["c1"], rtol=1e-8)
    np.testing.assert_allclose(r["c"][1], z["c2"], rtol=1e-8)
    np.testing.assert_allclose(r["hapProbs_t"][:2], z["hapProbs_t"][:2], rtol=1e-8, atol=1e-14)


@pytest.mark.parametrize("name", ["rtwin_shard_0.npz", "rtwin_shard_1.npz"])
def test_shard_passes_oracle_matches_r_twin(oracle, name):
    """the diploid call with its three shard passes; the fixture holds flips (asserted when it was made)."""
    z = np.load(os.path.join(GOLD, name))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], 255, z["ref_error"])
    s = sample_from_arrays(z["read_ptr"], z["u"], z["bq"], z["wif"])
    r = oracle.forwardBackwardGibbsNIPT(panel, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]), z["runif_shard"])
    assert z["flip_mode"].sum() >= 2
    assert np.array_equal(r["H"], z["H"]) and np.array_equal(r["H_class"], z["H_class"])
    np.testing.assert_allclose(r["alphaHat_t"][0], z["alphaHat_t1"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(r["betaHat_t"][1], z["betaHat_t2"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(r["eMatGrid_t"][0], z["eMatGrid_t1"], rtol=1e-8)
    np.testing.assert_allclose(r["c"][0], z["c1"], rtol=1e-8)
    np.testing.assert_allclose(r["c"][1], z["c2"], rtol=1e-8)
    np.testing.assert_allclose(r["hapProbs_t"][:2], z["hapProbs_t"][:2], rtol=1e-8, atol=1e-14)
    # without the shard passes the labels come out differently: the passes are what is being compared
    r0 = oracle.forwardBackwardGibbsNIPT(panel, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]), z["runif_shard"],
                                         perform_block_gibbs=False)
    assert not np.array_equal(r0["H"], z["H"])


@pytest.mark.parametrize("name", ["rtwin_block_0.npz", "rtwin_block_1.npz"])
def test_block_passes_oracle_matches_r_twin(oracle, name):
    """the NIPT call with block definition and three block passes (several blocks each, several
    different relabellings chosen -- asserted when the fixture was made)."""
    z = np.load(os.path.join(GOLD, name))
    panel = panel_from_rhb(z["rhb_t"], z["transMatRate_t"], z["nSNPs"], 255, z["ref_error"])
    s = sample_from_arrays(z["read_ptr"], z["u"], z["bq"], z["wif"])
    kw = dict(ff=float(z["ff"]), runif_block=z["runif_block"], runif_resample=z["runif_resample"], L_grid=z["L_grid"],
              block_gibbs_quantile_prob=float(z["quantile_prob"]), shuffle_bin_radius=int(z["shuffle_bin_radius"]))
    r = oracle.forwardBackwardGibbsNIPT(panel, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]),
                                        np.zeros(3 * (panel.nGrids - 1)), **kw)
    assert len(set(z["ir_chosen"].tolist())) >= 3 and z["n_blocks"].min() >= 4
    assert np.array_equal(r["H"], z["H"]) and np.array_equal(r["H_class"], z["H_class"])
    np.testing.assert_allclose(r["alphaHat_t"][0], z["alphaHat_t1"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(r["betaHat_t"][2], z["betaHat_t3"], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(r["eMatGrid_t"][1], z["eMatGrid_t2"], rtol=1e-8)
    np.testing.assert_allclose(r["c"][0], z["c1"], rtol=1e-8)
    np.testing.assert_allclose(r["c"][2], z["c3"], rtol=1e-8)
    np.testing.assert_allclose(r["hapProbs_t"], z["hapProbs_t"], rtol=1e-8, atol=1e-14)
    r0 = oracle.forwardBackwardGibbsNIPT(panel, s, z["which"], z["H0"], z["runif_reads"], int(z["first_read"]),
                                         np.zeros(3 * (panel.nGrids - 1)), perform_block_gibbs=False, **kw)
    assert not np.array_equal(r0["H"], z["H"])
