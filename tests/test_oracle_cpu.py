"""CPU tests that pin the oracle: the reference's RNG-independent known answers and the structural
invariants its own testthat suite asserts, transcribed."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import label_gl, thin_cols


def test_simple_binary_search_known_answers():
    # test-unit-reference-single.R:150-206
    rng = np.random.default_rng(1)
    vec = np.sort(rng.choice(100000, 1000, replace=False)).astype(np.int32)
    vec2 = np.sort(rng.choice(100000, 500, replace=False)).astype(np.int32)
    vec3 = np.sort(rng.choice(100000, 300, replace=False)).astype(np.int32)
    locations = [1, 37, 50, 900, 1000]
    for loc in locations:
        assert O.simple_binary_search(vec[loc - 1], vec) == loc - 1
    assert O.simple_binary_search(7, np.array([7], dtype=np.int32)) == 0
    for pad in (1, 2, 3):
        if pad == 1:
            mat = np.concatenate([np.stack([vec2, vec2 + 10], 1), np.stack([vec, vec + 10], 1), np.stack([vec3, vec3 + 10], 1)])
            s1, e1 = 501, 1500
        elif pad == 2:
            mat = np.concatenate([np.stack([vec2, vec2 + 10], 1), np.stack([vec, vec + 10], 1)])
            s1, e1 = 501, 1500
        else:
            mat = np.concatenate([np.stack([vec, vec + 10], 1), np.stack([vec3, vec3 + 10], 1)])
            s1, e1 = 1, 1000
        for loc in locations:
            assert O.simple_binary_matrix_search(vec[loc - 1], mat, s1, e1) == vec[loc - 1] + 10
    # Appendix A.7 quirk: a one-row range returns the integer 0
    assert O.simple_binary_matrix_search(5, np.array([[5, 99]], dtype=np.int32), 1, 1) == 0


def test_gl_bounding_rule():
    # test-unit-reference-single.R:31-59: larger member becomes 1, the other is floored at minGLValue
    gl = np.asfortranarray(np.array([[1e-30, 0.2, 0.5], [1e-3, 1e-40, 0.25]]))
    O.make_gl_bound(gl, 1e-10, np.array([0, 1, 2], dtype=np.int32))
    np.testing.assert_allclose(gl[:, 0], [1e-10, 1.0])
    np.testing.assert_allclose(gl[:, 1], [1.0, 1e-10])
    np.testing.assert_allclose(gl[:, 2], [1.0, 0.5])


def test_top_K_picker_definition():
    # test-unit-reference-single.R:102-145: distinct values -> the K largest, in k order; ties widen the set
    rng = np.random.default_rng(3)
    a = rng.random(500)
    b = rng.random(500)
    idx, val, g = O.get_top_K_or_more_matches(a, b, 5)
    want = np.sort(np.argsort(-(a * b))[:5])
    assert np.array_equal(idx, want)
    np.testing.assert_allclose(val, (a * b)[want])
    a2 = np.ones(50)
    idx2, _, _ = O.get_top_K_or_more_matches(a2, a2, 5)
    assert len(idx2) == 50


def test_panel_round_trip(small_panel, ragged_panel):
    # test-unit-reference-single.R:238-307
    from quilt_amd.panel import rebuild_rhb_t
    for p in (small_panel, ragged_panel):
        tabs = dict(hapMatcher=None, hapMatcherR=p.hapMatcherR, distinctHapsB=p.distinctHapsB,
                    eMatDH_special_matrix=p.eMatDH_special_matrix,
                    eMatDH_special_matrix_helper=p.eMatDH_special_matrix_helper)
        assert np.array_equal(rebuild_rhb_t(tabs, p.K, p.nGrids), p.rhb_t)
        assert (p.eMatDH_special_grid_which != 0).sum() == len(p.eMatDH_special_values_list)


@pytest.mark.parametrize("symbols", [False, True])
def test_fullpass_invariants(small_panel, symbols):
    # test-unit-reference-single.R:588-642, :940, :1015-1029
    from quilt_amd.synth import make_synthetic_sample
    p = small_panel
    s = make_synthetic_sample(p, seed=1001, n_reads=125)
    gl = label_gl(p, s, 1, O)
    cols = thin_cols(p.nGrids)
    lazy = O.haploid_dosage_versus_refs(p, gl, cols, return_gamma_t=True, return_gammaSmall_t=True,
                                        always_normalize=False, use_eMatDH_special_symbols=symbols)
    always = O.haploid_dosage_versus_refs(p, gl, cols, return_gamma_t=True, return_gammaSmall_t=True,
                                          always_normalize=True, use_eMatDH_special_symbols=symbols)
    np.testing.assert_allclose(lazy["gamma_t"].sum(axis=0), 1.0, atol=1e-12)
    np.testing.assert_allclose(lazy["dosage"], always["dosage"], atol=1e-12)
    np.testing.assert_allclose(lazy["gammaSmall_t"], always["gammaSmall_t"], atol=1e-12)
    np.testing.assert_allclose(np.log(lazy["c"]).sum(), np.log(always["c"]).sum(), rtol=1e-12)
    truth = s.truth_haps[0]
    assert np.corrcoef(lazy["dosage"], truth)[0, 1] ** 2 > 0.7
    # normalize_emissions on/off agree
    off = O.haploid_dosage_versus_refs(p, gl, cols, normalize_emissions=False, use_eMatDH_special_symbols=symbols)
    np.testing.assert_allclose(lazy["dosage"], off["dosage"], atol=1e-10)


def test_gibbs_invariants(medium_panel):
    from quilt_amd.synth import make_synthetic_sample
    p = medium_panel
    s = make_synthetic_sample(p, seed=3, n_reads=600)
    rng = np.random.default_rng(5)
    which = np.sort(rng.choice(p.K, 150, replace=False)).astype(np.int32) + 1
    R, G = s.nReads, p.nGrids
    H0 = rng.integers(1, 3, size=R)
    ru, rs = rng.random(R * 21), rng.random(3 * (G - 1))
    out = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, int(rng.integers(0, R)), rs, gibbs_initialize_iteratively=True)
    assert out["status"] == 0
    # Appendix B (12): after block + shard Gibbs the state equals a from-scratch forward-backward given H
    fresh = O.forwardBackwardGibbsNIPT(p, s, which, out["H"], ru, 0, rs, n_gibbs_burn_in_its=0, n_gibbs_sample_its=0,
                                       perform_block_gibbs=False)
    for h in range(2):
        np.testing.assert_allclose(out["eMatGrid_t"][h], fresh["eMatGrid_t"][h], rtol=1e-12)
        np.testing.assert_allclose(out["alphaHat_t"][h], fresh["alphaHat_t"][h], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(out["betaHat_t"][h], fresh["betaHat_t"][h], rtol=1e-9, atol=1e-300)
    # (9) sparse category-2/3 updates == dense: same labels on every read that is sampled
    a = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, 5, rs)
    b = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, 5, rs, disable_read_category_usage=True, sample_is_diploid=True)
    cat = a["read_category"]
    assert np.array_equal(a["H"][cat != 1], b["H"][cat != 1])
    np.testing.assert_allclose(out["genProbsM_t"].sum(axis=0), 1.0, atol=1e-12)
    # (6) packed-panel eMatRead == dense eMatRead built from the expanded haplotypes
    from quilt_amd.synth import panel_hap_bits
    e = O.make_eMatRead_t(p, s, which, rescale_eMatRead_t=False)
    bits = np.stack([panel_hap_bits(p, int(k) - 1) for k in which[:20]]).astype(np.float64)
    eh = np.where(bits == 1, 1 - p.ref_error, p.ref_error)
    dense = O.calculate_eMatRead_t_vs_haplotypes(s, list(eh), 1e10, rescale_eMatRead_t=False, Jmax=10000)
    np.testing.assert_allclose(e[:20], dense, rtol=1e-12)


def test_rare_common_restatement_matches_dense_expansion(small_panel):
    """Rare + common emissions and hapProbs (gibbs-small.cpp:270-460, :711-867) == the dense computation on the
    haplotypes expanded over all SNPs -- the all-SNP analogue of invariants (6) and (7)."""
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common, rare_common_hap_bits
    p = small_panel
    rc = make_rare_common(p, 5)
    s_com, s_all = make_synthetic_sample_rare_common(p, rc, 77, n_reads=200)
    assert rc.nSNPs_all == p.nSNPs + (rc.snp_is_common == 0).sum() and s_com.u.max() < p.nSNPs
    rng = np.random.default_rng(1)
    which = np.sort(rng.choice(p.K, 120, replace=False)).astype(np.int32) + 1
    bits = np.stack([rare_common_hap_bits(p, rc, int(k) - 1) for k in which]).astype(np.float64)
    eh = np.where(bits == 1, 1 - p.ref_error, p.ref_error)
    for rescale in (True, False):
        e = O.make_eMatRead_t_rare_common(p, rc, s_all, which, rescale_eMatRead_t=rescale)
        dense = O.calculate_eMatRead_t_vs_haplotypes(s_all, list(eh), 1e10, rescale_eMatRead_t=rescale, Jmax=10000)
        np.testing.assert_allclose(e, dense, rtol=1e-12)
    R, G = s_all.nReads, rc.nGrids_all
    H0 = rng.integers(1, 3, size=R)
    out = O.forwardBackwardGibbsNIPT(p, s_all, which, H0, rng.random(R * 21), 0, rng.random(3 * (G - 1)),
                                     disable_read_category_usage=True, rare_common=rc)
    assert out["status"] == 0
    for h in range(2):
        gam = out["alphaHat_t"][h] * out["betaHat_t"][h] / out["c"][h][None, :]
        np.testing.assert_allclose(gam.sum(axis=0), 1.0, atol=1e-12)
        d = (gam[:, np.arange(rc.nSNPs_all) // 32] * eh).sum(axis=0)
        np.testing.assert_allclose(out["hapProbs_t"][h], d, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(out["genProbsM_t"].sum(axis=0), 1.0, atol=1e-12)
