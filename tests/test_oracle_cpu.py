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


def test_nipt_block_pieces_known_answers():
    """test-unit-gibbs-block-nipt.R:35-42 (H_class log-probability), :92-130 (label / class swap table), :132-138
    (quantile), :238-304 (make_gibbs_considers: brute-force definition, no gaps, non-overlapping and spanning)."""
    ln = np.log
    assert O.get_log_p_H_class2(5, 10, 15, 20, 25, 30, 0.2) == pytest.approx(
        5 * ln(.5) + 10 * ln(.4) + 15 * ln(.1) + 20 * ln(.9) + 25 * ln(.6) + 30 * ln(.5), rel=1e-14)
    rx = np.array([[1, 2, 3], [1, 3, 2], [2, 1, 3], [3, 1, 2], [2, 3, 1], [3, 2, 1]])
    for ir in range(6):
        one_based_swap = np.r_[1, 1 + rx[ir], 8 - rx[ir][::-1], 8]
        assert np.array_equal(O.zero_based_swap(ir), one_based_swap - 1)
    rng = np.random.default_rng(7)
    x = rng.random(1000)
    assert O.simple_quantile(x, 0.90) == np.sort(x)[int(0.90 * 1000)]
    # sample(1:3, 1, prob): mass in decreasing order, first cumulative mass >= u
    assert [O.sample3((0.5, 0.4, 0.1), u) for u in (0.3, 0.5, 0.7, 0.95)] == [1, 1, 2, 3]
    assert [O.sample3((0, 0.4, 0.1), u) for u in (0.3, 0.85)] == [2, 3]
    for trial in range(30):
        G = int(rng.integers(20, 80))
        n_cut = int(rng.integers(1, 8))
        cuts = np.sort(rng.choice(np.arange(1, G), size=n_cut, replace=False))
        blocked = np.searchsorted(cuts, np.arange(G), side="right").astype(np.int32)
        R = int(rng.integers(10, 200))
        wif = np.sort(rng.integers(0, G, size=R)).astype(np.int32)
        out = O.make_gibbs_considers(blocked, wif)
        n = out["n_blocks"]
        gs, ge = out["consider_grid_start_0_based"], out["consider_grid_end_0_based"]
        rs, re_ = out["consider_reads_start_0_based"], out["consider_reads_end_0_based"]
        n_blocks0 = blocked[-1] + 1
        has = [np.isin(wif, np.arange(np.nonzero(blocked == b)[0][0], np.nonzero(blocked == b)[0][-1] + 1)) for b in range(n_blocks0)]
        # grids: no gaps, span -- except that removing a read-less LAST block leaves the final grid out (:1476-1483:
        # `x <- grid_end[last]; grid_end[s1 - 1] <- x - 1`), which the restatement keeps
        last_removed = out["consider_reads_start_0_based"] is not None and not has[-1].any() and n < n_blocks0
        assert gs[0] == 0 and np.all(gs[1:] - 1 == ge[:-1]) and ge[-1] == (G - 2 if last_removed else G - 1)
        assert np.array_equal(np.nonzero(out["consider_grid_where_0_based"] >= 0)[0], ge)
        if all(h.any() for h in has) and blocked[wif[-1]] == blocked[wif[-2]]:
            # nothing removed: the brute-force definition (:238-262)
            assert n == n_blocks0
            for b in range(n):
                w = np.nonzero(has[b])[0]
                assert (rs[b], re_[b]) == (w[0], w[-1])
            assert rs[0] == 0 and re_[-1] == R - 1 and np.all(rs[1:] - 1 == re_[:-1])   # reads: non-overlapping, span


def test_nipt_block_gibbs_invariants(medium_panel):
    """ff > 0 with the block resampler: the state left behind equals a from-scratch forward-backward given the labels
    (Appendix B (12)), the smoothing is a weighted mean, block definitions cover the grids."""
    from quilt_amd.synth import make_synthetic_sample
    p = medium_panel
    s = make_synthetic_sample(p, seed=3, n_reads=800, ff=0.2)
    rng = np.random.default_rng(5)
    which = np.sort(rng.choice(p.K, 100, replace=False)).astype(np.int32) + 1
    R, G = s.nReads, p.nGrids
    H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=R)
    ru, rs = rng.random(R * 21), rng.random(3 * (G - 1))
    rb, rr = rng.random(3 * R), rng.random(3 * R)
    # stop right after the last block pass (sweeps 0..9, block passes after sweeps 3, 6 and 9)
    out = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, 3, rs, ff=0.2, gibbs_initialize_iteratively=True,
                                     n_gibbs_burn_in_its=9, runif_block=rb, runif_resample=rr)
    assert out["status"] == 0 and set(np.unique(out["H"])) <= {1, 2, 3}
    fresh = O.forwardBackwardGibbsNIPT(p, s, which, out["H"], ru, 0, rs, ff=0.2, n_gibbs_burn_in_its=0,
                                       n_gibbs_sample_its=0, perform_block_gibbs=False)
    for h in range(3):
        np.testing.assert_allclose(out["eMatGrid_t"][h], fresh["eMatGrid_t"][h], rtol=1e-12)
        np.testing.assert_allclose(out["alphaHat_t"][h], fresh["alphaHat_t"][h], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(out["betaHat_t"][h], fresh["betaHat_t"][h], rtol=1e-9, atol=1e-300)
    without = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, 3, rs, ff=0.2, gibbs_initialize_iteratively=True,
                                         n_gibbs_burn_in_its=9, perform_block_gibbs=False)
    assert (without["H"] != out["H"]).any()          # the block passes do something
    sm = O.make_smoothed_rate(np.full(G - 1, 0.3), p.L_grid, 5000)
    np.testing.assert_allclose(sm, 0.3, rtol=1e-12)
    rate = rng.random(G - 1) * 0.02
    rate[[20, 21, 22, 60]] = 0.9
    blocked = O.define_blocked_grids(rate, p.L_grid)
    assert blocked[0] == 0 and np.all(np.diff(blocked) >= 0) and np.all(np.diff(blocked) <= 1) and blocked[-1] >= 1


# ---------------------------------------------------------------------------------------------------------------------------
# Armadillo's sum(): the order the oracle adds the reference's arma sum() sites in (oracle/quilt_oracle.h)
# ---------------------------------------------------------------------------------------------------------------------------
def _arma_accumulate(x):
    """arrayops::accumulate / accu_proxy_linear without -ffast-math, stated independently of the C macro: the even-indexed
    elements added left to right into one accumulator, the odd-indexed ones into a second, acc1 + acc2.  (np.cumsum is a
    sequential recurrence, unlike np.sum's pairwise tree.)"""
    x = np.asarray(x, dtype=np.float64)
    acc1 = np.cumsum(x[0::2])[-1] if len(x[0::2]) else 0.0
    acc2 = np.cumsum(x[1::2])[-1] if len(x[1::2]) else 0.0
    return np.float64(acc1) + np.float64(acc2)


def _left_to_right(x):
    return np.cumsum(np.asarray(x, dtype=np.float64))[-1]


@pytest.mark.parametrize("K_keep", [None, 501])   # an even and an odd length (the odd tail goes into the first accumulator)
def test_grid0_sum_is_armadillos_two_accumulator_sum(small_panel, K_keep):
    """c(0) = 1 / sum(alphaHat_t_col), reference-single.cpp:2347: alphaHat_t_col is an arma::colvec, sum() of it is
    arrayops::accumulate.  The oracle's default must be that order, its switch the left-to-right sum of rounds 1-5, and the
    two must be told apart by this input (they differ in the last bits)."""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    p = small_panel if K_keep is None else make_synthetic_panel(K=K_keep, nSNPs=320, seed=11)
    s = make_synthetic_sample(p, seed=1001, n_reads=125)
    gl = label_gl(p, s, 1, O)
    cols = thin_cols(p.nGrids)
    # alpha(0) before normalisation, from the oracle's own emission table: prob(k) / K
    e0 = O.build_eMatDH(p.distinctHapsB, gl, p.nGrids, p.nSNPs, p.ref_error, add_zero_row=True)[:, 0]
    code0 = p.hapMatcherR[:, 0].astype(np.int64)
    assert (code0 > 0).all(), "pick a panel whose grid 0 has no special haplotypes for this check"
    a0 = e0[code0] * (1 / float(p.K))
    assert O.get_sum_order() is False
    try:
        arma = O.haploid_dosage_versus_refs(p, gl, cols)["c"][0]
        O.set_sum_order(True)
        ltr = O.haploid_dosage_versus_refs(p, gl, cols)["c"][0]
    finally:
        O.set_sum_order(False)
    assert arma == 1 / _arma_accumulate(a0)
    assert ltr == 1 / _left_to_right(a0)


def test_gibbs_sums_follow_the_switch(medium_panel):
    """The sampler's Ks-wide sums are Armadillo sum() calls as well (gibbs-nipt.cpp:644-724, 854-974, 1270-1287;
    copied-from-stitch.cpp:367-435): with the same uniforms the two summation orders give the same labels and states that
    agree to rounding -- but not the same bits, which shows the switch reaches the sampler."""
    from quilt_amd.synth import make_synthetic_sample
    p = medium_panel
    s = make_synthetic_sample(p, seed=3, n_reads=400)
    rng = np.random.default_rng(5)
    which = np.sort(rng.choice(p.K, 150, replace=False)).astype(np.int32) + 1
    R, G = s.nReads, p.nGrids
    H0 = rng.integers(1, 3, size=R)
    ru, rs = rng.random(R * 21), rng.random(3 * (G - 1))
    try:
        a = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, 7, rs)
        O.set_sum_order(True)
        b = O.forwardBackwardGibbsNIPT(p, s, which, H0, ru, 7, rs)
    finally:
        O.set_sum_order(False)
    assert np.array_equal(a["H"], b["H"])
    np.testing.assert_allclose(a["alphaHat_t"][0], b["alphaHat_t"][0], rtol=1e-10, atol=1e-300)
    assert not np.array_equal(a["alphaHat_t"][0], b["alphaHat_t"][0])
