"""GPU parity at the HEADLINE dimensions (BASELINE.json configs[2..4]): single native calls against the fp64 CPU oracle.

The pipeline at K = 50 000 costs the oracle ~20 core-minutes per sample, so whole-pipeline parity at full size is checked by
properties only (tests/test_pipeline_gpu.py); but ONE call is seconds to a minute of CPU, and effects that only show over
2 000 grids / 20 000 reads (renormalisations, read streams crossing hundreds of 64-entry reloads, reciprocal refinement)
cannot be seen in small tests.  Each test here is one call of the production geometry:

  * one small-panel Gibbs call, Ks = 600, 2 000 grids, 20 000 short reads, shard passes: labels and H_class bit-identical,
    alpha / beta / eMatGrid / c to 1e-9 relative                               (gibbs-nipt.cpp:2395-3307)
  * one ONT-length read set (300 reads x 200-800 SNPs, every emission a dense column), Ks = 600: the same bar
  * one NIPT call (three labels, block Gibbs, ff = 0.2), Ks = 600, 20 000 reads: labels / classes identical, hapProbs 1e-9
  * one thin pass and one dosage pass at K = 50 000 x 2 000 grids: best-haplotype lists identical (values 1e-9), alpha at the
    thinned grids and c to 1e-9 (the fp64 ranking kernels follow the reference's lazy normalisation operation by operation),
    |dosage diff| <= 2e-6 (fp32 state) resp. <= 1e-9 (qa_panel_set_dosage_precision(64)), sum(log c) to 1e-7
                                                                                (reference-single.cpp:2189-2413)
"""
import numpy as np
import pytest

from tests.util import check_best_haps, r2

pytestmark = pytest.mark.gpu

RTOL = 1e-9
K_HEAD, T_HEAD, KS, R_HEAD = 50000, 64000, 600, 20000


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


@pytest.fixture(scope="module")
def head_panel():
    from quilt_amd.synth import make_synthetic_panel
    return make_synthetic_panel(K=K_HEAD, nSNPs=T_HEAD, seed=4916)


@pytest.fixture(scope="module")
def head_dev(head_panel):
    from quilt_amd.native import DevicePanel
    dev = DevicePanel(head_panel)
    yield dev
    dev.close()


def _gibbs_inputs(panel, seed, n_reads, mode="short", ff=0.0):
    from quilt_amd.synth import make_synthetic_sample
    s = make_synthetic_sample(panel, seed=seed, n_reads=n_reads, mode=mode, ff=ff)
    rng = np.random.default_rng(seed + 17)
    which = np.sort(rng.choice(panel.K, KS, replace=False)).astype(np.int32) + 1
    R = s.nReads
    if ff > 0:
        H0 = (rng.choice(3, size=R, p=[0.5, 0.5 - ff / 2, ff / 2]) + 1).astype(np.int32)
    else:
        H0 = rng.integers(1, 3, size=R).astype(np.int32)
    return s, which, H0, rng.random(R * 21), rng.random(3 * (panel.nGrids - 1)), int(rng.integers(0, R)), rng


def _compare_state(got, ref):
    assert not got["underflow_problem"] and ref["status"] == 0
    assert np.array_equal(got["H"], ref["H"]), f"{(got['H'] != ref['H']).sum()} labels differ"
    assert np.array_equal(got["H_class"], ref["H_class"])
    for h in range(2):
        np.testing.assert_allclose(got[f"eMatGrid_t{h + 1}"], ref["eMatGrid_t"][h], rtol=RTOL)
        np.testing.assert_allclose(got[f"alphaHat_t{h + 1}"], ref["alphaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"betaHat_t{h + 1}"], ref["betaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"c{h + 1}"], ref["c"][h], rtol=RTOL)
    np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)


@pytest.mark.parametrize("init_iter", [True, False])
def test_gibbs_call_headline_size(head_panel, head_dev, oracle, init_iter):
    """configs[2]: Ks = 600, G = 2 000, R = 20 000, 21 sweeps, shard passes after sweeps 3, 6, 9; first-round
    (iterative initialisation) and later-round form."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    s, which, H0, ru, rs, fr, _ = _gibbs_inputs(head_panel, 1001, R_HEAD)
    assert s.nReads == R_HEAD and head_panel.nGrids == 2000
    ref = oracle.forwardBackwardGibbsNIPT(head_panel, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter)
    got = rcpp_forwardBackwardGibbsNIPT(head_dev, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter,
                                        return_state=True)
    _compare_state(got, ref)
    assert (got["H"] != H0).sum() > 1000   # the sampler did move labels


def test_gibbs_call_ont_headline_size(head_panel, head_dev, oracle):
    """configs[3]: 300 reads of 200-800 SNPs, phred 5-15: every read emission is a dense Ks-column."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    s, which, H0, ru, rs, fr, _ = _gibbs_inputs(head_panel, 1003, 300, mode="ont")
    n = np.diff(s.read_ptr)
    assert s.nReads == 300 and n.min() >= 200 and n.max() <= 800
    ref = oracle.forwardBackwardGibbsNIPT(head_panel, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=True)
    got = rcpp_forwardBackwardGibbsNIPT(head_dev, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=True,
                                        return_state=True)
    _compare_state(got, ref)


def test_nipt_call_headline_size(head_panel, head_dev, oracle):
    """configs[4]: three labels, block Gibbs (blocks from the switch rate over 2 000 grids), ff = 0.2, 20 000 reads."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    ff = 0.2
    s, which, H0, ru, _, fr, rng = _gibbs_inputs(head_panel, 1005, R_HEAD, ff=ff)
    R = s.nReads
    rb, rr = rng.random(3 * R), rng.random(3 * R)
    ref = oracle.forwardBackwardGibbsNIPT(head_panel, s, which, H0, ru, fr, np.zeros(3 * head_panel.nGrids), ff=ff,
                                          gibbs_initialize_iteratively=True, runif_block=rb, runif_resample=rr)
    got = rcpp_forwardBackwardGibbsNIPT(head_dev, s, which, H0, ru, fr, None, ff=ff, gibbs_initialize_iteratively=True,
                                        runif_block=rb, runif_resample=rr)
    assert ref["status"] == 0 and not got["underflow_problem"]
    assert np.array_equal(got["H"], ref["H"]), f"{(got['H'] != ref['H']).sum()} labels differ"
    assert np.array_equal(got["H_class"], ref["H_class"])
    np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(got["genProbsM_t"], ref["genProbsM_t"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(got["genProbsF_t"], ref["genProbsF_t"], rtol=RTOL, atol=1e-14)


@pytest.fixture(scope="module")
def head_gl(head_panel, oracle):
    """gl of one read label of a 20 000-read sample (the labels a converged Gibbs chain would hold: the truth)."""
    from quilt_amd.synth import make_synthetic_sample
    from tests.util import label_gl
    s = make_synthetic_sample(head_panel, seed=1007, n_reads=R_HEAD)
    return label_gl(head_panel, s, 1, oracle)


def _run_gpu(dev, gl, cols, matrices=True, **kw):
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    P = dev.panel
    n_thin = int((cols >= 0).sum())
    out = dict(c=np.ones(P.nGrids), dosage=np.zeros(P.nSNPs), best_haps_stuff_list=[None] * n_thin)
    if matrices:
        out["alphaHat_t"] = np.zeros((P.K, P.nGrids), order="F")
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, return_gamma_t=False, return_betaHat_t=False,
                                    **out, **kw)
    return out


def test_thin_pass_headline_size(head_panel, head_dev, head_gl, oracle):
    """One thin pass (return_dosage = FALSE: best-haplotype lists at the 200 thinned grids) at K = 50 000 x 2 000 grids: the
    fp64 ranking kernels (lazy normalisation over 2 000 grids, fused / separate top-K)."""
    from quilt_amd.driver import thinned_grid_columns
    cols = thinned_grid_columns(head_panel.nGrids, 0.1)
    assert int((cols >= 0).sum()) == 200
    ref = oracle.haploid_dosage_versus_refs(head_panel, head_gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True)
    got = _run_gpu(head_dev, head_gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True, always_normalize=False)
    check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    # the reference's own (lazily normalised) alpha columns and c: same operations, only the order of the K-wide sums differs
    np.testing.assert_allclose(got["c"], ref["c"], rtol=RTOL)
    np.testing.assert_allclose(np.log(got["c"]).sum(), np.log(ref["c"]).sum(), rtol=1e-12)
    n_renorm = int((np.abs(ref["c"][1:] * head_panel.transMatRate_t[0] - 1) > 1e-9).sum())
    assert 5 < n_renorm < 1000, "the lazy schedule renormalises now and then, not every grid"
    for g in np.nonzero(cols >= 0)[0][::20]:
        np.testing.assert_allclose(got["alphaHat_t"][:, g], ref["alphaHat_t"][:, g], rtol=RTOL, atol=1e-300)


def test_validation_mode_headline_size(head_panel, head_dev, head_gl, oracle):
    """qa_panel_set_sum_order(1) at K = 50 000 x 2 000 grids (state in the pass's HBM scratch, one lane adding 50 000 values per grid in
    the reference's order): a thin pass and a dosage pass equal the oracle BIT FOR BIT -- lists, c, alpha at the thinned grids,
    dosage.  No tolerance."""
    from quilt_amd.driver import thinned_grid_columns
    cols = thinned_grid_columns(head_panel.nGrids, 0.1)
    head_dev.set_sum_order(True)
    try:
        ref = oracle.haploid_dosage_versus_refs(head_panel, head_gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True)
        got = _run_gpu(head_dev, head_gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True, always_normalize=False)
        assert np.array_equal(got["c"], ref["c"])
        for g, (oi, ov) in zip(got["best_haps_stuff_list"], ref["best_haps"]):
            assert np.array_equal(g["top_matches"], oi) and np.array_equal(g["top_matches_values"], ov)
        for g in np.nonzero(cols >= 0)[0][::20]:
            assert np.array_equal(got["alphaHat_t"][:, g], ref["alphaHat_t"][:, g])
        refd = oracle.haploid_dosage_versus_refs(head_panel, head_gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
        gotd = _run_gpu(head_dev, head_gl, cols, matrices=False, return_dosage=True, get_best_haps_from_thinned_sites=True,
                        always_normalize=False)
        assert np.array_equal(gotd["dosage"], refd["dosage"]) and np.array_equal(gotd["c"], refd["c"])
    finally:
        head_dev.set_sum_order(False)


def test_dosage_pass_headline_size(head_panel, head_dev, head_gl, oracle):
    """One dosage pass at K = 50 000 x 2 000 grids: fp32 state (default) and fp64 state (qa_panel_set_dosage_precision)."""
    from quilt_amd.driver import thinned_grid_columns
    cols = thinned_grid_columns(head_panel.nGrids, 0.1)
    ref = oracle.haploid_dosage_versus_refs(head_panel, head_gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    got = _run_gpu(head_dev, head_gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    assert np.abs(got["dosage"] - ref["dosage"]).max() <= 2e-6
    assert r2(got["dosage"], ref["dosage"]) >= 0.999999
    np.testing.assert_allclose(np.log(got["c"]).sum(), np.log(ref["c"]).sum(), rtol=1e-7)
    check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    head_dev.set_dosage_precision(64)
    try:
        # the call of a dosage round (no K x G matrices back): k_fwd64 + k_bwd64d, the lists from the ranking pass beside them
        got64 = _run_gpu(head_dev, head_gl, cols, matrices=False, return_dosage=True, get_best_haps_from_thinned_sites=True,
                         always_normalize=False)
        # with alphaHat_t asked for: the generic fp64 kernels
        got64g = _run_gpu(head_dev, head_gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    finally:
        head_dev.set_dosage_precision(32)
    assert np.abs(got64["dosage"] - ref["dosage"]).max() <= 1e-10
    np.testing.assert_allclose(got64["c"], ref["c"], rtol=1e-10)   # (elementwise: same renormalisation grids; K-wide sums in another order)
    check_best_haps(got64["best_haps_stuff_list"], ref["best_haps"])
    assert np.abs(got64g["dosage"] - ref["dosage"]).max() <= 1e-9
    np.testing.assert_allclose(np.log(got64g["c"]).sum(), np.log(ref["c"]).sum(), rtol=1e-12)
    check_best_haps(got64g["best_haps_stuff_list"], ref["best_haps"])


def test_haplotype_search_headline_size(head_panel, head_dev):
    """msPBWT mode's search at K = 50 000 x 2 000 grids (four interleaved indices, QUILT's defaults): the device's tables equal
    the numpy statement of the definition -- integer work, identical."""
    from quilt_amd.mspbwt import find_good_matches, match_tables_as_lists, rcpp_int_contract
    from quilt_amd.synth import make_truth_haplotype, panel_hap_bits
    from tests.oracle_backend import find_good_matches_bruteforce
    rng = np.random.default_rng(3)
    queries = [panel_hap_bits(head_panel, 12345), make_truth_haplotype(head_panel, rng)]
    noisy = queries[1].copy()
    flip = rng.random(len(noisy)) < 0.002
    noisy[flip] = 1 - noisy[flip]
    queries.append(noisy)
    Zs = np.stack([rcpp_int_contract(q) for q in queries])
    got = match_tables_as_lists(*find_good_matches(head_dev, Zs, 4, 1, 150))
    ref = find_good_matches_bruteforce(head_panel, Zs, 4, 1, 150)
    for q in range(len(Zs)):
        for i in range(4):
            assert np.array_equal(got[q][i], ref[q][i]), (q, i)
    assert got[0][0][:, 2].max() == len(range(0, head_panel.nGrids, 4)) or (np.asarray(head_panel.hapMatcherR)[12345, 0::4] == 0).any()


def test_mspbwt_neighbour_scan_headline_size(head_panel):
    """use_mspbwt = TRUE at K = 50 000 x 2 000 grids, QUILT's defaults (four indices, mspbwtL = 3, mspbwtM = 1): the product's
    query -- the panel's msPBWT indices and their neighbour scan, csrc/mspbwt.cpp -- returns the rows of the restated scan
    (tests/mspbwt_scan.py) and select_new_haps_mspbwt_v3 chooses the same next small panel from both (`selected` = 1.0)."""
    from quilt_amd.mspbwt import panel_mspbwt_index, rcpp_int_contract
    from quilt_amd.synth import make_truth_haplotype
    from tests.mspbwt_scan import find_good_matches_scan, selection_agreement
    rng = np.random.default_rng(3)
    clean = make_truth_haplotype(head_panel, rng)
    noisy = make_truth_haplotype(head_panel, rng)
    flip = rng.random(len(noisy)) < 0.002
    noisy[flip] = 1 - noisy[flip]
    Zs = np.stack([rcpp_int_contract(clean), rcpp_int_contract(noisy)])
    idx = panel_mspbwt_index(head_panel, 4)
    got = idx.find_good_matches(Zs, 3, 1)
    want = find_good_matches_scan(head_panel, Zs, 4, 3, 1)
    for q in range(2):
        for i in range(4):
            assert np.array_equal(got[q][i], want[q][i]), (q, i)
    a = selection_agreement(want, got, KS, head_panel.K, head_panel.nGrids)
    assert a["selected"] == 1.0 and a["longest"] == 1.0
    sel = idx.select_new_haps(Zs, 2, 3, 1, KS, [77])
    from quilt_amd.mspbwt import select_new_haps_mspbwt_v3
    assert np.array_equal(sel[0], select_new_haps_mspbwt_v3(want, KS, head_panel.K, head_panel.nGrids, 77))


def test_hrc_sized_panel(oracle):
    """The HRC panel's scale: K = 64 976 haplotypes (eight chunk rows: seven on chip, one streamed through HBM) over 500 grids
    with the lazily normalised schedule at work; one dosage pass with fp64 state and one thin pass through the single-pass entry
    point: dosage <= 1e-9, c to 1e-9, lists identical."""
    from quilt_amd.driver import thinned_grid_columns
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.util import label_gl
    panel = make_synthetic_panel(K=64976, nSNPs=16000, seed=64976)
    s = make_synthetic_sample(panel, seed=12, n_reads=5000)
    gl = label_gl(panel, s, 2, oracle)
    cols = thinned_grid_columns(panel.nGrids, 0.1)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True,
                                            always_normalize=False)
    got = _run_gpu(dev, gl, cols, matrices=False, return_dosage=True, get_best_haps_from_thinned_sites=True, always_normalize=False)
    dev.close()
    assert np.abs(got["dosage"] - ref["dosage"]).max() <= 1e-9
    np.testing.assert_allclose(got["c"], ref["c"], rtol=RTOL)
    check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    n_renorm = int((np.abs(ref["c"][1:] * panel.transMatRate_t[0] - 1) > 1e-9).sum())
    assert 1 < n_renorm < 400
