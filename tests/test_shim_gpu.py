"""shim/quilt_amd_shim.c EXECUTED on the device under the test runtime of R's C API subset (tests/c/mini_r.c, tests/mini_r.py):
the registered `.Call` routines are called by name with R-shaped objects, as R's `.Call` would, and their results compared
with the Python host side's calls of the same library.  (R itself is not in the image: this runs the shim's own marshalling,
caching, error and registration code -- what `make -C shim check` only type-checks.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from tests.mini_r import R as Runtime
    r = Runtime()
    yield r
    r.dotcall("qa_shim_release")
    r.reset()


@pytest.fixture(scope="module")
def panel():
    from quilt_amd.synth import make_synthetic_panel
    return make_synthetic_panel(K=1024, nSNPs=96 * 32, seed=21)


def _params(R, prm, **more):
    d = dict(nGibbsSamples=R.integer([prm.nGibbsSamples]), n_seek_its=R.integer([prm.n_seek_its]), Ksubset=R.integer([prm.Ksubset]),
             Knew=R.integer([prm.Knew]), K_top_matches=R.integer([prm.K_top_matches]), heuristic_match_thin=R.real([prm.heuristic_match_thin]),
             small_ref_panel_gibbs_iterations=R.integer([prm.small_ref_panel_gibbs_iterations]),
             small_ref_panel_block_gibbs_iterations=R.integer(list(prm.small_ref_panel_block_gibbs_iterations)),
             maxDifferenceBetweenReads=R.real([prm.maxDifferenceBetweenReads]), minGLValue=R.real([prm.minGLValue]), Jmax=R.integer([prm.Jmax]),
             seed=R.real([float(prm.seed)]), samples_per_launch_set=R.integer([2]))
    d.update(more)
    return R.named(d)


def _check(out, want, nL=2):
    T = want[0].dosage.shape[0]
    for i, w in enumerate(want):
        assert np.array_equal(out["dosage"][:, i], w.dosage)
        assert np.array_equal(out["gp_t"][:, i].reshape(3, T), w.gp_t)
        assert np.array_equal(out["phasing_haps"][:, i].reshape(nL, T).T, w.phasing_haps)
        assert np.array_equal(out["read_labels"][i], w.read_labels)
        assert out["nDosage"][i] == w.nDosage


def test_sample_range_call_equals_the_python_caller(R, panel):
    """`.Call("qa_impute_sample_range", ...)` with sampleReads lists and the prepared-panel objects == quilt_amd.impute on the
    same device, bit for bit; two host threads; the routine is registered with six arguments."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    samples = [make_synthetic_sample(panel, seed=810 + i, n_reads=300) for i in range(5)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=77)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    want = impute_samples([dev], samples, prm, sample_offset=4, samples_per_launch_set=2)
    dev.close()
    assert R.arity("qa_impute_sample_range") == 6
    out = R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s) for s in samples]), R.panel_objects(panel), _params(R, prm),
                    R.real([4.0]), R.integer([2]), R.nil)
    assert out["dosage"].shape == (panel.nSNPs, 5) and out["stats"][3] > 0
    _check(out, want)


def test_sample_range_call_quilt2_defaults(R, panel):
    """use_mspbwt = TRUE with impute_rare_common = TRUE through the routine: the msPBWT indices built inside the call, the
    all-SNP handle from special_rare_common_objects' entries, allSNP_sampleReads as the sixth argument."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    rc = make_rare_common(panel, 6)
    samples = [make_synthetic_sample_rare_common(panel, rc, 820 + i, n_reads=300)[0] for i in range(3)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=78, impute_rare_common=True, use_mspbwt=True, mspbwt_nindices=2)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    drc = DeviceRareCommon(dev, rc)
    want = impute_samples([dev], samples, prm, samples_per_launch_set=2, drcs=[drc])
    drc.close()
    dev.close()
    rare = [R.integer(rc.rare_snp[rc.rare_ptr[k]:rc.rare_ptr[k + 1]]) for k in range(panel.K)]
    rco = R.named(dict(snp_is_common=R.logical(rc.snp_is_common), rare_per_hap_info=R.list(rare), transMatRate_t=R.real(rc.transMatRate_t_all),
                       L_grid=R.integer(rc.L_grid_all)))
    out = R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s) for s in samples]), R.panel_objects(panel, rare_common=rco),
                    _params(R, prm, use_mspbwt=R.logical([1]), mspbwtL=R.integer([prm.mspbwtL]), mspbwtM=R.integer([prm.mspbwtM]),
                            mspbwt_nindices=R.integer([2]), impute_rare_common=R.logical([1])),
                    R.real([0.0]), R.integer([1]), R.list([R.sample_reads(s.all_snp) for s in samples]))
    assert out["dosage"].shape == (rc.nSNPs_all, 3)
    _check(out, want)


def test_sample_range_call_nipt(R, panel):
    """method = "nipt" through the routine: params$ff, panel_objects$L_grid, the fetus' outputs in the returned list."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    samples = [make_synthetic_sample(panel, seed=830 + i, n_reads=300, ff=0.1 + 0.05 * i) for i in range(3)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=79, method="nipt")
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    want = impute_samples([dev], samples, prm, samples_per_launch_set=2)
    dev.close()
    out = R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s) for s in samples]),
                    R.panel_objects(panel, L_grid=R.integer(panel.L_grid)),
                    _params(R, prm, method=R.string("nipt"), ff=R.real([s.ff for s in samples]), shuffle_bin_radius=R.integer([prm.shuffle_bin_radius])),
                    R.real([0.0]), R.integer([1]), R.nil)
    _check(out, want, nL=3)
    T = panel.nSNPs
    for i, w in enumerate(want):
        assert np.array_equal(out["fet_dosage"][:, i], w.fet_dosage)
        assert np.array_equal(out["fet_gp_t"][:, i].reshape(3, T), w.fet_gp_t)


def test_sample_range_call_raises_r_errors(R, panel):
    """What the library refuses comes back as an R error with its text (Rf_error), not as a crash or a silent result."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.mini_r import RError
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128)
    s = make_synthetic_sample(panel, seed=840, n_reads=100)
    with pytest.raises(RError, match="Incorrect number of arguments"):
        R.dotcall("qa_impute_sample_range", R.nil, R.nil, R.nil)
    with pytest.raises(RError, match="panel_objects needs"):
        R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s)]), R.named({}), _params(R, prm), R.real([0.0]), R.integer([1]), R.nil)
    with pytest.raises(RError, match="params.ff"):   # method = "nipt" without fetal fractions
        R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s)]), R.panel_objects(panel), _params(R, prm, method=R.string("nipt")),
                  R.real([0.0]), R.integer([1]), R.nil)
    with pytest.raises(RError, match="no reads"):
        R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s), R.list([])]), R.panel_objects(panel), _params(R, prm), R.real([0.0]),
                  R.integer([1]), R.nil)


def _entry_args(name):
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return json.load(open(os.path.join(root, "tests", "golden", "callentries.json")))[name]["args"]


def _call_by_name(R, name, vals):
    """`.Call(name, ...)` with the arguments in the reference's order (tests/golden/callentries.json, from RcppExports.cpp);
    arguments the test does not name are NULL."""
    order = _entry_args(name)
    assert set(vals) <= set(order), set(vals) - set(order)
    return R.dotcall(name, *[vals.get(a, R.nil) for a in order])


def _panel_args(R, panel):
    return dict(hapMatcher=R.integer(np.zeros((1, 1), dtype=np.int32)), hapMatcherR=R.raw(panel.hapMatcherR), use_hapMatcherR=R.logical([1]),
                distinctHapsB=R.integer(panel.distinctHapsB), distinctHapsIE=R.real(panel.distinctHapsIE),
                eMatDH_special_matrix_helper=R.integer(panel.eMatDH_special_matrix_helper), eMatDH_special_matrix=R.integer(panel.eMatDH_special_matrix),
                rhb_t=R.integer(panel.rhb_t), ref_error=R.real([panel.ref_error]))


def test_full_panel_entry_38_arguments(R, panel):
    """`.Call("_QUILT_Rcpp_haploid_dosage_versus_refs", <38 arguments in RcppExports.cpp's order>)`: R's matrices are written in
    place (dosage, c, alphaHat_t, betaHat_t, gamma_t, gammaSmall_t), best_haps_stuff_list is filled -- the same bytes as the Python
    mirror's call of the library; a second call re-uses the cached panel handle."""
    from quilt_amd.driver import thinned_grid_columns
    from quilt_amd.native import DevicePanel
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    K, G, T = panel.K, panel.nGrids, panel.nSNPs
    rng = np.random.default_rng(3)
    gl = np.asfortranarray(rng.random((2, T)) * 0.9 + 0.05)
    cols = thinned_grid_columns(G, 0.1)
    n_thin = int((cols >= 0).sum())
    dev = DevicePanel(panel)
    w = dict(alphaHat_t=np.zeros((K, G), order="F"), betaHat_t=np.zeros((K, G), order="F"), c=np.ones(G), gamma_t=np.zeros((K, G), order="F"),
             gammaSmall_t=np.zeros((K, n_thin), order="F"), dosage=np.zeros(T), best_haps_stuff_list=[None] * n_thin)
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, K_top_matches=5, return_gammaSmall_t=True,
                                    get_best_haps_from_thinned_sites=True, **w)
    dev.close()
    pa = _panel_args(R, panel)   # (the same R objects in both calls: the second one finds the cached handle by their identity)
    pa["transMatRate_t"] = R.real(panel.transMatRate_t)
    for rep in range(2):
        a = dict(gl=R.real(gl), arma_alphaHat_t=R.real(np.zeros((K, G))), eigen_alphaHat_t=R.real(np.zeros((1, 1))), betaHat_t=R.real(np.zeros((K, G))),
                 c=R.real(np.ones(G)), gamma_t=R.real(np.zeros((K, G))), gammaSmall_t=R.real(np.zeros((K, n_thin))),
                 best_haps_stuff_list=R.list([R.nil] * n_thin), dosage=R.real(np.zeros(T)),
                 use_eMatDH=R.logical([1]), use_eMatDH_special_symbols=R.logical([0]), gammaSmall_cols_to_get=R.integer(cols),
                 eMatDH_special_grid_which=R.integer(panel.eMatDH_special_grid_which), eMatDH_special_values_list=R.list([]),
                 K_top_matches=R.integer([5]), suppressOutput=R.integer([1]), min_emission_prob_normalization_threshold=R.real([1e-100]),
                 return_betaHat_t=R.logical([1]), return_dosage=R.logical([1]), return_gamma_t=R.logical([1]), return_gammaSmall_t=R.logical([1]),
                 get_best_haps_from_thinned_sites=R.logical([1]), is_version_2=R.logical([0]), is_version_3=R.logical([0]),
                 return_extra=R.logical([0]), always_normalize=R.logical([1]), use_eigen=R.logical([0]), normalize_emissions=R.logical([1]))
        a.update(pa)
        assert _call_by_name(R, "_QUILT_Rcpp_haploid_dosage_versus_refs", a) is None
        assert np.array_equal(R.value(a["dosage"]), w["dosage"]) and np.array_equal(R.value(a["c"]), w["c"])
        for r_name, p_name in (("arma_alphaHat_t", "alphaHat_t"), ("betaHat_t", "betaHat_t"), ("gamma_t", "gamma_t"), ("gammaSmall_t", "gammaSmall_t")):
            assert np.array_equal(R.value(a[r_name]), w[p_name]), r_name
        lists = R.value(a["best_haps_stuff_list"])
        for got, want in zip(lists, w["best_haps_stuff_list"]):
            assert np.array_equal(got["top_matches"], want["top_matches"]) and np.array_equal(got["top_matches_values"], want["top_matches_values"])
    assert w["dosage"].std() > 0 and all(len(x["top_matches"]) >= 5 for x in w["best_haps_stuff_list"])


@pytest.mark.parametrize("ff", [0.0, 0.2])
def test_gibbs_entry_63_arguments(R, panel, ff):
    """`.Call("_QUILT_rcpp_forwardBackwardGibbsNIPT", <63 arguments>)` as impute_one_sample makes it (functions.R:2385-2700): the shim
    draws the call's uniforms with unif_rand() in the reference's order (reads x sweeps, the first read, per block pass six + two
    vectors over the reads and -- diploid -- nGrids - 1 for the shard pass), between GetRNGstate / PutRNGstate; labels, H_class,
    hapProbs / genProbs and the state matrices R passed in equal the Python mirror's call with the same numbers.
    NIPT (round 6): after a block pass's relabelling the reference draws ONE uniform per read whose class leaves a choice
    (Rcpp::sample inside rcpp_sample_H_using_H_class) -- a count that depends on the pass's result.  The library asks the shim for
    those when the reference draws them (qa_gibbs_opts_t.draw_uniforms), so R's generator is consumed in the reference's order AND
    NUMBER: checked against the oracle run on the same stream (labels and classes identical, the count of uniforms exactly the
    oracle's)."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    G, T = panel.nGrids, panel.nSNPs
    s = make_synthetic_sample(panel, seed=850, n_reads=400, ff=ff)
    Rn, Ks, n_burn, n_samp, blocks = s.nReads, 64, 6, 1, np.array([2, 4], dtype=np.int32)
    n_its, nb = n_burn + n_samp, 2
    rng = np.random.default_rng(9)
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    start = rng.integers(1, 4 if ff else 3, size=Rn).astype(np.int32)
    per_block = 8 * Rn + (0 if ff else G - 1)
    U = rng.random(Rn * n_its + 1 + nb * (per_block + (Rn if ff else 0)))   # (NIPT: room for up to nReads re-draws per pass)
    ru = U[:Rn * n_its]
    first_read = min(int(U[Rn * n_its] * Rn), Rn - 1)
    base = Rn * n_its + 1
    n_expected = U.size
    if ff:
        # R's stream at a block iteration: 6 x nReads (runif_proposed), nReads (runif_block), nReads (runif_total), then the
        # re-draws.  The oracle takes [runif_block, re-draws] per pass as one stream; how many re-draws a pass makes is its own
        # result, so the stream is built pass by pass (a pass's count does not depend on what lies behind it).
        from oracle import oracle as O
        at, n_draw, stream = base, [], []
        for j in range(nb):
            stream_j = stream + [U[at + 6 * Rn: at + 7 * Rn], U[at + 8 * Rn: at + 9 * Rn]]   # this pass: block, then up to nReads draws
            want = O.forwardBackwardGibbsNIPT(panel, s, which, start, ru, first_read, np.zeros(max(nb * (G - 1), 1)), ff=ff,
                                              n_gibbs_burn_in_its=n_burn, n_gibbs_sample_its=n_samp, block_gibbs_iterations=blocks[:j + 1],
                                              runif_stream=np.concatenate(stream_j))
            n_draw.append(want["runif_stream_used"] - sum(Rn + n for n in n_draw) - Rn)
            stream += [U[at + 6 * Rn: at + 7 * Rn], U[at + 8 * Rn: at + 8 * Rn + n_draw[-1]]]
            at += 8 * Rn + n_draw[-1]
        assert all(0 < n < Rn for n in n_draw), n_draw   # (some reads' classes fix their label, some draw: the case worth testing)
        n_expected = at
    else:
        shard = np.concatenate([U[base + b * per_block + 8 * Rn: base + (b + 1) * per_block] for b in range(nb)])
        dev = DevicePanel(panel)
        want = rcpp_forwardBackwardGibbsNIPT(dev, s, which, start, ru, first_read, shard, ff=ff, n_gibbs_burn_in_its=n_burn,
                                             n_gibbs_sample_its=n_samp, block_gibbs_iterations=blocks, return_state=True)
        dev.close()
    assert not want["underflow_problem"]
    R.load_unif(U)
    z = lambda: R.real(np.zeros((Ks, G)))
    pl = dict(perform_block_gibbs=R.logical([1]), do_shard_block_gibbs=R.logical([0 if ff else 1]), return_hapProbs=R.logical([1]),
              return_genProbs=R.logical([1]), use_starting_read_labels=R.logical([1]), shard_check_every_pair=R.logical([1]),
              gibbs_initialize_iteratively=R.logical([0]), sample_is_diploid=R.logical([0 if ff else 1]), rescale_eMatRead_t=R.logical([1]))
    a = dict(sampleReads=R.sample_reads(s), transMatRate_tc_H=R.real(np.asarray(panel.transMatRate_t).reshape(2, G - 1, 1, order="F")), ff=R.real([ff]),
             alphaHat_t1=z(), betaHat_t1=z(), alphaHat_t2=z(), betaHat_t2=z(), eMatGrid_t1=z(), eMatGrid_t2=z(),
             which_haps_to_use=R.integer(which), wif0=R.integer(s.wif), L_grid=R.integer(panel.L_grid), param_list=R.named(pl),
             Jmax_local=R.integer([10000]), maxDifferenceBetweenReads=R.real([1e10]), generate_fb_snp_offsets=R.logical([0]),
             n_gibbs_starts=R.integer([1]), n_gibbs_sample_its=R.integer([n_samp]), n_gibbs_burn_in_its=R.integer([n_burn]),
             double_list_of_starting_read_labels=R.list([R.list([R.integer(start)])]), class_sum_cutoff=R.real([0.06]),
             shuffle_bin_radius=R.integer([5000]), block_gibbs_iterations=R.integer(blocks), block_gibbs_quantile_prob=R.real([0.95]))
    a.update(_panel_args(R, panel))
    out = _call_by_name(R, "_QUILT_rcpp_forwardBackwardGibbsNIPT", a)
    assert R.L.mini_r_unif_drawn() == n_expected and R.L.mini_r_rng_violations() == 0
    assert out["underflow_problem"][0] == 0
    assert np.array_equal(out["H"], want["H"]) and np.array_equal(out["H_class"], want["H_class"])
    assert np.array_equal(out["double_list_of_ending_read_labels"][0][0], want["H"])
    if ff:   # against the oracle (fp64 CPU, other summation order): to rounding
        np.testing.assert_allclose(out["hapProbs_t"], want["hapProbs_t"], rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(out["genProbsM_t"], want["genProbsM_t"], rtol=1e-9, atol=1e-14)
        for i, name in enumerate(("alphaHat_t1", "alphaHat_t2")):
            np.testing.assert_allclose(R.value(a[name]), want["alphaHat_t"][i], rtol=1e-9, atol=1e-300)
    else:    # against the Python mirror's call of the same library: bit for bit
        assert np.array_equal(out["hapProbs_t"], want["hapProbs_t"])
        assert np.array_equal(out["genProbsM_t"], want["genProbsM_t"]) and np.array_equal(out["genProbsF_t"], want["genProbsF_t"])
        for name in ("alphaHat_t1", "alphaHat_t2", "betaHat_t1", "betaHat_t2", "eMatGrid_t1", "eMatGrid_t2"):
            assert np.array_equal(R.value(a[name]), want[name]), name
    pit = out["per_it_likelihoods"]
    assert pit.shape == (n_its, 13) and np.array_equal(pit[:, 2], np.arange(1, n_its + 1)) and np.isfinite(pit[:, 7]).all()
    assert len(set(out["H"].tolist())) == (3 if ff else 2)


def _upload_panel_through_the_shim(R, panel):
    """A common-SNP call of the 63-argument entry: the shim uploads and caches the panel (what a QUILT() run has always done
    before the calls below are reached)."""
    from quilt_amd.synth import make_synthetic_sample
    G = panel.nGrids
    s = make_synthetic_sample(panel, seed=860, n_reads=60)
    Ks = 32
    R.load_unif(np.random.default_rng(1).random(s.nReads * 2 + 1))
    z = lambda: R.real(np.zeros((Ks, G)))
    pl = dict(perform_block_gibbs=R.logical([0]), return_hapProbs=R.logical([1]))
    a = dict(sampleReads=R.sample_reads(s), transMatRate_tc_H=R.real(np.asarray(panel.transMatRate_t).reshape(2, G - 1, 1, order="F")), ff=R.real([0.0]),
             alphaHat_t1=z(), betaHat_t1=z(), alphaHat_t2=z(), betaHat_t2=z(), eMatGrid_t1=z(), eMatGrid_t2=z(),
             which_haps_to_use=R.integer(np.arange(1, Ks + 1)), wif0=R.integer(s.wif), L_grid=R.integer(panel.L_grid), param_list=R.named(pl),
             Jmax_local=R.integer([10000]), maxDifferenceBetweenReads=R.real([1e10]), generate_fb_snp_offsets=R.logical([0]),
             n_gibbs_starts=R.integer([1]), n_gibbs_sample_its=R.integer([1]), n_gibbs_burn_in_its=R.integer([1]),
             double_list_of_starting_read_labels=R.list([R.list([R.integer(np.ones(s.nReads, dtype=np.int32))])]), class_sum_cutoff=R.real([0.06]),
             shuffle_bin_radius=R.integer([5000]), block_gibbs_iterations=R.integer([]), block_gibbs_quantile_prob=R.real([0.95]))
    pa = _panel_args(R, panel)
    a.update(pa)
    out = _call_by_name(R, "_QUILT_rcpp_forwardBackwardGibbsNIPT", a)
    assert out["underflow_problem"][0] == 0
    return pa


def test_read_likelihood_entry_15_arguments(R, panel):
    """`.Call("_QUILT_rcpp_make_eMatRead_t", <15 arguments>)` as calculate_eMatRead_t_vs_haplotypes makes it (functions.R:2975-3020):
    K = 2 sampled haplotypes, slice s of eHapsCurrent_tc, R's K x nReads matrix filled in place; what the entry does not cover
    is an R error with its reason."""
    import ctypes as C
    from quilt_amd.native import DevicePanel, check, lib, ptr
    from quilt_amd.synth import make_synthetic_sample
    from tests.mini_r import RError
    _upload_panel_through_the_shim(R, panel)
    T = panel.nSNPs
    s = make_synthetic_sample(panel, seed=861, n_reads=250)
    rng = np.random.default_rng(4)
    eh = rng.random((2, T, 2))   # K x nSNPs x S
    dev = DevicePanel(panel)
    want = np.zeros((s.nReads, 2))
    sl = np.ascontiguousarray(eh[:, :, 1].T)   # [SNP][K]: R's K x nSNPs slice, column-major
    read_off = np.array([0, s.nReads], dtype=np.int32)
    check(lib().qa_rcpp_make_eMatRead_t_nsnps(dev.handle, C.c_int32(T), C.c_int32(1), C.c_int32(2), ptr(sl), ptr(read_off),
                                              ptr(np.asarray(s.read_ptr, dtype=np.int32)), ptr(np.asarray(s.u, dtype=np.int32)),
                                              ptr(np.asarray(s.bq, dtype=np.int32)), C.c_double(1000.0), C.c_int32(1000), C.c_int32(0), ptr(want)))
    dev.close()
    em = R.real(np.zeros((2, s.nReads)))
    args = [em, R.sample_reads(s), R.real(eh), R.integer([1]), R.real([1000.0]), R.integer([1000]), R.real(np.zeros((1, 1))), R.real([0.0]),
            R.real([0.0]), R.integer([0]), R.integer([1]), R.string("N"), R.string("N"), R.logical([0]), R.logical([0])]
    assert R.dotcall("_QUILT_rcpp_make_eMatRead_t", *args) is None
    got = R.value(em)
    assert np.array_equal(got.T, want) and want.std() > 0
    args[13] = R.logical([1])
    with pytest.raises(RError, match="pseudo_haploid"):
        R.dotcall("_QUILT_rcpp_make_eMatRead_t", *args)
    args[13], args[0] = R.logical([0]), R.real(np.zeros((5, s.nReads)))
    with pytest.raises(RError, match="more than 3"):
        R.dotcall("_QUILT_rcpp_make_eMatRead_t", *args)


def test_gibbs_entry_63_arguments_rare_common(R, panel):
    """The all-SNP call of impute_final_gibbs_with_rare_common (rare_common.R:325-398): make_eMatRead_t_rare_common = TRUE in
    param_list, the all-SNP transition rates, snp_is_common and rare_per_hap_info as R holds them; the shim builds the all-SNP
    handle beside the cached panel; hapProbs_t comes back 3 x nSNPs_all and equals the Python mirror's call."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    pa = _upload_panel_through_the_shim(R, panel)
    rc = make_rare_common(panel, 6)
    smp = make_synthetic_sample_rare_common(panel, rc, 870, n_reads=300)[0]
    s = smp.all_snp
    Ga, Ta = rc.nGrids_all, rc.nSNPs_all
    Rn, Ks, n_burn, n_samp, blocks = s.nReads, 48, 4, 1, np.array([1], dtype=np.int32)
    n_its, nb = n_burn + n_samp, 1
    rng = np.random.default_rng(10)
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    start = rng.integers(1, 3, size=Rn).astype(np.int32)
    U = rng.random(Rn * n_its + 1 + nb * (8 * Rn + Ga - 1))
    base = Rn * n_its + 1
    dev = DevicePanel(panel)
    drc = DeviceRareCommon(dev, rc)
    want = rcpp_forwardBackwardGibbsNIPT(dev, s, which, start, U[:Rn * n_its], min(int(U[Rn * n_its] * Rn), Rn - 1), U[base + 8 * Rn:],
                                         n_gibbs_burn_in_its=n_burn, n_gibbs_sample_its=n_samp, block_gibbs_iterations=blocks,
                                         disable_read_category_usage=True, rare_common=drc)
    drc.close()
    dev.close()
    R.load_unif(U)
    z = lambda: R.real(np.zeros((1, 1)))
    pl = dict(perform_block_gibbs=R.logical([1]), do_shard_block_gibbs=R.logical([1]), return_hapProbs=R.logical([1]), return_genProbs=R.logical([1]),
              make_eMatRead_t_rare_common=R.logical([1]), disable_read_category_usage=R.logical([1]))
    rare = [R.integer(rc.rare_snp[rc.rare_ptr[k]:rc.rare_ptr[k + 1]]) for k in range(panel.K)]
    a = dict(sampleReads=R.sample_reads(s), transMatRate_tc_H=R.real(np.asarray(rc.transMatRate_t_all).reshape(2, Ga - 1, 1, order="F")), ff=R.real([0.0]),
             alphaHat_t1=z(), betaHat_t1=z(), alphaHat_t2=z(), betaHat_t2=z(), eMatGrid_t1=z(), eMatGrid_t2=z(),
             which_haps_to_use=R.integer(which), wif0=R.integer(s.wif), L_grid=R.integer(rc.L_grid_all), param_list=R.named(pl),
             Jmax_local=R.integer([10000]), maxDifferenceBetweenReads=R.real([1e10]), generate_fb_snp_offsets=R.logical([0]),
             n_gibbs_starts=R.integer([1]), n_gibbs_sample_its=R.integer([n_samp]), n_gibbs_burn_in_its=R.integer([n_burn]),
             double_list_of_starting_read_labels=R.list([R.list([R.integer(start)])]), class_sum_cutoff=R.real([0.06]),
             shuffle_bin_radius=R.integer([5000]), block_gibbs_iterations=R.integer(blocks), block_gibbs_quantile_prob=R.real([0.95]),
             rare_per_hap_info=R.list(rare), snp_is_common=R.logical(rc.snp_is_common))
    a.update(pa)
    out = _call_by_name(R, "_QUILT_rcpp_forwardBackwardGibbsNIPT", a)
    assert R.L.mini_r_unif_drawn() == U.size
    assert out["hapProbs_t"].shape == (3, Ta)
    assert np.array_equal(out["H"], want["H"]) and np.array_equal(out["hapProbs_t"], want["hapProbs_t"])
    assert np.array_equal(out["genProbsM_t"], want["genProbsM_t"])


@pytest.mark.parametrize("method", ["diploid", "nipt"])
def test_bam_range_call_from_paths_to_columns(R, tmp_path, method):
    """`.Call("qa_impute_bam_range", bam_files, sites, panel_objects, params, sample_index, n_handles)` -- what shim/quilt-amd.R's
    fast path calls: the BAM files are loaded, imputed, formatted and counted natively behind one call.
      * production mode: every sample's per_sample_vcf_col is the TEXT the Python BAM -> VCF path (quilt_amd.io.impute_bams_to_vcf:
        loader, Python driver on the same device, column writers) puts into its file, the file without reads comes back as not
        imputed, and the range's four count arrays equal SummaryCounts' bit for bit;
      * params$sum_order = 1 (the validation knob R reaches through QUILT_AMD_SUM_ORDER): the text equals the CPU path's.
    """
    from quilt_amd.driver import DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel
    from tests.oracle_backend import OracleBackend
    from tests.test_driver_host import _bam_to_vcf
    small = make_synthetic_panel(K=1000, nSNPs=640, seed=4916)
    ff = 0.2 if method == "nipt" else None
    prm = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, method=method)
    for d in ("gpu", "cpu"):
        (tmp_path / d).mkdir()
    dev = DevicePanel(small)
    dev.set_dosage_precision(64)
    rows_g, rec_g, _ = _bam_to_vcf(tmp_path / "gpu", small, HipBackend(dev), method=method, ff=ff, prm=prm)
    dev.close()
    rows_c, rec_c, _ = _bam_to_vcf(tmp_path / "cpu", small, OracleBackend(small), method=method, ff=ff, prm=prm)
    rng = np.random.default_rng(21)   # (the alleles test_driver_host._bam_to_vcf drew)
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(small.nSNPs)]
    ref, alt = [a for a, _ in alleles], [b for _, b in alleles]
    bams = [str(tmp_path / "gpu" / n) for n in ("s0.bam", "empty.bam", "s1.bam", "s2.bam")]
    sites = dict(chr=R.string("chr20"), L=R.integer(small.L), ref=R.strings(ref), alt=R.strings(alt),
                 grid=R.integer(np.arange(small.nSNPs) // 32), minimum_number_of_sample_reads=R.integer([2]),
                 output_gt_phased_genotypes=R.logical([1]), n_io_threads=R.integer([3]))
    more = dict(method=R.string("nipt"), ff=R.real([ff] * 4), shuffle_bin_radius=R.integer([prm.shuffle_bin_radius])) if ff else {}
    pan = lambda: R.panel_objects(small, **(dict(L_grid=R.real(np.asarray(small.L_grid, dtype=np.float64))) if ff else {}))
    assert R.arity("qa_impute_bam_range") == 6
    # impute_bams_to_vcf numbers the samples it keeps 0, 1, 2: the kept files' global indices here
    out = R.dotcall("qa_impute_bam_range", R.strings(bams), R.named(sites), pan(), _params(R, prm, **more), R.real([0.0, 99.0, 1.0, 2.0]),
                    R.integer([2]))
    assert out["sample_was_imputed"].tolist() == [1, 0, 1, 1] and out["n_reads"][1] == 0 and out["per_sample_vcf_col"][1] is None
    for i in (0, 2, 3):
        assert out["per_sample_vcf_col"][i] == [r[9 + i] for r in rows_g]
        assert np.array_equal(out["read_labels"][i], rec_g["results"][i].read_labels)
    for name in ("infoCount", "afCount", "hweCount", "alleleCount"):
        assert np.array_equal(out[name], getattr(rec_g["counts"], name)), name
    assert out["seconds"][3] >= out["seconds"][1] > 0
    val = R.dotcall("qa_impute_bam_range", R.strings(bams), R.named(sites), pan(), _params(R, prm, sum_order=R.integer([1]), **more),
                    R.real([0.0, 99.0, 1.0, 2.0]), R.integer([2]))
    for i in (0, 2, 3):
        assert val["per_sample_vcf_col"][i] == [r[9 + i] for r in rows_c]
    from tests.mini_r import RError
    with pytest.raises(RError, match="cannot load"):
        R.dotcall("qa_impute_bam_range", R.strings([str(tmp_path / "nope.bam")]), R.named(sites), pan(),
                  _params(R, prm, **(dict(more, ff=R.real([ff])) if ff else {})), R.real([0.0]), R.integer([1]))
    with pytest.raises(RError, match="sum_order"):
        R.dotcall("qa_impute_bam_range", R.strings(bams), R.named(sites), pan(), _params(R, prm, sum_order=R.integer([7]), **more),
                  R.real([0.0, 99.0, 1.0, 2.0]), R.integer([2]))


def test_bam_range_call_quilt2_defaults(R, panel, tmp_path):
    """`.Call("qa_impute_bam_range", ...)` in QUILT2's default mode (use_mspbwt + impute_rare_common): sites$L_all / ref_all / alt_all /
    grid_all name the all-SNP sites, panel_objects$rare_common the all-SNP side of the panel; every file is piled up twice by the
    native loader, the columns and the count arrays cover all SNPs -- equal to quilt_amd.impute.impute_bam_range on the device."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_bam_range
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    from tests import bamutil
    rc = make_rare_common(panel, 6)
    Ta = rc.nSNPs_all
    rng = np.random.default_rng(3)
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(Ta)]
    ref_all, alt_all = [a for a, _ in alleles], [b for _, b in alleles]
    common = np.flatnonzero(rc.snp_is_common == 1)
    ref, alt = [ref_all[i] for i in common], [alt_all[i] for i in common]
    grid_all = (np.arange(Ta) // 32).astype(np.int32)
    header = [("chr20", int(rc.L_all[-1]) + 1000)]
    files = []
    for i in range(3):
        s_all = make_synthetic_sample_rare_common(panel, rc, 880 + i, n_reads=300)[0].all_snp
        f = str(tmp_path / f"q{i}.bam")
        bamutil.write_bam(f, header, bamutil.sample_to_alignments(s_all, rc.L_all, ref_all, alt_all, rng))
        files.append(f)
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=78, impute_rare_common=True, use_mspbwt=True, mspbwt_nindices=2)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    drc = DeviceRareCommon(dev, rc)
    want = impute_bam_range([dev], files, "chr20", ref, alt, prm, sample_index=[5, 6, 7], drcs=[drc],
                            all_sites=(rc.L_all, ref_all, alt_all, grid_all), samples_per_launch_set=2, n_io_threads=2,
                            downsampleToCov=30, bqFilter=17)
    drc.close()
    dev.close()
    rare = [R.integer(rc.rare_snp[rc.rare_ptr[k]:rc.rare_ptr[k + 1]]) for k in range(panel.K)]
    rco = R.named(dict(snp_is_common=R.logical(rc.snp_is_common), rare_per_hap_info=R.list(rare), transMatRate_t=R.real(rc.transMatRate_t_all),
                       L_grid=R.integer(rc.L_grid_all)))
    sites = dict(chr=R.string("chr20"), L=R.integer(panel.L), ref=R.strings(ref), alt=R.strings(alt),
                 grid=R.integer(np.arange(panel.nSNPs) // 32), L_all=R.integer(rc.L_all), ref_all=R.strings(ref_all), alt_all=R.strings(alt_all),
                 grid_all=R.integer(grid_all), minimum_number_of_sample_reads=R.integer([2]), output_gt_phased_genotypes=R.logical([1]),
                 n_io_threads=R.integer([2]))
    out = R.dotcall("qa_impute_bam_range", R.strings(files), R.named(sites), R.panel_objects(panel, rare_common=rco),
                    _params(R, prm, use_mspbwt=R.logical([1]), mspbwtL=R.integer([prm.mspbwtL]), mspbwtM=R.integer([prm.mspbwtM]),
                            mspbwt_nindices=R.integer([2]), impute_rare_common=R.logical([1])),
                    R.real([5.0, 6.0, 7.0]), R.integer([1]))
    assert out["sample_was_imputed"].tolist() == [1, 1, 1]
    for i in range(3):
        assert len(out["per_sample_vcf_col"][i]) == Ta
        assert out["per_sample_vcf_col"][i] == want["columns"][i].tolist()
        assert np.array_equal(out["read_labels"][i], want["results"][i].read_labels)
    for name in ("infoCount", "afCount", "hweCount", "alleleCount"):
        assert out[name].shape[0] == Ta and np.array_equal(out[name], getattr(want["counts"], name)), name
