"""GPU parity: HIP full-panel forward/backward (through the C ABI) vs the fp64 CPU oracle.

Tolerances (stated).  Dosage / gamma / alpha / beta / c come from fp32 device state: dosage |diff| <= 2e-4 and
r2 >= 0.99999; colSums(gamma) = 1 +- 1e-4; sum(log c) relative 1e-5.  The best-haplotype lists come from the fp64-state
ranking passes and must be identical to the oracle's (same haplotypes, values to 1e-9); in the optional fp32 ranking
mode they agree up to rounding at the threshold (values relative 2e-4).
"""
import numpy as np
import pytest

from tests.util import check_best_haps, label_gl, r2, thin_cols

pytestmark = pytest.mark.gpu

DOSAGE_ATOL = 2e-4


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _run_gpu(dev, gl, cols, **kw):
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    P = dev.panel
    K, G, T = P.K, P.nGrids, P.nSNPs
    n_thin = int((cols >= 0).sum())
    out = dict(alphaHat_t=np.zeros((K, G), order="F"), c=np.ones(G), dosage=np.zeros(T),
               best_haps_stuff_list=[None] * n_thin)
    if kw.get("return_gamma_t"):
        out["gamma_t"] = np.zeros((K, G), order="F")
    if kw.get("return_betaHat_t"):
        out["betaHat_t"] = np.zeros((K, G), order="F")
    if kw.get("return_gammaSmall_t"):
        out["gammaSmall_t"] = np.zeros((K, n_thin), order="F")
    kw.setdefault("return_gamma_t", False)
    kw.setdefault("return_betaHat_t", False)
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, **out, **kw)
    return out


@pytest.mark.parametrize("panel_name,symbols", [("small_panel", False), ("small_panel", True),
                                                ("ragged_panel", False), ("ragged_panel", True),
                                                ("medium_panel", False)])
def test_dosage_pass_matches_oracle(request, oracle, panel_name, symbols):
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel, use_eMatDH_special_symbols=symbols)
    sample = make_synthetic_sample(panel, seed=1001, n_reads=max(40, panel.nSNPs // 4))
    cols = thin_cols(panel.nGrids)
    for label in (1, 2):
        gl = label_gl(panel, sample, label, oracle)
        ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, return_gamma_t=True, return_betaHat_t=True,
                                                always_normalize=True, get_best_haps_from_thinned_sites=True,
                                                use_eMatDH_special_symbols=symbols)
        got = _run_gpu(dev, gl, cols, return_dosage=True, return_gamma_t=True, return_betaHat_t=True,
                       get_best_haps_from_thinned_sites=True)
        assert np.abs(got["dosage"] - ref["dosage"]).max() <= DOSAGE_ATOL
        assert r2(got["dosage"], ref["dosage"]) >= 0.99999
        np.testing.assert_allclose(got["gamma_t"].sum(axis=0), 1.0, atol=1e-4)
        np.testing.assert_allclose(got["gamma_t"], ref["gamma_t"], atol=2e-5, rtol=2e-3)
        np.testing.assert_allclose(got["alphaHat_t"], ref["alphaHat_t"], atol=1e-6, rtol=2e-3)
        np.testing.assert_allclose(got["betaHat_t"], ref["betaHat_t"], atol=1e-30, rtol=2e-3)
        np.testing.assert_allclose(np.log(got["c"]).sum(), np.log(ref["c"]).sum(), rtol=1e-5)
        np.testing.assert_allclose(got["c"], ref["c"], rtol=1e-4)
        check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    dev.close()


def test_thin_pass_matches_oracle(small_panel, oracle):
    """return_dosage = FALSE: only best-haps at the thinned grids (functions.R:748)."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = small_panel
    dev = DevicePanel(panel)
    sample = make_synthetic_sample(panel, seed=5, n_reads=150)
    cols = thin_cols(panel.nGrids, every=3)
    gl = label_gl(panel, sample, 1, oracle)
    ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=False,
                                            get_best_haps_from_thinned_sites=True)
    got = _run_gpu(dev, gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True)
    check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    # alpha only at column 0 and the thinned columns (reference-single.cpp:2264-2268); normalised
    for g in np.nonzero(cols >= 0)[0]:
        a = got["alphaHat_t"][:, g]
        b = ref["alphaHat_t"][:, g]
        np.testing.assert_allclose(a, b / b.sum(), atol=1e-6, rtol=2e-3)
    assert np.all(got["alphaHat_t"][:, np.nonzero(cols < 0)[0][1:]] == 0)


def test_label_without_reads_returns_all_haplotypes(small_panel, oracle):
    """gl == 1 everywhere: all gammas tie, every haplotype is a top match (reference-single.cpp:150-151)."""
    from quilt_amd.native import DevicePanel
    panel = small_panel
    dev = DevicePanel(panel)
    gl = np.ones((2, panel.nSNPs), order="F")
    cols = thin_cols(panel.nGrids, every=5)
    got = _run_gpu(dev, gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, get_best_haps_from_thinned_sites=True)
    assert np.abs(got["dosage"] - ref["dosage"]).max() <= DOSAGE_ATOL
    for e in got["best_haps_stuff_list"]:
        assert len(e["top_matches"]) == panel.K


def test_batch_matches_single(medium_panel, oracle):
    """qa_fullpass_batch: mixed dosage / thin passes in one launch set == one-at-a-time results."""
    import ctypes as C
    from quilt_amd.native import DevicePanel, check, lib, ptr
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    dev = DevicePanel(panel)
    cols = thin_cols(panel.nGrids, every=10)
    n_thin = int((cols >= 0).sum())
    gls, want = [], []
    for i in range(6):
        s = make_synthetic_sample(panel, seed=100 + i, n_reads=1000)
        gls.append(label_gl(panel, s, 1 + i % 2, oracle))
        want.append(i % 3 != 0)
    gl = np.ascontiguousarray(np.stack([np.ascontiguousarray(g.T) for g in gls]))  # [P][T][2]
    wd = np.array(want, dtype=np.int32)
    dosage = np.zeros((len(gls), panel.nSNPs))
    bptr = np.zeros(len(gls) * n_thin + 1, dtype=np.int32)
    cap = len(gls) * n_thin * 64
    bidx = np.zeros(cap, dtype=np.int32)
    bval = np.zeros(cap)
    check(lib().qa_fullpass_batch(dev.handle, C.c_int32(len(gls)), ptr(gl), ptr(wd), ptr(cols), C.c_int32(5),
                                  ptr(dosage), ptr(bptr), ptr(bidx), ptr(bval), C.c_int64(cap)))
    for i, g in enumerate(gls):
        ref = oracle.haploid_dosage_versus_refs(panel, g, cols, return_dosage=bool(want[i]),
                                                get_best_haps_from_thinned_sites=True)
        if want[i]:
            assert np.abs(dosage[i] - ref["dosage"]).max() <= DOSAGE_ATOL
        got = [dict(top_matches=bidx[bptr[i * n_thin + j]:bptr[i * n_thin + j + 1]],
                    top_matches_values=bval[bptr[i * n_thin + j]:bptr[i * n_thin + j + 1]]) for j in range(n_thin)]
        check_best_haps(got, ref["best_haps"])


def test_fp32_ranking_mode(medium_panel, oracle):
    """qa_panel_set_ranking_precision(32): lists from the fp32-state pass, equal up to rounding at the threshold."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    dev = DevicePanel(panel)
    dev.set_ranking_precision(32)
    sample = make_synthetic_sample(panel, seed=77, n_reads=800)
    cols = thin_cols(panel.nGrids, every=10)
    gl = label_gl(panel, sample, 1, oracle)
    ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, get_best_haps_from_thinned_sites=True)
    got = _run_gpu(dev, gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    assert np.abs(got["dosage"] - ref["dosage"]).max() <= DOSAGE_ATOL
    check_best_haps(got["best_haps_stuff_list"], ref["best_haps"], exact=False)
    dev.close()


@pytest.mark.parametrize("K", [50000, 30011])
def test_production_k_geometry(oracle, K):
    """K = 50 000 (BASELINE.json configs[1]) on a short region: the launch geometries of the real workload
    (fp32: 448 threads x 7 chunks; fp64 ranking: 256 threads x 10 register + 3 LDS chunks) against the oracle;
    K = 30 011 adds a chunk that straddles K."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=K, nSNPs=640, seed=4242, nMaxDH=255)
    dev = DevicePanel(panel)
    sample = make_synthetic_sample(panel, seed=9, n_reads=200)
    cols = thin_cols(panel.nGrids, every=4)
    for label in (1, 2):
        gl = label_gl(panel, sample, label, oracle)
        ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, get_best_haps_from_thinned_sites=True, always_normalize=True)
        got = _run_gpu(dev, gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
        assert np.abs(got["dosage"] - ref["dosage"]).max() <= DOSAGE_ATOL
        np.testing.assert_allclose(got["c"], ref["c"], rtol=1e-4)
        check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    dev.close()


def _dosage_only(dev, gl, cols=None, lists=False):
    """The call a dosage round makes: dosage and c, no K x G matrices (with fp64 dosage precision: k_fwd64 + k_bwd64d)."""
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    P = dev.panel
    out = dict(c=np.ones(P.nGrids), dosage=np.zeros(P.nSNPs))
    if lists:
        out["best_haps_stuff_list"] = [None] * int((cols >= 0).sum())
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, return_dosage=True, return_gamma_t=False,
                                    return_betaHat_t=False, get_best_haps_from_thinned_sites=lists, always_normalize=False,
                                    **out)
    return out


@pytest.mark.parametrize("panel_name,symbols", [("small_panel", False), ("small_panel", True), ("ragged_panel", False),
                                                ("ragged_panel", True), ("medium_panel", False)])
def test_fp64_dosage_kernels_match_oracle(request, oracle, panel_name, symbols):
    """qa_panel_set_dosage_precision(64): the fp64-state dosage kernels follow the reference's arithmetic (lazy
    normalisation, reference-single.cpp:878-1131, :1781-2179) -- dosage to 1e-10, c elementwise to 1e-12, and the lists of
    the ranking pass beside them identical; grids with special haplotypes included (small / ragged panels)."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel, use_eMatDH_special_symbols=symbols)
    dev.set_dosage_precision(64)
    sample = make_synthetic_sample(panel, seed=1001, n_reads=max(40, panel.nSNPs // 4))
    cols = thin_cols(panel.nGrids)
    for label in (1, 2):
        gl = label_gl(panel, sample, label, oracle)
        ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, always_normalize=False, get_best_haps_from_thinned_sites=True,
                                                use_eMatDH_special_symbols=symbols)
        got = _dosage_only(dev, gl, cols, lists=True)
        assert np.abs(got["dosage"] - ref["dosage"]).max() <= 1e-10
        np.testing.assert_allclose(got["c"], ref["c"], rtol=1e-12)
        check_best_haps(got["best_haps_stuff_list"], ref["best_haps"])
    # a label without reads: gl == 1 everywhere, every grid takes the no-variant shortcut
    gl = np.ones((2, panel.nSNPs), order="F")
    ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, always_normalize=False)
    got = _dosage_only(dev, gl)
    assert np.abs(got["dosage"] - ref["dosage"]).max() <= 1e-10
    dev.close()


@pytest.mark.parametrize("K", [50000, 30011, 8192, 8193])
def test_fp64_dosage_kernels_production_geometry(oracle, K):
    """The fp64 dosage kernels at the launch geometries of the real workload (K = 50 000: four chunk rows in registers, two
    and a fraction in LDS), a K whose last chunk straddles K, and K at / just past a whole chunk row."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=K, nSNPs=640, seed=4242, nMaxDH=255)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    sample = make_synthetic_sample(panel, seed=9, n_reads=200)
    for label in (1, 2):
        gl = label_gl(panel, sample, label, oracle)
        ref = oracle.haploid_dosage_versus_refs(panel, gl, None, always_normalize=False)
        got = _dosage_only(dev, gl)
        assert np.abs(got["dosage"] - ref["dosage"]).max() <= 1e-10
        np.testing.assert_allclose(got["c"], ref["c"], rtol=1e-12)
    dev.close()


def test_zeroed_options_take_the_references_threshold(oracle):
    """A C caller that zero-initialises qa_fullpass_opts_t hands over min_emission_prob_normalization_threshold = 0, which read
    literally would mean "never renormalise between grid 0 and the last grid" in the lazily normalised fp64 passes (alpha
    underflows, NaN dosages over a long region).  A threshold that is not positive is read as the reference's default, 1e-100
    (reference-single.cpp:2216): the same bytes as a call that passes 1e-100, on a region long enough for the schedule to
    renormalise in between."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=2000, nSNPs=6400, seed=31)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    sample = make_synthetic_sample(panel, seed=9, n_reads=4000)
    gl = label_gl(panel, sample, 1, oracle)
    cols = thin_cols(panel.nGrids)
    res = {}
    for thr in (1e-100, 0.0, -1.0):
        out = dict(dosage=np.zeros(panel.nSNPs), c=np.ones(panel.nGrids), best_haps_stuff_list=[None] * int((cols >= 0).sum()))
        Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, return_dosage=True, return_gamma_t=False,
                                        return_betaHat_t=False, get_best_haps_from_thinned_sites=True, always_normalize=False,
                                        min_emission_prob_normalization_threshold=thr, **out)
        res[thr] = out
    sigma = panel.transMatRate_t[0]
    inner = res[1e-100]["c"][1:-1]
    assert (inner != 1 / sigma[:-1]).any(), "the region must be long enough for a renormalisation between the ends"
    for thr in (0.0, -1.0):
        assert np.isfinite(res[thr]["dosage"]).all()
        assert np.array_equal(res[thr]["dosage"], res[1e-100]["dosage"])
        assert np.array_equal(res[thr]["c"], res[1e-100]["c"])
    ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, always_normalize=False)
    assert np.abs(res[0.0]["dosage"] - ref["dosage"]).max() <= 1e-10
    dev.close()


def test_fp64_dosage_batch(medium_panel, oracle):
    """qa_fullpass_batch with fp64 dosage precision: dosage passes (k_fwd64 + k_bwd64d) and ranking passes in one call."""
    import ctypes as C
    from quilt_amd.native import DevicePanel, check, lib, ptr
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    cols = thin_cols(panel.nGrids, every=10)
    n_thin = int((cols >= 0).sum())
    gls, want = [], []
    for i in range(7):
        s = make_synthetic_sample(panel, seed=200 + i, n_reads=1000)
        gls.append(label_gl(panel, s, 1 + i % 2, oracle))
        want.append(i % 3 != 0)
    gl = np.ascontiguousarray(np.stack([np.ascontiguousarray(g.T) for g in gls]))
    wd = np.array(want, dtype=np.int32)
    dosage = np.zeros((len(gls), panel.nSNPs))
    bptr = np.zeros(len(gls) * n_thin + 1, dtype=np.int32)
    cap = len(gls) * n_thin * 64
    bidx = np.zeros(cap, dtype=np.int32)
    bval = np.zeros(cap)
    check(lib().qa_fullpass_batch(dev.handle, C.c_int32(len(gls)), ptr(gl), ptr(wd), ptr(cols), C.c_int32(5),
                                  ptr(dosage), ptr(bptr), ptr(bidx), ptr(bval), C.c_int64(cap)))
    for i, g in enumerate(gls):
        ref = oracle.haploid_dosage_versus_refs(panel, g, cols, return_dosage=bool(want[i]), get_best_haps_from_thinned_sites=True)
        if want[i]:
            assert np.abs(dosage[i] - ref["dosage"]).max() <= 1e-10
        got = [dict(top_matches=bidx[bptr[i * n_thin + j]:bptr[i * n_thin + j + 1]],
                    top_matches_values=bval[bptr[i * n_thin + j]:bptr[i * n_thin + j + 1]]) for j in range(n_thin)]
        check_best_haps(got, ref["best_haps"])
    dev.close()


@pytest.mark.parametrize("K", [57344, 57345, 65536, 70001, 131072])
def test_panels_beyond_the_on_chip_capacity(oracle, K):
    """K above 57 344 haplotypes (seven chunk rows of 8 192: what one compute unit's registers and LDS hold): the chunk rows past
    the seventh stream their state through HBM (PassParams::spill).  The reference has no limit on K
    (reference-single.cpp:878-1131, :1781-2179); HRC is 64 976 haplotypes.  Dosage passes (fp64 state; also what a handle with
    fp32 dosage passes falls back to beyond the fp32 kernels' 98 304) and ranking passes through the batched call: dosage to
    1e-9, best-haplotype lists identical, against the oracle.  57 344 / 57 345: either side of the boundary."""
    import ctypes as C
    from quilt_amd.native import DevicePanel, check, lib, ptr
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=K, nSNPs=640, seed=515, nMaxDH=255)
    s = make_synthetic_sample(panel, seed=3, n_reads=300)
    cols = thin_cols(panel.nGrids, every=4)
    n_thin = int((cols >= 0).sum())
    for fp64_dosage in ((True, False) if K in (65536, 131072) else (True,)):
        dev = DevicePanel(panel)
        if fp64_dosage:
            dev.set_dosage_precision(64)
        gls = [label_gl(panel, s, 1, oracle), label_gl(panel, s, 2, oracle)]
        want = [True, False]
        gl = np.ascontiguousarray(np.stack([np.ascontiguousarray(g.T) for g in gls]))
        wd = np.array(want, dtype=np.int32)
        dosage = np.zeros((2, panel.nSNPs))
        bptr = np.zeros(2 * n_thin + 1, dtype=np.int32)
        cap = 2 * n_thin * 64
        bidx, bval = np.zeros(cap, dtype=np.int32), np.zeros(cap)
        check(lib().qa_fullpass_batch(dev.handle, C.c_int32(2), ptr(gl), ptr(wd), ptr(cols), C.c_int32(5), ptr(dosage), ptr(bptr),
                                      ptr(bidx), ptr(bval), C.c_int64(cap)))
        for i, g in enumerate(gls):
            ref = oracle.haploid_dosage_versus_refs(panel, g, cols, return_dosage=bool(want[i]), get_best_haps_from_thinned_sites=True)
            if want[i]:
                tol = 1e-9 if (fp64_dosage or K > 98304) else 2e-4
                assert np.abs(dosage[i] - ref["dosage"]).max() <= tol
            got = [dict(top_matches=bidx[bptr[i * n_thin + j]:bptr[i * n_thin + j + 1]],
                        top_matches_values=bval[bptr[i * n_thin + j]:bptr[i * n_thin + j + 1]]) for j in range(n_thin)]
            check_best_haps(got, ref["best_haps"])
        dev.close()


def test_k_limit_that_remains_is_a_documented_status(oracle):
    """What is still limited above K = 57 344: the K x nGrids outputs of the single-pass entry point (alphaHat_t / betaHat_t /
    gamma_t come from kernels that hold the state on chip) -- QA_ERR_UNSUPPORTED with a text that says so (include/quilt_amd.h);
    dosage, c and the lists of the same entry point work."""
    from quilt_amd.native import DevicePanel, QuiltAmdError
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=65536, nSNPs=320, seed=99, nMaxDH=255)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    s = make_synthetic_sample(panel, seed=4, n_reads=150)
    gl = label_gl(panel, s, 1, oracle)
    dosage = np.zeros(panel.nSNPs)
    c = np.zeros(panel.nGrids)
    Rcpp_haploid_dosage_versus_refs(dev, gl, dosage=dosage, c=c, return_betaHat_t=False, return_gamma_t=False, always_normalize=False)
    ref = oracle.haploid_dosage_versus_refs(panel, gl, None, always_normalize=False)
    assert np.abs(dosage - ref["dosage"]).max() <= 1e-9
    np.testing.assert_allclose(c, ref["c"], rtol=1e-12)
    with pytest.raises(QuiltAmdError, match="57 344") as e:
        Rcpp_haploid_dosage_versus_refs(dev, gl, dosage=dosage, gamma_t=np.zeros((panel.K, panel.nGrids), order="F"),
                                        return_betaHat_t=False, return_gamma_t=True)
    assert e.value.status == -3   # QA_ERR_UNSUPPORTED
    dev.close()
