"""the per-grid dictionary compression of the packed panel on the device
(qa_panel_create_from_rhb) against the host restatement of STITCH::make_rhb_t_equality -- integer work, bit-exact --
and the hot path on a device-built panel against the oracle."""
import numpy as np
import pytest

from tests.util import label_gl, thin_cols

pytestmark = pytest.mark.gpu


def _check_tables(panel, nMaxDH):
    from quilt_amd.native import DevicePanel
    from quilt_amd.panel import make_rhb_t_equality
    ref = make_rhb_t_equality(panel.rhb_t, nMaxDH, panel.nSNPs, panel.ref_error)
    dev = DevicePanel.from_rhb(panel, nMaxDH=nMaxDH)
    hm, B, off, sk, sw = dev.export_tables()
    assert np.array_equal(hm, ref["hapMatcherR"])
    assert np.array_equal(B, ref["distinctHapsB"])
    # specials: per grid the haplotypes with code 0, ascending, with their words
    G = panel.nGrids
    for g in range(G):
        ks = np.nonzero(ref["hapMatcherR"][:, g] == 0)[0]
        assert np.array_equal(sk[off[g]:off[g + 1]], ks)
        assert np.array_equal(sw[off[g]:off[g + 1]], panel.rhb_t[ks, g])
    dev.close()


@pytest.mark.parametrize("nMaxDH", [255, 40, 3])
def test_device_tables_equal_host_restatement(ragged_panel, nMaxDH):
    _check_tables(ragged_panel, nMaxDH)


def test_grid_with_more_distinct_words_than_the_device_table():
    """> 4096 distinct words in one grid: that grid is ranked on the host, same rule; zero words and ties included."""
    from quilt_amd.synth import make_synthetic_panel
    panel = make_synthetic_panel(K=9000, nSNPs=96, seed=5, nMaxDH=255)
    rng = np.random.default_rng(0)
    rhb = np.array(panel.rhb_t, order="F", copy=True)
    rhb[:, 1] = rng.integers(-2**31, 2**31 - 1, size=panel.K, dtype=np.int64).astype(np.int32)   # ~9000 distinct
    rhb[::7, 1] = 0                                                                               # a frequent zero word
    rhb[1::7, 2] = 0
    panel.rhb_t = rhb
    _check_tables(panel, 255)


def test_fullpass_on_device_built_panel(medium_panel):
    from oracle import oracle as O
    from quilt_amd.native import DevicePanel
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    dev = DevicePanel.from_rhb(panel)
    sample = make_synthetic_sample(panel, seed=31, n_reads=800)
    cols = thin_cols(panel.nGrids, every=10)
    gl = label_gl(panel, sample, 1, O)
    ref = O.haploid_dosage_versus_refs(panel, gl, cols, get_best_haps_from_thinned_sites=True, always_normalize=True)
    n_thin = int((cols >= 0).sum())
    out = dict(alphaHat_t=np.zeros((panel.K, panel.nGrids), order="F"), c=np.ones(panel.nGrids), dosage=np.zeros(panel.nSNPs),
               best_haps_stuff_list=[None] * n_thin)
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, return_dosage=True,
                                    get_best_haps_from_thinned_sites=True, return_gamma_t=False, return_betaHat_t=False, **out)
    assert np.abs(out["dosage"] - ref["dosage"]).max() <= 2e-4
    for e, (idx, val) in zip(out["best_haps_stuff_list"], ref["best_haps"]):
        assert np.array_equal(e["top_matches"], idx)
    dev.close()


def test_special_symbols_mode_matches_host_created_panel(ragged_panel):
    """use_eMatDH_special_symbols: special words decoded as the reference's clamped search over the special matrix would
    (quirks included), identically whether the tables came from the host or were built on the device."""
    from quilt_amd.native import DevicePanel
    panel = ragged_panel
    a = DevicePanel(panel, use_eMatDH_special_symbols=True)
    b = DevicePanel.from_rhb(panel, use_eMatDH_special_symbols=True)
    ta, tb = a.export_tables(), b.export_tables()
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y)
    a.close()
    b.close()


def test_make_rhb_t_from_rhi_t():
    """STITCH::make_rhb_t_from_rhi_t on the device (test-drivers.R:393-394): K x nSNPs alleles -> K x ceil(nSNPs / 32) words,
    bit b of word g = allele at SNP 32 g + b; ragged last grid; non-zero counts as 1."""
    import ctypes as C
    from quilt_amd.native import check, lib, ptr
    rng = np.random.default_rng(8)
    for K, T in ((1000, 500), (37, 1), (513, 64), (2000, 3333)):
        rhi = np.asfortranarray(rng.integers(0, 2, size=(K, T)).astype(np.int32))
        if T > 3:
            rhi[:, 3] *= 7                     # any non-zero entry is an alternate allele
        G = (T + 31) // 32
        out = np.zeros((K, G), dtype=np.int32, order="F")
        lib().qa_make_rhb_t_from_rhi_t.restype = C.c_int
        check(lib().qa_make_rhb_t_from_rhi_t(ptr(rhi), C.c_int32(K), C.c_int32(T), ptr(out)))
        pad = np.zeros((K, 32 * G), dtype=np.uint64)
        pad[:, :T] = rhi != 0
        exp = (pad.reshape(K, G, 32) << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32).view(np.int32)
        assert np.array_equal(out, exp)
