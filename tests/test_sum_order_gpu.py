"""The validation mode of the full-panel passes (qa_panel_set_sum_order(panel, 1), csrc/fullpass_ref.hip): every K-wide sum
formed in the reference's order (QUILT/src/reference-single.cpp:1002-1075, :1899-1955, :2349-2353, :2083-2139).

What it proves.  The production kernels form those sums as block-wide trees; their last bits differ from a sequential sum's,
and on panels with exactly tied haplotypes (duplicates, or values absorbed into the recombination term) the last bits decide
which of the tied haplotypes make a best-haplotype list.  In validation mode the device equals the CPU restatement
(oracle/fullpass.c, a sequential sum like the reference's) BIT FOR BIT in c, alphaHat_t, betaHat_t, gamma_t, dosage and the
lists -- so the order of the sums is the ONLY difference between the production mode and the CPU path.  Tolerance: none
(array_equal), fp64.
"""
import numpy as np
import pytest

from tests.util import label_gl, thin_cols

pytestmark = pytest.mark.gpu


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
               best_haps_stuff_list=[None] * n_thin, gamma_t=np.zeros((K, G), order="F"),
               betaHat_t=np.zeros((K, G), order="F"))
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, **out, **kw)
    return out


def _lists_equal(got, ref):
    assert len(got) == len(ref)
    for g, (oi, ov) in zip(got, ref):
        assert np.array_equal(g["top_matches"], oi)
        assert np.array_equal(g["top_matches_values"], ov)


@pytest.mark.parametrize("panel_name,symbols", [("small_panel", False), ("small_panel", True),
                                                ("ragged_panel", False), ("ragged_panel", True),
                                                ("medium_panel", False)])
@pytest.mark.parametrize("always_normalize", [False, True])
def test_every_output_equals_the_oracle_bit_for_bit(request, oracle, panel_name, symbols, always_normalize):
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel, use_eMatDH_special_symbols=symbols)
    dev.set_sum_order(True)
    sample = make_synthetic_sample(panel, seed=1001, n_reads=max(40, panel.nSNPs // 4))
    cols = thin_cols(panel.nGrids)
    for label in (1, 2):
        gl = label_gl(panel, sample, label, oracle)
        ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, return_gamma_t=True, return_betaHat_t=True,
                                                always_normalize=always_normalize, get_best_haps_from_thinned_sites=True,
                                                use_eMatDH_special_symbols=symbols)
        got = _run_gpu(dev, gl, cols, return_dosage=True, return_gamma_t=True, return_betaHat_t=True,
                       get_best_haps_from_thinned_sites=True, always_normalize=always_normalize)
        assert np.array_equal(got["c"], ref["c"])
        assert np.array_equal(got["alphaHat_t"], ref["alphaHat_t"])
        assert np.array_equal(got["betaHat_t"], ref["betaHat_t"])
        assert np.array_equal(got["gamma_t"], ref["gamma_t"])
        assert np.array_equal(got["dosage"], ref["dosage"])
        _lists_equal(got["best_haps_stuff_list"], ref["best_haps"])
    dev.close()


def test_thin_pass_and_label_without_reads(small_panel, oracle):
    """Lists only (the driver's ranking passes); and a label without reads: every haplotype ties, every grid takes the
    reference's no-variant shortcut (:1078-1088, backward :1957-1967 -- grid 1 included, which only the forward pass forces)."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = small_panel
    dev = DevicePanel(panel)
    dev.set_sum_order(True)
    cols = thin_cols(panel.nGrids, every=3)
    sample = make_synthetic_sample(panel, seed=5, n_reads=150)
    for gl in (label_gl(panel, sample, 1, oracle), np.ones((2, panel.nSNPs), order="F")):
        ref = oracle.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=False, always_normalize=False,
                                                get_best_haps_from_thinned_sites=True)
        got = _run_gpu(dev, gl, cols, return_dosage=False, return_gamma_t=False, return_betaHat_t=False,
                       get_best_haps_from_thinned_sites=True, always_normalize=False)
        _lists_equal(got["best_haps_stuff_list"], ref["best_haps"])
        assert np.array_equal(got["c"], ref["c"])
        for g in np.nonzero(cols >= 0)[0]:
            assert np.array_equal(got["alphaHat_t"][:, g], ref["alphaHat_t"][:, g])
    dev.close()
