"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def label_gl(panel, sample, label, oracle):
    """gl (2 x T) of the reads carrying ``label`` (functions.R:2016-2024), via the oracle."""
    per_base = np.repeat(sample.truth_label, np.diff(sample.read_ptr))
    sel = (per_base == label) & (sample.bq != 0)
    return oracle.make_gl_from_u_bq(sample.u[sel], sample.bq[sel], panel.nSNPs)


def thin_cols(nGrids, every=4, start=1):
    cols = np.full(nGrids, -1, dtype=np.int32)
    w = np.arange(start, nGrids, every)
    cols[w] = np.arange(len(w), dtype=np.int32)
    return cols


def r2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.corrcoef(a, b)[0, 1] ** 2)


def check_best_haps(gpu_list, oracle_list, rtol=2e-4, exact=True):
    """Top-match lists of the HIP path vs the oracle's.

    exact (the default fp64 ranking passes): the same haplotypes in the same (ascending) order, values to 1e-9.
    Otherwise (fp32 ranking mode): lists agree up to fp32 rounding of gamma around the threshold -- every haplotype the
    oracle reports clearly above its threshold must be reported by the GPU path and vice versa; values agree to rtol.
    """
    assert len(gpu_list) == len(oracle_list)
    for got, (oi, ov) in zip(gpu_list, oracle_list):
        gi, gv = got["top_matches"], got["top_matches_values"]
        assert np.all(np.diff(gi) > 0), "top_matches must be ascending in k"
        if exact:
            assert np.array_equal(gi, oi)
            np.testing.assert_allclose(gv, ov, rtol=1e-9, atol=1e-300)
            continue
        thr = ov.min()
        sure = oi[ov > thr * (1 + 10 * rtol)]
        assert set(sure.tolist()) <= set(gi.tolist())
        gthr = gv.min()
        gsure = gi[gv > gthr * (1 + 10 * rtol)]
        assert set(gsure.tolist()) <= set(oi.tolist())
        common, ia, ib = np.intersect1d(gi, oi, return_indices=True)
        assert len(common) >= min(len(gi), len(oi)) - 2
        np.testing.assert_allclose(gv[ia], ov[ib], rtol=rtol, atol=1e-12)
