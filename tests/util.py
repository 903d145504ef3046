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


def panel_from_rhb(rhb_t, transMatRate_t, nSNPs, nMaxDH, ref_error):
    """A Panel (reference layouts) from the packed haplotypes of a fixture: per-grid dictionary compression restated
    from STITCH::make_rhb_t_equality (quilt_amd.panel)."""
    from quilt_amd.panel import Panel, make_rhb_t_equality
    rhb_t = np.asfortranarray(rhb_t, dtype=np.int32)
    K, G = rhb_t.shape
    t = make_rhb_t_equality(rhb_t, int(nMaxDH), int(nSNPs), float(ref_error), use_hapMatcherR=True)
    return Panel(K=K, nSNPs=int(nSNPs), nGrids=G, nMaxDH=t["nMaxDH"], ref_error=float(ref_error), rhb_t=rhb_t,
                 hapMatcher=t["hapMatcher"], hapMatcherR=t["hapMatcherR"], distinctHapsB=t["distinctHapsB"],
                 distinctHapsIE=t["distinctHapsIE"], eMatDH_special_grid_which=t["eMatDH_special_grid_which"],
                 eMatDH_special_values_list=t["eMatDH_special_values_list"], eMatDH_special_matrix=t["eMatDH_special_matrix"],
                 eMatDH_special_matrix_helper=t["eMatDH_special_matrix_helper"],
                 transMatRate_t=np.asfortranarray(transMatRate_t, dtype=np.float64))


def sample_from_arrays(read_ptr, u, bq, wif):
    from quilt_amd.synth import SampleReads
    return SampleReads(read_ptr=np.asarray(read_ptr, dtype=np.int32), u=np.asarray(u, dtype=np.int32),
                       bq=np.asarray(bq, dtype=np.int32), wif=np.asarray(wif, dtype=np.int32))


def underflowing_sample(panel, seed=78, n_reads=300, per_type=115, n_snps=12):
    """A sample whose small-panel forward underflows at maxDifferenceBetweenReads = 1e10: on one grid, three read types that
    contradict each other pairwise over n_snps SNPs (all ref / all alt / alternating) at base quality 40.  Two labels cannot
    separate three types, so one label always holds >= per_type/2 reads no haplotype explains: every haplotype's emission
    product falls below 1e-308 until the retry loop has brought the cap down (functions.R:2704-2715)."""
    from quilt_amd.synth import make_synthetic_sample
    s = make_synthetic_sample(panel, seed=seed, n_reads=n_reads)
    g0 = int(panel.nGrids // 2)
    snps = 32 * g0 + np.arange(n_snps)
    assert snps.max() < panel.nSNPs
    pat = [np.full(n_snps, -40), np.full(n_snps, 40), np.where(np.arange(n_snps) % 2 == 0, 40, -40)]
    reads = [(int(s.wif[r]), s.u[s.read_ptr[r]:s.read_ptr[r + 1]], s.bq[s.read_ptr[r]:s.read_ptr[r + 1]])
             for r in range(s.nReads)]
    for i in range(3 * per_type):
        reads.append((g0, snps.copy(), pat[i % 3].copy()))
    reads.sort(key=lambda t: t[0])   # stable: reads stay ordered by grid (the ABI's contract)
    ptr = np.r_[0, np.cumsum([len(t[1]) for t in reads])]
    return sample_from_arrays(ptr, np.concatenate([t[1] for t in reads]), np.concatenate([t[2] for t in reads]),
                              [t[0] for t in reads])
