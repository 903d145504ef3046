"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``quilt_amd/``) must never do so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.qo_haploid_dosage_versus_refs.restype = C.c_int
        _LIB.qo_simple_binary_search.restype = C.c_int
        _LIB.qo_simple_binary_matrix_search.restype = C.c_int
        _LIB.qo_get_top_K_or_more_matches_while_building_gamma.restype = C.c_int
    return _LIB


def set_sum_order(left_to_right: bool) -> None:
    """How the oracle adds the sums the reference forms with Armadillo's ``sum()`` (quilt_oracle.h, "Armadillo's sum()"):
    False (the default) = Armadillo's two accumulators over the even / odd elements; True = left to right, the form rounds 1-5
    used.  Process-wide (a C global): tests that flip it restore it."""
    lib().qo_set_sum_order(C.c_int(1 if left_to_right else 0))


def get_sum_order() -> bool:
    return bool(lib().qo_get_sum_order())


def _p(a, ctype=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class _Panel(C.Structure):
    _fields_ = [
        ("K", C.c_int), ("nGrids", C.c_int), ("nSNPs", C.c_int), ("nMaxDH", C.c_int),
        ("rhb_t", C.c_void_p), ("hapMatcher", C.c_void_p), ("hapMatcherR", C.c_void_p),
        ("distinctHapsB", C.c_void_p), ("distinctHapsIE", C.c_void_p),
        ("eMatDH_special_grid_which", C.c_void_p), ("special_values_ptr", C.c_void_p),
        ("special_values", C.c_void_p), ("eMatDH_special_matrix_helper", C.c_void_p),
        ("eMatDH_special_matrix", C.c_void_p), ("eMatDH_special_matrix_nrow", C.c_int),
        ("use_eMatDH_special_symbols", C.c_int), ("transMatRate_t", C.c_void_p),
        ("ref_error", C.c_double),
    ]


class _FullOpts(C.Structure):
    _fields_ = [
        ("K_top_matches", C.c_int), ("min_emission_prob_normalization_threshold", C.c_double),
        ("return_betaHat_t", C.c_int), ("return_dosage", C.c_int), ("return_gamma_t", C.c_int),
        ("return_gammaSmall_t", C.c_int), ("get_best_haps_from_thinned_sites", C.c_int),
        ("always_normalize", C.c_int), ("normalize_emissions", C.c_int),
    ]


def panel_struct(panel, use_eMatDH_special_symbols: Optional[bool] = None):
    """Returns (struct, keepalive)."""
    ptr, vals = panel.special_csr()
    if use_eMatDH_special_symbols is None:
        use_eMatDH_special_symbols = panel.rhb_t is None
    keep = [ptr, vals]
    s = _Panel(
        panel.K, panel.nGrids, panel.nSNPs, panel.nMaxDH,
        _p(panel.rhb_t), _p(panel.hapMatcher), _p(panel.hapMatcherR),
        _p(panel.distinctHapsB), _p(panel.distinctHapsIE),
        _p(panel.eMatDH_special_grid_which), _p(ptr), _p(vals),
        _p(panel.eMatDH_special_matrix_helper), _p(panel.eMatDH_special_matrix),
        int(panel.eMatDH_special_matrix.shape[0]), int(bool(use_eMatDH_special_symbols)),
        _p(panel.transMatRate_t), float(panel.ref_error),
    )
    return s, keep


def make_gl_from_u_bq(u, bq, nSNPs, minGLValue=1e-10):
    u = np.ascontiguousarray(u, dtype=np.int32)
    bq = np.ascontiguousarray(bq, dtype=np.int32)
    gl = np.ones((2, nSNPs), dtype=np.float64, order="F")
    lib().qo_make_gl_from_u_bq(_p(u), _p(bq), C.c_int(len(u)), C.c_int(nSNPs), C.c_double(minGLValue), _p(gl))
    return gl


def make_gl_bound(gl, minGLValue, to_fix):
    to_fix = np.ascontiguousarray(to_fix, dtype=np.int32)
    lib().qo_make_gl_bound(_p(gl), C.c_double(minGLValue), _p(to_fix), C.c_int(len(to_fix)))
    return gl


def build_eMatDH(distinctHapsB, gl, nGrids, nSNPs, ref_error, add_zero_row=False):
    nMaxDH = distinctHapsB.shape[0]
    out = np.zeros((nMaxDH + int(add_zero_row), nGrids), dtype=np.float64, order="F")
    lib().qo_build_eMatDH(_p(distinctHapsB), _p(gl), C.c_int(nMaxDH), C.c_int(nGrids), C.c_int(nSNPs),
                          C.c_double(ref_error), C.c_int(int(add_zero_row)), _p(out))
    return out


def simple_binary_search(val, vec):
    vec = np.ascontiguousarray(vec, dtype=np.int32)
    return lib().qo_simple_binary_search(C.c_int(int(val)), _p(vec), C.c_int(len(vec)))


def simple_binary_matrix_search(val, mat, s1, e1):
    mat = np.asfortranarray(mat, dtype=np.int32)
    return lib().qo_simple_binary_matrix_search(C.c_int(int(val)), _p(mat), C.c_int(mat.shape[0]),
                                                C.c_int(int(s1)), C.c_int(int(e1)))


def get_top_K_or_more_matches(alpha_col, beta_col, K_top_matches, mult=1.0):
    a = np.ascontiguousarray(alpha_col, dtype=np.float64)
    b = np.ascontiguousarray(beta_col, dtype=np.float64)
    K = len(a)
    g = np.zeros(K)
    idx = np.zeros(K, dtype=np.int32)
    val = np.zeros(K)
    n = lib().qo_get_top_K_or_more_matches_while_building_gamma(
        _p(a), _p(b), _p(g), C.c_int(K), C.c_int(K_top_matches), C.c_double(mult), _p(idx), _p(val))
    return idx[:n].copy(), val[:n].copy(), g


def haploid_dosage_versus_refs(panel, gl, gammaSmall_cols_to_get=None, *, K_top_matches=5,
                               return_betaHat_t=False, return_dosage=True, return_gamma_t=False,
                               return_gammaSmall_t=False, get_best_haps_from_thinned_sites=False,
                               always_normalize=False, normalize_emissions=True,
                               min_emission_prob_normalization_threshold=1e-100,
                               use_eMatDH_special_symbols=None):
    """Oracle twin of ``Rcpp_haploid_dosage_versus_refs`` (reference-single.cpp:2189-2413).

    Returns a dict with alphaHat_t, c, dosage and whichever optional outputs were asked for;
    ``best_haps`` is a list (one per thinned column) of (top_matches 0-based, values).
    """
    K, G, T = panel.K, panel.nGrids, panel.nSNPs
    if gammaSmall_cols_to_get is None:
        gammaSmall_cols_to_get = np.full(G, -1, dtype=np.int32)
    cols = np.ascontiguousarray(gammaSmall_cols_to_get, dtype=np.int32)
    n_thin = int((cols >= 0).sum())
    ps, keep = panel_struct(panel, use_eMatDH_special_symbols)
    opts = _FullOpts(K_top_matches, min_emission_prob_normalization_threshold, int(return_betaHat_t),
                     int(return_dosage), int(return_gamma_t), int(return_gammaSmall_t),
                     int(get_best_haps_from_thinned_sites), int(always_normalize), int(normalize_emissions))
    gl = np.asfortranarray(gl, dtype=np.float64)
    alpha = np.zeros((K, G), dtype=np.float64, order="F")
    beta = np.zeros((K, G), dtype=np.float64, order="F") if return_betaHat_t else None
    gamma = np.zeros((K, G), dtype=np.float64, order="F") if return_gamma_t else None
    gsmall = np.zeros((K, max(n_thin, 1)), dtype=np.float64, order="F") if return_gammaSmall_t else None
    c = np.ones(G, dtype=np.float64)
    dosage = np.zeros(T, dtype=np.float64)
    cap = max(1, n_thin) * K if get_best_haps_from_thinned_sites else 1
    cap = min(cap, 1 << 27)
    bptr = np.zeros(n_thin + 1, dtype=np.int32)
    bidx = np.zeros(cap, dtype=np.int32)
    bval = np.zeros(cap, dtype=np.float64)
    st = lib().qo_haploid_dosage_versus_refs(
        C.byref(ps), C.byref(opts), _p(gl), _p(cols), _p(alpha), _p(beta), _p(c), _p(gamma), _p(gsmall),
        _p(dosage), _p(bptr), _p(bidx), _p(bval), C.c_int64(cap))
    if st != 0:
        raise RuntimeError("oracle best-haps capacity too small")
    best = [(bidx[bptr[i]:bptr[i + 1]].copy(), bval[bptr[i]:bptr[i + 1]].copy()) for i in range(n_thin)]
    return dict(alphaHat_t=alpha, betaHat_t=beta, gamma_t=gamma, gammaSmall_t=gsmall, c=c, dosage=dosage,
                best_haps=best if get_best_haps_from_thinned_sites else None)


class _GibbsArgs(C.Structure):
    _fields_ = [
        ("Ks", C.c_int), ("nReads", C.c_int), ("which_haps_to_use_1based", C.c_void_p),
        ("read_ptr", C.c_void_p), ("u", C.c_void_p), ("bq", C.c_void_p), ("wif", C.c_void_p),
        ("grid_has_read", C.c_void_p), ("ff", C.c_double), ("Jmax", C.c_int),
        ("maxDifferenceBetweenReads", C.c_double), ("n_gibbs_burn_in_its", C.c_int),
        ("n_gibbs_sample_its", C.c_int), ("block_gibbs_iterations", C.c_void_p),
        ("n_block_gibbs_iterations", C.c_int), ("perform_block_gibbs", C.c_int),
        ("do_shard_block_gibbs", C.c_int), ("gibbs_initialize_iteratively", C.c_int),
        ("sample_is_diploid", C.c_int), ("disable_read_category_usage", C.c_int),
        ("rescale_eMatRead_t", C.c_int), ("class_sum_cutoff", C.c_double),
        ("runif_reads", C.c_void_p), ("first_read", C.c_int), ("runif_shard", C.c_void_p),
        ("rc", C.c_void_p),
        ("L_grid", C.c_void_p), ("shuffle_bin_radius", C.c_int), ("block_gibbs_quantile_prob", C.c_double),
        ("runif_block", C.c_void_p), ("runif_resample", C.c_void_p),
        ("runif_stream", C.c_void_p), ("runif_stream_used", C.c_void_p),
    ]


class _RareCommon(C.Structure):
    _fields_ = [
        ("nSNPs_all", C.c_int), ("nGrids_all", C.c_int), ("snp_is_common", C.c_void_p),
        ("common_snp_index", C.c_void_p), ("rare_ptr", C.c_void_p), ("rare_snp_1based", C.c_void_p),
        ("transMatRate_t_all", C.c_void_p),
    ]


def rare_common_struct(rc):
    keep = dict(a=np.ascontiguousarray(rc.snp_is_common, dtype=np.uint8),
                b=np.ascontiguousarray(rc.common_snp_index, dtype=np.int32),
                c=np.ascontiguousarray(rc.rare_ptr, dtype=np.int64),
                d=np.ascontiguousarray(rc.rare_snp, dtype=np.int32),
                e=np.asfortranarray(rc.transMatRate_t_all, dtype=np.float64))
    st = _RareCommon(int(rc.nSNPs_all), int(rc.nGrids_all), _p(keep["a"]), _p(keep["b"]), _p(keep["c"]), _p(keep["d"]),
                     _p(keep["e"]))
    return st, keep


def make_eMatRead_t_rare_common(panel, rc, sample_all, which_haps_to_use, maxDifferenceBetweenReads=1e10, Jmax=10000,
                                rescale_eMatRead_t=True, use_eMatDH_special_symbols=None):
    """``Rcpp_make_eMatRead_t_for_final_rare_common_gibbs_using_objects`` (gibbs-small.cpp:270-460)."""
    which = np.ascontiguousarray(which_haps_to_use, dtype=np.int32)
    Ks, R = len(which), sample_all.nReads
    ps, keep = panel_struct(panel, use_eMatDH_special_symbols)
    rs, keep2 = rare_common_struct(rc)
    e = np.ones((Ks, R), dtype=np.float64, order="F")
    lib().qo_make_eMatRead_t_rare_common(
        C.byref(ps), C.byref(rs), _p(which), C.c_int(Ks), C.c_int(R), _p(sample_all.read_ptr), _p(sample_all.u),
        _p(sample_all.bq), C.c_int(int(rescale_eMatRead_t)), C.c_int(Jmax), C.c_double(maxDifferenceBetweenReads), _p(e))
    return e


def grid_has_read_of(sample, nGrids):
    g = np.zeros(nGrids, dtype=np.uint8)
    g[sample.wif] = 1
    return g


def make_eMatRead_t(panel, sample, which_haps_to_use, maxDifferenceBetweenReads=1e10, Jmax=10000,
                    rescale_eMatRead_t=True, use_eMatDH_special_symbols=None):
    which = np.ascontiguousarray(which_haps_to_use, dtype=np.int32)
    Ks, R = len(which), sample.nReads
    ps, keep = panel_struct(panel, use_eMatDH_special_symbols)
    e = np.ones((Ks, R), dtype=np.float64, order="F")
    lib().qo_make_eMatRead_t_for_gibbs_using_objects(
        C.byref(ps), _p(which), C.c_int(Ks), C.c_int(R), _p(sample.read_ptr), _p(sample.u), _p(sample.bq),
        C.c_int(int(rescale_eMatRead_t)), C.c_int(Jmax), C.c_double(maxDifferenceBetweenReads), _p(e))
    return e


def evaluate_read_variability(eMatRead_t):
    Ks, R = eMatRead_t.shape
    n = np.zeros(R, dtype=np.int32)
    idx = np.zeros((Ks, R), dtype=np.int32, order="F")
    cat = np.zeros(R, dtype=np.int32)
    lib().qo_evaluate_read_variability(_p(np.asfortranarray(eMatRead_t)), C.c_int(Ks), C.c_int(R), _p(n), _p(idx), _p(cat))
    return n, idx, cat


def forwardBackwardGibbsNIPT(panel, sample, which_haps_to_use, H, runif_reads, first_read, runif_shard, *,
                             ff=0.0, n_gibbs_burn_in_its=20, n_gibbs_sample_its=1,
                             block_gibbs_iterations=(3, 6, 9), perform_block_gibbs=True,
                             gibbs_initialize_iteratively=False, sample_is_diploid=None,
                             disable_read_category_usage=False, maxDifferenceBetweenReads=1e10, Jmax=10000,
                             class_sum_cutoff=0.06, use_eMatDH_special_symbols=None, rare_common=None,
                             runif_block=None, runif_resample=None, shuffle_bin_radius=5000,
                             block_gibbs_quantile_prob=0.95, L_grid=None, runif_stream=None):
    """Oracle twin of ``rcpp_forwardBackwardGibbsNIPT`` (gibbs-nipt.cpp:2395-3307), production path.

    ``H``: starting labels (1-based); returns a dict holding the ending labels and every state
    matrix the reference mutates in place.  ``rare_common``: the all-SNP side of the panel for the final
    rare + common Gibbs (``make_eMatRead_t_rare_common = TRUE``); ``sample`` then holds the all-SNP reads.
    NIPT (``ff`` > 0) with block Gibbs: ``runif_block`` / ``runif_resample`` hold ``len(block_gibbs_iterations) x
    nReads`` uniforms (gibbs-nipt.cpp:3016; gibbs-nipt-block.cpp:226-243); or ``runif_stream``: ONE stream consumed in the
    reference's order -- per block pass nReads of runif_block, then one uniform per read whose class leaves a choice
    (``out["runif_stream_used"]`` = how many were consumed).
    """
    lib().qo_gibbs.restype = C.c_int
    which = np.ascontiguousarray(which_haps_to_use, dtype=np.int32)
    Ks, R, G, T = len(which), sample.nReads, panel.nGrids, panel.nSNPs
    rs = keep_rc = None
    if rare_common is not None:
        G, T = rare_common.nGrids_all, rare_common.nSNPs_all
        rs, keep_rc = rare_common_struct(rare_common)
    if sample_is_diploid is None:
        sample_is_diploid = ff == 0
    blocks = np.ascontiguousarray(block_gibbs_iterations, dtype=np.int32)
    ghr = grid_has_read_of(sample, G)
    runif_reads = np.ascontiguousarray(runif_reads, dtype=np.float64)
    runif_shard = np.ascontiguousarray(runif_shard, dtype=np.float64)
    n_its = n_gibbs_burn_in_its + n_gibbs_sample_its
    assert runif_reads.size >= R * n_its
    assert runif_shard.size >= len(blocks) * (G - 1)
    args = _GibbsArgs(Ks, R, _p(which), _p(sample.read_ptr), _p(sample.u), _p(sample.bq), _p(sample.wif),
                      _p(ghr), float(ff), int(Jmax), float(maxDifferenceBetweenReads), int(n_gibbs_burn_in_its),
                      int(n_gibbs_sample_its), _p(blocks), len(blocks), int(perform_block_gibbs),
                      int(ff == 0), int(gibbs_initialize_iteratively), int(sample_is_diploid),
                      int(disable_read_category_usage), 1, float(class_sum_cutoff), _p(runif_reads),
                      int(first_read), _p(runif_shard), C.cast(C.pointer(rs), C.c_void_p) if rs is not None else None,
                      None, int(shuffle_bin_radius), float(block_gibbs_quantile_prob), None, None, None, None)
    keep_nipt = None
    stream_used = np.zeros(1, dtype=np.int64)
    if ff != 0 and perform_block_gibbs and len(blocks) and runif_stream is not None:
        Lg = np.ascontiguousarray(panel.L_grid if L_grid is None else L_grid, dtype=np.int32)
        # (padded with NaN: a pass may draw up to nReads uniforms, a count only its own result fixes; a stream that is too short
        # shows as NaN-driven labels and a `runif_stream_used` beyond its length instead of a read past the buffer)
        rst = np.concatenate([np.asarray(runif_stream, dtype=np.float64).ravel(), np.full(len(blocks) * 2 * R, np.nan)])
        assert len(Lg) == G
        keep_nipt = (Lg, rst)
        args.L_grid, args.runif_stream, args.runif_stream_used = Lg.ctypes.data, rst.ctypes.data, stream_used.ctypes.data
    if ff != 0 and perform_block_gibbs and len(blocks) and runif_block is not None:
        Lg = np.ascontiguousarray(panel.L_grid if L_grid is None else L_grid, dtype=np.int32)
        rb = np.ascontiguousarray(runif_block, dtype=np.float64)
        rr_ = np.ascontiguousarray(runif_resample, dtype=np.float64)
        assert rb.size >= len(blocks) * R and rr_.size >= len(blocks) * R and len(Lg) == G
        keep_nipt = (Lg, rb, rr_)
        args.L_grid, args.runif_block, args.runif_resample = Lg.ctypes.data, rb.ctypes.data, rr_.ctypes.data
    ps, keep = panel_struct(panel, use_eMatDH_special_symbols)
    Hout = np.array(H, dtype=np.int32).copy()
    Hc = np.zeros(R, dtype=np.int32)
    mats = {n: [np.zeros((Ks, G), order="F") for _ in range(3)] for n in ("alpha", "beta", "eg")}
    cs = [np.zeros(G) for _ in range(3)]

    def arr3(lst):
        return (C.c_void_p * 3)(*[a.ctypes.data for a in lst])

    eMatRead = np.zeros((Ks, R), order="F")
    cat = np.zeros(R, dtype=np.int32)
    hap = np.zeros((3, T), order="F")
    gm = np.zeros((3, T), order="F")
    gf = np.zeros((3, T), order="F")
    st = lib().qo_gibbs(C.byref(ps), C.byref(args), _p(Hout), _p(Hc), arr3(mats["alpha"]), arr3(mats["beta"]),
                        arr3(mats["eg"]), arr3(cs), _p(eMatRead), _p(cat), _p(hap), _p(gm), _p(gf))
    return dict(status=st, underflow_problem=(st == 1), H=Hout, H_class=Hc, alphaHat_t=mats["alpha"],
                betaHat_t=mats["beta"], eMatGrid_t=mats["eg"], c=cs, eMatRead_t=eMatRead, read_category=cat,
                hapProbs_t=hap, genProbsM_t=gm, genProbsF_t=gf, runif_stream_used=int(stream_used[0]))


def calculate_eMatRead_t_vs_haplotypes(sample, haps, maxDifferenceBetweenReads, rescale_eMatRead_t=False, Jmax=1000):
    """``calculate_eMatRead_t_vs_haplotypes`` (functions.R:2975-3020) via the dense ``rcpp_make_eMatRead_t``."""
    K = len(haps)
    e = np.asfortranarray(np.stack([np.asarray(h, dtype=np.float64) for h in haps], axis=0))
    out = np.ones((K, sample.nReads), order="F")
    lib().qo_make_eMatRead_t_dense(_p(e), C.c_int(K), C.c_int(sample.nReads), _p(sample.read_ptr), _p(sample.u),
                                   _p(sample.bq), C.c_double(maxDifferenceBetweenReads), C.c_int(Jmax),
                                   C.c_int(int(rescale_eMatRead_t)), _p(out))
    return out


# ---- pieces of the NIPT block Gibbs (gibbs-nipt-block.cpp) with known answers / defining properties ----

def simple_quantile(x, q):
    lib().qo_simple_quantile.restype = C.c_double
    x = np.ascontiguousarray(x, dtype=np.float64)
    return float(lib().qo_simple_quantile(_p(x), C.c_int(len(x)), C.c_double(q)))


def get_log_p_H_class2(n1, n2, n3, n4, n5, n6, ff):
    lib().qo_get_log_p_H_class2.restype = C.c_double
    return float(lib().qo_get_log_p_H_class2(*(C.c_int(int(v)) for v in (n1, n2, n3, n4, n5, n6)), C.c_double(ff)))


def zero_based_swap(ir_chosen_0based):
    out = (C.c_int * 8)()
    lib().qo_zero_based_swap(C.c_int(ir_chosen_0based), out)
    return np.array(list(out), dtype=np.int64)


def sample3(probs, u):
    lib().qo_sample3.restype = C.c_int
    return int(lib().qo_sample3((C.c_double * 3)(*probs), C.c_double(u)))


def make_smoothed_rate(sigma_rate, L_grid, shuffle_bin_radius):
    L = np.ascontiguousarray(L_grid, dtype=np.int32)
    r = np.ascontiguousarray(sigma_rate, dtype=np.float64)
    out = np.zeros(len(L) - 1)
    lib().qo_make_smoothed_rate(_p(r), _p(L), C.c_int(len(L)), C.c_int(shuffle_bin_radius), _p(out))
    return out


def define_blocked_grids(rate2, L_grid, shuffle_bin_radius=5000, block_gibbs_quantile_prob=0.95):
    L = np.ascontiguousarray(L_grid, dtype=np.int32)
    r = np.ascontiguousarray(rate2, dtype=np.float64)
    out = np.zeros(len(L), dtype=np.int32)
    lib().qo_define_blocked_grids(_p(r), _p(L), C.c_int(len(L)), C.c_int(shuffle_bin_radius),
                                  C.c_double(block_gibbs_quantile_prob), _p(out))
    return out


def make_gibbs_considers(blocked_grid, wif0):
    b = np.ascontiguousarray(blocked_grid, dtype=np.int32)
    w = np.ascontiguousarray(wif0, dtype=np.int32)
    G = len(b)
    arrs = [np.zeros(G, dtype=np.int32) for _ in range(5)]
    lib().qo_make_gibbs_considers.restype = C.c_int
    n = lib().qo_make_gibbs_considers(_p(b), C.c_int(G), _p(w), C.c_int(len(w)), *[_p(a) for a in arrs])
    names = ("consider_grid_start_0_based", "consider_grid_end_0_based", "consider_reads_start_0_based",
             "consider_reads_end_0_based")
    out = {k: a[:n].copy() for k, a in zip(names, arrs)}
    out["consider_grid_where_0_based"] = arrs[4]
    out["n_blocks"] = n
    return out
