"""Host-side mirror of the reference's full-panel interface (QUILT/R/RcppExports.R stubs).

Same names, argument meaning and in-place behaviour as the reference's
``Rcpp_haploid_dosage_versus_refs`` (QUILT/src/reference-single.cpp:2189-2413, called from
QUILT/R/functions.R:2034-2070): the function returns nothing and writes into the caller's
``dosage``, ``c``, ``*_alphaHat_t``, ``betaHat_t``, ``gamma_t``, ``gammaSmall_t`` and
``best_haps_stuff_list``.  The panel tables the reference passes on every call are replaced
by the device-resident handle (``panel``: :class:`quilt_amd.native.DevicePanel`), uploaded
once per process.  All arithmetic runs in the HIP library; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import native
from .native import DevicePanel, FullpassOpts, check, lib, ptr


def Rcpp_make_gl_bound(gl: np.ndarray, minGLValue: float, to_fix: np.ndarray) -> None:
    """reference-single.cpp:68-94 (in place; ``to_fix`` 0-based)."""
    assert gl.flags.f_contiguous and gl.dtype == np.float64
    to_fix = np.ascontiguousarray(to_fix, dtype=np.int32)
    check(lib().qa_Rcpp_make_gl_bound(ptr(gl), C.c_double(minGLValue), ptr(to_fix), C.c_int32(len(to_fix))))


def Rcpp_haploid_dosage_versus_refs(
    panel: DevicePanel,
    gl: np.ndarray,
    *,
    alphaHat_t: Optional[np.ndarray] = None,
    betaHat_t: Optional[np.ndarray] = None,
    c: Optional[np.ndarray] = None,
    gamma_t: Optional[np.ndarray] = None,
    gammaSmall_t: Optional[np.ndarray] = None,
    best_haps_stuff_list: Optional[List] = None,
    dosage: Optional[np.ndarray] = None,
    gammaSmall_cols_to_get: Optional[np.ndarray] = None,
    K_top_matches: int = 5,
    suppressOutput: int = 1,
    min_emission_prob_normalization_threshold: float = 1e-100,
    return_betaHat_t: bool = True,
    return_dosage: bool = True,
    return_gamma_t: bool = True,
    return_gammaSmall_t: bool = False,
    get_best_haps_from_thinned_sites: bool = False,
    always_normalize: bool = True,
    normalize_emissions: bool = True,
) -> None:
    P = panel.panel
    K, G, T = P.K, P.nGrids, P.nSNPs
    gl = np.asfortranarray(gl, dtype=np.float64)
    assert gl.shape == (2, T)
    if gammaSmall_cols_to_get is None:
        gammaSmall_cols_to_get = np.full(G, -1, dtype=np.int32)
    cols = np.ascontiguousarray(gammaSmall_cols_to_get, dtype=np.int32)
    n_thin = int((cols >= 0).sum())

    def _chk(a, shape, name):
        if a is None:
            return None
        if a.dtype != np.float64 or not a.flags.f_contiguous or a.shape != shape:
            raise ValueError(f"{name} must be a float64 column-major array of shape {shape}")
        return a

    _chk(alphaHat_t, (K, G), "alphaHat_t")
    _chk(betaHat_t, (K, G), "betaHat_t")
    _chk(gamma_t, (K, G), "gamma_t")
    if return_gammaSmall_t:
        _chk(gammaSmall_t, (K, n_thin), "gammaSmall_t")
    if return_dosage and dosage is None:
        raise ValueError("return_dosage needs a dosage buffer")
    if return_gamma_t and gamma_t is None:
        raise ValueError("return_gamma_t needs a gamma_t buffer")
    if return_betaHat_t and betaHat_t is None:
        raise ValueError("return_betaHat_t needs a betaHat_t buffer")
    cbuf = c if c is not None else np.ones(G, dtype=np.float64)
    opts = FullpassOpts(int(K_top_matches), int(return_betaHat_t), int(return_dosage), int(return_gamma_t),
                        int(return_gammaSmall_t), int(get_best_haps_from_thinned_sites), int(always_normalize),
                        int(normalize_emissions), float(min_emission_prob_normalization_threshold),
                        int(suppressOutput))
    bptr = np.zeros(n_thin + 1, dtype=np.int32)
    cap = max(64 * max(n_thin, 1), 1)
    for _ in range(2):
        bidx = np.zeros(cap, dtype=np.int32)
        bval = np.zeros(cap, dtype=np.float64)
        st = lib().qa_Rcpp_haploid_dosage_versus_refs(
            panel.handle, ptr(gl), ptr(cols), C.byref(opts), ptr(alphaHat_t), ptr(betaHat_t), ptr(cbuf),
            ptr(gamma_t), ptr(gammaSmall_t), ptr(dosage), ptr(bptr), ptr(bidx), ptr(bval), C.c_int64(cap))
        if st == native.QA_ERR_CAPACITY:
            cap = int(bptr[-1])
            continue
        check(st)
        break
    if get_best_haps_from_thinned_sites and best_haps_stuff_list is not None:
        if len(best_haps_stuff_list) != n_thin:
            raise ValueError("best_haps_stuff_list must have one entry per thinned grid")
        for i in range(n_thin):
            s, e = bptr[i], bptr[i + 1]
            best_haps_stuff_list[i] = dict(top_matches=bidx[s:e].copy(), top_matches_values=bval[s:e].copy())
