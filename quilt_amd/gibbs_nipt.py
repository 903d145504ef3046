"""Host-side mirror of the reference's Gibbs entry point (QUILT/R/RcppExports.R stub of
``rcpp_forwardBackwardGibbsNIPT``; kernel QUILT/src/gibbs-nipt.cpp:2395-3307).

Same argument names and meaning for what the production caller varies
(QUILT/R/functions.R:2566-2678); the panel tables are replaced by the device handle and the
uniforms the reference draws from R's RNG inside the call are explicit arguments (SURVEY.md
8(b)): the caller draws them in the same order (``runif(nReads * n_its)``, ``sample(nReads, 1)``,
then per block-Gibbs sweep ``runif(nGrids - 1)`` for the shard pass).  Returns a dict with the
reference's list names.  All arithmetic runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import native
from .native import DevicePanel, check, lib, ptr
from .trace import span


class GibbsOpts(C.Structure):
    _fields_ = [
        ("Ks", C.c_int32), ("ff", C.c_double), ("sample_is_diploid", C.c_int32), ("Jmax", C.c_int32),
        ("maxDifferenceBetweenReads", C.c_double), ("rescale_eMatRead_t", C.c_int32),
        ("n_gibbs_burn_in_its", C.c_int32), ("n_gibbs_sample_its", C.c_int32),
        ("block_gibbs_iterations", C.c_void_p), ("n_block_gibbs_iterations", C.c_int32),
        ("perform_block_gibbs", C.c_int32), ("do_shard_block_gibbs", C.c_int32),
        ("gibbs_initialize_iteratively", C.c_int32), ("disable_read_category_usage", C.c_int32),
        ("class_sum_cutoff", C.c_double),
        ("L_grid", C.c_void_p), ("shuffle_bin_radius", C.c_int32), ("block_gibbs_quantile_prob", C.c_double),
        ("ff_chain", C.c_void_p), ("per_it_out", C.c_void_p), ("hap_words_out", C.c_void_p),
        ("hap_major_out", C.c_void_p), ("hap_major_labels", C.c_int32), ("reads_same_as", C.c_void_p),
        ("draw_uniforms", C.c_void_p), ("draw_uniforms_ctx", C.c_void_p),
    ]


def forwardBackwardGibbsNIPT_batch(panel: DevicePanel, samples: Sequence, which_haps_to_use: Sequence[np.ndarray],
                                   starting_read_labels: Sequence[np.ndarray], runif_reads: Sequence[np.ndarray],
                                   first_read: Sequence[int], runif_shard: Sequence[np.ndarray], *,
                                   ff: float = 0.0, n_gibbs_burn_in_its: int = 20, n_gibbs_sample_its: int = 1,
                                   block_gibbs_iterations=(3, 6, 9), perform_block_gibbs: bool = True,
                                   gibbs_initialize_iteratively: bool = False,
                                   disable_read_category_usage: bool = False,
                                   maxDifferenceBetweenReads: float = 1e10, Jmax_local: int = 10000,
                                   class_sum_cutoff: float = 0.06, return_state: bool = False,
                                   seed_reads=None, seed_shard=None, return_hapProbs: bool = True,
                                   return_genProbs: bool = True, rare_common=None, runif_block=None,
                                   runif_resample=None, L_grid=None, shuffle_bin_radius: int = 5000,
                                   block_gibbs_quantile_prob: float = 0.95, return_per_it: bool = False,
                                   return_hap_words: bool = False, hap_major_out: Optional[np.ndarray] = None):
    """``n_chain`` independent calls of ``rcpp_forwardBackwardGibbsNIPT`` in one launch set.

    ``samples[c]`` is a :class:`quilt_amd.synth.SampleReads`-like object (``read_ptr``, ``u``, ``bq``,
    ``wif``); the other sequences hold one entry per chain.  Returns a list of dicts.

    ``rare_common`` (a :class:`quilt_amd.native.DeviceRareCommon`): the call the reference makes with
    ``make_eMatRead_t_rare_common = TRUE`` (QUILT/R/rare_common.R:325-398) -- ``samples`` then hold the all-SNP reads
    (``allSNP_sampleReads``) and the outputs cover all SNPs.

    NIPT (``ff`` > 0) with ``perform_block_gibbs``: ``runif_block[c]`` / ``runif_resample[c]`` hold
    ``len(block_gibbs_iterations) x nReads`` uniforms each (QUILT/src/gibbs-nipt.cpp:3016; gibbs-nipt-block.cpp:226-243)
    unless seeds are used; ``L_grid`` defaults to the panel's (the all-SNP grid's with ``rare_common``).

    ``hap_major_out``: a C-contiguous float64 array [n_chain, n_label, nSNPs] (ideally from ``native.pinned_empty``) that
    receives the haploid dosages label by label -- the layout the driver accumulates from -- instead of a per-chain
    ``hapProbs_t``; the dicts then carry views of its rows.
    """
    lib().qa_gibbs_batch.restype = C.c_int
    lib().qa_gibbs_batch_rare_common.restype = C.c_int
    import os, time
    ffc = None
    if np.ndim(ff) > 0:   # one fetal fraction per chain (NIPT)
        ffc = np.ascontiguousarray(ff, dtype=np.float64)
        ff = float(ffc[0])
    _t0 = time.perf_counter()
    P = panel.panel
    G, T = P.nGrids, P.nSNPs
    if rare_common is not None:
        G, T = rare_common.rc.nGrids_all, rare_common.rc.nSNPs_all
    Cn = len(samples)
    Ks = len(which_haps_to_use[0])
    n_its = n_gibbs_burn_in_its + n_gibbs_sample_its
    blocks = np.ascontiguousarray(block_gibbs_iterations, dtype=np.int32)
    nb = len(blocks) if perform_block_gibbs else 0
    read_off = np.zeros(Cn + 1, dtype=np.int32)
    for c, s in enumerate(samples):
        read_off[c + 1] = read_off[c] + s.nReads
    which = np.ascontiguousarray(np.stack([np.asarray(w, dtype=np.int32) for w in which_haps_to_use]))
    assert which.shape == (Cn, Ks)
    read_ptr = np.concatenate([np.asarray(s.read_ptr, dtype=np.int32) for s in samples])
    u = np.concatenate([np.asarray(s.u, dtype=np.int32) for s in samples])
    bq = np.concatenate([np.asarray(s.bq, dtype=np.int32) for s in samples])
    wif = np.concatenate([np.asarray(s.wif, dtype=np.int32) for s in samples])
    use_seeds = seed_reads is not None
    if use_seeds:
        ru = None
        sr = np.ascontiguousarray(seed_reads, dtype=np.uint64)
        ss = np.ascontiguousarray(seed_shard, dtype=np.uint64)
    else:
        sr = ss = None
        ru = np.concatenate([np.ascontiguousarray(r, dtype=np.float64).ravel()[: samples[c].nReads * n_its]
                             for c, r in enumerate(runif_reads)])
    fr = np.ascontiguousarray(first_read, dtype=np.int32)
    if ff != 0:
        # the block passes' uniforms, per chain [pass][block choice | label re-draw][read] (include/quilt_amd.h)
        if nb > 0 and not use_seeds:
            if runif_block is None or runif_resample is None:
                raise ValueError("NIPT block Gibbs: pass runif_block and runif_resample (or seeds)")
            parts = []
            for c, smp in enumerate(samples):
                R = smp.nReads
                rb = np.ascontiguousarray(runif_block[c], dtype=np.float64).ravel()[: len(blocks) * R].reshape(len(blocks), R)
                rr = np.ascontiguousarray(runif_resample[c], dtype=np.float64).ravel()[: len(blocks) * R].reshape(len(blocks), R)
                parts.append(np.stack([rb, rr], axis=1).ravel())
            rs = np.concatenate(parts)
        else:
            rs = np.zeros(1)
    elif nb > 0 and not use_seeds and perform_block_gibbs:   # (ff == 0: the shard passes, do_shard_block_gibbs below)
        parts = [np.ascontiguousarray(r, dtype=np.float64).ravel()[: nb * (G - 1)] for r in runif_shard]
        if len(parts) != Cn or any(len(x) != nb * (G - 1) for x in parts):   # (the library reads all of them)
            raise ValueError(f"runif_shard: {nb * (G - 1)} uniforms per chain (block iterations x (nGrids - 1))")
        rs = np.concatenate(parts)
    else:
        rs = np.zeros(1)   # (not read: no shard pass draws)
    if L_grid is None:
        L_grid = rare_common.rc.L_grid_all if rare_common is not None else P.L_grid
    Lg = np.ascontiguousarray(L_grid, dtype=np.int32) if L_grid is not None else None
    H = np.concatenate([np.asarray(h, dtype=np.int32) for h in starting_read_labels]).copy()
    if ff == 0 and H.size and (H.min() < 1 or H.max() > 2):
        raise ValueError("diploid read labels must be 1 or 2")
    Hc = np.zeros_like(H)
    # (np.empty: the library writes every entry of the rows it is asked for; zero-filling ~1 GB per call costs more than the copy)
    if hap_major_out is not None:
        if (hap_major_out.dtype != np.float64 or not hap_major_out.flags.c_contiguous or hap_major_out.ndim != 3 or
                hap_major_out.shape[0] != Cn or hap_major_out.shape[2] != T or hap_major_out.shape[1] not in (2, 3)):
            raise ValueError("hap_major_out must be a C-contiguous float64 array [n_chain, 2 or 3, nSNPs]")
        return_hapProbs = False
    hap = np.empty((Cn, T, 3)) if return_hapProbs else None
    gm = np.empty((Cn, T, 3)) if return_genProbs else None
    gf = np.empty((Cn, T, 3)) if return_genProbs else None
    uf = np.zeros(Cn, dtype=np.int32)
    state = np.zeros(6 * Ks * G + 3 * G) if (return_state and Cn == 1) else None
    per_it = np.zeros((Cn, n_its, 8)) if return_per_it else None
    words = np.zeros((Cn, 3, G), dtype=np.int32) if return_hap_words else None   # use_mspbwt: rounded, packed hapProbs
    opts = GibbsOpts(Ks, float(ff), int(ff == 0), int(Jmax_local), float(maxDifferenceBetweenReads), 1,
                     int(n_gibbs_burn_in_its), int(n_gibbs_sample_its), ptr(blocks), int(len(blocks)),
                     int(perform_block_gibbs), int(ff == 0), int(gibbs_initialize_iteratively),
                     int(disable_read_category_usage), float(class_sum_cutoff), ptr(Lg), int(shuffle_bin_radius),
                     float(block_gibbs_quantile_prob), ptr(ffc), ptr(per_it), ptr(words), ptr(hap_major_out),
                     0 if hap_major_out is None else int(hap_major_out.shape[1]))
    _t1 = time.perf_counter()
    tail = (C.byref(opts), C.c_int32(Cn), ptr(which), ptr(read_off), ptr(read_ptr), ptr(u), ptr(bq), ptr(wif), ptr(ru),
            ptr(fr), ptr(rs), ptr(H), ptr(Hc), ptr(hap), ptr(gm), ptr(gf), ptr(uf), ptr(state), ptr(sr), ptr(ss))
    if rare_common is not None:
        with span("device:gibbs_rare_common"):
            st = lib().qa_gibbs_batch_rare_common(panel.handle, rare_common.handle, *tail)
    else:
        with span("device:gibbs"):
            st = lib().qa_gibbs_batch(panel.handle, *tail)
    check(st)
    if os.environ.get("QA_TIMING"):
        print(f"[gibbs_batch py C={Cn}] marshal {_t1 - _t0:.3f} s, native call {time.perf_counter() - _t1:.3f} s", flush=True)
    out = []
    for c in range(Cn):
        s, e = read_off[c], read_off[c + 1]
        d = dict(underflow_problem=bool(uf[c]), H=H[s:e].copy(),
                 double_list_of_ending_read_labels=[[H[s:e].copy()]], H_class=Hc[s:e].copy())
        if per_it is not None:
            d["per_it"] = per_it[c].copy()
        if words is not None:
            d["hap_words"] = words[c]
        if hap is not None:
            d["hapProbs_t"] = np.asfortranarray(hap[c].T)
        if hap_major_out is not None:
            d["hapProbs_t"] = hap_major_out[c]   # [n_label, nSNPs] view
            d["hap_major_all"] = hap_major_out   # (the whole [chain, label, SNP] array: lets a caller skip per-chain copies)
        if gm is not None:
            d["genProbsM_t"] = np.asfortranarray(gm[c].T)
            d["genProbsF_t"] = np.asfortranarray(gf[c].T)
        if state is not None:
            m = state[: 6 * Ks * G].reshape(6, G, Ks)
            names = ("alphaHat_t1", "alphaHat_t2", "betaHat_t1", "betaHat_t2", "eMatGrid_t1", "eMatGrid_t2")
            for i, n in enumerate(names):
                d[n] = np.asfortranarray(m[i].T)
            cs = state[6 * Ks * G:].reshape(3, G)
            d["c1"], d["c2"], d["c3"] = cs[0].copy(), cs[1].copy(), cs[2].copy()
        out.append(d)
    return out


def rcpp_forwardBackwardGibbsNIPT(panel: DevicePanel, sampleReads, which_haps_to_use, starting_read_labels,
                                  runif_reads, first_read_for_gibbs_initialization, runif_shard, **kw):
    """Single-chain form with the reference's name; see :func:`forwardBackwardGibbsNIPT_batch`."""
    for k in ("runif_block", "runif_resample"):
        if kw.get(k) is not None:
            kw[k] = [kw[k]]
    return forwardBackwardGibbsNIPT_batch(panel, [sampleReads], [which_haps_to_use], [starting_read_labels],
                                          [runif_reads], [first_read_for_gibbs_initialization], [runif_shard],
                                          **kw)[0]


def calculate_eMatRead_t_vs_haplotypes_batch(panel: DevicePanel, samples: Sequence, haps: Sequence,
                                             maxDifferenceBetweenReads: float, rescale_eMatRead_t: bool = False,
                                             Jmax: int = 1000, nSNPs: Optional[int] = None, hap_major: bool = False):
    """``calculate_eMatRead_t_vs_haplotypes`` (QUILT/R/functions.R:2975-3020) for a batch: ``haps[c]`` is the
    list of K dense haplotype dosages of chain ``c``.  Returns one K x nReads matrix per chain.  ``nSNPs``: the
    length of the dosages when it is not the panel's (all-SNP reads, QUILT/R/rare_common.R:61-107)."""
    lib().qa_rcpp_make_eMatRead_t_nsnps.restype = C.c_int
    lib().qa_rcpp_make_eMatRead_t_hap_major.restype = C.c_int
    Cn = len(samples)
    T = panel.panel.nSNPs if nSNPs is None else int(nSNPs)
    fn = lib().qa_rcpp_make_eMatRead_t_nsnps
    if isinstance(haps, np.ndarray) and hap_major:   # [chain, haplotype, SNP], as the full-panel call returns dosages: no copy
        e = np.ascontiguousarray(haps, dtype=np.float64)
        K = e.shape[1]
        assert e.shape == (Cn, K, T)
        fn = lib().qa_rcpp_make_eMatRead_t_hap_major
    elif isinstance(haps, np.ndarray):   # already [chain, SNP, haplotype]
        e = np.ascontiguousarray(haps, dtype=np.float64)
        K = e.shape[2]
        assert e.shape == (Cn, T, K)
    else:
        K = len(haps[0])
        e = np.ascontiguousarray(np.stack([np.stack([np.asarray(h, dtype=np.float64) for h in hs], axis=1) for hs in haps]))
        assert e.shape == (Cn, T, K)
    read_off = np.zeros(Cn + 1, dtype=np.int32)
    for c, s in enumerate(samples):
        read_off[c + 1] = read_off[c] + s.nReads
    read_ptr = np.concatenate([np.asarray(s.read_ptr, dtype=np.int32) for s in samples])
    u = np.concatenate([np.asarray(s.u, dtype=np.int32) for s in samples])
    bq = np.concatenate([np.asarray(s.bq, dtype=np.int32) for s in samples])
    out = np.zeros((int(read_off[-1]), K))
    with span("device:read_confidence"):
        check(fn(panel.handle, C.c_int32(T), C.c_int32(Cn), C.c_int32(K), ptr(e), ptr(read_off),
                 ptr(read_ptr), ptr(u), ptr(bq), C.c_double(maxDifferenceBetweenReads), C.c_int32(Jmax),
                 C.c_int32(int(rescale_eMatRead_t)), ptr(out)))
    return [np.asfortranarray(out[read_off[c]:read_off[c + 1]].T) for c in range(Cn)]
