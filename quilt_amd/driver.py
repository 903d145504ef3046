"""Per-sample driver loop: host-side mirror of ``get_and_impute_one_sample`` (QUILT/R/functions.R:3-1500),
``impute_using_everything`` (:1922-2157) and their helpers, restructured for a GPU.

The reference walks one sample at a time through ``(nGibbsSamples + 1) x n_seek_its`` rounds of
[small-panel Gibbs -> full-panel pass per read label -> choose new haplotypes].  The Gibbs samples
of one sample are independent until the consensus "phasing" pass (functions.R:1170-1205), and
samples are independent of each other, so this driver advances *all chains of a batch of samples in
lock-step*: every round is ONE batched Gibbs launch set and ONE batched full-pass launch set over
``n_samples x nGibbsSamples`` chains (then ``n_samples`` chains for the phasing pass).  The host keeps
only what the reference keeps on the R side: read labels, haplotype subsets, the selection logic and
the dosage / genotype-probability accumulators.

The compute is delegated to a *backend* (``gibbs_batch`` / ``fullpass_batch``): the product backend is
:class:`HipBackend` (the C-ABI HIP library; no fallback).  Tests plug in a CPU-oracle backend to check
the whole pipeline.

R's Mersenne-Twister stream cannot be reproduced without R; every random draw the reference makes
(functions.R:580, 584, 746, 2294, 2301; gibbs-nipt.cpp:2845-2848; gibbs-nipt-block.cpp:2054) comes
from a counter-based stream keyed by (seed, sample, chain) -- see :func:`chain_rng`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import trace
from .trace import span


@dataclass
class DriverParams:
    """Defaults of ``QUILT()`` (QUILT/R/quilt.R:97-186) for the arguments the hot path sees."""

    nGibbsSamples: int = 7
    n_seek_its: int = 3
    n_burn_in_seek_its: Optional[int] = None      # NA -> n_seek_its - 1 (quilt.R:248-250)
    Ksubset: int = 600
    Knew: int = 600
    K_top_matches: int = 5
    heuristic_match_thin: float = 0.1
    small_ref_panel_gibbs_iterations: int = 20
    n_gibbs_sample_its: int = 1
    small_ref_panel_block_gibbs_iterations: Sequence[int] = (3, 6, 9)
    maxDifferenceBetweenReads: float = 1e10
    minGLValue: float = 1e-10
    Jmax: int = 10000
    seed: int = 1
    impute_rare_common: bool = False   # quilt.R:180: finish every Gibbs sample with a Gibbs call over ALL SNPs
    method: str = "diploid"            # or "nipt": mother + fetus, three read labels, fetal fraction per sample (sample.ff)
    shuffle_bin_radius: int = 5000     # quilt.R:134 (block definition of the NIPT block Gibbs)
    # Block Gibbs of DIPLOID samples (Rcpp_block_gibbs_resampler with ff = 0).  "reference_noop": what the reference does in
    # production -- the third label's c is all zero, the relabelling scores are NaN, no block is ever relabelled, the pass
    # only re-runs the backward sweep (gibbs-nipt-block.cpp:1819-1821, :661-675, :741-752, :830; derivation in
    # oracle/gibbs.c).  An "active" two-label block pass -- which the reference never executes -- is not built; the name is
    # here so that the behaviour is a stated choice rather than an omission (NIPT samples, ff > 0, run the real block pass).
    diploid_block_gibbs: str = "reference_noop"
    # use_mspbwt = TRUE (quilt.R:170-174): no full-panel pass; the next small panel comes from long matches of the Gibbs call's
    # rounded haploid dosages against the panel (mspbwt.R:225-474), the dosages from the Gibbs call itself (functions.R:784-893)
    use_mspbwt: bool = False
    mspbwtL: int = 3                   # neighbours scanned up and down per grid in the reference's index; here it scales how
    mspbwtM: int = 1                   # many matches a search returns (mspbwt_max_matches); mspbwtM: minimum match length
    mspbwt_nindices: int = 4
    # the query behind select_new_haps_mspbwt_v3: "scan" = the msPBWT neighbour scan of the panel's indices (the reference's
    # mspbwt::Rcpp_find_good_matches_without_a semantics: mspbwtL neighbours up and down per grid; host, csrc/mspbwt.cpp);
    # "exhaustive" = every haplotype's longest run, searched on the device (csrc/match.hip; this library's own definition)
    mspbwt_search: str = "scan"
    mspbwt_max_matches: Optional[int] = None   # matches per (haplotype, index) from the device search; None: 50 * mspbwtL

    def resolved(self, K: int) -> "DriverParams":
        p = DriverParams(**self.__dict__)
        if p.n_burn_in_seek_its is None:
            p.n_burn_in_seek_its = p.n_seek_its - 1
        if K < p.Ksubset:                           # quilt.R:453-463
            p.n_seek_its, p.n_burn_in_seek_its, p.Ksubset, p.Knew = 1, 0, K, K
        if p.Knew > p.Ksubset:                      # quilt.R:467-471
            p.Knew = p.Ksubset
        # validate_n_seek_its_and_n_burn_in_seek_its (quilt.R; STITCH-style argument checks)
        if p.n_seek_its < 1 or int(p.n_seek_its) != p.n_seek_its:
            raise ValueError("n_seek_its must be an integer >= 1")
        if p.n_burn_in_seek_its < 0 or p.n_burn_in_seek_its >= p.n_seek_its:
            raise ValueError("n_burn_in_seek_its must be in 0 .. n_seek_its - 1 (at least one seek iteration must count towards "
                             "the dosages)")
        if p.nGibbsSamples < 1 or p.Ksubset < 1 or p.Knew < 1:
            raise ValueError("nGibbsSamples, Ksubset and Knew must be >= 1")
        if p.K_top_matches < 1:
            raise ValueError("K_top_matches must be >= 1")
        if p.diploid_block_gibbs != "reference_noop":
            raise ValueError("diploid_block_gibbs: only 'reference_noop' exists (the reference's diploid block pass never "
                             "relabels; an 'active' two-label block Gibbs is not built)")
        if p.use_mspbwt and p.Knew != p.Ksubset:
            raise ValueError("use_mspbwt: select_new_haps_mspbwt_v3 returns the whole next small panel, so Knew must equal "
                             "Ksubset (the reference's defaults: 600 / 600)")
        if p.use_mspbwt and (p.mspbwt_nindices < 1 or p.mspbwtM < 1 or p.mspbwtL < 1):
            raise ValueError("mspbwt_nindices, mspbwtM and mspbwtL must be >= 1")
        if p.mspbwt_search not in ("scan", "exhaustive"):
            raise ValueError("mspbwt_search is 'scan' (the msPBWT neighbour scan) or 'exhaustive' (the device search)")
        if p.use_mspbwt and p.mspbwt_search == "scan" and p.mspbwtL > 64:
            raise ValueError("mspbwtL <= 64")
        return p


def thinned_grid_columns(nGrids: int, heuristic_match_thin: float) -> np.ndarray:
    """``full_gammaSmall_cols_to_get`` (quilt.R:719-721): R's ``seq(1, nGrids, length.out = n)`` used as
    an index vector (fractional indices truncate)."""
    n = max(1, int(round(heuristic_match_thin * nGrids)))
    ww = np.linspace(1, nGrids, n) if n > 1 else np.array([1.0])
    idx = np.floor(ww + 1e-9).astype(np.int64) - 1
    cols = np.full(nGrids, -1, dtype=np.int32)
    for i, g in enumerate(idx):   # later duplicates overwrite, as R's assignment would
        cols[g] = i
    # R: cols[ww] <- 0:(n-1); duplicated indices keep the last value, unused column numbers vanish.
    # Re-number densely so that column ids are 0..n_thin-1 in grid order (what the kernels expect).
    used = np.nonzero(cols >= 0)[0]
    cols[used] = np.arange(len(used), dtype=np.int32)
    return cols


def chain_rng(seed: int, i_sample: int, i_chain: int):
    """Counter-based stream of one (sample, Gibbs chain): quilt_amd/rng.py::ChainStream, the draws csrc/impute.cpp makes."""
    from .rng import ChainStream
    return ChainStream(seed, i_sample, i_chain)


# ---------------------------------------------------------------------------------------------
# host logic restated from the R driver
# ---------------------------------------------------------------------------------------------

def _phred_eps_table() -> np.ndarray:
    """10^(-q / 10) for q = 0 .. 255 from the C library's pow(), entry by entry -- what R's ``10^x`` evaluates to
    (convertScaledBQtoProbs), what the library's tables hold (csrc: "eps tables from the host libm") and what csrc/impute.cpp
    uses.  numpy's vectorised ``10.0 ** x`` is NOT that function: it differs from pow() in the last bit at q = 22, 50, 61, 78 on
    this machine, and a last bit in a genotype likelihood is enough to break an exact tie between two identical panel
    haplotypes at a best-haplotype list's threshold differently (scripts/check_pipeline_seeds.py found it: 6 of 96 runs)."""
    import math
    return np.array([math.pow(10.0, -q / 10.0) for q in range(256)])


PHRED_EPS = _phred_eps_table()


def phred_eps(bq) -> np.ndarray:
    """Error probability of signed base qualities (|bq| <= 255, the ABI's bound), libm-exact: see ``_phred_eps_table``."""
    return PHRED_EPS[np.abs(np.asarray(bq)).astype(np.int64)]


def make_gl_from_u_bq(u: np.ndarray, bq: np.ndarray, nSNPs: int, minGLValue: float, make_gl_bound) -> np.ndarray:
    """reference-single.R:19-42: per-label genotype likelihoods from the reads' bases (host code in the
    reference too); ``make_gl_bound`` is the native ``Rcpp_make_gl_bound``."""
    gl = np.ones((2, nSNPs), dtype=np.float64, order="F")
    # a base with bq == 0 carries no allele (neither ref nor alt in the signed-quality convention): the caller filters
    # such bases (functions.R:2018-2020) and the device kernel skips them (k_make_gl); here too they contribute nothing
    keep = np.asarray(bq) != 0
    u, bq = np.asarray(u)[keep], np.asarray(bq)[keep]
    if len(u) == 0:
        return gl
    eps = phred_eps(bq)
    ref = bq < 0
    pR = np.where(ref, 1 - eps, eps / 3)
    pA = np.where(ref, eps / 3, 1 - eps)
    np.multiply.at(gl[0], u, pR)
    np.multiply.at(gl[1], u, pA)
    if minGLValue > 0:
        to_fix = np.nonzero((gl < minGLValue).any(axis=0))[0].astype(np.int32)
        if len(to_fix):
            make_gl_bound(gl, minGLValue, to_fix)
    return gl


def everything_per_hap_rejig_haps(best_haps_stuff_list) -> List[np.ndarray]:
    """functions.R:2161-2170: per thinned grid, 1-based haplotypes by descending value (stable)."""
    out = []
    for x in best_haps_stuff_list:
        tm = np.asarray(x["top_matches"]) + 1
        order = np.argsort(-np.asarray(x["top_matches_values"]), kind="stable")
        out.append(tm[order])
    return out


class ListsTruncated(Exception):
    """everything_select_good_haps reached its exhausted branch (all entries of all lists, functions.R:2278-2281) while at
    least one best-haplotype list had been cut to the batched call's ``top_width``: the full lists are needed."""


def _unique_in_order(a: np.ndarray) -> np.ndarray:
    _, idx = np.unique(a, return_index=True)
    return a[np.sort(idx)]


def everything_select_good_haps(Knew: int, K_top_matches: int, new_haps: List[List[np.ndarray]],
                                previously_selected_haplotypes: np.ndarray, K: int, seed_select: int) -> np.ndarray:
    """functions.R:2262-2310.  ``new_haps[label][thinned grid]`` = 1-based haplotypes, best first."""
    width = max((len(y) for x in new_haps for y in x), default=0)
    dense = np.zeros((len(new_haps), max((len(x) for x in new_haps), default=0), max(width, 1)), dtype=np.int64)
    for a, x in enumerate(new_haps):
        for b, y in enumerate(x):
            dense[a, b, : len(y)] = y
    return everything_select_good_haps_dense(Knew, K_top_matches, dense, previously_selected_haplotypes, K, seed_select)


def previously_selected(which_haps_to_use: np.ndarray, n_keep: int, seed_select: int) -> np.ndarray:
    """functions.R:2286: ``sample(which_haps_to_use, Ksubset - Knew)``, drawn as the device draws it (csrc/select.hip)."""
    from .rng import SELECT_OFFSET_PREV, keyed_subset
    return which_haps_to_use[keyed_subset(seed_select, len(which_haps_to_use), n_keep, SELECT_OFFSET_PREV)]


def everything_select_good_haps_dense(Knew: int, K_top_matches: int, top: np.ndarray,
                                      previously_selected_haplotypes: np.ndarray, K: int,
                                      seed_select: int, truncated: bool = False) -> np.ndarray:
    """functions.R:2262-2310 on a dense table ``top[label, thinned grid, rank]`` of 1-based haplotypes (0 = no
    entry), each list ordered best first (functions.R:2161-2170).  Rank by rank, the distinct candidates (label-
    major, grid order: R's ``unlist(sapply(new_haps, ...))``) are added until ``Knew`` are found; the last rank
    is subsampled at random.  The reference draws with R's ``sample``; here every draw is "the smallest keys of the
    counter stream ``seed_select``" (quilt_amd/rng.py), the rule the device kernel csrc/select.hip implements, so the host
    and the device selection are the same function.  ``truncated``: some list is longer than the table is wide (ties at its
    threshold); the ranks up to ``K_top_matches`` are still exact, the exhausted branch is not (see :class:`ListsTruncated`)."""
    from .rng import SELECT_OFFSET_POOL, SELECT_OFFSET_RANK, keyed_subset
    i = 1
    to_keep = np.zeros(0, dtype=np.int64)
    prev = np.asarray(previously_selected_haplotypes, dtype=np.int64)
    done = False
    while not done:
        if i <= K_top_matches and i <= top.shape[2]:
            vals = top[:, :, i - 1].ravel()
        else:
            if truncated:   # raised before any random draw: the caller re-runs the selection on the full lists
                raise ListsTruncated()
            vals = top.reshape(-1)
            done = True
        vals = vals[vals > 0]
        new = _unique_in_order(vals.astype(np.int64)) if len(vals) else np.zeros(0, dtype=np.int64)
        new = new[~np.isin(new, prev)]
        new = new[~np.isin(new, to_keep)]
        if len(new) < Knew - len(to_keep):
            to_keep = np.concatenate([to_keep, new])
            i += 1
        else:
            toadd = Knew - len(to_keep)
            pick = keyed_subset(seed_select, len(new), toadd, SELECT_OFFSET_RANK)
            to_keep = np.concatenate([to_keep, new[pick]])
            done = True
    if len(to_keep) < Knew:
        pool = np.setdiff1d(np.arange(1, K + 1), np.concatenate([to_keep, prev]))
        extra = pool[keyed_subset(seed_select, len(pool), Knew - len(to_keep), SELECT_OFFSET_POOL)]
        to_keep = np.concatenate([to_keep, extra])
    if len(to_keep) != Knew:
        raise RuntimeError("Have returned too many haps")
    return to_keep.astype(np.int32)


def assess_ability_of_reads_to_be_confident(p: np.ndarray, minrp: float = 0.95) -> np.ndarray:
    """functions.R:1615-1660: ``p`` = 2 x nReads read likelihoods against (hap1, hap2), or 3 x nReads (NIPT)."""
    if p.shape[0] == 3:
        with np.errstate(invalid="ignore", divide="ignore"):
            q = p / p.sum(axis=0)
        mp = q[0].copy()
        w = q[1] > q[0]
        mp[w] = q[1][w]
        w = q[2] > mp
        mp[w] = q[2][w]
        mp[np.isnan(mp)] = 1 / 3
        return mp > minrp
    p1, p2 = p[0], p[1]
    with np.errstate(invalid="ignore", divide="ignore"):
        mp = p1 / (p1 + p2)
    mp[np.isnan(mp)] = 0.5
    mp = np.where(mp < 0.5, 1 - mp, mp)
    return mp > minrp


def determine_best_read_label_so_far(read_label_matrix_all: np.ndarray, read_label_matrix_conf: np.ndarray,
                                     nReads: int, nGibbsSamples: int, can_hap: int) -> np.ndarray:
    """functions.R:1680-1784 (including that the final flips start at the LAST change point, counted in the filtered
    row space: ``w <- s1[i]:nReads`` with the leftover loop variable).

    The reference rewrites the suffix ``w`` of its working matrix at every change point; rows are only ever read at
    later change points, so the rewrites are kept here as flip parities per column (and one for the canonical
    haplotype) and applied when a row is looked at: O(change points x columns) instead of O(change points x reads).
    ``_determine_best_read_label_so_far_literal`` is the line-by-line form the tests compare against.
    """
    rl = read_label_matrix_all.astype(np.int64)
    default = rl[:, can_hap - 1].astype(np.int32)
    keep = read_label_matrix_conf.all(axis=1)
    L0 = rl[keep]
    if L0.shape[0] < 10:
        return default
    can0 = L0[:, can_hap - 1]
    a0 = L0 - can0[:, None]
    s = np.nonzero(np.diff(np.abs(a0).sum(axis=1)) != 0)[0] + 1  # R's which(), 1-based
    if len(s) == 0:
        return default
    s1 = np.concatenate([[1], s + 1])
    fc = np.zeros(nGibbsSamples, dtype=bool)   # label-flip parity of each column over the current suffix
    fcan = False                               # flip parity of the canonical labels
    flipped = np.zeros(nGibbsSamples, dtype=bool)
    half = nGibbsSamples / 2
    for r in (s1[1:] - 1).tolist():            # R: for(i in 2:length(s1))
        lab = np.where(fc, 3 - L0[r], L0[r])
        cur = lab - ((3 - can0[r]) if fcan else can0[r])
        changed = np.nonzero(cur != 0)[0]
        if len(changed) > 0:
            if len(changed) > half:
                changed = np.nonzero(cur == 0)[0]
                fcan = not fcan
            fc[changed] ^= True
        flipped[changed] = True
    out = default.copy()
    if flipped[can_hap - 1]:
        w0 = int(s1[-1]) - 1                   # s1[i] with the loop's last i; an index into the UNFILTERED reads
        if w0 < nReads:
            out[w0:nReads] = 3 - out[w0:nReads]
    return out


def _determine_best_read_label_so_far_literal(read_label_matrix_all: np.ndarray, read_label_matrix_conf: np.ndarray,
                                     nReads: int, nGibbsSamples: int, can_hap: int) -> np.ndarray:
    """functions.R:1680-1784, line by line (including that the final flips start at the LAST change point,
    counted in the filtered row space: ``w <- s1[i]:nReads`` with the leftover loop variable)."""
    rl = read_label_matrix_all.astype(np.int64).copy()
    default = rl[:, can_hap - 1].astype(np.int32)
    a = rl.astype(np.float64)
    a[~read_label_matrix_conf] = np.nan
    a = a[~np.isnan(a).any(axis=1)]
    if a.shape[0] < 10:
        return default
    can = a[:, can_hap - 1].copy()
    a = a - can[:, None]
    s = np.nonzero(np.diff(np.abs(a).sum(axis=1)) != 0)[0] + 1  # R's which(), 1-based
    if len(s) == 0:
        return default
    s1 = np.concatenate([[1], s + 1])
    e1 = np.concatenate([s1[1:] - 1, [a.shape[0]]])
    flip = np.zeros((len(s1), nGibbsSamples), dtype=bool)
    nrow = a.shape[0]
    for i in range(2, len(s1) + 1):           # R: for(i in 2:length(s1))
        cur = a[s1[i - 1] - 1, :]
        changed = np.nonzero(cur != 0)[0]     # 0-based columns
        w = np.arange(s1[i - 1] - 1, nrow)
        if len(changed) > 0:
            if len(changed) <= nGibbsSamples / 2:
                for c1 in changed:
                    reverted = 3 - (a[w, c1] + can[w])
                    a[w, c1] = reverted - can[w]
            else:
                changed = np.nonzero(cur == 0)[0]
                for c1 in changed:
                    reverted = 3 - (a[w, c1] + can[w])
                    a[w, c1] = reverted - can[w]
                reverted = a[w, :] + can[w][:, None]
                can[w] = 3 - can[w]
                a[w, :] = reverted - can[w][:, None]
        flip[i - 1, changed] = True
    i_last = len(s1)                          # value of R's `i` after the loop
    for col in range(nGibbsSamples):
        if flip[:, col].any():
            w0 = s1[i_last - 1] - 1           # s1[i], 1-based -> 0-based row of the read matrix
            if w0 < nReads:
                rl[w0:nReads, col] = 3 - rl[w0:nReads, col]
    return rl[:, can_hap - 1].astype(np.int32)


def recast_haps(hd1: np.ndarray, hd2: np.ndarray, gp: np.ndarray):
    """functions.R:3180-3209: make the phased haplotype dosages agree with argmax genotype probability.
    ``gp`` is nSNPs x 3."""
    hd1 = hd1.copy()
    hd2 = hd2.copy()
    r_round = lambda x: np.round(x)  # R rounds half to even, as numpy does
    gt1 = r_round(hd1) + r_round(hd2)
    max_val = gp[:, 0].copy()
    gt3 = np.zeros(len(hd1))
    for i in (1, 2):
        w = gp[:, i] > max_val
        gt3[w] = i
        max_val[w] = gp[w, i]
    ch = gt3 != gt1
    z = ch & (gt3 == 0)
    hd1[z] = 0; hd2[z] = 0
    t = ch & (gt3 == 2)
    hd1[t] = 1; hd2[t] = 1
    o = ch & (gt3 == 1)
    a1, a2 = hd1[o].copy(), hd2[o].copy()
    hd1[o] = np.where(a1 > a2, 1.0, 0.0)
    hd2[o] = np.where(a1 > a2, 0.0, 1.0)
    return hd1, hd2


def determine_best_read_label_so_far_nipt(read_label_matrix_all: np.ndarray, read_label_matrix_conf: np.ndarray,
                                          nReads: int, nGibbsSamples: int, can_hap: int) -> np.ndarray:
    """functions.R:1788-1829: label 3 folded into 2 (and called not confident) for the consensus, then put back."""
    rl = read_label_matrix_all.copy()
    conf = read_label_matrix_conf.copy()
    three = rl == 3
    conf[three] = False
    rl[three] = 2
    out = determine_best_read_label_so_far(rl, conf, nReads, nGibbsSamples, can_hap)
    out = out.copy()
    out[three[:, can_hap - 1]] = 3
    return out


def recast_nipt_haps(hap1, hap2, hap3, mat_gp_t, fet_gp_t):
    """functions.R:3214-3287: phased haplotypes of mother and fetus made to agree with the argmax genotypes
    (``*_gp_t`` are 3 x nSNPs)."""
    hap1, hap2, hap3 = hap1.copy(), hap2.copy(), hap3.copy()
    gtMT = np.zeros(mat_gp_t.shape[1])
    gtFT = np.zeros(mat_gp_t.shape[1])
    mxA, mxB = mat_gp_t[0].copy(), fet_gp_t[0].copy()
    for i in (1, 2):
        w = mat_gp_t[i] > mxA
        gtMT[w] = i
        mxA[w] = mat_gp_t[i][w]
        w = fet_gp_t[i] > mxB
        gtFT[w] = i
        mxB[w] = fet_gp_t[i][w]
    conv = [(0, 0, 0, 0, 0), (0, 1, 0, 0, 1), (0, 2, 0, 0, 1), (1, 0, 0, 1, 0), (1, 2, 1, 0, 1), (2, 0, 1, 1, 0),
            (2, 1, 1, 1, 0), (2, 2, 1, 1, 1)]
    for m, f, a, b, c in conv:
        w = (gtMT == m) & (gtFT == f)
        hap1[w], hap2[w], hap3[w] = a, b, c
    w1 = np.nonzero((gtMT == 1) & (gtFT == 1))[0]
    r1, r2, r3 = np.round(hap1[w1]), np.round(hap2[w1]), np.round(hap3[w1])
    w2 = (r1 == 1) & (r2 == 0) & (r3 == 0)
    w3 = (r1 == 0) & (r2 == 1) & (r3 == 1)
    w4 = ~w2 & ~w3
    hap1[w1[w2]], hap2[w1[w2]], hap3[w1[w2]] = 1, 0, 0
    hap1[w1[w3]], hap2[w1[w3]], hap3[w1[w3]] = 0, 1, 1
    hap1[w1[w4]] = np.round(hap1[w1[w4]])
    hap2[w1[w4]] = np.round(hap2[w1[w4]])
    hap3[w1[w4]] = 1 - hap1[w1[w4]]
    return np.round(hap1), np.round(hap2), np.round(hap3)


# ---------------------------------------------------------------------------------------------
# the lock-step driver
# ---------------------------------------------------------------------------------------------

@dataclass
class ChainState:
    sample: object                    # the chain's SampleReads
    i_sample: int                     # index of the sample within its batch
    i_chain: int                      # 1..nGibbsSamples, nGibbsSamples + 1 = phasing
    rng: object                       # quilt_amd.rng.ChainStream
    which_haps_to_use: Optional[np.ndarray] = None   # 1-based
    read_labels: Optional[np.ndarray] = None
    hap: Optional[List[np.ndarray]] = None           # dosage1, dosage2 of the latest full pass
    hap_all: Optional[List[np.ndarray]] = None       # hap1_all, hap2_all of the rare + common call (all SNPs)

    _phasing: bool = False

    @property
    def phasing(self) -> bool:
        return self._phasing


@dataclass
class SampleResult:
    dosage: np.ndarray
    gp_t: np.ndarray                  # 3 x nSNPs
    phasing_haps: np.ndarray          # nSNPs x 2
    read_labels: np.ndarray           # consensus labels used by the phasing pass
    nDosage: int
    n_underflow_retries: int = 0
    # method = "nipt": dosage / gp_t are the mother's; the fetus':
    fet_dosage: Optional[np.ndarray] = None
    fet_gp_t: Optional[np.ndarray] = None


@dataclass
class _Batch:
    """One batch of samples in flight: its Gibbs chains and the accumulators of get_and_impute_one_sample."""
    samples: list
    offset: int                       # global index of samples[0] (keys the random streams)
    chains: List[ChainState]
    dosage: np.ndarray
    gp_t: np.ndarray
    nDosage: np.ndarray
    phasing: Optional[List[ChainState]] = None
    consensus: Optional[List[np.ndarray]] = None
    fet_dosage: Optional[np.ndarray] = None     # method = "nipt": dosage / gp_t hold the mother's
    fet_gp_t: Optional[np.ndarray] = None
    dosage_all: Optional[np.ndarray] = None     # impute_rare_common: the same accumulators over ALL SNPs
    gp_t_all: Optional[np.ndarray] = None
    nDosage_all: Optional[np.ndarray] = None


def get_initial_read_labels(e: np.ndarray, runif: np.ndarray) -> np.ndarray:
    """rare_common.R:61-107 (diploid): ``e`` = 2 x nReads likelihoods of the all-SNP reads against (hap1, hap2) spread
    over all SNPs (0.5 at the rare ones); ``H <- as.integer(runif(nReads) < e[1, ] / colSums(e)) + 1``."""
    return (runif < (e[0] / (e[0] + e[1]))).astype(np.int32) + 1


def preserve_round(x: np.ndarray) -> np.ndarray:
    """gibbs-nipt.R:1779-1789: round to integers keeping the sum (the largest fractional parts go up)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.floor(x)
    n_up = int(np.round(x.sum()) - y.sum())
    if n_up > 0:
        idx = np.argsort(x - y, kind="stable")[-n_up:]
        y[idx] += 1
    if abs(x.sum() - y.sum()) > 0.1:
        raise RuntimeError("preserve round has not worked")
    return y.astype(np.int64)


def get_initial_read_labels_nipt(e: np.ndarray, ff: float, rng) -> np.ndarray:
    """rare_common.R:104-105 with get_read_groupings_given_fetal_fraction_and_cov and sample_H_for_NIPT_given_groupings
    (gibbs-nipt.R:1655-1777, :1796-1849): ``e`` = 3 x nReads rescaled likelihoods of the all-SNP reads against
    (hap1, hap2, hap3).  A read is grouped by which haplotypes it fits (> 0.5); reads that fit exactly one take its label,
    reads shared by two are split between them in the ratio of the label priors (rounded, sum preserved) and drawn with
    those proportions, the rest follow the prior.  (The reference also draws a down-sampled read set here whose result
    is not used; it is not drawn.)"""
    frp = np.array([0.5, 0.5 - ff / 2, ff / 2])
    with np.errstate(invalid="ignore"):
        m = [np.where(np.isnan(e[i]), False, e[i] > 0.5) for i in range(3)]
    n_fit = m[0].astype(int) + m[1].astype(int) + m[2].astype(int)
    R = e.shape[1]
    H = np.zeros(R, dtype=np.int32)
    every = (n_fit == 3) | (n_fit == 0)
    H[every] = rng.choice(3, size=int(every.sum()), p=frp) + 1
    for i in range(3):
        H[(n_fit == 1) & m[i]] = i + 1
    for i in range(2):
        for j in range(i + 1, 3):
            both = (n_fit == 2) & m[i] & m[j]
            n = int(both.sum())
            if n == 0:
                continue
            f1 = frp[i] / (frp[i] + frp[j])
            a, b = preserve_round(n * np.array([f1, 1 - f1]))
            H[both] = np.where(rng.random(n) < a / (a + b), i + 1, j + 1)
    return H


class PhasingTail:
    """Meeting point of the host threads of one device for the END of their streams.

    A thread's stream pipelines batch i's phasing rounds with batch i + 1's main rounds, so the last batch of every thread
    is left with three rounds of one chain per sample -- a Gibbs launch that costs a chain's serial time for an eighth of
    the device.  With several threads these tails would run one after the other, each between the launches of the threads
    still in their main rounds.  Here a thread whose stream has drained leaves its last batch and waits; the thread that
    drains LAST runs all the tails together (one launch per round) and hands every batch back to its owner, who finishes
    it (``Driver._finish``) on its own thread.  Results are unchanged: every chain owns its random stream."""

    def __init__(self, n_threads: int):
        import threading
        self.lock = threading.Lock()
        self.n_active = n_threads       # threads that have not drained yet
        self.waiting = []               # (batch, Future) left by drained threads
        self.failed: Optional[BaseException] = None

    def drained(self, batch):
        """Called once per thread when its stream has no main rounds left; ``batch``: its last batch (phasing rounds
        pending) or None.  Returns (run, fut): the last thread gets ``run`` = every waiting (batch, Future) and runs them with
        its own; an earlier thread gets ``fut``, resolved (to the batch itself) once its rounds have been run."""
        from concurrent.futures import Future
        with self.lock:
            if self.failed is not None:
                raise RuntimeError("another host thread of this device failed") from self.failed
            self.n_active -= 1
            if self.n_active <= 0:
                run, self.waiting = self.waiting, []
                return run, None
            if batch is None:
                return [], None
            fut = Future()
            self.waiting.append((batch, fut))
            return None, fut

    def abort(self, exc: BaseException):
        """A thread failed: nobody may wait for it (or for rounds it had taken over)."""
        with self.lock:
            if self.failed is None:
                self.failed = exc
            waiting, self.waiting = self.waiting, []
        for _, fut in waiting:
            if not fut.done():
                fut.set_exception(exc)


class Driver:
    """Runs ``get_and_impute_one_sample`` (quilt.R:688-996, functions.R:420-1259) for batches of samples, all chains of
    a batch in lock-step.

    Batches are software-pipelined: the phasing rounds of batch i (one chain per sample) share their launches with the
    main rounds of batch i + 1 (nGibbsSamples chains per sample).  A Gibbs chain is a serial walk over the reads, so a
    launch costs the same whether it carries 128 or 1024 chains (one per SIMD); fusing the two halves the number of
    latency-bound launches.  Results do not depend on the batching: every chain owns its random stream.
    """

    def __init__(self, panel, backend, params: Optional[DriverParams] = None, rare_common=None):
        """``rare_common`` (:class:`quilt_amd.panel.RareCommon`), with ``params.impute_rare_common``: every sample then
        carries its all-SNP reads as ``sample.all_snp`` and the results cover all SNPs (functions.R:1306-1307)."""
        self.panel = panel
        self.backend = backend
        self.params = (params or DriverParams()).resolved(panel.K)
        self.rare_common = rare_common
        if self.params.impute_rare_common and rare_common is None:
            raise ValueError("impute_rare_common needs the panel's rare/common tables")
        if self.params.method not in ("diploid", "nipt"):
            raise ValueError("method is 'diploid' or 'nipt'")
        self.n_label = 3 if self.params.method == "nipt" else 2
        self.cols = thinned_grid_columns(panel.nGrids, self.params.heuristic_match_thin)
        self.n_thin = int((self.cols >= 0).sum())
        self.top_width = max(8, self.params.K_top_matches)   # entries kept per (label, thinned grid) list
        self.timing = {"gibbs": 0.0, "fullpass": 0.0, "host": 0.0, "consensus": 0.0, "finish": 0.0, "accumulate": 0.0,
                       "new_batch": 0.0}
        self.n_full_list_refetches = 0   # chains whose selection needed the untruncated best-haplotype lists
        self.n_underflow_retries = 0     # Gibbs calls repeated with a smaller maxDifferenceBetweenReads
        self.n_gibbs_chain_calls = 0     # chains handed to the Gibbs entry point, retries included
        self.n_device_selections = 0     # chains whose next small panel was chosen by csrc/select.hip
        self._zero_hap = None
        self.gibbs_gate = None           # workers.PairGate shared by the host threads of a device, or None
        self.phasing_tail = None         # PhasingTail shared by the host threads of a device, or None
        self.on_first_launch = None      # called once, right before this driver's next Gibbs call (workers: staggered start)
        self._round_dosages = None
        self._round_dosage_chains = []

    # -- one [Gibbs -> full pass -> select] round over a set of chains (main and / or phasing chains, same i_it)
    def _round(self, chains: List[ChainState], i_it: int):
        import time
        P = self.params
        K, G, T = self.panel.K, self.panel.nGrids, self.panel.nSNPs
        t0 = time.perf_counter()
        any_first = False
        starts, seed_reads, first_reads, seed_shards = [], [], [], []
        for ch in chains:
            R = ch.sample.nReads
            if R < 1:   # the reference drops such samples before imputing (functions.R:300-310: "sample has no reads")
                raise ValueError("a sample without reads cannot be imputed (no read intersects a SNP of the region)")
            first = (i_it == 1) and not ch.phasing
            any_first |= first
            if first:
                # functions.R:579-585
                ch.which_haps_to_use = np.sort(ch.rng.choice(K, size=P.Ksubset, replace=False) + 1).astype(np.int32)
                if P.method == "nipt":   # functions.R:586
                    ff = ch.sample.ff
                    H0 = (ch.rng.choice(3, size=R, p=[0.5, 0.5 - ff / 2, ff / 2]) + 1).astype(np.int32)
                else:
                    H0 = ch.rng.integers(1, 3, size=R).astype(np.int32)
            else:
                H0 = ch.read_labels
            starts.append(H0)
            # seeds of the counter-based streams that stand in for Rcpp::runif (quilt_amd/rng.py)
            seed_reads.append(int(ch.rng.integers(0, 2 ** 63)))
            fr = int(ch.rng.integers(0, R))
            # only first-round chains initialise iteratively (functions.R:2369); in a launch that also carries
            # later-round (phasing) chains a negative first_read marks those (include/quilt_amd.h)
            first_reads.append(fr if first else -1)
            seed_shards.append(int(ch.rng.integers(0, 2 ** 63)))
        if not any_first:
            first_reads = [0] * len(chains)   # unused without gibbs_initialize_iteratively
        t1 = time.perf_counter()
        self.timing["host"] += t1 - t0
        trace.add("round:prepare", t0, t1)
        # ---- small-panel Gibbs (impute_one_sample, functions.R:2313-2774), with the underflow retry
        # use_mspbwt: the haplotype search wants the call's rounded, packed haploid dosages (formed on the device); the
        # dosages themselves only on the rounds that accumulate them (the last round's also feed the read-confidence step)
        extra = {"return_hap_words": True, "return_hapProbs": i_it > P.n_burn_in_seek_its} if P.use_mspbwt else {}
        results = self._gibbs_with_retry(chains, [ch.sample for ch in chains], starts, seed_reads, first_reads, seed_shards,
                                         gibbs_initialize_iteratively=any_first, **extra)
        t2 = time.perf_counter()
        self.timing["gibbs"] += t2 - t1
        trace.add("round:gibbs_call", t1, t2)
        # ---- full-panel pass per read label (impute_using_everything, functions.R:1922-2157)
        return_dosage = i_it > P.n_burn_in_seek_its
        for ch, res in zip(chains, results):
            ch.read_labels = res["double_list_of_ending_read_labels"][0][0].astype(np.int32)
        if P.use_mspbwt:
            self._round_mspbwt(chains, results, i_it, return_dosage)
            self.timing["fullpass"] += time.perf_counter() - t2
            return return_dosage
        uniq, sample_list = {}, []
        for ch in chains:
            if id(ch.sample) not in uniq:
                uniq[id(ch.sample)] = len(sample_list)
                sample_list.append(ch.sample)
        t3 = time.perf_counter()
        self.timing["host"] += t3 - t2
        trace.add("round:labels", t2, t3)
        # The reference asks for the best haplotypes on every call (functions.R:738-743), but the selection made from
        # them is read again only by a later round of the same chain or -- the last chain's final selection -- by the
        # phasing rounds (which_haps_to_use carried over): skip the lists nobody reads.
        # (with impute_rare_common every chain's final selection feeds its all-SNP Gibbs call)
        want_top = [i_it < P.n_seek_its or P.impute_rare_common or (not ch.phasing and ch.i_chain == P.nGibbsSamples)
                    for ch in chains]
        # everything_select_good_haps: one selection stream per chain and round; its draws are keyed (quilt_amd/rng.py), so
        # the device (csrc/select.hip, behind the full-panel call) and the host make the same choice
        seed_sel = [int(ch.rng.integers(0, 2 ** 63)) for ch in chains]
        on_device = bool(getattr(self.backend, "select_on_device", False)) and any(want_top)
        select = None
        if on_device:
            select = dict(Ksubset=P.Ksubset, Knew=P.Knew, which=[ch.which_haps_to_use for ch in chains], seeds=seed_sel)
        out = self.backend.fullpass_reads_batch(
            sample_list, [uniq[id(ch.sample)] for ch in chains], [ch.read_labels for ch in chains],
            [return_dosage] * len(chains), want_top, self.cols, P.K_top_matches, P.minGLValue, self.top_width,
            n_label=self.n_label, **({"select": select} if on_device else {}))
        dosages, top, top_cnt = out[:3]
        which_next, sel_status = out[3:5] if on_device else (None, None)
        t4 = time.perf_counter()
        self.timing["fullpass"] += t4 - t3
        trace.add("round:fullpass_call", t3, t4)
        if return_dosage and (dosages.min() < -1e-5 or dosages.max() > 1 + 1e-5):   # functions.R:2072-2075
            raise RuntimeError("Dosage observed outside of range of 0 to 1 on forward-backward full iteration")
        self._round_dosages = dosages if return_dosage else None   # [chain, label, T]: run_stream accumulates from it
        self._round_dosage_chains = list(chains) if return_dosage else []
        if self._zero_hap is None or len(self._zero_hap) != T:
            self._zero_hap = np.zeros(T)
            self._zero_hap.flags.writeable = False
        for ci, ch in enumerate(chains):
            if return_dosage:
                ch.hap = [dosages[ci][l] for l in range(self.n_label)]
            else:
                ch.hap = [self._zero_hap] * self.n_label
            if not want_top[ci]:
                continue
            if on_device and sel_status[ci] == 0:
                ch.which_haps_to_use = which_next[ci].astype(np.int32)
                self.n_device_selections += 1
                continue
            prev_sel = previously_selected(ch.which_haps_to_use, P.Ksubset - P.Knew, seed_sel[ci])
            try:
                if on_device:   # the device ran out of ranked candidates (status 1): the lists did not come back
                    raise ListsTruncated()
                sel = everything_select_good_haps_dense(P.Knew, P.K_top_matches, top[ci].astype(np.int64) + 1, prev_sel, K,
                                                        seed_sel[ci], truncated=bool((top_cnt[ci] > self.top_width).any()))
            except ListsTruncated:
                # the reference's lists hold every haplotype at or above the threshold (reference-single.cpp:129-194); the
                # batched call returns their first top_width entries.  Ties made a list longer and the selection ran out
                # of ranked candidates: fetch this chain's full lists and select from them (functions.R:2278-2281)
                self.n_full_list_refetches += 1
                new_haps = self._full_lists(ch)
                sel = everything_select_good_haps(P.Knew, P.K_top_matches, new_haps, prev_sel, K, seed_sel[ci])
            ch.which_haps_to_use = np.concatenate([prev_sel, sel]).astype(np.int32)
        t5 = time.perf_counter()
        self.timing["host"] += t5 - t4
        trace.add("round:select", t4, t5)
        return return_dosage

    def _round_mspbwt(self, chains: List[ChainState], results, i_it: int, return_dosage: bool):
        """use_mspbwt = TRUE (functions.R:784-893): the haploid dosages are the Gibbs call's hapProbs_t; the next small panel
        comes from the long matches of their rounded form against the whole panel (select_new_haps_mspbwt_v3,
        mspbwt.R:225-474) -- searched on the device for all chains and labels at once."""
        from .mspbwt import int_contract_rows, select_new_haps_mspbwt_batch
        P = self.params
        T, nL = self.panel.nSNPs, self.n_label
        if return_dosage:
            allh = results[0].get("hap_major_all")
            if (allh is not None and allh.shape == (len(chains), nL, T) and
                    all(r.get("hap_major_all") is allh for r in results)):
                dosages = allh   # the backend's [chain, label, SNP] buffer itself, in chain order: nothing to copy
            else:
                dosages = np.empty((len(chains), nL, T))
                for ci, res in enumerate(results):
                    dosages[ci] = np.asarray(res["hapProbs_t"])[:nL]
            for ci, ch in enumerate(chains):
                ch.hap = [dosages[ci][l] for l in range(nL)]
        else:
            dosages = None
            if self._zero_hap is None or len(self._zero_hap) != T:
                self._zero_hap = np.zeros(T)
                self._zero_hap.flags.writeable = False
            for ch in chains:
                ch.hap = [self._zero_hap] * nL
        self._round_dosages = dosages
        self._round_dosage_chains = list(chains) if return_dosage else []
        want = [i_it < P.n_seek_its or P.impute_rare_common or (not ch.phasing and ch.i_chain == P.nGibbsSamples)
                for ch in chains]
        idx = [ci for ci, w in enumerate(want) if w]
        if not idx:
            return
        seed_sel = {ci: int(chains[ci].rng.integers(0, 2 ** 63)) for ci in idx}
        # per chain its nL haplotypes, rounded and packed (mspbwt.R:271-272)
        if "hap_words" in results[idx[0]]:
            Zs = np.concatenate([np.asarray(results[ci]["hap_words"])[:nL] for ci in idx])
        else:
            Zs = int_contract_rows(np.concatenate([np.asarray(results[ci]["hapProbs_t"])[:nL] for ci in idx]))
        if P.mspbwt_search == "scan":
            # the reference's query: the neighbour scan of the panel's msPBWT indices, then the selection, natively per chain
            new = self.backend.mspbwt_select(Zs, nL, P.mspbwt_nindices, P.mspbwtL, P.mspbwtM, P.Knew,
                                             [seed_sel[ci] for ci in idx])
        else:
            n_max = P.mspbwt_max_matches or 50 * P.mspbwtL
            match, n_match = self.backend.find_good_matches(Zs, P.mspbwt_nindices, P.mspbwtM, n_max)
            new = select_new_haps_mspbwt_batch(match, n_match, nL, P.Knew, self.panel.K, self.panel.nGrids,
                                               [seed_sel[ci] for ci in idx])
        for a, ci in enumerate(idx):
            chains[ci].which_haps_to_use = new[a].copy()

    def _full_lists(self, ch: ChainState) -> List[List[np.ndarray]]:
        """``new_haps`` of one chain from complete best-haplotype lists: per read label its gl (make_gl_from_u_bq,
        reference-single.R:19-42), a thin full-panel pass returning the whole lists, ordered per thinned grid as
        everything_per_hap_rejig_haps does."""
        P = self.params
        s = ch.sample
        per_base = np.repeat(ch.read_labels, np.diff(s.read_ptr))
        gls = []
        for l in range(1, self.n_label + 1):
            sel = (per_base == l) & (s.bq != 0)
            gls.append(make_gl_from_u_bq(s.u[sel], s.bq[sel], self.panel.nSNPs, P.minGLValue, self.backend.make_gl_bound))
        _, best = self.backend.fullpass_batch(gls, [0] * len(gls), self.cols, P.K_top_matches)
        return [everything_per_hap_rejig_haps(b) for b in best]

    def _gibbs_with_retry(self, chains, samples, starts, seed_reads, first_reads, seed_shards, **kw):
        """impute_one_sample's loop (functions.R:2612-2716): a chain whose call reports underflow is re-run with
        maxDifferenceBetweenReads / 10."""
        P = self.params
        pending = list(range(len(chains)))
        maxdiff = [P.maxDifferenceBetweenReads] * len(chains)
        results = [None] * len(chains)
        n_try = 0
        while pending:
            groups = {}
            for i in pending:
                groups.setdefault(maxdiff[i], []).append(i)
            nxt = []
            for md, idx in groups.items():
                if P.method == "nipt":   # every chain carries its sample's fetal fraction (functions.R:128)
                    kw = dict(kw, ff=[float(chains[i].sample.ff) for i in idx], shuffle_bin_radius=P.shuffle_bin_radius)
                if self.gibbs_gate is not None and n_try == 0:
                    self.gibbs_gate.wait()   # start together with the other host thread's launch (workers.PairGate)
                self.n_gibbs_chain_calls += len(idx)
                if self.on_first_launch is not None:
                    cb, self.on_first_launch = self.on_first_launch, None
                    cb()
                out = self.backend.gibbs_batch(
                    [samples[i] for i in idx], [chains[i].which_haps_to_use for i in idx],
                    [starts[i] for i in idx], [seed_reads[i] for i in idx], [first_reads[i] for i in idx],
                    [seed_shards[i] for i in idx], n_gibbs_burn_in_its=P.small_ref_panel_gibbs_iterations,
                    n_gibbs_sample_its=P.n_gibbs_sample_its,
                    block_gibbs_iterations=P.small_ref_panel_block_gibbs_iterations,
                    maxDifferenceBetweenReads=md, Jmax_local=P.Jmax, **kw)
                if len(groups) > 1 or n_try > 0:
                    # the backend's dosage buffer is reused by its next call: results of a call that is not the round's only
                    # one keep copies
                    for o in out:
                        if o.get("hap_major_all") is not None:
                            o["hapProbs_t"] = np.array(o["hapProbs_t"])
                            o["hap_major_all"] = None
                for i, o in zip(idx, out):
                    if o["underflow_problem"]:
                        maxdiff[i] = max(1.0, maxdiff[i] / 10)   # functions.R:2704-2715
                        nxt.append(i)
                        self.n_underflow_retries += 1
                    else:
                        results[i] = o
            if nxt:   # another call follows and reuses the backend's dosage buffer: what has been accepted keeps copies
                for o in results:
                    if o is not None and o.get("hap_major_all") is not None:
                        o["hapProbs_t"] = np.array(o["hapProbs_t"])
                        o["hap_major_all"] = None
            pending = nxt
            n_try += 1
            if n_try > 10 and pending:
                raise RuntimeError("There were consecutive underflow problems (functions.R:2710)")
        return results

    def _rare_common_round(self, chains: List[ChainState]):
        """impute_final_gibbs_with_rare_common (rare_common.R:109-420), once per Gibbs sample after its seek
        iterations (functions.R:1042-1098): starting labels from the all-SNP reads against the latest (hap1, hap2)
        (get_initial_read_labels), then one Gibbs call over ALL SNPs with the haplotypes selected last."""
        import time
        P, rc = self.params, self.rare_common
        t0 = time.perf_counter()
        common = rc.snp_is_common == 1
        reads = [ch.sample.all_snp for ch in chains]
        nL = self.n_label
        # eHapsCurrent_tc of get_initial_read_labels (rare_common.R:76-81) for every chain at once: [chain, SNP, haplotype],
        # 0.5 at the rare SNPs, the latest full-pass dosages at the common ones
        haps = np.full((len(chains), rc.nSNPs_all, nL), 0.5)
        cidx = np.flatnonzero(common)
        for l in range(nL):
            haps[:, cidx, l] = np.stack([ch.hap[l] for ch in chains])
        lik = self.backend.read_likelihood_all_snps_batch(reads, haps, P.maxDifferenceBetweenReads)
        starts, seed_reads, seed_shards = [], [], []
        for ch, e in zip(chains, lik):
            if P.method == "nipt":
                starts.append(get_initial_read_labels_nipt(e, ch.sample.ff, ch.rng))
            else:
                starts.append(get_initial_read_labels(e, ch.rng.random(e.shape[1])))
            seed_reads.append(int(ch.rng.integers(0, 2 ** 63)))
            seed_shards.append(int(ch.rng.integers(0, 2 ** 63)))
        t1 = time.perf_counter()
        self.timing["host"] += t1 - t0
        # impute_one_sample's defaults for this call (functions.R:2385-2409): labels given, read categories off
        results = self._gibbs_with_retry(chains, reads, starts, seed_reads, [0] * len(chains), seed_shards,
                                         gibbs_initialize_iteratively=False, rare_common=True)
        for ch, res in zip(chains, results):
            ch.hap_all = [res["hapProbs_t"][l] for l in range(nL)]
        self.timing["gibbs"] += time.perf_counter() - t1

    def _new_batch(self, samples, offset: int) -> _Batch:
        P = self.params
        T = self.panel.nSNPs
        N = len(samples)
        chains = [ChainState(samples[i], i, c, chain_rng(P.seed, offset + i, c))
                  for i in range(N) for c in range(1, P.nGibbsSamples + 1)]
        b = _Batch(list(samples), offset, chains, np.zeros((N, T)), np.zeros((N, 3, T)), np.zeros(N, dtype=np.int64))
        if P.method == "nipt" and not P.impute_rare_common:
            b.fet_dosage, b.fet_gp_t = np.zeros((N, T)), np.zeros((N, 3, T))
        if P.impute_rare_common:
            Ta = self.rare_common.nSNPs_all
            b.dosage_all, b.gp_t_all, b.nDosage_all = np.zeros((N, Ta)), np.zeros((N, 3, Ta)), np.zeros(N, dtype=np.int64)
            if P.method == "nipt":
                b.fet_dosage, b.fet_gp_t = np.zeros((N, Ta)), np.zeros((N, 3, Ta))
        return b

    def _start_phasing(self, b: _Batch):
        """Read confidence per chain and consensus labels (functions.R:1144-1205); one phasing chain per sample."""
        P = self.params
        # the chains' haploid dosages: rows of the last round's [chain, label, SNP] array (b's chains come first in a round)
        rd, rc = self._round_dosages, self._round_dosage_chains
        n = len(b.chains)
        if isinstance(rd, np.ndarray) and len(rc) >= n and all(x is y for x, y in zip(b.chains, rc)):
            haps = rd[:n]
        else:
            haps = [ch.hap for ch in b.chains]
        conf = self.backend.read_confidence_batch([ch.sample for ch in b.chains], haps, P.maxDifferenceBetweenReads)
        b.phasing = []
        from .io import consensus_read_labels
        by_sample: dict = {}
        for k, ch in enumerate(b.chains):
            by_sample.setdefault(ch.i_sample, []).append(k)
        for i, smp in enumerate(b.samples):
            mine = by_sample.get(i, [])
            # assess_ability_of_reads_to_be_confident + determine_best_read_label_so_far(_nipt) (functions.R:1615-1660, :1680-1829),
            # native; the numpy text of both stays in this module as the tested statement of what it does
            labels = consensus_read_labels(np.stack([b.chains[k].read_labels for k in mine]),
                                           np.stack([np.asarray(conf[k], dtype=np.float64) for k in mine]),
                                           can_hap=P.nGibbsSamples)
            last = b.chains[mine[-1]]
            b.phasing.append(ChainState(smp, i, P.nGibbsSamples + 1, chain_rng(P.seed, b.offset + i, P.nGibbsSamples + 1),
                                        which_haps_to_use=last.which_haps_to_use.copy(), read_labels=labels, _phasing=True))
        b.consensus = [ph.read_labels.copy() for ph in b.phasing]
        b.chains = []   # the main chains are done

    def _finish(self, b: _Batch) -> List[SampleResult]:
        out = []
        rc = self.params.impute_rare_common
        for i in range(len(b.samples)):
            if rc:   # the switch-over to all SNPs (functions.R:1232-1252, 1305-1317)
                d = b.dosage_all[i] / b.nDosage_all[i]
                g = b.gp_t_all[i] / b.nDosage_all[i]
                if self.params.method == "nipt":
                    fd, fg = b.fet_dosage[i] / b.nDosage_all[i], b.fet_gp_t[i] / b.nDosage_all[i]
                    ph = b.phasing[i].hap_all
                    h1, h2, h3 = recast_nipt_haps(ph[0], ph[1], ph[2], g, fg)
                    out.append(SampleResult(d, g, np.stack([h1, h2, h3], axis=1), b.consensus[i], int(b.nDosage_all[i]),
                                            fet_dosage=fd, fet_gp_t=fg))
                    continue
                h1, h2 = recast_haps(b.phasing[i].hap_all[0], b.phasing[i].hap_all[1], g.T)
                out.append(SampleResult(d, g, np.stack([h1, h2], axis=1), b.consensus[i], int(b.nDosage_all[i])))
                continue
            d = b.dosage[i] / b.nDosage[i]
            g = b.gp_t[i] / b.nDosage[i]
            if self.params.method == "nipt":   # functions.R:1218-1231, 1313-1317
                fd, fg = b.fet_dosage[i] / b.nDosage[i], b.fet_gp_t[i] / b.nDosage[i]
                ph = b.phasing[i].hap
                h1, h2, h3 = recast_nipt_haps(ph[0], ph[1], ph[2], g, fg)
                out.append(SampleResult(d, g, np.stack([h1, h2, h3], axis=1), b.consensus[i], int(b.nDosage[i]),
                                        fet_dosage=fd, fet_gp_t=fg))
                continue
            h1, h2 = recast_haps(b.phasing[i].hap[0], b.phasing[i].hap[1], g.T)   # functions.R:1207-1217
            out.append(SampleResult(d, g, np.stack([h1, h2], axis=1), b.consensus[i], int(b.nDosage[i])))
        return out

    def run_stream(self, batches):
        """``batches``: iterable of ``(samples, sample_offset)``; yields one list of SampleResult per batch, in order.
        The phasing rounds of a batch run fused with the main rounds of the next one; the last batch's run alone, or -- with
        a :class:`PhasingTail` shared by the host threads of a device -- together with the other threads' last batches."""
        import time
        P = self.params
        prev: Optional[_Batch] = None
        it = iter(batches)
        tail = self.phasing_tail
        reported = False          # this thread has told the tail that its stream drained
        taken = []                # (batch, Future) of other threads' last batches run here
        try:
            while True:
                nxt = next(it, None)
                t_nb = time.perf_counter()
                cur = self._new_batch(*nxt) if nxt is not None else None
                self.timing["new_batch"] += time.perf_counter() - t_nb
                taken = []            # (batch, Future) of other threads' last batches run here
                if cur is None and tail is not None and not reported:
                    reported = True
                    run, fut = tail.drained(prev)
                    if fut is not None:   # a later thread runs prev's phasing rounds with its own
                        if self.gibbs_gate is not None and not getattr(self, "_gate_left", False):
                            self.gibbs_gate.leave()
                            self._gate_left = True
                        with span("tail:wait"):
                            b = fut.result()
                        t0 = time.perf_counter()
                        with span("finish"):
                            done = self._finish(b)
                        self.timing["finish"] += time.perf_counter() - t0
                        yield done
                        return
                    taken = run
                if cur is None and prev is None and not taken:
                    return
                others = [ch for b, _ in taken for ch in b.phasing]
                phasing = (prev.phasing if prev else []) + others
                for i_it in range(1, P.n_seek_its + 1):
                    chains = (cur.chains if cur else []) + phasing
                    with span("round"):
                        stored = self._round(chains, i_it)
                    if stored and cur:   # functions.R:999-1020 (1009-1016: fetus = maternal transmitted + paternal transmitted)
                        from .io import accumulate_dosage
                        t_acc = time.perf_counter()
                        n_cur = len(cur.chains)           # the round's chains: cur's first, then the phasing chains
                        fetal = P.method == "nipt" and not P.impute_rare_common
                        accumulate_dosage(np.ascontiguousarray(self._round_dosages[:n_cur], dtype=np.float64),
                                          [ch.i_sample for ch in cur.chains], cur.dosage, cur.gp_t,
                                          cur.fet_dosage if fetal else None, cur.fet_gp_t if fetal else None)
                        for ch in cur.chains:
                            cur.nDosage[ch.i_sample] += 1
                        self.timing["accumulate"] += time.perf_counter() - t_acc
                if P.impute_rare_common:   # functions.R:1042-1123
                    self._rare_common_round((cur.chains if cur else []) + phasing)
                    for ch in (cur.chains if cur else []):
                        h1, h2 = ch.hap_all[0], ch.hap_all[1]
                        cur.dosage_all[ch.i_sample] += h1 + h2
                        cur.gp_t_all[ch.i_sample] += np.stack([(1 - h1) * (1 - h2), (1 - h1) * h2 + h1 * (1 - h2), h1 * h2])
                        if P.method == "nipt":   # functions.R:1113-1120
                            h3 = ch.hap_all[2]
                            cur.fet_dosage[ch.i_sample] += h1 + h3
                            cur.fet_gp_t[ch.i_sample] += np.stack([(1 - h1) * (1 - h3), (1 - h1) * h3 + h1 * (1 - h3), h1 * h3])
                        cur.nDosage_all[ch.i_sample] += 1
                # The other threads' batches go back to their owners, who finish them on their own threads.  Their phasing
                # chains' dosages are rows of THIS backend's transfer buffer: it is not written again before the owners are
                # done (this thread has no round left, and a later stream starts only after every result was handed out).
                for b, fut in taken:
                    fut.set_result(b)
                t0 = time.perf_counter()
                with span("finish"):
                    done = self._finish(prev) if prev else None
                t1 = time.perf_counter()
                if cur:
                    with span("start_phasing"):
                        self._start_phasing(cur)
                self.timing["finish"] += t1 - t0
                self.timing["consensus"] += time.perf_counter() - t1
                prev = cur
                if done is not None:
                    yield done
        except BaseException as e:
            # Batches this thread took over live only in ``taken`` (the tail's waiting list is empty by then): their owners
            # wait on these futures, so a failure inside the fused tail rounds must reach them here.
            err = e if not isinstance(e, GeneratorExit) else RuntimeError(
                "the thread running the last batches' phasing rounds stopped early")
            for _, fut in taken:
                if not fut.done():
                    fut.set_exception(err)
            if tail is not None and not isinstance(e, GeneratorExit):
                tail.abort(e)
            raise
        finally:
            if tail is not None and not reported:   # (a consumer that stopped early: the others must not wait for this thread)
                try:
                    run, _ = tail.drained(None)
                    if run:
                        err = RuntimeError("the thread that would have run the last batches' phasing rounds stopped early")
                        for _, fut in run:
                            fut.set_exception(err)
                except RuntimeError:
                    pass

    def run(self, samples, sample_offset: int = 0) -> List[SampleResult]:
        """One batch.  ``sample_offset``: global index of ``samples[0]`` (keys the random streams, so that a sample
        gets the same draws whichever rank / batch it lands in)."""
        return next(self.run_stream([(samples, sample_offset)]))


# ---------------------------------------------------------------------------------------------
# product backend: the HIP library (no fallback)
# ---------------------------------------------------------------------------------------------

class HipBackend:
    select_on_device = True   # everything_select_good_haps behind the full-panel call (csrc/select.hip)

    def __init__(self, device_panel, device_rare_common=None):
        self.dev = device_panel
        self.drc = device_rare_common   # quilt_amd.native.DeviceRareCommon, for impute_rare_common
        self._dosage_buf = None         # pinned host buffer of the dosage rounds (fullpass_reads_batch)
        self._hap_buf = None            # the same for the Gibbs call's own haploid dosages (use_mspbwt = TRUE)

    def make_gl_bound(self, gl, minGLValue, to_fix):
        from .reference_single import Rcpp_make_gl_bound
        Rcpp_make_gl_bound(gl, minGLValue, to_fix)

    def gibbs_batch(self, samples, which, starts, seed_reads, first_reads, seed_shards, rare_common=False, **kw):
        from .gibbs_nipt import forwardBackwardGibbsNIPT_batch
        if rare_common:   # the all-SNP call: hapProbs_t is its result (rare_common.R:401-407)
            if self.drc is None:
                raise ValueError("this backend was created without the rare/common tables")
            return forwardBackwardGibbsNIPT_batch(self.dev, samples, which, starts, None, first_reads, None,
                                                  seed_reads=seed_reads, seed_shard=seed_shards, return_hapProbs=True,
                                                  return_genProbs=False, disable_read_category_usage=True,
                                                  rare_common=self.drc, **kw)
        want_hap = bool(kw.pop("return_hapProbs", False))   # use_mspbwt = TRUE: the call's own haploid dosages
        hap_out = None
        if want_hap:
            # label by label into this backend's pinned buffer (no staging copy, no per-chain transposes on the host): valid
            # until the next call that asks for dosages -- the driver consumes a round before it starts the next
            n_label = 3 if np.ndim(kw.get("ff", 0.0)) > 0 or kw.get("ff", 0.0) != 0 else 2
            need = len(samples) * n_label * self.dev.panel.nSNPs
            if self._hap_buf is None or self._hap_buf.size < need:
                from .native import pinned_empty
                self._hap_buf = None
                self._hap_buf = pinned_empty((need + need // 8,))
            hap_out = self._hap_buf[:need].reshape(len(samples), n_label, self.dev.panel.nSNPs)
        return forwardBackwardGibbsNIPT_batch(self.dev, samples, which, starts, None, first_reads, None,
                                              seed_reads=seed_reads, seed_shard=seed_shards, return_hapProbs=False,
                                              return_hap_words=bool(kw.pop("return_hap_words", False)),
                                              return_genProbs=False, hap_major_out=hap_out, **kw)

    def find_good_matches(self, Zs, nindices, min_len, max_matches):
        from .mspbwt import find_good_matches
        return find_good_matches(self.dev, Zs, nindices, min_len, max_matches)

    def mspbwt_select(self, Zs, n_label, nindices, L, M, Knew, seeds):
        from .mspbwt import panel_mspbwt_index
        with span("device:mspbwt_scan"):   # (host code: the name keeps the trace's columns)
            return panel_mspbwt_index(self.dev.panel, nindices).select_new_haps(Zs, n_label, L, M, Knew, seeds)

    def read_likelihood_all_snps_batch(self, samples_all, haps, maxDifferenceBetweenReads):
        """rcpp_make_eMatRead_t as get_initial_read_labels calls it (rare_common.R:82-98): rescaled, Jmax = 100.
        ``haps``: [chain, SNP, haplotype] array over all SNPs."""
        from .gibbs_nipt import calculate_eMatRead_t_vs_haplotypes_batch
        return calculate_eMatRead_t_vs_haplotypes_batch(self.dev, samples_all, haps, maxDifferenceBetweenReads,
                                                        rescale_eMatRead_t=True, Jmax=100, nSNPs=self.drc.rc.nSNPs_all)

    def fullpass_batch(self, gls, want_dosage, cols, K_top_matches):
        import ctypes as C
        from .native import QA_ERR_CAPACITY, check, lib, ptr
        P = self.dev.panel
        n = len(gls)
        T = P.nSNPs
        n_thin = int((cols >= 0).sum())
        gl = np.ascontiguousarray(np.stack([np.ascontiguousarray(g.T) for g in gls]))
        wd = np.ascontiguousarray(want_dosage, dtype=np.int32)
        dosage = np.zeros((n, T))
        bptr = np.zeros(n * n_thin + 1, dtype=np.int32)
        cap = n * n_thin * 16
        for _ in range(2):
            bidx = np.zeros(cap, dtype=np.int32)
            bval = np.zeros(cap)
            st = lib().qa_fullpass_batch(self.dev.handle, C.c_int32(n), ptr(gl), ptr(wd), ptr(np.ascontiguousarray(cols, dtype=np.int32)),
                                         C.c_int32(K_top_matches), ptr(dosage), ptr(bptr), ptr(bidx), ptr(bval),
                                         C.c_int64(cap))
            if st == QA_ERR_CAPACITY:
                cap = int(bptr[-1])
                continue
            check(st)
            break
        best = []
        for p in range(n):
            best.append([dict(top_matches=bidx[bptr[p * n_thin + j]:bptr[p * n_thin + j + 1]],
                              top_matches_values=bval[bptr[p * n_thin + j]:bptr[p * n_thin + j + 1]])
                         for j in range(n_thin)])
        return list(dosage), best

    def fullpass_reads_batch(self, samples, chain_sample, labels, want_dosage, want_top, cols, K_top_matches, minGLValue,
                             top_width, n_label=2, select=None):
        """impute_using_everything for every chain: returns dosage [n_chain, n_label, T], the ordered top matches
        [n_chain, n_label, n_thin, top_width] (0-based, -1 padded; only for chains with want_top) and the full list lengths.
        With ``select`` (Ksubset, Knew, which, seeds) the re-selection of the small panels runs on the device behind the
        passes (qa_fullpass_reads_select_batch): the lists stay there (``top`` is None) and the call also returns
        which_next [n_chain, Ksubset] and the per-chain selection status.

        The dosage array is a view of this backend's pinned transfer buffer (``qa_host_alloc``): it is valid until the
        next call that asks for dosages -- copy what must outlive that (the driver consumes a round before the next)."""
        import ctypes as C
        from .native import check, lib, ptr
        lib().qa_fullpass_reads_batch.restype = C.c_int
        lib().qa_fullpass_reads_select_batch.restype = C.c_int
        P = self.dev.panel
        T = P.nSNPs
        n_chain, n_sample = len(chain_sample), len(samples)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        n_thin = int((cols >= 0).sum())
        read_off = np.zeros(n_sample + 1, dtype=np.int32)
        for i, s in enumerate(samples):
            read_off[i + 1] = read_off[i] + s.nReads
        read_ptr = np.concatenate([np.asarray(s.read_ptr, dtype=np.int32) for s in samples])
        u = np.concatenate([np.asarray(s.u, dtype=np.int32) for s in samples])
        bq = np.concatenate([np.asarray(s.bq, dtype=np.int32) for s in samples])
        H = np.concatenate([np.asarray(h, dtype=np.int32) for h in labels])
        cs = np.ascontiguousarray(chain_sample, dtype=np.int32)
        wd = np.ascontiguousarray(want_dosage, dtype=np.int32)
        wt = np.ascontiguousarray(want_top, dtype=np.int32)
        # The round's dosages land in this backend's pinned buffer (qa_host_alloc: no staging copy, no fresh pages for ~1 GB
        # per dosage round) -- valid until the next dosage round of this backend; the driver consumes a round's rows
        # (accumulation, read confidence, the phasing haplotypes) before it starts the next one.
        dosage = None
        if wd.any():
            need = n_chain * n_label * T
            if self._dosage_buf is None or self._dosage_buf.size < need:
                from .native import pinned_empty
                self._dosage_buf = None   # (release the smaller one first)
                self._dosage_buf = pinned_empty((need + need // 8,))
            dosage = self._dosage_buf[:need].reshape(n_chain, n_label, T)
            if not wd.all():
                dosage[...] = 0.0
        cnt = np.zeros((n_chain, n_label, n_thin), dtype=np.int32)
        head = (self.dev.handle, C.c_int32(n_chain), C.c_int32(n_label), C.c_int32(n_sample), ptr(cs), ptr(read_off),
                ptr(read_ptr), ptr(u), ptr(bq), ptr(H), ptr(wd), ptr(wt), ptr(cols), C.c_int32(K_top_matches),
                C.c_double(minGLValue), ptr(dosage), C.c_int32(top_width))
        if select is not None:
            Ks = int(select["Ksubset"])
            which = np.ascontiguousarray(np.stack([np.asarray(w, dtype=np.int32) for w in select["which"]]))
            seeds = np.ascontiguousarray(select["seeds"], dtype=np.uint64)
            nxt = np.zeros((n_chain, Ks), dtype=np.int32)
            status = np.full(n_chain, -1, dtype=np.int32)
            with span("device:fullpass"):
                check(lib().qa_fullpass_reads_select_batch(*head, None, None, ptr(cnt), C.c_int32(Ks),
                                                           C.c_int32(int(select["Knew"])), ptr(which), ptr(seeds), ptr(nxt),
                                                           ptr(status)))
            return dosage, None, cnt, nxt, status
        top = np.full((n_chain, n_label, n_thin, top_width), -1, dtype=np.int32)
        val = np.zeros((n_chain, n_label, n_thin, top_width), dtype=np.float32)
        with span("device:fullpass"):
            check(lib().qa_fullpass_reads_batch(*head, ptr(top), ptr(val), ptr(cnt)))
        return dosage, top, cnt

    def read_confidence_batch(self, samples, haps, maxDifferenceBetweenReads):
        """``haps``: per chain the list of its haploid dosages, or one [chain, label, SNP] array."""
        from .gibbs_nipt import calculate_eMatRead_t_vs_haplotypes_batch
        return calculate_eMatRead_t_vs_haplotypes_batch(self.dev, samples, haps, maxDifferenceBetweenReads,
                                                        hap_major=isinstance(haps, np.ndarray))
