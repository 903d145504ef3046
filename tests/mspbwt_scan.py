"""Test infrastructure: the neighbour scan of a multi-symbol positional BWT, the query the reference's msPBWT mode makes.

``select_new_haps_mspbwt_v3`` (QUILT/R/mspbwt.R:297-310) calls ``mspbwt::Rcpp_find_good_matches_without_a(Z, all_symbols,
usge_all, egs, pbwtL = mspbwtL, pbwtM = mspbwtM, hapMatcherR, ...)`` once per interleaved index.  The mspbwt package
(rwdavies/mspbwt, an R dependency of QUILT; DESCRIPTION pins mspbwt >= 0.1.0) is NOT in the reference tree, so its text
cannot be followed line by line: PARITY UNPINNED.  What is restated here is the published algorithm the call's interface and
QUILT's option help describe (quilt.R: ``mspbwtL``: "How many neighbouring haplotypes to scan up and down at each grid",
``mspbwtM``: "Minimum long grids matches"; Durbin 2014, Bioinformatics 30:1266 for the positional prefix order and the
query's insertion point; the multi-symbol form of it in the QUILT2 paper's Methods):

  * an index covers the grids i, i + n, i + 2 n, ...; a haplotype's symbol at a grid is its row of the grid's dictionary
    (hapMatcherR; 0 = not in the dictionary, which matches nothing here, as in csrc/match.hip)
  * after position t the panel's haplotypes are ordered by their reversed prefixes (symbols at t, t - 1, ..., 0); the query
    is inserted into that order (Durbin's update of f: the haplotypes with a smaller symbol at t, plus those before the old
    insertion point with the same symbol)
  * the ``L`` haplotypes above and the ``L`` below the insertion point are scanned; one whose match with the query ending
    at t has at least ``M`` positions is reported as (haplotype0, start0, len1)
  * the same (haplotype, start) seen at several positions keeps its longest report (mspbwt.R:330-345 drops the others)

By the order's defining property the scanned neighbours are the haplotypes with the longest matches ending at t
(``check=True`` asserts that of the adjacent pair): the scan returns the locally longest matches at EVERY position, where
the device search (csrc/match.hip) returns every haplotype's single longest run, the longest ones first.  tests and bench.py
report how far the next small panels chosen from the two agree.  Only tests/ and bench.py's checking legs import this.
"""
from __future__ import annotations

from typing import List

import numpy as np


def query_symbols(panel, Zs: np.ndarray) -> np.ndarray:
    """[query, grid] dictionary rows (1-based) of the queries' 32-SNP words; 0: the word is not in the grid's dictionary."""
    B = np.asarray(panel.distinctHapsB)                     # nMaxDH x G
    Zs = np.asarray(Zs)
    qc = np.zeros(Zs.shape, dtype=np.int64)
    for g in range(Zs.shape[1]):
        eq = B[:, g][None, :] == Zs[:, g][:, None]          # query x row: the first matching row is the symbol
        hit = eq.any(axis=1)
        qc[hit, g] = eq[hit].argmax(axis=1) + 1
    return qc


def neighbour_scan_index(sym: np.ndarray, zq: np.ndarray, L: int, M: int, check: bool = False) -> List[np.ndarray]:
    """One index.  ``sym`` [positions, K] panel symbols, ``zq`` [query, positions] query symbols (0: no symbol).
    Returns per query the (haplotype0, start0, len1) rows, haplotype order, one row per (haplotype, start)."""
    Tp, K = sym.shape
    Q = zq.shape[0]
    a = np.arange(K)                                        # the order before position 0: haplotype order
    run = np.zeros((Q, K), dtype=np.int32)                  # by haplotype: positions matched up to and including t
    f = np.zeros(Q, dtype=np.int64)                         # the queries' insertion points in `a`
    rep = [dict() for _ in range(Q)]
    for t in range(Tp):
        s = sym[t]
        s_a = s[a]
        z = zq[:, t]
        n_sym = int(max(s.max(), z.max())) + 2
        below = np.concatenate([[0], np.cumsum(np.bincount(s_a, minlength=n_sym))])     # haplotypes with a smaller symbol
        for q in range(Q):
            if z[q] == 0:          # a word the panel does not hold: no symbol, placed before every haplotype
                f[q] = 0
            else:
                f[q] = below[z[q]] + np.count_nonzero(s_a[:f[q]] == z[q])
        a = a[np.argsort(s_a, kind="stable")]
        match = (s[None, :] == z[:, None]) & (s[None, :] != 0)
        run = np.where(match, run + 1, 0).astype(np.int32)
        for q in range(Q):
            lo, hi = max(int(f[q]) - L, 0), min(int(f[q]) + L, K)
            hap = a[lo:hi]
            ln = run[q, hap]
            if check:
                adj = [run[q, a[int(f[q]) - 1]] if f[q] > 0 else 0, run[q, a[int(f[q])]] if f[q] < K else 0]
                assert max(adj) == run[q].max(), (t, q, adj, run[q].max())
                # and the lengths fall away from the insertion point on either side
                up, dn = run[q, a[lo:int(f[q])]], run[q, a[int(f[q]):hi]]
                assert (np.diff(up) >= 0).all() and (np.diff(dn) <= 0).all()
            for k, n in zip(hap[ln >= M], ln[ln >= M]):
                key = (int(k), t - int(n) + 1)
                if rep[q].get(key, 0) < n:
                    rep[q][key] = int(n)
    out = []
    for q in range(Q):
        rows = sorted((k, s0, n) for (k, s0), n in rep[q].items())
        out.append(np.array(rows, dtype=np.int32).reshape(-1, 3))
    return out


def find_good_matches_scan(panel, Zs: np.ndarray, nindices: int, L: int, M: int, check: bool = False):
    """The scan for every query and index: ``out[query][index]`` = (haplotype0, start0, len1) rows, the shape
    quilt_amd.mspbwt.select_new_haps_mspbwt_v3 takes."""
    hm = np.asarray(panel.hapMatcherR if panel.hapMatcherR is not None else panel.hapMatcher)     # K x G
    qc = query_symbols(panel, Zs)
    out = [[None] * nindices for _ in range(len(qc))]
    for i in range(nindices):
        sym = np.ascontiguousarray(hm[:, i::nindices].T).astype(np.int64)
        res = neighbour_scan_index(sym, qc[:, i::nindices], L, M, check=check)
        for q in range(len(qc)):
            out[q][i] = res[q]
    return out


def selection_agreement(found_a, found_b, Knew: int, Kfull: int, nGrids: int, seed: int = 1) -> dict:
    """How far two searches agree on what the driver does with them, for one chain (``found_x[label][index]``):
    ``selected``: share of the next small panel (select_new_haps_mspbwt_v3, same selection stream) chosen from both;
    ``longest``: share of the haplotypes holding a's ``Knew`` longest matches that b reports at all;
    ``length``: the same weighted by match length."""
    from quilt_amd.mspbwt import matches_to_mtm, select_new_haps_mspbwt_v3
    sa = select_new_haps_mspbwt_v3(found_a, Knew, Kfull, nGrids, seed)
    sb = select_new_haps_mspbwt_v3(found_b, Knew, Kfull, nGrids, seed)
    haps_b = set()
    for per_index in found_b:
        for m in per_index:
            haps_b.update(np.asarray(m)[:, 0].tolist())
    best = {}
    for per_index in found_a:
        mtm = matches_to_mtm(per_index, nGrids)
        for k1, n in zip(mtm[:, 0], mtm[:, 3]):
            best[int(k1) - 1] = max(best.get(int(k1) - 1, 0), int(n))
    top = sorted(best.items(), key=lambda kv: (-kv[1], kv[0]))[:Knew]
    in_b = [(k, n) for k, n in top if k in haps_b]
    return dict(selected=len(np.intersect1d(sa, sb)) / float(Knew),
                longest=len(in_b) / max(len(top), 1),
                length=sum(n for _, n in in_b) / max(sum(n for _, n in top), 1),
                n_scan=len(best), n_other=len(haps_b))
