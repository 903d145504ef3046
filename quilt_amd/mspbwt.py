"""The small-panel re-selection of the msPBWT mode (``use_mspbwt = TRUE``), SURVEY.md 8(f) rank 2(b).

``select_new_haps_mspbwt_v3`` restates QUILT/R/mspbwt.R:225-474 line by line (heuristic_approach "A"); the positional-BWT
query it calls -- ``mspbwt::Rcpp_find_good_matches_without_a``, a third-party package that is not in the reference tree --
is the neighbour scan of the panel's msPBWT indices in the library's host code (:class:`MsPbwtIndex`, csrc/mspbwt.cpp: the
published algorithm, stated in tests/mspbwt_scan.py; UNPINNED against the package) or, as an option, the exhaustive device
search ``qa_find_good_matches`` (csrc/match.hip: every haplotype's longest run at HBM rate).  R's ``sample()`` draws are keyed draws from
the chain's selection stream (quilt_amd/rng.py), as everywhere in the driver.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from .rng import SELECT_OFFSET_POOL, keyed_subset


def rcpp_int_contract(hap: np.ndarray) -> np.ndarray:
    """STITCH::rcpp_int_contract: 0 / 1 alleles -> one int32 word per 32 SNPs, bit b of word g = allele at SNP 32 g + b."""
    hap = np.asarray(hap).astype(np.uint64)
    G = (len(hap) + 31) // 32
    pad = np.zeros(32 * G, dtype=np.uint64)
    pad[:len(hap)] = hap
    return (pad.reshape(G, 32) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32).view(np.int32)


def find_good_matches(dev, Zs: np.ndarray, nindices: int, min_len: int, max_matches: int):
    """``qa_find_good_matches``: dense tables ``match`` [query, index, max_matches, 3] of (index0, start0, len1) in haplotype
    order and ``n`` [query, index], the rows in use."""
    from .native import check, lib, ptr
    lib().qa_find_good_matches.restype = C.c_int
    Zs = np.ascontiguousarray(Zs, dtype=np.int32)
    nq, G = Zs.shape
    match = np.zeros((nq, nindices, max_matches, 3), dtype=np.int32)
    n = np.zeros((nq, nindices), dtype=np.int32)
    check(lib().qa_find_good_matches(dev.handle, C.c_int32(nq), ptr(Zs), C.c_int32(nindices), C.c_int32(min_len),
                                     C.c_int32(max_matches), ptr(match), ptr(n)))
    return match, n


class MsPbwtIndex:
    """The panel's msPBWT indices (``ms_indices`` of quilt-prepare-reference: mspbwt::ms_BuildIndices_Algorithm5) and the
    neighbour scan that queries them -- ``mspbwt::Rcpp_find_good_matches_without_a`` as mspbwt.R:297-310 calls it -- in the
    library's host code (csrc/mspbwt.cpp; the published algorithm as tests/mspbwt_scan.py states it, unpinned against the
    package, which is not in the reference tree)."""

    def __init__(self, panel, nindices: int):
        from .native import lib
        if panel.hapMatcherR is None:
            raise ValueError("the msPBWT index is built from hapMatcherR (nMaxDH <= 255)")
        L = lib()
        L.qa_mspbwt_create.restype = C.c_void_p
        L.qa_mspbwt_bytes.restype = C.c_int64
        hm = np.asfortranarray(panel.hapMatcherR, dtype=np.uint8)
        B = np.asfortranarray(panel.distinctHapsB, dtype=np.int32)
        self.K, self.nGrids, self.nindices = int(hm.shape[0]), int(hm.shape[1]), int(nindices)
        h = L.qa_mspbwt_create(C.c_int32(self.K), C.c_int32(self.nGrids), hm.ctypes.data_as(C.c_void_p), C.c_int32(B.shape[0]),
                               B.ctypes.data_as(C.c_void_p), C.c_int32(nindices))
        if not h:
            L.qa_last_error.restype = C.c_char_p
            raise ValueError((L.qa_last_error() or b"qa_mspbwt_create failed").decode())
        self.handle = C.c_void_p(h)
        self.bytes = int(L.qa_mspbwt_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            from .native import lib
            lib().qa_mspbwt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find_good_matches(self, Zs: np.ndarray, L: int, M: int) -> List[List[np.ndarray]]:
        """``out[query][index]``: (haplotype0, start0, len1) rows, as tests/mspbwt_scan.py::find_good_matches_scan."""
        from .native import lib
        lb = lib()
        lb.qa_mspbwt_find_good_matches.restype = C.c_int64
        Zs = np.ascontiguousarray(Zs, dtype=np.int32)
        nq = Zs.shape[0]
        row_ptr = np.zeros(nq * self.nindices + 1, dtype=np.int64)
        cap = max(1024, 64 * nq * self.nindices)
        while True:
            rows = np.zeros((cap, 3), dtype=np.int32)
            n = lb.qa_mspbwt_find_good_matches(self.handle, C.c_int32(nq), Zs.ctypes.data_as(C.c_void_p), C.c_int32(L), C.c_int32(M),
                                               row_ptr.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), C.c_int64(cap))
            if n < 0:
                raise ValueError("qa_mspbwt_find_good_matches: invalid arguments")
            if n <= cap:
                break
            cap = int(n)
        return [[rows[row_ptr[q * self.nindices + i]:row_ptr[q * self.nindices + i + 1]].copy() for i in range(self.nindices)]
                for q in range(nq)]

    def select_new_haps(self, Zs: np.ndarray, n_label: int, L: int, M: int, Knew: int, seeds: Sequence[int]) -> np.ndarray:
        """Scan + ``select_new_haps_mspbwt_v3`` for every chain of a round (``Zs``: the chains' ``n_label`` packed haplotypes back
        to back).  Returns [chain, Knew] 1-based haplotypes."""
        from .native import lib
        lb = lib()
        lb.qa_mspbwt_select_new_haps.restype = C.c_int
        Zs = np.ascontiguousarray(Zs, dtype=np.int32)
        n_chain = Zs.shape[0] // n_label
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        out = np.zeros((n_chain, Knew), dtype=np.int32)
        st = lb.qa_mspbwt_select_new_haps(self.handle, C.c_int32(n_chain), C.c_int32(n_label), Zs.ctypes.data_as(C.c_void_p),
                                          C.c_int32(L), C.c_int32(M), C.c_int32(Knew), sd.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p))
        if st != 0:
            raise ValueError("qa_mspbwt_select_new_haps failed (status %d)" % st)
        return out


_INDEX_LOCK = __import__("threading").Lock()


def panel_mspbwt_index(panel, nindices: int) -> MsPbwtIndex:
    """The panel's indices, built on first use and shared by every host thread working on the panel (queries only read them)."""
    with _INDEX_LOCK:
        cache = panel.__dict__.setdefault("_mspbwt_indices", {})
        if nindices not in cache:
            cache[nindices] = MsPbwtIndex(panel, nindices)
        return cache[nindices]


def match_tables_as_lists(match: np.ndarray, n: np.ndarray) -> List[List[np.ndarray]]:
    return [[match[q, i, :n[q, i]].copy() for i in range(match.shape[1])] for q in range(match.shape[0])]


def match_lists_as_tables(found: Sequence[Sequence[np.ndarray]], max_matches: int):
    nq, ni = len(found), len(found[0])
    match = np.zeros((nq, ni, max_matches, 3), dtype=np.int32)
    n = np.zeros((nq, ni), dtype=np.int32)
    for q in range(nq):
        for i in range(ni):
            m = np.asarray(found[q][i], dtype=np.int32).reshape(-1, 3)
            n[q, i] = len(m)
            match[q, i, :len(m)] = m
    return match, n


def select_new_haps_mspbwt_batch(match: np.ndarray, n: np.ndarray, n_label: int, Knew: int, Kfull: int, nGrids: int,
                                 seeds: Sequence[int]) -> np.ndarray:
    """``select_new_haps_mspbwt_v3`` for every chain of a round (native: csrc/hostio.cpp, the same text as the numpy function
    below, tested equal): ``match`` / ``n`` as :func:`find_good_matches` returns them, the chains' ``n_label`` haplotypes back to
    back.  Returns [chain, Knew] 1-based haplotypes."""
    from .native import lib, ptr
    lib().qa_select_new_haps_mspbwt.restype = C.c_int
    match = np.ascontiguousarray(match, dtype=np.int32)
    n = np.ascontiguousarray(n, dtype=np.int32)
    nq, ni, mm, _ = match.shape
    n_chain = nq // n_label
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    out = np.zeros((n_chain, Knew), dtype=np.int32)
    st = lib().qa_select_new_haps_mspbwt(C.c_int32(n_chain), C.c_int32(n_label), C.c_int32(ni), C.c_int32(mm), ptr(match), ptr(n),
                                         C.c_int32(Knew), C.c_int32(Kfull), C.c_int32(nGrids), ptr(sd), ptr(out))
    if st != 0:
        raise ValueError("qa_select_new_haps_mspbwt: bad match tables")
    return out


def match_weights(start1: np.ndarray, end1: np.ndarray) -> np.ndarray:
    """mspbwt.R:418-427 (native: the loop is sequential in the matches)."""
    from .native import lib, ptr
    lib().qa_mspbwt_weights.restype = C.c_int
    s1 = np.ascontiguousarray(start1, dtype=np.int64)
    e1 = np.ascontiguousarray(end1, dtype=np.int64)
    w = np.zeros(len(s1))
    if lib().qa_mspbwt_weights(C.c_int32(len(s1)), ptr(s1), ptr(e1), ptr(w)) != 0:
        raise ValueError("bad match coordinates")
    return w


def int_contract_rows(dosages: np.ndarray) -> np.ndarray:
    """rcpp_int_contract(round(x)) for every row of a 2-D array at once (R's round: 0.5 -> 0, so the bit is x > 0.5)."""
    n, T = dosages.shape
    G = (T + 31) // 32
    bits = np.zeros((n, 32 * G), dtype=np.uint8)
    bits[:, :T] = dosages > 0.5
    return np.packbits(bits, axis=1, bitorder="little").view("<u4").astype(np.uint32).view(np.int32).reshape(n, G)


def _order_stable(*keys) -> np.ndarray:
    """R's order(k1, k2, ...): ascending, stable."""
    return np.lexsort(tuple(reversed(keys)))


def matches_to_mtm(per_index: Sequence[np.ndarray], nGrids: int) -> np.ndarray:
    """mspbwt.R:303-371 for one haplotype: columns index1, start1, end1, len1, key, n -- per index the matches with
    1-based coordinates, duplicates of (haplotype, start) dropped keeping the longest, then all indices ordered by length."""
    parts = []
    for i_index, m in enumerate(per_index, start=1):
        if m is None or len(m) == 0:
            continue
        m = np.asarray(m, dtype=np.int64)
        mtm = np.column_stack([m[:, 0] + 1, m[:, 1] + 1, m[:, 1] + m[:, 2], m[:, 2]])
        if len(mtm) > 1:
            mtm = mtm[_order_stable(mtm[:, 0], -mtm[:, 2], -mtm[:, 1])]
            dup = np.r_[False, (np.diff(mtm[:, 0]) == 0) & (np.diff(mtm[:, 1]) == 0)]
            mtm = mtm[~dup]
        key = nGrids * mtm[:, 1] + mtm[:, 2]
        parts.append(np.column_stack([mtm, key, np.full(len(mtm), i_index)]))
    if not parts:
        return np.zeros((0, 6), dtype=np.int64)
    mtm = np.concatenate(parts)
    return mtm[_order_stable(-mtm[:, 3])]


def select_new_haps_mspbwt_v3(matches: Sequence[Sequence[np.ndarray]], Knew: int, Kfull: int, nGrids: int,
                              seed_select: int) -> np.ndarray:
    """mspbwt.R:375-474.  ``matches[ihap][iIndex]``: the search results of the sample's 2 (3: NIPT) haplotypes.
    Returns ``Knew`` 1-based haplotypes."""
    out = [matches_to_mtm(per_index, nGrids) for per_index in matches]
    first = np.concatenate([o[:, 0] for o in out]) if out else np.zeros(0, dtype=np.int64)
    _, idx = np.unique(first, return_index=True)
    unique_haps = first[np.sort(idx)]
    if len(unique_haps) == 0:                       # "special fluke case": sample(1:Kfull, Knew)
        return (keyed_subset(seed_select, Kfull, Knew, SELECT_OFFSET_POOL) + 1).astype(np.int32)
    if len(unique_haps) <= Knew:                    # the identified haplotypes, topped up from the rest of the panel
        pool = np.setdiff1d(np.arange(1, Kfull + 1), unique_haps)
        extra = pool[keyed_subset(seed_select, len(pool), Knew - len(unique_haps), SELECT_OFFSET_POOL)]
        return np.concatenate([unique_haps, extra]).astype(np.int32)
    # prioritise by length and new-ness, per haplotype (mspbwt.R:416-446)
    results = []
    for mtm in out:
        if len(mtm) == 0:
            results.append(np.zeros(0, dtype=np.int64))
            continue
        results.append(mtm[_order_stable(-match_weights(mtm[:, 1], mtm[:, 2])), 0])
    a = max(len(r) for r in results)
    padded = np.full((len(results), a), -1, dtype=np.int64)
    for i, r in enumerate(results):
        padded[i, :len(r)] = r
    inter = padded.T.reshape(-1)                    # c(t(cbind(x, y[, z]))): x1, y1, x2, y2, ...
    inter = inter[inter > 0]
    _, idx = np.unique(inter, return_index=True)
    unique_ordered = inter[np.sort(idx)]
    if len(unique_ordered) >= Knew:
        return unique_ordered[:Knew].astype(np.int32)
    new_haps = np.concatenate([np.setdiff1d(unique_ordered, unique_haps), unique_haps])[:Knew]
    return new_haps.astype(np.int32)
