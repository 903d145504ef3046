"""Prepared-reference panel containers in the reference's own layouts.

The hot path consumes the objects `QUILT_prepare_reference` saves
(QUILT/R/quilt-prepare-reference.R:484-525): the bit-packed panel ``rhb_t``, its
per-grid dictionary compression (``hapMatcher(R)``, ``distinctHapsB``,
``distinctHapsIE``, the "special" side tables) and the transition rates.  The
producers live in the un-vendored STITCH package (1.8.4); their layouts are pinned by
QUILT/tests/testthat/test-unit-reference-single.R:210-309 (round trip) and restated here
so that synthetic panels can be built without R.

All matrices are numpy arrays in Fortran (column-major) order with R's shapes, so their
buffers can be handed to the C ABI unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


def make_rhb_t_from_rhi_t(rhi_t: np.ndarray) -> np.ndarray:
    """Pack a K x T 0/1 matrix into K x ceil(T/32) int32 words, SNP 32g+b -> bit b.

    Layout as read by QUILT/src/reference-single.cpp:309-315 (``tmp & (1<<b)``).
    """
    K, T = rhi_t.shape
    G = (T + 31) // 32
    padded = np.zeros((K, G * 32), dtype=np.uint32)
    padded[:, :T] = rhi_t.astype(np.uint32)
    weights = (np.uint32(1) << np.arange(32, dtype=np.uint32))
    words = (padded.reshape(K, G, 32) * weights[None, None, :]).sum(axis=2, dtype=np.uint64)
    return np.asfortranarray(words.astype(np.uint32).view(np.int32).reshape(K, G))


def int_expand(words: np.ndarray, n_bits: int = 32) -> np.ndarray:
    """Bits of int32 words, LSB first (STITCH::int_expand semantics)."""
    w = np.asarray(words).astype(np.int32).view(np.uint32)
    return ((w[..., None] >> np.arange(n_bits, dtype=np.uint32)) & 1).astype(np.int8)


@dataclass
class Panel:
    """One prepared reference panel (region), in the reference's layouts."""

    K: int
    nSNPs: int
    nGrids: int
    nMaxDH: int
    ref_error: float
    rhb_t: Optional[np.ndarray]            # int32 K x G (F); None in msPBWT mode
    hapMatcher: Optional[np.ndarray]       # int32 K x G (F)
    hapMatcherR: Optional[np.ndarray]      # uint8 K x G (F)
    distinctHapsB: np.ndarray              # int32 nMaxDH x G (F)
    distinctHapsIE: np.ndarray             # float64 nMaxDH x T (F)
    eMatDH_special_grid_which: np.ndarray  # int32 G; 0 = none else 1-based list id
    eMatDH_special_values_list: List[np.ndarray]  # 0-based k, ascending, per special grid
    eMatDH_special_matrix: np.ndarray      # int32 n x 2 (F): k (0-based), word
    eMatDH_special_matrix_helper: np.ndarray  # int32 G x 2 (F): 1-based first/last row
    transMatRate_t: np.ndarray             # float64 2 x (G-1) (F): sigma, 1 - sigma
    L: Optional[np.ndarray] = None         # SNP positions (bp)
    L_grid: Optional[np.ndarray] = None    # grid positions (bp)
    grid: Optional[np.ndarray] = None      # int32 T: 0-based grid of each SNP
    smooth_cm: Optional[np.ndarray] = None
    extra: dict = field(default_factory=dict)

    @property
    def use_hapMatcherR(self) -> bool:
        return self.hapMatcherR is not None

    def special_csr(self):
        """(ptr, values) CSR over ``eMatDH_special_values_list``."""
        n = len(self.eMatDH_special_values_list)
        ptr = np.zeros(n + 1, dtype=np.int32)
        for i, v in enumerate(self.eMatDH_special_values_list):
            ptr[i + 1] = ptr[i] + len(v)
        vals = (np.concatenate(self.eMatDH_special_values_list).astype(np.int32)
                if n else np.zeros(0, dtype=np.int32))
        return ptr, vals


@dataclass
class RareCommon:
    """The all-SNP side of a QUILT2 panel (``special_rare_common_objects``, prepare_reference_functions.R:172-247):
    the panel tables cover the common SNPs; every haplotype lists the rare SNPs it carries the alt of
    (``rare_per_hap_info``); the all-SNP grid is 32 SNPs wide like the common one."""

    nSNPs_all: int
    nGrids_all: int
    snp_is_common: np.ndarray        # uint8 T_all
    common_snp_index: np.ndarray     # int32 T_all: 1-based index among the common SNPs, 0 for rare (rare_common.R:222-223)
    rare_ptr: np.ndarray             # int64 K + 1: CSR over rare_per_hap_info
    rare_snp: np.ndarray             # int32: 1-based all-SNP indices, ascending within a haplotype
    transMatRate_t_all: np.ndarray   # float64 2 x (G_all - 1) (F)
    L_all: Optional[np.ndarray] = None
    L_grid_all: Optional[np.ndarray] = None


def make_rhb_t_equality(rhb_t: np.ndarray, nMaxDH: Optional[int], nSNPs: int, ref_error: float,
                        use_hapMatcherR: bool = True) -> dict:
    """Per-grid dictionary compression of the packed panel.

    Restates STITCH::make_rhb_t_equality (STITCH 1.8.4, not vendored; call sites
    quilt-prepare-reference.R:416-428, test-drivers.R:398).  Per grid the distinct 32-bit
    words are ranked by descending frequency (ties: ascending signed value); the first
    ``nMaxDH`` get 1-based ids in ``distinctHapsB``/``hapMatcher``; the rest get id 0 and are
    listed in the "special" side tables.  Pinned by the rebuild-``rhb_t`` round trip of
    test-unit-reference-single.R:238-307.
    """
    K, G = rhb_t.shape
    if nMaxDH is None:
        # infer: enough rows for the 90th percentile of per-grid distinct counts, <= 255
        counts = np.array([len(np.unique(rhb_t[:, g])) for g in range(G)])
        nMaxDH = int(min(255, max(1, np.quantile(counts, 0.9))))
    if use_hapMatcherR and nMaxDH > 255:
        raise ValueError("hapMatcherR needs nMaxDH <= 255")
    distinctHapsB = np.zeros((nMaxDH, G), dtype=np.int32, order="F")
    hm = np.zeros((K, G), dtype=np.int32, order="F")
    for g in range(G):
        col = rhb_t[:, g]
        vals, inv, cnt = np.unique(col, return_inverse=True, return_counts=True)
        order = np.lexsort((vals, -cnt))  # descending count, ties ascending value
        keep = order[:nMaxDH]
        distinctHapsB[: len(keep), g] = vals[keep]
        rank = np.zeros(len(vals), dtype=np.int32)
        rank[keep] = np.arange(1, len(keep) + 1, dtype=np.int32)
        # a word equal to the zero padding of distinctHapsB must still map to its own row
        hm[:, g] = rank[inv]
    bits = int_expand(distinctHapsB)  # nMaxDH x G x 32
    ie = np.where(bits.reshape(nMaxDH, G * 32)[:, :nSNPs] == 1, 1.0 - ref_error, ref_error)
    distinctHapsIE = np.asfortranarray(ie.astype(np.float64))
    # special tables
    which = np.zeros(G, dtype=np.int32)
    values_list: List[np.ndarray] = []
    rows = []
    helper = np.zeros((G, 2), dtype=np.int32, order="F")
    n_rows = 0
    for g in range(G):
        ks = np.nonzero(hm[:, g] == 0)[0].astype(np.int32)
        if len(ks) == 0:
            continue
        values_list.append(ks)
        which[g] = len(values_list)
        rows.append(np.stack([ks, rhb_t[ks, g]], axis=1))
        helper[g, 0] = n_rows + 1
        helper[g, 1] = n_rows + len(ks)
        n_rows += len(ks)
    if rows:
        special_matrix = np.asfortranarray(np.concatenate(rows, axis=0).astype(np.int32))
    else:
        special_matrix = np.zeros((1, 2), dtype=np.int32, order="F")
    out = dict(
        distinctHapsB=distinctHapsB,
        distinctHapsIE=distinctHapsIE,
        hapMatcher=None if use_hapMatcherR else hm,
        hapMatcherR=np.asfortranarray(hm.astype(np.uint8)) if use_hapMatcherR else None,
        eMatDH_special_grid_which=which,
        eMatDH_special_values_list=values_list,
        eMatDH_special_matrix=special_matrix,
        eMatDH_special_matrix_helper=helper,
        nrow_which_hapMatcher_0=n_rows,
        nMaxDH=nMaxDH,
    )
    return out


def simple_binary_matrix_search(val: int, mat: np.ndarray, s1: int, e1: int) -> int:
    """Host twin of gibbs-small.cpp:69-105 (incl. the one-row -> 0 quirk)."""
    nori = e1 - s1 + 1
    if nori == 1:
        return 0
    n = nori
    i = n // 2
    n = n // 4
    for _ in range(100):
        key = int(mat[s1 - 1 + i, 0])
        if key == val:
            return int(mat[s1 - 1 + i, 1])
        if key < val:
            i += n
        else:
            i -= n
        n = max(n // 2, 1)
        i = min(max(i, 0), nori - 1)
    return int(mat[s1, 1])


def rebuild_rhb_t(panel_tables: dict, K: int, G: int, exact_single_special: bool = True) -> np.ndarray:
    """Invert :func:`make_rhb_t_equality` (the reference's round-trip invariant).

    With ``exact_single_special`` a grid holding exactly one special haplotype is decoded
    from the stored word (the reference's C++ search returns 0 there, Appendix A.7).
    """
    hm = panel_tables["hapMatcher"] if panel_tables["hapMatcher"] is not None else panel_tables["hapMatcherR"]
    B = panel_tables["distinctHapsB"]
    out = np.zeros((K, G), dtype=np.int32, order="F")
    for g in range(G):
        ids = hm[:, g].astype(np.int64)
        col = np.where(ids > 0, B[np.maximum(ids - 1, 0), g], 0).astype(np.int32)
        s1, e1 = panel_tables["eMatDH_special_matrix_helper"][g]
        for k in np.nonzero(ids == 0)[0]:
            if exact_single_special and s1 == e1:
                col[k] = panel_tables["eMatDH_special_matrix"][s1 - 1, 1]
            else:
                col[k] = simple_binary_matrix_search(int(k), panel_tables["eMatDH_special_matrix"], int(s1), int(e1))
        out[:, g] = col
    return out
