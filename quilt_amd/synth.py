"""Seeded synthetic panels and low-coverage read sets (no R, no BAMs, no network).

Mimics the reference's own test-data recipes:

* panel: ``make_reference_single_test_package`` (QUILT/R/test-drivers.R:324-462): per
  32-SNP grid a few "founder" words with Beta(0.1, 0.1) allele frequencies, a Poisson
  number of derived words with 4 flipped bits, every panel haplotype assigned to one of
  them; a couple of "stress" grids with more than ``nMaxDH`` distinct words exercise the
  special-haplotype tables.
* reads: ``make_quilt_fb_test_package`` (QUILT/R/test-drivers.R:127-319) /
  STITCH ``sampleReads`` convention (copied-from-stitch.cpp:153-160): per read
  ``J`` (= #SNPs - 1), ``wif`` (0-based grid of the central SNP), signed base
  qualities ``bq`` (< 0 REF, > 0 ALT, |bq| = phred) and 0-based SNP indices ``u``.

Reads are kept flattened (CSR) because that is what the C ABI takes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .panel import Panel, RareCommon, int_expand, make_rhb_t_equality


def _sigma_from_positions(L_grid, nGen, expRate, minRate=0.1, maxRate=100.0):
    """prepare_reference_functions.R:89-108 with a flat genetic map of ``expRate`` cM/Mb."""
    dl = np.diff(L_grid).astype(np.float64)
    rate = nGen * dl / 1e6 * (expRate / 100.0)
    lo = nGen * dl / 1e6 * (minRate / 100.0)
    hi = nGen * dl / 1e6 * (maxRate / 100.0)
    rate = np.minimum(np.maximum(rate, lo), hi)
    return np.exp(-rate)


def make_synthetic_panel(K: int, nSNPs: int, seed: int = 4916, nMaxDH: int = 255,
                         ref_error: float = 1e-3, nGen: float = 100.0, expRate: float = 1.0,
                         region_bp: Optional[int] = None, stress_grids: Sequence[int] = (2, 9),
                         use_hapMatcherR: bool = True, keep_rhb_t: bool = True) -> Panel:
    rng = np.random.default_rng(seed)
    G = (nSNPs + 31) // 32
    if region_bp is None:
        region_bp = nSNPs * 47  # ~64 000 SNPs on 3 Mb
    L = np.sort(rng.choice(np.arange(1, region_bp + 1), size=nSNPs, replace=False)).astype(np.int64)
    grid = (np.arange(nSNPs) // 32).astype(np.int32)
    starts = np.arange(0, nSNPs, 32)
    L_grid = (np.add.reduceat(L, starts) // np.diff(np.r_[starts, nSNPs])).astype(np.int64)
    sigma = _sigma_from_positions(L_grid, nGen, expRate)
    transMatRate_t = np.asfortranarray(np.stack([sigma, 1.0 - sigma], axis=0))

    rhb_t = np.zeros((K, G), dtype=np.int32, order="F")
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))
    for g in range(G):
        stress = g in stress_grids and K > nMaxDH + 20
        y = rng.poisson(10, size=K)
        nLocal = 4
        if stress:
            pick = rng.choice(K, size=nMaxDH + 20, replace=False)
            y[pick] = np.arange(1, nMaxDH + 21)
            nLocal = 6
        n = int(y.max()) + 1
        z = max(int(rng.poisson(6)), nLocal)
        z = min(z, n)
        af = rng.beta(1, 1, size=32) if stress else rng.beta(0.1, 0.1, size=32)
        y2 = np.zeros((32, n), dtype=np.uint8)
        y2[:, :z] = (rng.random((32, z)) < af[:, None]).astype(np.uint8)
        for i in range(z, n):
            y2[:, i] = y2[:, rng.integers(0, z)]
            flip = rng.choice(32, size=nLocal, replace=False)
            y2[flip, i] = rng.integers(0, 2, size=nLocal)
        nbits = min(32, nSNPs - 32 * g)
        y2[nbits:, :] = 0
        words = (y2.astype(np.uint64) * weights[:, None]).sum(axis=0).astype(np.uint32).view(np.int32)
        rhb_t[:, g] = words[y]
    t = make_rhb_t_equality(rhb_t, nMaxDH, nSNPs, ref_error, use_hapMatcherR=use_hapMatcherR)
    return Panel(
        K=K, nSNPs=nSNPs, nGrids=G, nMaxDH=t["nMaxDH"], ref_error=ref_error,
        rhb_t=rhb_t if keep_rhb_t else None,
        hapMatcher=t["hapMatcher"], hapMatcherR=t["hapMatcherR"],
        distinctHapsB=t["distinctHapsB"], distinctHapsIE=t["distinctHapsIE"],
        eMatDH_special_grid_which=t["eMatDH_special_grid_which"],
        eMatDH_special_values_list=t["eMatDH_special_values_list"],
        eMatDH_special_matrix=t["eMatDH_special_matrix"],
        eMatDH_special_matrix_helper=t["eMatDH_special_matrix_helper"],
        transMatRate_t=transMatRate_t, L=L, L_grid=L_grid, grid=grid,
        extra=dict(seed=seed, nGen=nGen, expRate=expRate),
    )


def make_1000g_like_panel(K: int = 5008, nSNPs: int = 3200, seed: int = 2504, nMaxDH: int = 255, ref_error: float = 1e-3,
                          nGen: float = 100.0, expRate: float = 1.0, block: int = 256) -> Panel:
    """A panel with the shape of the quick-start's (BASELINE configs[0]: 1000 Genomes phase 3, 2 504 samples = 5 008
    haplotypes): a site-frequency spectrum dominated by rare variants -- the derived-allele count i of a SNP is drawn with
    P(i) ~ 1 / i, the neutral expectation 1000G roughly follows -- and linkage disequilibrium from shared ancestry: within a
    block of ``block`` SNPs the haplotypes have a fixed order and a SNP's carriers are a contiguous run of that order (nested and
    overlapping clades); from block to block segments of the order are cut and moved (recombination).  No real data: the
    stand-in for the panel whose file cannot be downloaded here."""
    rng = np.random.default_rng(seed)
    G = (nSNPs + 31) // 32
    region_bp = nSNPs * 47
    L = np.sort(rng.choice(np.arange(1, region_bp + 1), size=nSNPs, replace=False)).astype(np.int64)
    grid = (np.arange(nSNPs) // 32).astype(np.int32)
    starts = np.arange(0, nSNPs, 32)
    L_grid = (np.add.reduceat(L, starts) // np.diff(np.r_[starts, nSNPs])).astype(np.int64)
    sigma = _sigma_from_positions(L_grid, nGen, expRate)
    transMatRate_t = np.asfortranarray(np.stack([sigma, 1.0 - sigma], axis=0))
    counts = np.arange(1, K)
    p_count = (1.0 / counts) / (1.0 / counts).sum()
    order = rng.permutation(K)
    hap = np.zeros((K, nSNPs), dtype=np.uint8)
    for t in range(nSNPs):
        if t % block == 0 and t > 0:   # a few recombinations: segments of the order move
            for _ in range(6):
                a, b = sorted(rng.integers(0, K, size=2))
                if b - a > 1:
                    seg = order[a:b].copy()
                    rest = np.concatenate([order[:a], order[b:]])
                    at = int(rng.integers(0, len(rest) + 1))
                    order = np.concatenate([rest[:at], seg, rest[at:]])
        i = int(rng.choice(counts, p=p_count))
        a = int(rng.integers(0, K - i + 1))
        hap[order[a:a + i], t] = 1
    pad = np.zeros((K, G * 32), dtype=np.uint8)
    pad[:, :nSNPs] = hap
    # bit b of word g = the allele at SNP 32 g + b, least significant bit first (packbits: 1 byte per 8 SNPs; 4 bytes = a word)
    rhb_t = np.asfortranarray(np.packbits(pad, axis=1, bitorder="little").view("<u4").view(np.int32))
    t = make_rhb_t_equality(rhb_t, nMaxDH, nSNPs, ref_error, use_hapMatcherR=True)
    return Panel(
        K=K, nSNPs=nSNPs, nGrids=G, nMaxDH=t["nMaxDH"], ref_error=ref_error, rhb_t=rhb_t,
        hapMatcher=t["hapMatcher"], hapMatcherR=t["hapMatcherR"],
        distinctHapsB=t["distinctHapsB"], distinctHapsIE=t["distinctHapsIE"],
        eMatDH_special_grid_which=t["eMatDH_special_grid_which"],
        eMatDH_special_values_list=t["eMatDH_special_values_list"],
        eMatDH_special_matrix=t["eMatDH_special_matrix"],
        eMatDH_special_matrix_helper=t["eMatDH_special_matrix_helper"],
        transMatRate_t=transMatRate_t, L=L, L_grid=L_grid, grid=grid,
        extra=dict(seed=seed, nGen=nGen, expRate=expRate, spectrum="1/i"),
    )


@dataclass
class SampleReads:
    """Flattened ``sampleReads`` (one sample)."""

    read_ptr: np.ndarray   # int32 R+1
    u: np.ndarray          # int32 sum(J+1): 0-based SNP index
    bq: np.ndarray         # int32 sum(J+1): signed phred
    wif: np.ndarray        # int32 R: 0-based grid of the central SNP (non-decreasing)
    truth_label: Optional[np.ndarray] = None  # int32 R: 1-based generating haplotype
    truth_haps: Optional[np.ndarray] = None   # int8 H x T
    ff: float = 0.0
    all_snp: Optional["SampleReads"] = None   # the same sample read over ALL SNPs (impute_rare_common)

    @property
    def nReads(self) -> int:
        return len(self.wif)

    def as_list(self):
        """R-style list of (J, wif, bq, u) for readability in tests."""
        out = []
        for r in range(self.nReads):
            s, e = self.read_ptr[r], self.read_ptr[r + 1]
            out.append((int(e - s - 1), int(self.wif[r]), self.bq[s:e].copy(), self.u[s:e].copy()))
        return out


def panel_hap_bits(panel: Panel, k: int) -> np.ndarray:
    """0/1 alleles of panel haplotype ``k`` over all SNPs."""
    assert panel.rhb_t is not None
    return int_expand(panel.rhb_t[k, :]).reshape(-1)[: panel.nSNPs]


def make_truth_haplotype(panel: Panel, rng, n_segments=(3, 6)) -> np.ndarray:
    nseg = int(rng.integers(n_segments[0], n_segments[1] + 1))
    cuts = np.sort(rng.choice(np.arange(1, panel.nSNPs), size=nseg - 1, replace=False)) if nseg > 1 else np.array([], dtype=int)
    bounds = np.r_[0, cuts, panel.nSNPs]
    hap = np.zeros(panel.nSNPs, dtype=np.int8)
    for i in range(nseg):
        k = int(rng.integers(0, panel.K))
        hap[bounds[i]:bounds[i + 1]] = panel_hap_bits(panel, k)[bounds[i]:bounds[i + 1]]
    return hap


def make_synthetic_sample(panel: Panel, seed: int, mode: str = "short", n_reads: Optional[int] = None,
                          ff: float = 0.0) -> SampleReads:
    """One sample's reads.  ``mode``: "short" (1x Illumina-like), "ont" (long noisy reads);
    ``ff`` > 0 makes an NIPT read mixture over (mat-T, mat-U, pat-T) as functions.R:586."""
    rng = np.random.default_rng(seed)
    T = panel.nSNPs
    H = 3 if ff > 0 else 2
    truth = np.stack([make_truth_haplotype(panel, rng) for _ in range(H)], axis=0)
    if mode == "short":
        if n_reads is None:
            n_reads = max(2, int(round(T * 20000 / 64000)))
        nsnp = np.minimum(1 + rng.poisson(2, size=n_reads), 8)
        phred_lo, phred_hi = 20, 40
    elif mode == "ont":
        if n_reads is None:
            n_reads = max(2, int(round(T * 300 / 64000)))
        nsnp = rng.integers(200, 801, size=n_reads)
        phred_lo, phred_hi = 5, 15
    else:
        raise ValueError(mode)
    nsnp = np.minimum(nsnp, T)
    start = np.array([rng.integers(0, T - n + 1) for n in nsnp], dtype=np.int64)
    central = start + (nsnp - 1) // 2
    order = np.argsort(central, kind="stable")
    start, nsnp, central = start[order], nsnp[order], central[order]
    if H == 2:
        label = rng.integers(1, 3, size=n_reads)
    else:
        label = rng.choice([1, 2, 3], p=[0.5, 0.5 - ff / 2, ff / 2], size=n_reads)
    read_ptr = np.zeros(n_reads + 1, dtype=np.int32)
    read_ptr[1:] = np.cumsum(nsnp)
    total = int(read_ptr[-1])
    u = np.zeros(total, dtype=np.int32)
    bq = np.zeros(total, dtype=np.int32)
    for r in range(n_reads):
        s, e = read_ptr[r], read_ptr[r + 1]
        idx = np.arange(start[r], start[r] + nsnp[r])
        u[s:e] = idx
        phred = rng.integers(phred_lo, phred_hi + 1, size=nsnp[r])
        allele = truth[label[r] - 1, idx].astype(np.int32)
        err = rng.random(nsnp[r]) < 10.0 ** (-phred / 10.0)
        allele = np.where(err, 1 - allele, allele)
        bq[s:e] = np.where(allele == 1, phred, -phred)
    wif = (central // 32).astype(np.int32)
    return SampleReads(read_ptr=read_ptr, u=u, bq=bq, wif=wif, truth_label=label.astype(np.int32),
                       truth_haps=truth, ff=ff)


def make_rare_common(panel: Panel, seed: int, n_rare: Optional[int] = None, carriers=(0, 4)) -> RareCommon:
    """Rare SNPs scattered between the panel's (common) SNPs, as ``QUILT_prepare_reference(impute_rare_common=TRUE)``
    leaves them (quilt-prepare-reference.R: ``rare_per_hap_info``, ``snp_is_common``): each rare SNP gets
    ``carriers[0]..carriers[1]`` panel haplotypes that carry its alt allele."""
    rng = np.random.default_rng(seed)
    T = panel.nSNPs
    if n_rare is None:
        n_rare = 2 * T
    lo, hi = int(panel.L[0]), int(panel.L[-1])
    free = np.setdiff1d(np.arange(lo, hi + 1), panel.L)
    rare_pos = np.sort(rng.choice(free, size=min(n_rare, len(free)), replace=False))
    L_all = np.sort(np.concatenate([panel.L, rare_pos])).astype(np.int64)
    T_all = len(L_all)
    snp_is_common = np.isin(L_all, panel.L).astype(np.uint8)
    common_snp_index = np.zeros(T_all, dtype=np.int32)
    common_snp_index[snp_is_common == 1] = np.arange(1, T + 1, dtype=np.int32)
    G_all = (T_all + 31) // 32
    starts = np.arange(0, T_all, 32)
    L_grid_all = (np.add.reduceat(L_all, starts) // np.diff(np.r_[starts, T_all])).astype(np.int64)
    ex = panel.extra
    sigma = _sigma_from_positions(L_grid_all, ex.get("nGen", 100.0), ex.get("expRate", 1.0))
    tm = np.asfortranarray(np.stack([sigma, 1.0 - sigma], axis=0))
    rare_idx = np.flatnonzero(snp_is_common == 0)
    n_car = rng.integers(carriers[0], carriers[1] + 1, size=len(rare_idx))
    snp = np.repeat(rare_idx, n_car).astype(np.int64)
    hap = rng.integers(0, panel.K, size=len(snp)).astype(np.int64)
    pairs = np.unique(hap * T_all + snp)            # distinct (haplotype, SNP), sorted by haplotype then SNP
    hap, snp = pairs // T_all, pairs % T_all
    rare_ptr = np.zeros(panel.K + 1, dtype=np.int64)
    rare_ptr[1:] = np.cumsum(np.bincount(hap, minlength=panel.K))
    rare_snp = (snp + 1).astype(np.int32)
    return RareCommon(nSNPs_all=T_all, nGrids_all=G_all, snp_is_common=snp_is_common,
                      common_snp_index=common_snp_index, rare_ptr=rare_ptr, rare_snp=rare_snp,
                      transMatRate_t_all=tm, L_all=L_all, L_grid_all=L_grid_all)


def rare_common_hap_bits(panel: Panel, rc: RareCommon, k: int) -> np.ndarray:
    """0/1 alleles of panel haplotype ``k`` over all (common + rare) SNPs."""
    out = np.zeros(rc.nSNPs_all, dtype=np.int8)
    out[rc.snp_is_common == 1] = panel_hap_bits(panel, k)
    out[rc.rare_snp[rc.rare_ptr[k]:rc.rare_ptr[k + 1]] - 1] = 1
    return out


def make_synthetic_sample_rare_common(panel: Panel, rc: RareCommon, seed: int, n_reads: Optional[int] = None,
                                      ff: float = 0.0):
    """One diploid sample read twice, as get_and_impute_one_sample does with ``impute_rare_common``
    (functions.R:130-175): over all SNPs (``allSNP_sampleReads``) and over the common SNPs only.  Returns
    ``(sample_common, sample_all)``; reads without a common SNP are absent from the first."""
    rng = np.random.default_rng(seed)
    T_all = rc.nSNPs_all
    truth = []
    for _ in range(3 if ff > 0 else 2):
        nseg = int(rng.integers(3, 7))
        cuts = np.sort(rng.choice(np.arange(1, T_all), size=nseg - 1, replace=False))
        bounds = np.r_[0, cuts, T_all]
        hap = np.zeros(T_all, dtype=np.int8)
        for i in range(nseg):
            k = int(rng.integers(0, panel.K))
            hap[bounds[i]:bounds[i + 1]] = rare_common_hap_bits(panel, rc, k)[bounds[i]:bounds[i + 1]]
        truth.append(hap)
    truth = np.stack(truth, axis=0)
    if n_reads is None:
        n_reads = max(2, int(round(panel.nSNPs * 20000 / 64000)))
    nsnp = np.minimum(1 + rng.poisson(2 * T_all / panel.nSNPs, size=n_reads), 24)
    nsnp = np.minimum(nsnp, T_all)
    start = np.array([rng.integers(0, T_all - n + 1) for n in nsnp], dtype=np.int64)
    central = start + (nsnp - 1) // 2
    order = np.argsort(central, kind="stable")
    start, nsnp, central = start[order], nsnp[order], central[order]
    label = (rng.choice(3, size=n_reads, p=[0.5, 0.5 - ff / 2, ff / 2]) + 1) if ff > 0 else rng.integers(1, 3, size=n_reads)
    us, bqs = [], []
    for r in range(n_reads):
        idx = np.arange(start[r], start[r] + nsnp[r])
        phred = rng.integers(20, 41, size=nsnp[r])
        allele = truth[label[r] - 1, idx].astype(np.int32)
        err = rng.random(nsnp[r]) < 10.0 ** (-phred / 10.0)
        allele = np.where(err, 1 - allele, allele)
        us.append(idx.astype(np.int32))
        bqs.append(np.where(allele == 1, phred, -phred).astype(np.int32))

    def pack(us, bqs, labels, T):
        keep = [i for i, x in enumerate(us) if len(x)]
        us, bqs = [us[i] for i in keep], [bqs[i] for i in keep]
        cen = np.array([x[(len(x) - 1) // 2] for x in us], dtype=np.int64)
        o = np.argsort(cen, kind="stable")
        us, bqs, cen = [us[i] for i in o], [bqs[i] for i in o], cen[o]
        ptr = np.zeros(len(us) + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([len(x) for x in us])
        return SampleReads(read_ptr=ptr, u=np.concatenate(us).astype(np.int32), bq=np.concatenate(bqs).astype(np.int32),
                           wif=(cen // 32).astype(np.int32), truth_label=np.asarray(labels)[keep][o].astype(np.int32))

    s_all = pack(us, bqs, label, T_all)
    s_all.truth_haps = truth
    com_u, com_bq = [], []
    for x, b in zip(us, bqs):
        m = rc.snp_is_common[x] == 1
        com_u.append(rc.common_snp_index[x[m]] - 1)
        com_bq.append(b[m])
    s_com = pack(com_u, com_bq, label, panel.nSNPs)
    s_com.truth_haps = truth[:, rc.snp_is_common == 1]
    s_com.ff = s_all.ff = ff
    s_com.all_snp = s_all
    return s_com, s_all


# ---------------------------------------------------------------------------------------------------------------------------
# synthetic BAM files (input-side data generator for `bench.py --bam`): one plain-M alignment per read
# ---------------------------------------------------------------------------------------------------------------------------
def synthetic_alleles(nSNPs: int, seed: int = 0):
    """A reference / alternate base per SNP (two different letters of ACGT)."""
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=nSNPs)
    alt = (ref + rng.integers(1, 4, size=nSNPs)) % 4
    letters = np.array(list("ACGT"))
    return "".join(letters[ref]), "".join(letters[alt])


def write_synthetic_bam(path: str, sample: SampleReads, L: np.ndarray, ref: str, alt: str, chrom: str = "chr20",
                        seed: int = 0, mapq: int = 60) -> None:
    """The reads of ``sample`` as a coordinate-sorted BAM (BGZF, SAM spec 4.1-4.2): read r becomes one alignment spanning its
    first to last SNP, showing the ref / alt allele with quality |bq| at its SNPs and a third base at every other SNP it
    crosses, so that a pile-up over ``L`` gives back exactly the sample's (u, bq) lists."""
    import struct
    import zlib
    rng = np.random.default_rng(seed)
    L = np.asarray(L, dtype=np.int64)
    code = {"=": 0, "A": 1, "C": 2, "G": 4, "T": 8}
    lut = np.zeros(256, dtype=np.uint8)
    for k, v in code.items():
        lut[ord(k)] = v
    refb, altb = np.frombuffer(ref.encode(), dtype=np.uint8), np.frombuffer(alt.encode(), dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    # a base that is neither allele, per SNP
    other = np.array([next(c for c in acgt if c != a and c != b) for a, b in zip(refb, altb)], dtype=np.uint8)
    recs = []
    for r in range(sample.nReads):
        a, b = sample.read_ptr[r], sample.read_ptr[r + 1]
        us, bqs = sample.u[a:b], sample.bq[a:b]
        start, end = int(L[us[0]]), int(L[us[-1]])
        n = end - start + 1
        seq = acgt[rng.integers(0, 4, size=n)].copy()
        qual = rng.integers(20, 41, size=n).astype(np.uint8)
        lo, hi = np.searchsorted(L, start), np.searchsorted(L, end, side="right")
        at = (L[lo:hi] - start).astype(np.int64)
        seq[at] = other[lo:hi]
        at_u = (L[us] - start).astype(np.int64)
        seq[at_u] = np.where(bqs > 0, altb[us], refb[us])
        qual[at_u] = np.abs(bqs).astype(np.uint8)
        nib = lut[seq]
        if n & 1:
            nib = np.append(nib, 0)
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)
        name = b"r%d\0" % r
        body = (struct.pack("<iiBBHHHIiii", 0, start - 1, len(name), mapq, 4680, 1, 0, n, -1, -1, 0) + name +
                struct.pack("<I", (n << 4) | 0) + packed.tobytes() + qual.tobytes())
        recs.append((start, struct.pack("<i", len(body)) + body))
    recs.sort(key=lambda t: t[0])
    text = f"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:{chrom}\tLN:{int(L[-1]) + 1000}\n".encode()
    nm = chrom.encode() + b"\0"
    raw = (b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", len(nm)) + nm +
           struct.pack("<i", int(L[-1]) + 1000) + b"".join(x for _, x in recs))

    def block(data: bytes) -> bytes:
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
    with open(path, "wb") as f:
        for i in range(0, len(raw), 0xff00):
            f.write(block(raw[i:i + 0xff00]))
        f.write(block(b""))
