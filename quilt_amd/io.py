"""Host-side formats either side of the hot path (include/quilt_amd_io.h), named after the
reference functions they stand in for:

  * ``loadBamAndConvert`` + ``snap_sampleReads_to_grid``  (QUILT/R/functions.R:243-298)   BAM -> flattened ``SampleReads``
  * ``make_per_sample_vcf_col`` / ``make_per_sample_vcf_col_nipt``  (functions.R:1408-1463)  one sample's VCF column
  * ``per_sample_counts`` / ``SummaryCounts``  (functions.R:1382-1418, quilt.R:955-961)   the four cross-sample count arrays
  * ``make_and_write_output_file``  (QUILT/R/writers.R:1-128)                             header, INFO, body, BGZF

The string building, BGZF framing, BAM parsing and the exact HWE test run in the native library (csrc/hostio.cpp); this
module only marshals.  There is no pure-Python fallback: without ``libquilt_amd.so`` the calls raise.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from .native import QA_ERR_CAPACITY, QuiltAmdError, lib, ptr
from .synth import SampleReads


class BamOpts(C.Structure):
    _fields_ = [("bqFilter", C.c_int32), ("iSizeUpperLimit", C.c_int32), ("useSoftClippedBases", C.c_int32),
                ("downsampleToCov", C.c_int32), ("chrStart", C.c_int32), ("chrEnd", C.c_int32),
                ("merge_mates", C.c_int32), ("seed", C.c_uint64)]


def _io_lib():
    L = lib()
    if not getattr(L, "_io_ready", False):
        for name in ("qa_bam_load_sample_reads", "qa_sample_reads_n_reads", "qa_sample_reads_export", "qa_vcf_column_diploid",
                     "qa_vcf_column_nipt", "qa_vcf_info_column", "qa_vcf_write_body", "qa_vcf_write_text", "qa_hwe_exact",
                     "qa_accumulate_dosage", "qa_consensus_read_labels"):
            getattr(L, name).restype = C.c_int
        L.qa_sample_reads_n_bases.restype = C.c_int64
        L.qa_vcf_missing_entry.restype = C.c_char_p
        L.qa_sample_reads_destroy.restype = None
        L.qa_sample_reads_stats.restype = None
        L.qa_bam_opts_default.restype = None
        L._io_ready = True
    return L


def _check(st: int, what: str):
    if st < 0:
        detail = ""
        if st == -3:   # QA_ERR_UNSUPPORTED carries the library's explanation (e.g. CRAM input and how to convert it)
            try:
                lb = _io_lib()
                lb.qa_last_error.restype = C.c_char_p
                detail = ": " + lb.qa_last_error().decode()
            except Exception:
                pass
        raise QuiltAmdError(st, what + detail)


def accumulate_dosage(hap: np.ndarray, chain_sample: np.ndarray, dosage: np.ndarray, gp_t: np.ndarray,
                      fet_dosage: Optional[np.ndarray] = None, fet_gp_t: Optional[np.ndarray] = None) -> None:
    """functions.R:999-1020 for all chains of a round (native, one pass): ``hap`` [n_chain, n_label, T] haploid dosages,
    ``dosage`` [n_sample, T] and ``gp_t`` [n_sample, 3, T] running sums, updated in place."""
    n_chain, n_label, T = hap.shape
    for a in (hap, dosage, gp_t, fet_dosage, fet_gp_t):
        if a is not None and (a.dtype != np.float64 or not a.flags.c_contiguous):
            raise ValueError("accumulate_dosage takes C-contiguous float64 arrays")
    cs = np.ascontiguousarray(chain_sample, dtype=np.int32)
    _check(_io_lib().qa_accumulate_dosage(C.c_int32(n_chain), C.c_int32(n_label), C.c_int32(T), ptr(hap), ptr(cs),
                                          C.c_int32(dosage.shape[0]), ptr(dosage), ptr(gp_t), ptr(fet_dosage), ptr(fet_gp_t)),
           "qa_accumulate_dosage")


def consensus_read_labels(labels: np.ndarray, p: np.ndarray, can_hap: int, minrp: float = 0.95) -> np.ndarray:
    """Read confidence + consensus labels of one sample before its phasing pass (functions.R:1615-1660, :1680-1784, NIPT
    :1788-1829), native: ``labels`` [n Gibbs samples, nReads], ``p`` [n, K, nReads] read likelihoods against each Gibbs sample's
    K haplotypes (K = 3: NIPT).  The numpy text of the same functions is in quilt_amd/driver.py (tested equal)."""
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float64)
    n, R = labels.shape
    if p.shape[0] != n or p.shape[2] != R or p.shape[1] not in (2, 3):
        raise ValueError("p must be [n, 2 or 3, nReads]")
    out = np.zeros(R, dtype=np.int32)
    _check(_io_lib().qa_consensus_read_labels(C.c_int32(R), C.c_int32(n), ptr(labels), ptr(p), C.c_int32(p.shape[1]),
                                              C.c_double(minrp), C.c_int32(can_hap), ptr(out)), "qa_consensus_read_labels")
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# f3
# ---------------------------------------------------------------------------------------------------------------------------
def loadBamAndConvert(bam_file: str, chr: str, L: np.ndarray, ref: Sequence[str], alt: Sequence[str],
                      grid: Optional[np.ndarray] = None, *, bqFilter: int = 17, iSizeUpperLimit: float = 1e6,
                      useSoftClippedBases: bool = False, downsampleToCov: int = 30, chrStart: int = 0, chrEnd: int = 0,
                      merge_mates: bool = True, seed: int = 1, return_stats: bool = False):
    """One sample's reads over the region's SNPs, already snapped to the grid (functions.R:243-298).
    ``L`` 1-based ascending positions, ``ref`` / ``alt`` one character per SNP, ``grid`` 0-based grid per SNP (default
    ``snp // 32``, STITCH::assign_positions_to_grid with gridWindowSize = 32 SNPs as QUILT uses it)."""
    lb = _io_lib()
    L = np.ascontiguousarray(L, dtype=np.int32)
    T = len(L)
    grid = (np.arange(T, dtype=np.int32) // 32) if grid is None else np.ascontiguousarray(grid, dtype=np.int32)
    refb = "".join(ref).encode() if not isinstance(ref, (bytes, bytearray)) else bytes(ref)
    altb = "".join(alt).encode() if not isinstance(alt, (bytes, bytearray)) else bytes(alt)
    if len(refb) != T or len(altb) != T or len(grid) != T:
        raise ValueError("L, ref, alt and grid must describe the same SNPs (one character per allele)")
    o = BamOpts(int(bqFilter), int(min(iSizeUpperLimit, 2**31 - 1)), int(bool(useSoftClippedBases)), int(downsampleToCov),
                int(chrStart), int(chrEnd), int(bool(merge_mates)), int(seed))
    h = C.c_void_p()
    _check(lb.qa_bam_load_sample_reads(bam_file.encode(), chr.encode(), C.c_int32(T), ptr(L), refb, altb, ptr(grid),
                                       C.byref(o), C.byref(h)), f"cannot load {bam_file} ({chr})")
    try:
        R = lb.qa_sample_reads_n_reads(h)
        nb = lb.qa_sample_reads_n_bases(h)
        read_ptr = np.zeros(R + 1, dtype=np.int32)
        u, bq = np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32)
        wif = np.zeros(R, dtype=np.int32)
        _check(lb.qa_sample_reads_export(h, ptr(read_ptr), ptr(u), ptr(bq), ptr(wif), None), "export")
        stats = np.zeros(8, dtype=np.int64)
        lb.qa_sample_reads_stats(h, ptr(stats))
    finally:
        lb.qa_sample_reads_destroy(h)
    s = SampleReads(read_ptr=read_ptr, u=u, bq=bq, wif=wif)
    if return_stats:
        names = ("alignments_on_chr", "used", "low_mapq", "insert_size", "flagged", "removed_by_coverage_cap",
                 "mates_merged", "no_site")
        return s, dict(zip(names, (int(x) for x in stats)))
    return s


# ---------------------------------------------------------------------------------------------------------------------------
# f4
# ---------------------------------------------------------------------------------------------------------------------------
@dataclass
class VcfColumn:
    """One sample's column: entries back to back, NUL-terminated; ``off[t]`` is entry t's start."""
    buf: np.ndarray
    off: np.ndarray

    def __getitem__(self, t: int) -> str:
        return bytes(self.buf[self.off[t]:self.off[t + 1] - 1]).decode()

    def __len__(self) -> int:
        return len(self.off) - 1

    def tolist(self) -> List[str]:
        return bytes(self.buf[:self.off[-1]]).decode().split("\0")[:-1]


def _two_pass(fn, T: int, *args) -> VcfColumn:
    off = np.zeros(T + 1, dtype=np.int64)
    need = C.c_int64(0)
    cap = 48 * T + 64
    while True:
        buf = np.zeros(cap, dtype=np.uint8)
        st = fn(C.c_int32(T), *args, ptr(buf), C.c_int64(cap), ptr(off), C.byref(need))
        if st == QA_ERR_CAPACITY:
            cap = int(need.value)
            continue
        _check(st, fn.__name__)
        return VcfColumn(buf, off)


def _f64(a, order="C"):
    return np.ascontiguousarray(a, dtype=np.float64) if order == "C" else np.asfortranarray(a, dtype=np.float64)


def make_per_sample_vcf_col(gp_t: np.ndarray, phasing_haps: np.ndarray, output_gt_phased_genotypes: bool = True) -> VcfColumn:
    """functions.R:1420-1441 (``STITCH::rcpp_make_column_of_vcf`` + the phased-GT paste).  ``gp_t`` 3 x T,
    ``phasing_haps`` T x 2."""
    gp = _f64(gp_t, "F")
    hd = _f64(phasing_haps, "F")
    T = gp.shape[1]
    if gp.shape[0] != 3 or hd.shape != (T, 2):
        raise ValueError("gp_t must be 3 x T and phasing_haps T x 2")
    return _two_pass(_io_lib().qa_vcf_column_diploid, T, ptr(gp), ptr(hd), C.c_int32(int(bool(output_gt_phased_genotypes))))


def make_per_sample_vcf_col_nipt(mat_gp_t, fet_gp_t, phasing_haps, mat_dosage, fet_dosage) -> VcfColumn:
    """functions.R:1443-1459: GT:MGP:MDS:FGP:FDS with R's paste0(round(x, 3)) numbers."""
    m, f, hd = _f64(mat_gp_t, "F"), _f64(fet_gp_t, "F"), _f64(phasing_haps, "F")
    T = m.shape[1]
    if m.shape[0] != 3 or f.shape != m.shape or hd.shape != (T, 3):
        raise ValueError("mat_gp_t / fet_gp_t must be 3 x T and phasing_haps T x 3")
    return _two_pass(_io_lib().qa_vcf_column_nipt, T, ptr(m), ptr(f), ptr(hd), ptr(_f64(mat_dosage)), ptr(_f64(fet_dosage)))


def missing_entry() -> str:
    return _io_lib().qa_vcf_missing_entry().decode()


def per_sample_counts(gp_t: np.ndarray, sample: SampleReads, nSNPs: int):
    """eij, fij, max_gen and the pileup allele counts of one sample (functions.R:1382-1418).  ``max_gen`` is the 0-based
    most likely genotype per SNP (STITCH::get_max_gen_rapid's column index; first maximum on ties)."""
    gp_t = np.asarray(gp_t, dtype=np.float64)
    eij = np.round(gp_t[1] + 2 * gp_t[2], 3)
    fij = np.round(gp_t[1] + 4 * gp_t[2], 3)
    max_gen = np.argmax(gp_t, axis=0)
    from .driver import phred_eps
    eps = phred_eps(sample.bq)   # (pow() of the C library, as R's 10^x and the native code: driver._phred_eps_table)
    p_ref = np.where(sample.bq < 0, 1 - eps, eps / 3)   # STITCH::convertScaledBQtoProbs column 1
    p_alt = np.where(sample.bq < 0, eps / 3, 1 - eps)   # column 2
    c1 = np.bincount(sample.u, weights=p_ref, minlength=nSNPs)
    c2 = np.bincount(sample.u, weights=p_alt, minlength=nSNPs)
    return eij, fij, max_gen, np.stack([c2, c1 + c2], axis=1)


@dataclass
class SummaryCounts:
    """The four per-SNP arrays every worker sums over its samples and the writer adds up across workers -- the only
    cross-shard reduction of a run (quilt.R:697-699, 955-961; writers.R:38-47)."""
    nSNPs: int
    N: int = 0
    hweCount: np.ndarray = field(default=None)
    infoCount: np.ndarray = field(default=None)
    afCount: np.ndarray = field(default=None)
    alleleCount: np.ndarray = field(default=None)

    def __post_init__(self):
        T = self.nSNPs
        self.hweCount = np.zeros((T, 3)) if self.hweCount is None else self.hweCount
        self.infoCount = np.zeros((T, 2)) if self.infoCount is None else self.infoCount
        self.afCount = np.zeros(T) if self.afCount is None else self.afCount
        self.alleleCount = np.zeros((T, 2)) if self.alleleCount is None else self.alleleCount

    def add_sample(self, eij, fij, max_gen, per_sample_alleleCount):
        self.infoCount[:, 0] += eij
        self.infoCount[:, 1] += fij - eij ** 2
        self.afCount += eij / 2
        self.hweCount[np.arange(self.nSNPs), max_gen] += 1
        self.alleleCount += per_sample_alleleCount

    def merge(self, other: "SummaryCounts"):
        for name in ("hweCount", "infoCount", "afCount", "alleleCount"):
            getattr(self, name).__iadd__(getattr(other, name))
        return self

    def as_vector(self) -> np.ndarray:
        """One flat buffer for the cross-rank sum (sharding.reduce_counts)."""
        return np.concatenate([self.hweCount.ravel(), self.infoCount.ravel(), self.afCount, self.alleleCount.ravel()])

    def from_vector(self, v: np.ndarray):
        T = self.nSNPs
        self.hweCount = v[:3 * T].reshape(T, 3).copy()
        self.infoCount = v[3 * T:5 * T].reshape(T, 2).copy()
        self.afCount = v[5 * T:6 * T].copy()
        self.alleleCount = v[6 * T:8 * T].reshape(T, 2).copy()
        return self

    def finalize(self, N: int):
        """writers.R:48-58: allele counts with their ratio, INFO score, estimated allele frequency, HWE p-value."""
        with np.errstate(divide="ignore", invalid="ignore"):
            ac = np.column_stack([self.alleleCount, self.alleleCount[:, 0] / self.alleleCount[:, 1]])
            thetaHat = self.infoCount[:, 0] / 2 / N
            denom = 2 * N * thetaHat * (1 - thetaHat)
            info = 1 - self.infoCount[:, 1] / denom
        r2 = np.round(thetaHat, 2)
        info[(r2 == 0) | (r2 == 1)] = 1
        info[info < 0] = 0
        eaf = self.afCount / N
        hwe = np.zeros(self.nSNPs)
        counts = np.asfortranarray(self.hweCount, dtype=np.float64)
        _check(_io_lib().qa_hwe_exact(C.c_int32(self.nSNPs), ptr(counts), ptr(hwe)), "qa_hwe_exact")
        return dict(alleleCount=ac, info=info, estimatedAlleleFrequency=eaf, hwe=hwe)


_INFO_LINES = (
    ("INFO_SCORE", "Info score", "Info score from maternal genotype posteriors"),
    ("EAF", "Estimated allele frequency", "Estimated allele frequency"),
    ("HWE", "Hardy-Weinberg p-value", "Hardy-Weinberg p-value from maternal genotypes"),
    ("ERC", "Estimated number of copies of the reference allele from the pileup",) * 2,
    ("EAC", "Estimated number of copies of the alternate allele from the pileup",) * 2,
    ("PAF", "Estimated allele frequency using the pileup of reference and alternate alleles",) * 2,
)


def vcf_header(sampleNames: Sequence[str], method: str = "diploid", output_gt_phased_genotypes: bool = True) -> str:
    """The meta lines and column header of writers.R:215-285 (VCFv4.0; the same IDs, types and descriptions, so files read
    by the reference's consumers read the same)."""
    nipt = method == "nipt"
    lines = ["##fileformat=VCFv4.0"]
    for row in _INFO_LINES:
        ident, desc = row[0], (row[-1] if nipt else row[1])
        lines.append(f'##INFO=<ID={ident},Number=.,Type=Float,Description="{desc}">')
    fmt = lambda i, n, t, d: f'##FORMAT=<ID={i},Number={n},Type={t},Description="{d}">'
    if nipt:
        lines += [fmt("GT", 1, "String", "Phased genotypes in order of maternal transmitted, maternal untransmitted, and fetal transmitted"),
                  fmt("MGP", 3, "Float", "Maternal Posterior genotype probability of 0/0, 0/1, and 1/1"),
                  fmt("MDS", 1, "Float", "Maternal Diploid dosage"),
                  fmt("FGP", 3, "Float", "Fetal Posterior genotype probability of 0/0, 0/1, and 1/1"),
                  fmt("FDS", 1, "Float", "Fetal Diploid dosage")]
    else:
        gt = "Phased genotypes" if output_gt_phased_genotypes else \
            "Most likely genotype, given posterior probability of at least 0.90"
        lines += [fmt("GT", 1, "String", gt),
                  fmt("GP", 3, "Float", "Posterior genotype probability of 0/0, 0/1, and 1/1"),
                  fmt("DS", 1, "Float", "Diploid dosage"),
                  fmt("HD", 2, "Float", "Haploid dosages")]
    cols = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + list(sampleNames)
    return "\n".join(lines) + "\n" + "\t".join(cols) + "\n"


def make_and_write_output_file(output_filename: str, sampleNames: Sequence[str], chr: str, pos_bp: np.ndarray,
                               ref: Sequence[str], alt: Sequence[str], columns: Sequence[Optional[VcfColumn]],
                               counts: SummaryCounts, inRegion2: Optional[np.ndarray] = None, method: str = "diploid",
                               output_gt_phased_genotypes: bool = True) -> dict:
    """writers.R:1-128: header, INFO from the summed counts, one line per SNP inside the region; ``.gz`` names are written
    as BGZF (what ``bgzip`` would make of the text).  A ``None`` column is a sample that was not imputed."""
    lb = _io_lib()
    T = counts.nSNPs
    N = len(sampleNames)
    if len(columns) != N:
        raise ValueError("one column per sample name")
    bgzf = int(output_filename.endswith(".gz"))
    fin = counts.finalize(N)
    info = _two_pass(lb.qa_vcf_info_column, T, ptr(_f64(fin["estimatedAlleleFrequency"])), ptr(_f64(fin["info"])),
                     ptr(_f64(fin["hwe"])), ptr(_f64(fin["alleleCount"], "F")))
    head = vcf_header(sampleNames, method, output_gt_phased_genotypes).encode()
    _check(lb.qa_vcf_write_text(output_filename.encode(), bgzf, 1, head, C.c_int64(len(head))), "header")
    fmt = b"GT:MGP:MDS:FGP:FDS" if method == "nipt" else b"GT:GP:DS:HD"
    col_ptrs = (C.c_void_p * max(N, 1))(*[None if c is None else c.buf.ctypes.data for c in columns])
    off_ptrs = (C.c_void_p * max(N, 1))(*[None if c is None else c.off.ctypes.data for c in columns])
    keep = None if inRegion2 is None else np.ascontiguousarray(inRegion2, dtype=np.uint8)
    pos_bp = np.ascontiguousarray(pos_bp, dtype=np.int32)
    _check(lb.qa_vcf_write_body(output_filename.encode(), bgzf, 1, chr.encode(), C.c_int32(T), ptr(pos_bp),
                                "".join(ref).encode(), "".join(alt).encode(), ptr(keep), ptr(info.buf), ptr(info.off), fmt,
                                C.c_int32(N), col_ptrs, off_ptrs), "body")
    return fin


def impute_bams_to_vcf(panel, backend, bam_files: Sequence[str], sampleNames: Sequence[str], chr: str, ref: Sequence[str],
                       alt: Sequence[str], output_filename: str, params=None, inRegion2: Optional[np.ndarray] = None,
                       minimum_number_of_sample_reads: int = 2, ff: Optional[Sequence[float]] = None,
                       output_gt_phased_genotypes: bool = True, **bam_opts):
    """The per-sample path end to end for one region: BAM -> sampleReads (f3) -> the driver loop on `backend` -> VCF
    columns and file (f4).  What get_and_impute_one_sample does between its ``loadBamAndConvert`` call and its return
    value, plus the writer (functions.R:243-298, 1408-1477; writers.R).  Samples with fewer than
    ``minimum_number_of_sample_reads`` reads are written as missing and left out of the counts (functions.R:274-287)."""
    from .driver import Driver, DriverParams
    params = params or DriverParams()
    if panel.L is None:
        raise ValueError("the panel carries no SNP positions (Panel.L)")
    grid = panel.grid if panel.grid is not None else np.arange(panel.nSNPs, dtype=np.int32) // 32
    samples, imputed = [], []
    for i, path in enumerate(bam_files):
        s = loadBamAndConvert(path, chr, panel.L, ref, alt, grid, **bam_opts)
        if ff is not None:
            s.ff = float(ff[i])
        if s.nReads < minimum_number_of_sample_reads:
            continue
        samples.append(s)
        imputed.append(i)
    results = Driver(panel, backend, params).run(samples) if samples else []
    counts = SummaryCounts(panel.nSNPs)
    cols: List[Optional[VcfColumn]] = [None] * len(bam_files)
    for i, s, r in zip(imputed, samples, results):
        counts.add_sample(*per_sample_counts(r.gp_t, s, panel.nSNPs))
        if params.method == "nipt":
            cols[i] = make_per_sample_vcf_col_nipt(r.gp_t, r.fet_gp_t, r.phasing_haps, r.dosage, r.fet_dosage)
        else:
            cols[i] = make_per_sample_vcf_col(r.gp_t, r.phasing_haps, output_gt_phased_genotypes)
    make_and_write_output_file(output_filename, sampleNames, chr, panel.L, ref, alt, cols, counts, inRegion2=inRegion2,
                               method=params.method, output_gt_phased_genotypes=output_gt_phased_genotypes)
    return dict(results=dict(zip(imputed, results)), columns=cols, counts=counts)
