"""Thin caller of the native per-sample driver ``qa_impute_samples`` (csrc/impute.cpp, include/quilt_amd.h): the body of
the reference's loop over a core's sample range (QUILT/R/quilt.R:688-996, ``get_and_impute_one_sample``
QUILT/R/functions.R:3-1500) as ONE native call.  Everything between the native compute calls -- the round loop, the
hand-over of ``which_haps_to_use``, accumulation, consensus labels, ``recast_haps``, the host threads per device -- is C++
there; this module only flattens ``SampleReads`` objects and wraps the outputs.  ``quilt_amd/driver.py`` keeps the same loop
in Python as the tested statement the native loop must equal bit for bit (tests/test_native_driver_cpu.py, tests/test_native_driver_gpu.py)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from .driver import DriverParams, SampleResult
from .native import check, lib, ptr


class ImputeParams(C.Structure):
    _fields_ = [
        ("nGibbsSamples", C.c_int32), ("n_seek_its", C.c_int32), ("n_burn_in_seek_its", C.c_int32),
        ("Ksubset", C.c_int32), ("Knew", C.c_int32), ("K_top_matches", C.c_int32),
        ("heuristic_match_thin", C.c_double),
        ("small_ref_panel_gibbs_iterations", C.c_int32), ("n_gibbs_sample_its", C.c_int32),
        ("small_ref_panel_block_gibbs_iterations", C.c_void_p), ("n_block_gibbs_iterations", C.c_int32),
        ("maxDifferenceBetweenReads", C.c_double), ("minGLValue", C.c_double), ("Jmax", C.c_int32),
        ("seed", C.c_uint64),
        ("use_mspbwt", C.c_int32), ("mspbwtL", C.c_int32), ("mspbwtM", C.c_int32),
        ("mspbwt_index", C.c_void_p),
        ("samples_per_launch_set", C.c_int32), ("no_fused_tails", C.c_int32),
        ("rare_common", C.c_void_p), ("nipt", C.c_void_p), ("sample_index", C.c_void_p),
        ("on_samples_done", C.c_void_p), ("on_samples_done_ctx", C.c_void_p), ("sample_source", C.c_void_p),
    ]


class SampleView(C.Structure):
    """qa_sample_view_t (include/quilt_amd.h): one sample's reads as a qa_sample_source_t hands them over."""
    _fields_ = [("n_reads", C.c_int32), ("read_ptr", C.c_void_p), ("u", C.c_void_p), ("bq", C.c_void_p), ("wif", C.c_void_p),
                ("n_reads_all", C.c_int32), ("read_ptr_all", C.c_void_p), ("u_all", C.c_void_p), ("bq_all", C.c_void_p),
                ("wif_all", C.c_void_p), ("read_labels", C.c_void_p)]


ACQUIRE_SAMPLE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(SampleView))
QA_END_OF_SAMPLES = 2


class SampleSource(C.Structure):
    """qa_sample_source_t: ``acquire(ctx, s, view)`` blocks until sample ``s`` of the call is there (QA_OK, view filled), reports
    that the range ended before it (QA_END_OF_SAMPLES) or fails (< 0)."""
    _fields_ = [("acquire", ACQUIRE_SAMPLE), ("ctx", C.c_void_p)]


def sample_source_over(samples, labels_out, n_available=None, fail_at=None, order_log=None):
    """A SampleSource over samples already in memory (what a loader thread would hand over one by one): sample ``s`` of the call is
    ``samples[s]``, its labels go to ``labels_out[s]`` (int32 arrays of nReads); the range ends at ``n_available`` (default: all).
    Returns (SampleSource, keep-alive objects)."""
    n_av = len(samples) if n_available is None else int(n_available)
    hold = {}

    def arrays(r):
        return tuple(np.ascontiguousarray(getattr(r, k), dtype=np.int32) for k in ("read_ptr", "u", "bq", "wif"))

    def acquire(_ctx, s, view):
        if order_log is not None:
            order_log.append(int(s))
        if fail_at is not None and s == fail_at:
            return -2
        if s >= n_av:
            return QA_END_OF_SAMPLES
        smp = samples[s]
        a = hold[s] = arrays(smp)
        v = view.contents
        v.n_reads = int(smp.nReads)
        v.read_ptr, v.u, v.bq, v.wif = (x.ctypes.data for x in a)
        v.read_labels = labels_out[s].ctypes.data
        al = getattr(smp, "all_snp", None)
        if al is not None:
            b = hold[(s, "all")] = arrays(al)
            v.n_reads_all = int(al.nReads)
            v.read_ptr_all, v.u_all, v.bq_all, v.wif_all = (x.ctypes.data for x in b)
        return 0

    cb = ACQUIRE_SAMPLE(acquire)
    return SampleSource(cb, None), (cb, hold, labels_out)


class ImputeNipt(C.Structure):
    _fields_ = [("ff", C.c_void_p), ("L_grid", C.c_void_p), ("shuffle_bin_radius", C.c_int32), ("fet_dosage", C.c_void_p),
                ("fet_gp_t", C.c_void_p)]


class ImputeRareCommon(C.Structure):
    _fields_ = [("handles", C.c_void_p), ("nSNPs_all", C.c_int32), ("nGrids_all", C.c_int32), ("snp_is_common", C.c_void_p),
                ("read_off", C.c_void_p), ("read_ptr", C.c_void_p), ("u", C.c_void_p), ("bq", C.c_void_p), ("wif", C.c_void_p),
                ("L_grid_all", C.c_void_p)]


STAT_NAMES = ("underflow_retries", "full_list_refetches", "device_selections", "gibbs_chain_calls", "gibbs_launches",
              "ms_gibbs", "ms_fullpass", "ms_host", "ms_consensus", "ms_finish", "ms_accumulate")


def flatten_samples(samples: Sequence):
    """``sampleReads`` of a range of samples in the flattened form of include/quilt_amd.h: read_off [n + 1], per sample
    R + 1 read_ptr entries, bases and wif back to back."""
    n = len(samples)
    read_off = np.zeros(n + 1, dtype=np.int32)
    for i, s in enumerate(samples):
        read_off[i + 1] = read_off[i] + s.nReads
    cat = lambda name: (np.concatenate([np.asarray(getattr(s, name), dtype=np.int32) for s in samples])
                        if n else np.zeros(0, dtype=np.int32))
    return read_off, cat("read_ptr"), cat("u"), cat("bq"), cat("wif")


def make_rare_common(rc, rc_handles, samples):
    """(ImputeRareCommon, keep-alive objects): the all-SNP side of the call -- ``rc`` a quilt_amd.panel.RareCommon, ``rc_handles``
    one native handle (c_void_p) per panel handle, the samples' ``all_snp`` reads flattened."""
    read_off, read_ptr, u, bq, wif = flatten_samples([s.all_snp for s in samples])
    is_common = np.ascontiguousarray(rc.snp_is_common, dtype=np.uint8)
    hs = (C.c_void_p * len(rc_handles))(*rc_handles)
    Lg = np.ascontiguousarray(rc.L_grid_all, dtype=np.int32)
    q = ImputeRareCommon(C.cast(hs, C.c_void_p), rc.nSNPs_all, rc.nGrids_all, ptr(is_common), ptr(read_off), ptr(read_ptr), ptr(u),
                         ptr(bq), ptr(wif), ptr(Lg))
    return q, (hs, is_common, read_off, read_ptr, u, bq, wif, Lg)


def make_nipt(panel, samples, shuffle_bin_radius: int, nSNPs_out: Optional[int] = None):
    """(ImputeNipt, the fetus' output arrays, keep-alive objects) for method = "nipt": the samples' fetal fractions, the
    panel's grid positions; ``nSNPs_out``: what the outputs cover (all SNPs with impute_rare_common)."""
    n, T = len(samples), (panel.nSNPs if nSNPs_out is None else int(nSNPs_out))
    ff = np.ascontiguousarray([float(s.ff) for s in samples], dtype=np.float64)
    Lg = np.ascontiguousarray(panel.L_grid, dtype=np.int32)
    fd, fg = np.zeros((n, T)), np.zeros((n, 3, T))
    return ImputeNipt(ptr(ff), ptr(Lg), int(shuffle_bin_radius), ptr(fd), ptr(fg)), fd, fg, (ff, Lg)


def make_params(P: DriverParams, samples_per_launch_set: int, mspbwt_index=None, fuse_tails: bool = True, rare_common=None, nipt=None):
    """(ImputeParams, keep-alive objects) from the Python driver's parameters; ``rare_common``: an ImputeRareCommon (with
    impute_rare_common); ``nipt``: an ImputeNipt (method = "nipt")."""
    if P.method == "nipt" and nipt is None:
        raise ValueError("method = 'nipt' needs make_nipt(...)")
    if P.impute_rare_common and rare_common is None:
        raise ValueError("impute_rare_common needs the all-SNP side (make_rare_common)")
    blocks = np.ascontiguousarray(P.small_ref_panel_block_gibbs_iterations, dtype=np.int32)
    if P.use_mspbwt and P.mspbwt_search != "scan":
        raise ValueError("qa_impute_samples runs the msPBWT neighbour scan (mspbwt_search = 'scan')")
    q = ImputeParams(P.nGibbsSamples, P.n_seek_its, -1 if P.n_burn_in_seek_its is None else P.n_burn_in_seek_its,
                     P.Ksubset, P.Knew, P.K_top_matches, P.heuristic_match_thin, P.small_ref_panel_gibbs_iterations,
                     P.n_gibbs_sample_its, ptr(blocks), len(blocks), P.maxDifferenceBetweenReads, P.minGLValue, P.Jmax,
                     P.seed, int(P.use_mspbwt), P.mspbwtL, P.mspbwtM,
                     mspbwt_index.handle if (P.use_mspbwt and mspbwt_index is not None) else None,
                     int(samples_per_launch_set), 0 if fuse_tails else 1,
                     C.cast(C.pointer(rare_common), C.c_void_p) if (P.impute_rare_common and rare_common is not None) else None,
                     C.cast(C.pointer(nipt), C.c_void_p) if (P.method == "nipt" and nipt is not None) else None)
    return q, (blocks, mspbwt_index, rare_common, nipt)


def wrap_results(samples, dosage, gp_t, haps, labels, nDosage, read_off, fet_dosage=None, fet_gp_t=None) -> List[SampleResult]:
    """One SampleResult per sample over the call's output arrays: rows and slices of them, no copies (``phasing_haps`` is the
    nSNPs x n_label transposed VIEW of the library's n_label x nSNPs rows)."""
    return [SampleResult(dosage[i], gp_t[i], haps[i].T, labels[read_off[i]:read_off[i + 1]],
                         int(nDosage[i]), fet_dosage=None if fet_dosage is None else fet_dosage[i],
                         fet_gp_t=None if fet_gp_t is None else fet_gp_t[i]) for i in range(len(samples))]


class PreparedRange:
    """A sample range in the form the C ABI takes it -- the reads of all samples back to back (``flatten_samples``), the
    parameter structs, the output arrays (allocated, not touched: the library zeroes an accumulator row when it starts the
    sample's launch set) -- so that a caller who already holds flat buffers (the R shim flattens R's lists in C) can be timed
    from there: ``prepare_range`` + ``run_prepared`` = ``impute_samples``."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def prepare_range(devs: Sequence, samples: Sequence, params: Optional[DriverParams] = None, sample_offset: int = 0,
                  samples_per_launch_set: int = 256, fuse_tails: bool = True, drcs: Sequence = ()) -> PreparedRange:
    panel = devs[0].panel
    P = (params or DriverParams())
    idx = None
    if P.use_mspbwt:
        from .mspbwt import panel_mspbwt_index
        idx = panel_mspbwt_index(panel, P.mspbwt_nindices)
    rcq = keep_rc = None
    if P.impute_rare_common:
        if len(drcs) != len(devs):
            raise ValueError("impute_rare_common: one DeviceRareCommon per DevicePanel")
        rcq, keep_rc = make_rare_common(drcs[0].rc, [d.handle for d in drcs], samples)
    n, T = len(samples), (drcs[0].rc.nSNPs_all if P.impute_rare_common else panel.nSNPs)
    nq = fd = fg = keep_n = None
    if P.method == "nipt":
        nq, fd, fg, keep_n = make_nipt(panel, samples, P.shuffle_bin_radius, T)
    q, keep = make_params(P, samples_per_launch_set, idx, fuse_tails, rcq, nq)
    read_off, read_ptr, u, bq, wif = flatten_samples(samples)
    nL = 3 if P.method == "nipt" else 2
    return PreparedRange(devs=list(devs), samples=list(samples), q=q, keep=(keep, keep_rc, keep_n), sample_offset=int(sample_offset),
                         read_off=read_off, read_ptr=read_ptr, u=u, bq=bq, wif=wif, n=n, T=T, nL=nL, fd=fd, fg=fg,
                         handles=(C.c_void_p * len(devs))(*[d.handle for d in devs]))


def run_prepared(r: PreparedRange, return_stats: bool = False, one_by_one: bool = False):
    """``one_by_one``: the samples' reads are handed to the call through a qa_sample_source_t (include/quilt_amd.h: each sample
    when the launch set holding it is taken) instead of the flat arrays -- same results."""
    n, T = r.n, r.T
    dosage, gp_t, haps = np.empty((n, T)), np.empty((n, 3, T)), np.empty((n, r.nL, T))
    labels = np.empty(int(r.read_off[-1]), dtype=np.int32)
    nDosage = np.zeros(n, dtype=np.int32)
    stats = np.zeros(11, dtype=np.int64)
    L = lib()
    L.qa_impute_samples.restype = C.c_int
    flat = (ptr(r.read_off), ptr(r.read_ptr), ptr(r.u), ptr(r.bq), ptr(r.wif))
    keep_s = None
    if one_by_one:
        views = [labels[r.read_off[i]:r.read_off[i + 1]] for i in range(n)]   # (contiguous slices of the flat label array)
        src, keep_s = sample_source_over(r.samples, views)
        r.q.sample_source = C.cast(C.pointer(src), C.c_void_p)
        flat = (None,) * 5
    try:
        check(L.qa_impute_samples(r.handles, C.c_int32(len(r.devs)), C.byref(r.q), C.c_int32(n), C.c_int64(r.sample_offset), *flat,
                                  ptr(dosage), ptr(gp_t), ptr(haps), None if one_by_one else ptr(labels), ptr(nDosage), ptr(stats)))
    finally:
        r.q.sample_source = None
        del keep_s
    out = wrap_results(r.samples, dosage, gp_t, haps, labels, nDosage, r.read_off, r.fd, r.fg)
    return (out, dict(zip(STAT_NAMES, stats.tolist()))) if return_stats else out


def impute_samples(devs: Sequence, samples: Sequence, params: Optional[DriverParams] = None, sample_offset: int = 0,
                   samples_per_launch_set: int = 256, fuse_tails: bool = True, return_stats: bool = False, drcs: Sequence = (),
                   one_by_one: bool = False):
    """``devs``: one :class:`quilt_amd.native.DevicePanel` per host thread (replicas of one panel on one device; with more
    than one, switch ``set_exclusive`` on).  ``drcs`` (with ``params.impute_rare_common``): one
    :class:`quilt_amd.native.DeviceRareCommon` per entry of ``devs``; every sample then carries its all-SNP reads as
    ``sample.all_snp`` and the results cover all SNPs.  Returns one SampleResult per sample (and the native counters)."""
    return run_prepared(prepare_range(devs, samples, params, sample_offset, samples_per_launch_set, fuse_tails, drcs), return_stats, one_by_one)


# ---------------------------------------------------------------------------------------------------------------------------
# a sample range from BAM paths to VCF columns in one native call (qa_impute_bam_range, include/quilt_amd_io.h)
# ---------------------------------------------------------------------------------------------------------------------------
class BamRangeIo(C.Structure):
    from .io import BamOpts as _BamOpts
    _fields_ = [("chr", C.c_char_p), ("nSNPs", C.c_int32), ("L", C.c_void_p), ("ref", C.c_char_p), ("alt", C.c_char_p),
                ("grid", C.c_void_p), ("nSNPs_all", C.c_int32), ("L_all", C.c_void_p), ("ref_all", C.c_char_p),
                ("alt_all", C.c_char_p), ("grid_all", C.c_void_p), ("bam", _BamOpts), ("minimum_number_of_sample_reads", C.c_int32),
                ("output_gt_phased_genotypes", C.c_int32), ("n_io_threads", C.c_int32), ("discard_sample_arrays", C.c_int32)]


def impute_bam_range(devs: Sequence, bam_files: Sequence[str], chr: str, ref, alt, params: Optional[DriverParams] = None,
                     sample_index: Optional[Sequence[int]] = None, ff: Optional[Sequence[float]] = None, *,
                     minimum_number_of_sample_reads: int = 2, output_gt_phased_genotypes: bool = True, n_io_threads: int = 0,
                     samples_per_launch_set: int = 256, fuse_tails: bool = True, drcs: Sequence = (), all_sites=None,
                     bqFilter: int = 17, iSizeUpperLimit: float = 1e6, useSoftClippedBases: bool = False, downsampleToCov: int = 30,
                     chrStart: int = 0, chrEnd: int = 0, merge_mates: bool = True, seed: int = 1, copy_out: Optional[Sequence[int]] = None,
                     discard_sample_arrays: bool = False, _entry=None) -> dict:
    """The body of QUILT()'s loop over a core's sample range (quilt.R:832-982) as ONE native call: the BAM files are loaded on
    host threads, the samples with enough reads imputed together on the device, their VCF columns formatted on host threads and
    the four per-SNP count arrays summed over the range.  ``sample_index``: the files' global 0-based sample indices (default
    0 .. n - 1).  ``all_sites`` (with ``params.impute_rare_common`` and ``drcs``): ``(L_all, ref_all, alt_all, grid_all)``.
    ``copy_out``: the files whose columns and arrays are copied out of the library into numpy objects (default: all; a caller
    that only wants the counts or a few samples saves the copies -- 2.5 MB of text and 4 MB of numbers per sample).
    ``discard_sample_arrays``: the library gives a sample's dosage / gp_t / phasing_haps rows back to the system once its column is
    formatted (what the R fast path asks for); ``results`` then carry the read labels and ``nDosage`` only.
    Returns dict(imputed, n_reads, columns [VcfColumn or None], results {file index: SampleResult}, counts SummaryCounts,
    seconds {load, impute, format, total}, stats)."""
    from .io import BamOpts, SummaryCounts, VcfColumn
    panel = devs[0].panel
    P = params or DriverParams()
    if panel.L is None:
        raise ValueError("the panel carries no SNP positions (Panel.L)")
    n = len(bam_files)
    T = panel.nSNPs
    Lc = np.ascontiguousarray(panel.L, dtype=np.int32)
    grid = np.ascontiguousarray(panel.grid if panel.grid is not None else np.arange(T, dtype=np.int32) // 32, dtype=np.int32)
    as_bytes = lambda a: bytes(a) if isinstance(a, (bytes, bytearray)) else "".join(a).encode()
    refb, altb = as_bytes(ref), as_bytes(alt)
    if len(refb) != T or len(altb) != T:
        raise ValueError("ref / alt: one character per SNP of the panel")
    idx = None
    if P.use_mspbwt:
        from .mspbwt import panel_mspbwt_index
        idx = panel_mspbwt_index(panel, P.mspbwt_nindices)
    rcq = nq = None
    keep = []
    T_out = T
    io = BamRangeIo()
    if P.impute_rare_common:
        if len(drcs) != len(devs) or all_sites is None:
            raise ValueError("impute_rare_common: one DeviceRareCommon per DevicePanel, and the all-SNP sites")
        rc = drcs[0].rc
        T_out = rc.nSNPs_all
        hs = (C.c_void_p * len(drcs))(*[d.handle for d in drcs])
        is_common = np.ascontiguousarray(rc.snp_is_common, dtype=np.uint8)
        Lga = np.ascontiguousarray(rc.L_grid_all, dtype=np.int32)
        rcq = ImputeRareCommon(C.cast(hs, C.c_void_p), rc.nSNPs_all, rc.nGrids_all, ptr(is_common), None, None, None, None, None, ptr(Lga))
        La, refa, alta, grida = all_sites
        La = np.ascontiguousarray(La, dtype=np.int32)
        grida = np.ascontiguousarray(grida, dtype=np.int32)
        refa, alta = as_bytes(refa), as_bytes(alta)
        io.nSNPs_all, io.L_all, io.ref_all, io.alt_all, io.grid_all = T_out, La.ctypes.data, refa, alta, grida.ctypes.data
        keep += [hs, is_common, Lga, La, grida, refa, alta]
    if P.method == "nipt":
        if ff is None or len(ff) != n:
            raise ValueError("method = 'nipt': one fetal fraction per file")
        Lg = np.ascontiguousarray(panel.L_grid, dtype=np.int32)
        nq = ImputeNipt(None, ptr(Lg), int(P.shuffle_bin_radius), None, None)
        keep.append(Lg)
    q, keep_q = make_params(P, samples_per_launch_set, idx, fuse_tails, rcq, nq)
    io.chr, io.nSNPs, io.L, io.ref, io.alt, io.grid = chr.encode(), T, Lc.ctypes.data, refb, altb, grid.ctypes.data
    io.bam = BamOpts(int(bqFilter), int(min(iSizeUpperLimit, 2**31 - 1)), int(bool(useSoftClippedBases)), int(downsampleToCov),
                     int(chrStart), int(chrEnd), int(bool(merge_mates)), int(seed))
    io.minimum_number_of_sample_reads = int(minimum_number_of_sample_reads)
    io.output_gt_phased_genotypes = int(bool(output_gt_phased_genotypes))
    io.n_io_threads = int(n_io_threads)
    io.discard_sample_arrays = int(bool(discard_sample_arrays))
    paths = (C.c_char_p * max(n, 1))(*[p.encode() for p in bam_files])
    sidx = np.ascontiguousarray(np.arange(n) if sample_index is None else sample_index, dtype=np.int64)
    ffv = None if ff is None else np.ascontiguousarray(ff, dtype=np.float64)
    handles = (C.c_void_p * len(devs))(*[getattr(d, "handle", None) for d in devs])
    L = lib()
    for name in ("qa_impute_bam_range", "qa_bam_range_column", "qa_bam_range_sample", "qa_bam_range_counts", "qa_bam_range_imputed",
                 "qa_bam_range_n_reads", "qa_bam_range_n_snps", "qa_bam_range_n_samples"):
        getattr(L, name).restype = C.c_int
    L.qa_bam_range_destroy.restype = None
    L.qa_bam_range_timings.restype = None
    h = C.c_void_p()
    if _entry is not None:   # (tests: the same native host code with its imputation step on a checker -- impute_testhook.h)
        _entry(q, io, n, paths, sidx, ffv, h)
    else:
        check(L.qa_impute_bam_range(handles, C.c_int32(len(devs)), C.byref(q), C.byref(io), C.c_int32(n), paths, ptr(sidx), ptr(ffv),
                                    C.byref(h)))
    try:
        assert L.qa_bam_range_n_snps(h) == T_out and L.qa_bam_range_n_samples(h) == n
        nL = 3 if P.method == "nipt" else 2
        imputed = [bool(L.qa_bam_range_imputed(h, C.c_int32(i))) for i in range(n)]
        n_reads = [int(L.qa_bam_range_n_reads(h, C.c_int32(i))) for i in range(n)]
        columns, results = [None] * n, {}
        wanted = set(range(n)) if copy_out is None else set(int(i) for i in copy_out)
        for i in range(n):
            if not imputed[i] or i not in wanted:
                continue
            buf, off = C.c_void_p(), C.c_void_p()
            check(L.qa_bam_range_column(h, C.c_int32(i), C.byref(buf), C.byref(off)))
            o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_int64)), shape=(T_out + 1,)).copy()
            b = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(int(o[-1]),)).copy()
            columns[i] = VcfColumn(b, o)
            pd, pg, ph, pfd, pfg, pl = (C.c_void_p() for _ in range(6))
            nl, nd = C.c_int32(), C.c_int32()
            check(L.qa_bam_range_sample(h, C.c_int32(i), C.byref(pd), C.byref(pg), C.byref(ph), C.byref(pfd), C.byref(pfg), C.byref(pl),
                                        C.byref(nl), C.byref(nd)))
            arr = lambda p, shape, t=C.c_double: (np.ctypeslib.as_array(C.cast(p, C.POINTER(t)), shape=shape).copy() if p.value else None)
            hp = arr(ph, (nL, T_out))
            results[i] = SampleResult(arr(pd, (T_out,)), arr(pg, (3, T_out)), None if hp is None else hp.T,
                                      arr(pl, (nl.value,), C.c_int32) if nl.value else np.zeros(0, dtype=np.int32), int(nd.value),
                                      fet_dosage=arr(pfd, (T_out,)) if pfd.value else None,
                                      fet_gp_t=arr(pfg, (3, T_out)) if pfg.value else None)
        info, af, hwe, ac = np.zeros((T_out, 2), order="F"), np.zeros(T_out), np.zeros((T_out, 3), order="F"), np.zeros((T_out, 2), order="F")
        check(L.qa_bam_range_counts(h, ptr(info), ptr(af), ptr(hwe), ptr(ac)))
        counts = SummaryCounts(T_out, hweCount=np.ascontiguousarray(hwe), infoCount=np.ascontiguousarray(info), afCount=af,
                               alleleCount=np.ascontiguousarray(ac))
        sec, st, ls = np.zeros(4), np.zeros(11, dtype=np.int64), np.zeros(8, dtype=np.int64)
        L.qa_bam_range_timings(h, ptr(sec), ptr(st), ptr(ls))
    finally:
        L.qa_bam_range_destroy(h)
    del keep, keep_q
    return dict(imputed=imputed, n_reads=n_reads, columns=columns, results=results, counts=counts,
                seconds=dict(zip(("load", "impute", "format", "total"), sec.tolist())), stats=dict(zip(STAT_NAMES, st.tolist())),
                load_stats=ls.tolist())
