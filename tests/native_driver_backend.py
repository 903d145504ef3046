"""Test infrastructure: a ``qa_impute_backend_t`` (quilt_amd/csrc/impute_testhook.h, a private header) whose entries are the CPU oracle, so that the native
driver loop of csrc/impute.cpp -- the product's host code -- can be run WITHOUT a device through ``qa_impute_samples_backend``
and compared with quilt_amd/driver.py on the oracle backend and with tests/r_driver_twin.py.  Each callback unflattens the
C arrays into what tests/oracle_backend.py::OracleBackend takes; the selection behind the full-panel call is the host text
of quilt_amd/driver.py (which tests/test_select_gpu.py holds equal to csrc/select.hip)."""
import ctypes as C

import numpy as np

from quilt_amd.driver import (ListsTruncated, everything_select_good_haps_dense, previously_selected)
from quilt_amd.gibbs_nipt import GibbsOpts
from quilt_amd.impute import ImputeParams, STAT_NAMES, flatten_samples, make_nipt, make_params, make_rare_common, wrap_results
from quilt_amd.native import lib, ptr
from tests.oracle_backend import OracleBackend

I32P, F64P, U64P, F32P = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_float)

GIBBS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(GibbsOpts), C.c_int32, I32P, I32P, I32P, I32P, I32P, I32P, F64P, I32P, F64P,
                       I32P, I32P, F64P, F64P, F64P, I32P, F64P, U64P, U64P)
SELECT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, I32P, I32P, I32P, I32P, I32P, I32P, I32P, I32P, I32P,
                        C.c_int32, C.c_double, F64P, C.c_int32, I32P, F32P, I32P, C.c_int32, C.c_int32, I32P, U64P, I32P, I32P)
FULLPASS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, F64P, I32P, I32P, C.c_int32, F64P, I32P, I32P, F64P, C.c_int64)
EMAT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, F64P, I32P, I32P, I32P, I32P, C.c_double, C.c_int32,
                      C.c_int32, F64P)
GIBBS_RC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(GibbsOpts), C.c_int32, I32P, I32P, I32P, I32P, I32P, I32P, F64P, I32P,
                          F64P, I32P, I32P, F64P, F64P, F64P, I32P, F64P, U64P, U64P)
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t)
FREE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class Backend(C.Structure):
    _fields_ = [("gibbs_batch", GIBBS_FN), ("fullpass_reads_select_batch", SELECT_FN), ("fullpass_batch", FULLPASS_FN),
                ("make_eMatRead_t_hap_major", EMAT_FN), ("mspbwt_select_new_haps", C.c_void_p), ("accumulate_dosage", C.c_void_p),
                ("consensus_read_labels", C.c_void_p), ("host_alloc", ALLOC_FN), ("host_free", FREE_FN), ("bind_thread", C.c_void_p),
                ("gibbs_batch_rare_common", GIBBS_RC_FN), ("make_eMatRead_t_nsnps", EMAT_FN),
                ("make_eMatRead_t_rare_common", C.c_void_p)]   # optional: left NULL, the loop takes its expansion path


def _arr(p, n, dtype):
    return np.ctypeslib.as_array(p, shape=(int(n),)) if n else np.zeros(0, dtype=dtype)


class _Reads:   # the SampleReads surface the oracle needs
    def __init__(self, read_ptr, u, bq, wif=None):
        self.read_ptr, self.u, self.bq, self.wif = read_ptr, u, bq, wif
        self.nReads = len(read_ptr) - 1


class OracleTable:
    """Owns the callbacks (ctypes keeps no reference to them) and the oracle they call."""

    def __init__(self, panel, fail_at_call=None, rare_common=None):
        self.panel = panel
        self.rare_common = rare_common
        self.ob = OracleBackend(panel, rare_common)
        self.calls = {"gibbs": 0, "select": 0, "fullpass": 0, "emat": 0}
        self.fail_at_call = fail_at_call   # ("gibbs", n): the n-th call of that entry reports a hard error (error-path tests)
        self.error = None
        L = lib()
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        libc.free.argtypes = [C.c_void_p]
        self._malloc, self._free = libc.malloc, libc.free
        self.calls.update(gibbs_rc=0, emat_all=0)
        self._cbs = (GIBBS_FN(self._guard(self.gibbs, "gibbs")), SELECT_FN(self._guard(self.select, "select")),
                     FULLPASS_FN(self._guard(self.fullpass, "fullpass")), EMAT_FN(self._guard(self.emat, "emat")),
                     ALLOC_FN(lambda n: self._malloc(max(int(n), 1))), FREE_FN(lambda p: (self._free(p), 0)[1]),
                     GIBBS_RC_FN(self._guard(self.gibbs_rc, "gibbs_rc")), EMAT_FN(self._guard(self.emat_all, "emat_all")))
        addr = lambda f: C.cast(f, C.c_void_p)
        self.table = Backend(self._cbs[0], self._cbs[1], self._cbs[2], self._cbs[3], addr(L.qa_mspbwt_select_new_haps),
                             addr(L.qa_accumulate_dosage), addr(L.qa_consensus_read_labels), self._cbs[4], self._cbs[5], None,
                             self._cbs[6], self._cbs[7])

    def _guard(self, f, name):
        def g(*a):
            try:
                self.calls[name] += 1
                if self.fail_at_call == (name, self.calls[name]):
                    return -4
                return f(*a)
            except BaseException as e:   # an exception must not cross the C frames
                self.error = e
                return -2
        return g

    # ---- qa_gibbs_batch_rare_common: the same unflattening, the oracle's all-SNP call
    def gibbs_rc(self, handle, rc, o, n, which, read_off, read_ptr, u, bq, wif, ru, fr, rs, H, Hc, hp, gm, gf, uf, state, sr, ss):
        return self.gibbs(handle, o, n, which, read_off, read_ptr, u, bq, wif, ru, fr, rs, H, Hc, hp, gm, gf, uf, state, sr, ss, rare=True)

    def emat_all(self, handle, nSNPs, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxdiff, Jmax, rescale, out):
        """qa_rcpp_make_eMatRead_t_nsnps: eHaps [chain][SNP][K] (R's K x nSNPs matrices, column-major)."""
        ro = _arr(read_off, n_chain + 1, np.int32)
        rp = _arr(read_ptr, int(ro[n_chain]) + n_chain, np.int32)
        e = np.ctypeslib.as_array(eHaps, shape=(n_chain, nSNPs, K))
        o = np.ctypeslib.as_array(out, shape=(int(ro[n_chain]), K))
        base = 0
        samples = []
        for c in range(n_chain):
            R = int(ro[c + 1] - ro[c])
            p = rp[ro[c] + c: ro[c] + c + R + 1]
            nb = int(p[-1])
            samples.append(_Reads(p.copy(), _arr(C.cast(C.addressof(u.contents) + 4 * base, I32P), nb, np.int32).copy(),
                                  _arr(C.cast(C.addressof(bq.contents) + 4 * base, I32P), nb, np.int32).copy()))
            base += nb
        assert rescale == 1 and Jmax == 100
        lik = self.ob.read_likelihood_all_snps_batch(samples, e, maxdiff)
        for c, m in enumerate(lik):
            o[ro[c]:ro[c + 1]] = np.asarray(m).T
        return 0

    # ---- qa_gibbs_batch on the oracle
    def gibbs(self, handle, o, n, which, read_off, read_ptr, u, bq, wif, ru, fr, rs, H, Hc, hp, gm, gf, uf, state, sr, ss, rare=False):
        o = o.contents
        ro = _arr(read_off, n + 1, np.int32)
        totR = int(ro[n])
        rp = _arr(read_ptr, totR + n, np.int32)
        wh = _arr(which, n * o.Ks, np.int32).reshape(n, o.Ks)
        Ha = _arr(H, totR, np.int32)
        wf = _arr(wif, totR, np.int32)
        blocks = ([int(x) for x in np.ctypeslib.as_array(C.cast(o.block_gibbs_iterations, I32P), shape=(o.n_block_gibbs_iterations,))]
                  if o.n_block_gibbs_iterations > 0 else [])   # (no block passes: the pointer may be null)
        base = 0
        samples, starts = [], []
        for c in range(n):
            R = int(ro[c + 1] - ro[c])
            p = rp[ro[c] + c: ro[c] + c + R + 1]
            nb = int(p[-1])
            samples.append(_Reads(p.copy(), _arr(C.cast(C.addressof(u.contents) + 4 * base, I32P), nb, np.int32).copy(),
                                  _arr(C.cast(C.addressof(bq.contents) + 4 * base, I32P), nb, np.int32).copy(),
                                  wf[ro[c]:ro[c + 1]].copy()))
            starts.append(Ha[ro[c]:ro[c + 1]].copy())
            base += nb
        nipt = {}
        if o.ff != 0:   # method = "nipt": one fetal fraction per chain, the block definition's radius (L_grid is the panel's)
            ffc = np.ctypeslib.as_array(C.cast(o.ff_chain, F64P), shape=(n,)) if o.ff_chain else np.full(n, o.ff)
            assert not o.do_shard_block_gibbs and not o.sample_is_diploid and o.L_grid
            Lg = self.rare_common.L_grid_all if rare else self.panel.L_grid   # (the all-SNP call's blocks are cut on its own grid)
            assert np.array_equal(np.ctypeslib.as_array(C.cast(o.L_grid, I32P), shape=(len(Lg),)), Lg)
            nipt = dict(ff=[float(x) for x in ffc], shuffle_bin_radius=int(o.shuffle_bin_radius))
        res = self.ob.gibbs_batch(samples, [wh[c].copy() for c in range(n)], starts, [int(sr[c]) for c in range(n)],
                                  [int(fr[c]) for c in range(n)], [int(ss[c]) for c in range(n)], **nipt,
                                  n_gibbs_burn_in_its=o.n_gibbs_burn_in_its, n_gibbs_sample_its=o.n_gibbs_sample_its,
                                  block_gibbs_iterations=blocks, gibbs_initialize_iteratively=bool(o.gibbs_initialize_iteratively),
                                  maxDifferenceBetweenReads=o.maxDifferenceBetweenReads, Jmax_local=o.Jmax, rare_common=rare)
        G, T = self.panel.nGrids, (self.rare_common.nSNPs_all if rare else self.panel.nSNPs)
        if rare:
            assert o.disable_read_category_usage == 1 and not o.gibbs_initialize_iteratively
        any_uf = False
        for c, r in enumerate(res):
            uf[c] = int(bool(r["underflow_problem"]))
            any_uf |= bool(r["underflow_problem"])
            Ha[ro[c]:ro[c + 1]] = r["H"]
            if o.hap_words_out:
                from quilt_amd.mspbwt import int_contract_rows
                w = np.ctypeslib.as_array(C.cast(o.hap_words_out, I32P), shape=(n, 3, G))
                w[c] = int_contract_rows(np.asarray(r["hapProbs_t"])[:3])
            if o.hap_major_out:
                h = np.ctypeslib.as_array(C.cast(o.hap_major_out, F64P), shape=(n, o.hap_major_labels, T))
                h[c] = np.asarray(r["hapProbs_t"])[:o.hap_major_labels]
        return 1 if any_uf else 0

    # ---- qa_fullpass_reads_select_batch: the oracle's passes, then quilt_amd/driver.py's host selection
    def select(self, handle, n_chain, n_label, n_sample, cs, read_off, read_ptr, u, bq, H, wd, wt, cols, Ktop, minGL, dosage, top_width,
               top_idx, top_val, top_cnt, Ksubset, Knew, which, seed, which_next, status):
        T, G, K = self.panel.nSNPs, self.panel.nGrids, self.panel.K
        ro = _arr(read_off, n_sample + 1, np.int32)
        rp = _arr(read_ptr, int(ro[n_sample]) + n_sample, np.int32)
        base = 0
        samples = []
        for s in range(n_sample):
            R = int(ro[s + 1] - ro[s])
            p = rp[ro[s] + s: ro[s] + s + R + 1]
            nb = int(p[-1])
            samples.append(_Reads(p.copy(), _arr(C.cast(C.addressof(u.contents) + 4 * base, I32P), nb, np.int32).copy(),
                                  _arr(C.cast(C.addressof(bq.contents) + 4 * base, I32P), nb, np.int32).copy()))
            base += nb
        csa = _arr(cs, n_chain, np.int32)
        labels, at = [], 0
        allH = None
        for c in range(n_chain):
            R = samples[csa[c]].nReads
            if allH is None:
                allH = np.ctypeslib.as_array(H, shape=(sum(samples[csa[k]].nReads for k in range(n_chain)),))
            labels.append(allH[at:at + R].copy())
            at += R
        wda, wta = _arr(wd, n_chain, np.int32), _arr(wt, n_chain, np.int32)
        colsa = _arr(cols, G, np.int32)
        n_thin = int((colsa >= 0).sum())
        dos, top, cnt = self.ob.fullpass_reads_batch(samples, list(csa), labels, list(wda), list(wta), colsa, Ktop, minGL, top_width,
                                                     n_label=n_label)
        if dosage and wda.any():
            d = np.ctypeslib.as_array(dosage, shape=(n_chain, n_label, T))
            for c in range(n_chain):
                if wda[c]:
                    d[c] = dos[c]
        np.ctypeslib.as_array(top_cnt, shape=(n_chain, n_label, n_thin))[...] = cnt
        wh = _arr(which, n_chain * Ksubset, np.int32).reshape(n_chain, Ksubset)
        nx = np.ctypeslib.as_array(which_next, shape=(n_chain, Ksubset))
        st = np.ctypeslib.as_array(status, shape=(n_chain,))
        for c in range(n_chain):
            if not wta[c]:
                st[c] = -1
                continue
            prev = previously_selected(wh[c], Ksubset - Knew, int(seed[c]))
            try:
                # csrc/select.hip reports status 1 whenever the ranks up to K_top_matches do not yield Knew new haplotypes
                sel = everything_select_good_haps_dense(Knew, Ktop, top[c].astype(np.int64) + 1, prev, K, int(seed[c]), truncated=True)
            except ListsTruncated:
                st[c] = 1
                continue
            nx[c] = np.concatenate([prev, sel])
            st[c] = 0
        return 0

    def fullpass(self, handle, n_pass, gl, wd, cols, Ktop, dosage, bptr, bidx, bval, cap):
        T, G = self.panel.nSNPs, self.panel.nGrids
        g = np.ctypeslib.as_array(gl, shape=(n_pass, T, 2))
        colsa = _arr(cols, G, np.int32)
        n_thin = int((colsa >= 0).sum())
        wda = _arr(wd, n_pass, np.int32)
        _, best = self.ob.fullpass_batch([np.asfortranarray(g[p].T) for p in range(n_pass)], list(wda), colsa, Ktop)
        bp = np.ctypeslib.as_array(bptr, shape=(n_pass * n_thin + 1,))
        bp[0] = 0
        k = 0
        for p in range(n_pass):
            for j in range(n_thin):
                bp[k + 1] = bp[k] + len(best[p][j]["top_matches"])
                k += 1
        if bp[-1] > cap:
            return -5
        bi = np.ctypeslib.as_array(bidx, shape=(int(cap),))
        bv = np.ctypeslib.as_array(bval, shape=(int(cap),))
        k = 0
        for p in range(n_pass):
            for j in range(n_thin):
                bi[bp[k]:bp[k + 1]] = best[p][j]["top_matches"]
                bv[bp[k]:bp[k + 1]] = best[p][j]["top_matches_values"]
                k += 1
        return 0

    def emat(self, handle, nSNPs, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxdiff, Jmax, rescale, out):
        from oracle import oracle as O
        ro = _arr(read_off, n_chain + 1, np.int32)
        rp = _arr(read_ptr, int(ro[n_chain]) + n_chain, np.int32)
        e = np.ctypeslib.as_array(eHaps, shape=(n_chain, K, nSNPs))
        o = np.ctypeslib.as_array(out, shape=(int(ro[n_chain]), K))
        base = 0
        for c in range(n_chain):
            R = int(ro[c + 1] - ro[c])
            p = rp[ro[c] + c: ro[c] + c + R + 1]
            nb = int(p[-1])
            s = _Reads(p.copy(), _arr(C.cast(C.addressof(u.contents) + 4 * base, I32P), nb, np.int32).copy(),
                       _arr(C.cast(C.addressof(bq.contents) + 4 * base, I32P), nb, np.int32).copy())
            base += nb
            m = O.calculate_eMatRead_t_vs_haplotypes(s, [e[c, k].copy() for k in range(K)], maxdiff, rescale_eMatRead_t=bool(rescale),
                                                     Jmax=Jmax)
            o[ro[c]:ro[c + 1]] = np.asarray(m).T
        return 0


def impute_samples_on_oracle(panel, samples, params, sample_offset=0, samples_per_launch_set=256, n_threads=1, fuse_tails=True,
                             fail_at_call=None, rare_common=None, source=None):
    """qa_impute_samples_backend over the oracle table: (results, native counters, the table).

    ``source``: None = the reads in the call's flat arrays; a dict(n_upper=..., fail_at=..., order_log=...) = the same samples handed
    over one by one through a qa_sample_source_t (params->sample_source), the call's n_sample being the upper bound ``n_upper``
    (default: the samples) and the range ending where the samples do."""
    P = params
    idx = None
    if P.use_mspbwt:
        from quilt_amd.mspbwt import panel_mspbwt_index
        idx = panel_mspbwt_index(panel, P.mspbwt_nindices)
    rcq = keep_rc = None
    if P.impute_rare_common:   # (the checker needs no native all-SNP handle: any non-null value per thread)
        rcq, keep_rc = make_rare_common(rare_common, [C.c_void_p(100 + w) for w in range(n_threads)], samples)
    nq = fd = fg = keep_n = None
    if P.method == "nipt":
        nq, fd, fg, keep_n = make_nipt(panel, samples, P.shuffle_bin_radius, rare_common.nSNPs_all if P.impute_rare_common else None)
    q, keep = make_params(P, samples_per_launch_set, idx, fuse_tails, rcq, nq)
    tab = OracleTable(panel, fail_at_call=fail_at_call, rare_common=rare_common)
    read_off, read_ptr, u, bq, wif = flatten_samples(samples)
    n, T = len(samples), (rare_common.nSNPs_all if P.impute_rare_common else panel.nSNPs)
    keep_s = None
    if source is not None:
        from quilt_amd.impute import sample_source_over
        n_given = n
        n = int(source.get("n_upper", n))   # (the call's n_sample: an upper bound; the output arrays are sized by it)
        per_sample_labels = [np.zeros(s.nReads, dtype=np.int32) for s in samples]
        src, keep_s = sample_source_over(samples, per_sample_labels, n_available=n_given, fail_at=source.get("fail_at"),
                                         order_log=source.get("order_log"))
        q.sample_source = C.cast(C.pointer(src), C.c_void_p)
        if rcq is not None:   # (the flat all-SNP arrays are not read with a source)
            rcq.read_off = rcq.read_ptr = rcq.u = rcq.bq = rcq.wif = None
        if nq is not None and n != n_given:
            ffu = np.full(n, np.nan)
            ffu[:n_given] = [float(s.ff) for s in samples]
            fd, fg = np.zeros((n, T)), np.zeros((n, 3, T))
            nq.ff, nq.fet_dosage, nq.fet_gp_t = ptr(ffu), ptr(fd), ptr(fg)
            keep_s = (keep_s, ffu)
    dosage, gp_t, haps = np.zeros((n, T)), np.zeros((n, 3, T)), np.zeros((n, 3 if P.method == "nipt" else 2, T))
    labels = np.zeros(int(read_off[-1]), dtype=np.int32)
    nDosage = np.zeros(n, dtype=np.int32)
    stats = np.zeros(11, dtype=np.int64)
    handles = (C.c_void_p * n_threads)(*[C.c_void_p(w + 1) for w in range(n_threads)])
    L = lib()
    L.qa_impute_samples_backend.restype = C.c_int
    L.qa_last_error.restype = C.c_char_p
    st = L.qa_impute_samples_backend(C.byref(tab.table), handles, C.c_int32(n_threads), C.c_int32(panel.K), C.c_int32(panel.nGrids),
                                     C.c_int32(panel.nSNPs), C.byref(q), C.c_int32(n), C.c_int64(sample_offset),
                                     *((None,) * 5 if source is not None else (ptr(read_off), ptr(read_ptr), ptr(u), ptr(bq), ptr(wif))),
                                     ptr(dosage), ptr(gp_t), ptr(haps), None if source is not None else ptr(labels), ptr(nDosage), ptr(stats))
    if source is not None:
        labels = np.concatenate(per_sample_labels) if per_sample_labels else labels
    del keep, keep_rc, keep_n, keep_s
    if tab.error is not None:
        raise tab.error
    if st != 0:
        raise RuntimeError(f"qa_impute_samples_backend: status {st}: {L.qa_last_error().decode()}")
    return wrap_results(samples, dosage, gp_t, haps, labels, nDosage, read_off, fd, fg), dict(zip(STAT_NAMES, stats.tolist())), tab


def impute_bam_range_on_oracle(panel, bam_files, chr, ref, alt, params, n_threads=1, rare_common=None, **kw):
    """qa_impute_bam_range_backend (csrc/bamrange.cpp through the private test hook): the product's loader, kept-sample
    bookkeeping, column formatting and count arrays, with the imputation step on the oracle table."""
    from quilt_amd.impute import impute_bam_range

    class _Dev:   # (what impute_bam_range reads of a DevicePanel: the panel; there is no native handle on this path)
        def __init__(self, p):
            self.panel, self.handle = p, None

    tab = OracleTable(panel, rare_common=rare_common)
    handles = (C.c_void_p * n_threads)(*[C.c_void_p(w + 1) for w in range(n_threads)])
    if rare_common is not None:   # (the checker needs no native all-SNP handle: any non-null value per thread)
        class _Drc:
            def __init__(self, w):
                self.rc, self.handle = rare_common, C.c_void_p(100 + w)
        kw = dict(kw, drcs=[_Drc(w) for w in range(n_threads)])
    L = lib()
    L.qa_impute_bam_range_backend.restype = C.c_int
    L.qa_last_error.restype = C.c_char_p

    def entry(q, io, n, paths, sidx, ffv, h):
        st = L.qa_impute_bam_range_backend(C.byref(tab.table), handles, C.c_int32(n_threads), C.c_int32(panel.K), C.c_int32(panel.nGrids),
                                           C.byref(q), C.byref(io), C.c_int32(n), paths, ptr(sidx), ptr(ffv), C.byref(h))
        if tab.error is not None:
            raise tab.error
        if st != 0:
            raise RuntimeError(f"qa_impute_bam_range_backend: status {st}: {L.qa_last_error().decode()}")

    return impute_bam_range([_Dev(panel)] * n_threads, bam_files, chr, ref, alt, params, _entry=entry, **kw)
