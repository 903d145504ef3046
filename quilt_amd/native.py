"""Loader of ``libquilt_amd.so`` (the C-ABI HIP library) -- there is no fallback.

If the shared library is missing or no gfx950 device is visible, every compute entry point
raises: the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libquilt_amd.so")
if os.environ.get("QUILT_AMD_LIB") and os.path.realpath(os.environ["QUILT_AMD_LIB"]) != os.path.realpath(LIB_PATH):
    # another build of the library (developer A/B builds under build/, scripts/decompose_gibbs.sh): only on request -- some of
    # those builds switch parts of the computation off, and a stray environment variable must not select one silently
    if os.environ.get("QA_DEV") != "1":
        raise ImportError(f"QUILT_AMD_LIB={os.environ['QUILT_AMD_LIB']} is not this tree's library ({LIB_PATH}); set QA_DEV=1 to "
                          "load a developer build")
    LIB_PATH = os.environ["QUILT_AMD_LIB"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "quilt_amd.h")

QA_OK = 0
QA_UNDERFLOW = 1
QA_ERR_NO_DEVICE = -1
QA_ERR_INVALID = -2
QA_ERR_UNSUPPORTED = -3
QA_ERR_HIP = -4
QA_ERR_CAPACITY = -5


class QuiltAmdError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libquilt_amd status {status}: {message}")
        self.status = status


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


preload_note = ""   # what _preload_hip_runtime decided, for diagnostics


def _soname_in(path: str, stem: bytes):
    """The versioned soname ``<stem>.so.N`` a shared object carries (its own SONAME or a DT_NEEDED entry): the first match in
    the file's string table."""
    import re
    try:
        with open(path, "rb") as f:
            m = re.search(re.escape(stem) + rb"\.so\.\d+", f.read())
        return m.group(0) if m else None
    except OSError:
        return None


def _preload_hip_runtime():
    """One HIP runtime per process.  libquilt_amd.so needs libamdhip64; PyTorch-ROCm ships its own copy and loads it when
    torch is imported.  Whichever copy is mapped first serves both -- but if THIS library came first with the system's copy,
    torch's later import ended up with a second runtime that saw no device (the load-order trap of round 2).  So when a
    torch installation exists, has not been imported yet AND its runtime has the soname this library was linked against, its
    copy is mapped here first: this library binds to it, and a later ``import torch`` finds it already there.  A torch built
    against another ROCm major (another soname) is left alone -- two different runtimes cannot serve one process, and this
    library must run on the one it was built with; callers that also need that torch import it first themselves.
    ``QUILT_AMD_TORCH_HIP=0`` switches the preload off, ``=1`` forces it.  Hosts without torch (R) are not affected."""
    import importlib.util
    import sys
    global preload_note
    want = os.environ.get("QUILT_AMD_TORCH_HIP", "")
    if want == "0":
        preload_note = "off (QUILT_AMD_TORCH_HIP=0)"
        return
    if "torch" in sys.modules:
        preload_note = "torch already imported"
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        preload_note = "no torch installation"
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    theirs = _soname_in(os.path.join(libdir, "libamdhip64.so"), b"libamdhip64")
    ours = _soname_in(LIB_PATH, b"libamdhip64")
    if want != "1" and (theirs is None or ours is None or theirs != ours):
        preload_note = f"skipped: torch ships {theirs}, this library needs {ours}"
        return
    for name in ("libhsa-runtime64.so", "libamd_comgr.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                preload_note = f"could not map {path}"
                return
    preload_note = f"torch's {theirs.decode() if theirs else 'runtime'} mapped first"


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QuiltAmdError(QA_ERR_NO_DEVICE, f"{LIB_PATH} is missing: run quilt_amd.native.build() "
                                "(or __graft_entry__.build()); there is no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.qa_last_error.restype = C.c_char_p
        for name in ("qa_abi_version", "qa_device_count", "qa_set_device", "qa_panel_create", "qa_gibbs_batch",
                 "qa_rcpp_make_eMatRead_t", "qa_profile_reset", "qa_profile_get", "qa_fullpass_reads_batch",
                     "qa_Rcpp_haploid_dosage_versus_refs", "qa_Rcpp_make_gl_bound", "qa_fullpass_batch",
                     "qa_last_fullpass_timing_ms", "qa_panel_set_ranking_precision", "qa_panel_set_device_share", "qa_profile_get_busy", "qa_panel_create_from_rhb",
                     "qa_panel_export_tables", "qa_rcpp_make_eMatRead_t_nsnps", "qa_rare_common_create", "qa_profile_count",
                     "qa_profile_get_work", "qa_panel_set_dosage_precision", "qa_panel_set_sum_order",
                     "qa_gibbs_batch_rare_common", "qa_nipt_block_table", "qa_panel_set_cu_partition"):
            getattr(L, name).restype = C.c_int
        L.qa_profile_name.restype = C.c_char_p
        L.qa_panel_destroy.restype = None
        L.qa_rare_common_destroy.restype = None
        _lib = L
    return _lib


def check(status: int) -> int:
    if status < 0:
        raise QuiltAmdError(status, lib().qa_last_error().decode())
    return status


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """An uninitialised array in a ``qa_host_alloc`` buffer (pinned, device-visible): transfers from / to it skip the
    library's staging copy.  The buffer is released when the array and every view of it are gone."""
    import weakref
    if os.environ.get("QUILT_AMD_PAGEABLE"):   # measurement hook (bench.py --pageable): what a caller that cannot use
        return np.empty(shape, dtype=dtype)    # qa_host_alloc gets -- R allocates its own vectors -- the staged path
    L = lib()
    L.qa_host_alloc.restype = C.c_void_p
    L.qa_host_alloc.argtypes = [C.c_size_t]
    L.qa_host_free.argtypes = [C.c_void_p]
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    p = L.qa_host_alloc(C.c_size_t(max(n, 1)))
    if not p:
        raise QuiltAmdError(QA_ERR_HIP, L.qa_last_error().decode())
    raw = (C.c_char * max(n, 1)).from_address(p)
    weakref.finalize(raw, L.qa_host_free, C.c_void_p(p))   # numpy keeps ``raw`` alive as the base of the array and its views
    return np.frombuffer(raw, dtype=dt, count=n // dt.itemsize).reshape(shape)


class PanelDesc(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("nGrids", C.c_int32), ("nSNPs", C.c_int32), ("nMaxDH", C.c_int32),
        ("hapMatcherR", C.c_void_p), ("hapMatcher", C.c_void_p), ("rhb_t", C.c_void_p),
        ("distinctHapsB", C.c_void_p), ("distinctHapsIE", C.c_void_p),
        ("eMatDH_special_grid_which", C.c_void_p), ("eMatDH_special_matrix_helper", C.c_void_p),
        ("eMatDH_special_matrix", C.c_void_p), ("eMatDH_special_matrix_nrow", C.c_int32),
        ("use_eMatDH_special_symbols", C.c_int32), ("transMatRate_t", C.c_void_p),
        ("ref_error", C.c_double),
    ]


class FullpassOpts(C.Structure):
    _fields_ = [
        ("K_top_matches", C.c_int32), ("return_betaHat_t", C.c_int32), ("return_dosage", C.c_int32),
        ("return_gamma_t", C.c_int32), ("return_gammaSmall_t", C.c_int32),
        ("get_best_haps_from_thinned_sites", C.c_int32), ("always_normalize", C.c_int32),
        ("normalize_emissions", C.c_int32), ("min_emission_prob_normalization_threshold", C.c_double),
        ("suppressOutput", C.c_int32),
    ]


class DevicePanel:
    """Device-resident copy of a :class:`quilt_amd.panel.Panel` (upload once per process)."""

    def __init__(self, panel, use_eMatDH_special_symbols=None):
        self.panel = panel
        if use_eMatDH_special_symbols is None:
            use_eMatDH_special_symbols = panel.rhb_t is None
        d = PanelDesc(
            panel.K, panel.nGrids, panel.nSNPs, panel.nMaxDH,
            ptr(panel.hapMatcherR), ptr(panel.hapMatcher), ptr(panel.rhb_t),
            ptr(panel.distinctHapsB), ptr(panel.distinctHapsIE),
            ptr(panel.eMatDH_special_grid_which), ptr(panel.eMatDH_special_matrix_helper),
            ptr(panel.eMatDH_special_matrix), int(panel.eMatDH_special_matrix.shape[0]),
            int(bool(use_eMatDH_special_symbols)), ptr(panel.transMatRate_t), float(panel.ref_error))
        h = C.c_void_p()
        check(lib().qa_panel_create(C.byref(d), C.byref(h)))
        self.handle = h

    @classmethod
    def from_rhb(cls, panel, nMaxDH=None, use_eMatDH_special_symbols=False):
        """Build the device panel straight from ``panel.rhb_t`` (K x nGrids int32): the per-grid dictionary compression
        (STITCH::make_rhb_t_equality) runs on the GPU.  ``panel`` supplies rhb_t, nSNPs, transMatRate_t, ref_error."""
        self = cls.__new__(cls)
        self.panel = panel
        rhb = np.asfortranarray(panel.rhb_t, dtype=np.int32)
        h = C.c_void_p()
        check(lib().qa_panel_create_from_rhb(ptr(rhb), C.c_int32(rhb.shape[0]), C.c_int32(rhb.shape[1]), C.c_int32(panel.nSNPs),
                                             C.c_int32(nMaxDH if nMaxDH is not None else panel.nMaxDH), ptr(panel.transMatRate_t),
                                             C.c_double(panel.ref_error), C.c_int32(int(bool(use_eMatDH_special_symbols))),
                                             C.byref(h)))
        self.handle = h
        self._nMaxDH = int(nMaxDH if nMaxDH is not None else panel.nMaxDH)
        return self

    def export_tables(self):
        """(hapMatcherR K x G uint8 F, distinctHapsB nMaxDH x G int32 F, special_off, special_k, special_word)."""
        K, G = self.panel.K, self.panel.nGrids
        nMaxDH = getattr(self, "_nMaxDH", self.panel.nMaxDH)
        hm = np.zeros((K, G), dtype=np.uint8, order="F")
        B = np.zeros((nMaxDH, G), dtype=np.int32, order="F")
        off = np.zeros(G + 1, dtype=np.int32)
        check(lib().qa_panel_export_tables(self.handle, ptr(hm), ptr(B), ptr(off), None, None, C.c_int64(0)))
        n = int(off[-1])
        sk = np.zeros(max(n, 1), dtype=np.int32)
        sw = np.zeros(max(n, 1), dtype=np.int32)
        check(lib().qa_panel_export_tables(self.handle, None, None, None, ptr(sk), ptr(sw), C.c_int64(max(n, 1))))
        return hm, B, off, sk[:n], sw[:n]

    def set_ranking_precision(self, bits: int):
        """64 (default): best-haplotype lists from fp64-state passes (the reference's arithmetic); 32: from the
        fp32-state pass (faster; near-ties may be ordered differently)."""
        check(lib().qa_panel_set_ranking_precision(self.handle, C.c_int32(bits)))

    def set_dosage_precision(self, bits: int):
        """32 (default): dosage / alpha / beta / gamma outputs from fp32 state (fp64 emissions and sums); 64: fp64 state
        throughout, as the reference (a verification mode, several times slower)."""
        check(lib().qa_panel_set_dosage_precision(self.handle, C.c_int32(bits)))

    def set_sum_order(self, reference_order):
        """VALIDATION MODE (``True`` / 1): the full-panel passes form every K-wide sum in the order the reference's code adds
        it, by one lane (fullpass_ref.hip): the explicit loops with the grid's special haplotypes first, then k = 0 .. K-1;
        grid 0's ``sum(alphaHat_t_col)`` as Armadillo's ``sum()`` adds it (two accumulators over the even / odd k).  20-50x
        slower; lists, c, alpha / beta and dosage then equal the CPU path's bit for bit.  2: the same with grid 0's sum left to
        right as well (the library's order before round 6; include/quilt_amd.h says how a maintainer with R decides between
        the two).  ``False`` / 0 (default): the production kernels."""
        check(lib().qa_panel_set_sum_order(self.handle, C.c_int32(int(reference_order))))

    def set_device_share(self, n_sharers: int):
        """This handle is one of ``n_sharers`` working on the device concurrently (one per host thread)."""
        check(lib().qa_panel_set_device_share(self.handle, C.c_int32(n_sharers)))

    def set_cu_partition(self, index: int, count: int):
        """Confine this handle's Gibbs launches to the index-th of ``count`` equal CU partitions (count = 1: no mask)."""
        check(lib().qa_panel_set_cu_partition(self.handle, C.c_int32(index), C.c_int32(count)))

    def set_exclusive(self, on: bool = True):
        """Exclusive device phases: this handle's launch sets run with the device to themselves, out of the device-wide arena,
        queueing in arrival order behind those of the other handles that opted in (include/quilt_amd.h)."""
        lib().qa_panel_set_exclusive.restype = C.c_int
        check(lib().qa_panel_set_exclusive(self.handle, C.c_int32(int(on))))

    def set_pass_priority(self, on: bool = True):
        """Full-panel calls of this handle on a highest-priority stream (several handles sharing the device)."""
        lib().qa_panel_set_pass_priority.restype = C.c_int
        check(lib().qa_panel_set_pass_priority(self.handle, C.c_int32(int(on))))

    def close(self):
        if self.handle:
            lib().qa_panel_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceRareCommon:
    """Device-resident all-SNP side of a QUILT2 panel (:class:`quilt_amd.panel.RareCommon`) for the final rare + common
    Gibbs call; belongs to one :class:`DevicePanel`."""

    def __init__(self, device_panel: DevicePanel, rc):
        self.rc = rc
        self.device_panel = device_panel
        is_common = np.ascontiguousarray(rc.snp_is_common, dtype=np.uint8)
        rare_ptr = np.ascontiguousarray(rc.rare_ptr, dtype=np.int64)
        rare_snp = np.ascontiguousarray(rc.rare_snp, dtype=np.int32)
        tm = np.asfortranarray(rc.transMatRate_t_all, dtype=np.float64)
        h = C.c_void_p()
        lib().qa_rare_common_create.restype = C.c_int
        check(lib().qa_rare_common_create(device_panel.handle, C.c_int32(rc.nSNPs_all), ptr(is_common), ptr(rare_ptr),
                                          ptr(rare_snp if rare_snp.size else np.zeros(1, dtype=np.int32)), ptr(tm),
                                          C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            lib().qa_rare_common_destroy.restype = None
            lib().qa_rare_common_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gate_stats(device: int = 0, reset: bool = False) -> dict:
    """Exclusive device phases (DevicePanel.set_exclusive): ms some handle held the device, ms callers queued, holds."""
    out = (C.c_double * 7)()
    lib().qa_gate_stats.restype = C.c_int
    check(lib().qa_gate_stats(C.c_int32(device), out))
    if reset:
        lib().qa_gate_stats_reset(C.c_int32(device))
    return dict(held_ms=out[0], queued_ms=out[1], holds=int(out[2]), exclusive_ms=out[3], gibbs_slot_ms=out[4],
                gibbs_holds=int(out[5]), gibbs_slots=int(out[6]))


def gate_trace(device: int = 0, on: bool = True, read: bool = True) -> np.ndarray:
    """qa_gate_trace: the finished holds since the last read as rows (request, admit, kernels done, release [ms], SIMD slots
    (0: exclusive), thread); ``read=False`` only switches the trace on / off."""
    lib().qa_gate_trace.restype = C.c_int
    if not read:
        check(min(lib().qa_gate_trace(C.c_int32(device), C.c_int32(int(on)), None, C.c_int32(0)), 0))
        return np.zeros((0, 6))
    n = lib().qa_gate_trace(C.c_int32(device), C.c_int32(1), None, C.c_int32(0))
    check(min(n, 0))
    rows = np.zeros((max(n, 1) + 64, 6))
    n = lib().qa_gate_trace(C.c_int32(device), C.c_int32(int(on)), ptr(rows), C.c_int32(len(rows)))
    return rows[:min(n, len(rows))]


def last_fullpass_timing_ms():
    out = (C.c_double * 5)()
    check(lib().qa_last_fullpass_timing_ms(out))
    return dict(emat=out[0], forward=out[1], backward=out[2], dosage=out[3], total=out[4])
