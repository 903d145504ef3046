"""TEST INFRASTRUCTURE: Python side of tests/c/mini_r.c -- R-shaped objects for shim/quilt_amd_shim.c executed without R.

``R`` wraps the shared object built from the shim and the test runtime (tests/c/Makefile: libshim_mini_r.so): vectors and
matrices from numpy arrays (column-major, R's shapes), named lists from dicts, ``dotcall(name, *args)`` = R's
``.Call(name, ...)`` with its arity check, results back as numpy arrays / dicts / lists."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, RAWSXP = 10, 13, 14, 16, 19, 24


class RError(RuntimeError):
    pass


class R:
    def __init__(self):
        out = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c"), "libshim_mini_r.so"], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError(out.stdout + out.stderr)
        L = C.CDLL(os.path.join(ROOT, "tests", "c", "libshim_mini_r.so"))
        self.L = L
        P = C.c_void_p
        for name, res, args in [("Rf_allocVector", P, [C.c_uint, C.c_ssize_t]), ("mini_r_data", P, [P]), ("mini_r_nil", P, []),
                                ("mini_r_set_dim", None, [P, C.c_int, C.c_int]), ("mini_r_set_dim3", None, [P, C.c_int, C.c_int, C.c_int]),
                                ("mini_r_set_names", None, [P, C.c_int, C.POINTER(C.c_char_p)]), ("SET_VECTOR_ELT", P, [P, C.c_ssize_t, P]),
                                ("VECTOR_ELT", P, [P, C.c_ssize_t]), ("STRING_ELT", P, [P, C.c_ssize_t]), ("CHAR", C.c_char_p, [P]),
                                ("TYPEOF", C.c_int, [P]), ("Rf_xlength", C.c_ssize_t, [P]), ("Rf_nrows", C.c_int, [P]),
                                ("Rf_ncols", C.c_int, [P]), ("Rf_getAttrib", P, [P, P]), ("Rf_mkString", P, [C.c_char_p]), ("Rf_mkChar", P, [C.c_char_p]),
                                ("SET_STRING_ELT", None, [P, C.c_ssize_t, P]),
                                ("mini_r_dotcall", P, [C.c_char_p, C.c_int, C.POINTER(P)]), ("mini_r_last_error", C.c_char_p, []),
                                ("mini_r_arity", C.c_int, [C.c_char_p]), ("mini_r_load_unif", None, [C.POINTER(C.c_double), C.c_size_t]),
                                ("mini_r_unif_drawn", C.c_size_t, []), ("mini_r_rng_violations", C.c_int, []),
                                ("mini_r_gc_violations", C.c_int, []), ("mini_r_gc_report", C.c_char_p, []),
                                ("mini_r_protect_imbalance", C.c_int, []), ("mini_r_gc_selftest", C.c_int, []),
                                ("mini_r_init", None, []), ("mini_r_reset", None, [])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.mini_r_init()
        self.nil = C.c_void_p(L.mini_r_nil())
        self.names_symbol = C.c_void_p.in_dll(L, "R_NamesSymbol")
        self.dim_symbol = C.c_void_p.in_dll(L, "R_DimSymbol")

    # ---- objects in
    def _vec(self, sxp, a, dtype):
        a = np.asarray(a)
        dims = a.shape
        flat = np.ascontiguousarray(a.astype(dtype, copy=False).ravel(order="F"))
        v = C.c_void_p(self.L.Rf_allocVector(sxp, flat.size))
        if flat.size:
            C.memmove(self.L.mini_r_data(v), flat.ctypes.data, flat.nbytes)
        if len(dims) == 2:
            self.L.mini_r_set_dim(v, dims[0], dims[1])
        elif len(dims) == 3:
            self.L.mini_r_set_dim3(v, dims[0], dims[1], dims[2])
        return v

    def integer(self, a):
        return self._vec(INTSXP, a, np.int32)

    def real(self, a):
        return self._vec(REALSXP, a, np.float64)

    def raw(self, a):
        return self._vec(RAWSXP, a, np.uint8)

    def logical(self, a):
        return self._vec(LGLSXP, a, np.int32)

    def string(self, s: str):
        return C.c_void_p(self.L.Rf_mkString(s.encode()))

    def strings(self, items):
        """A character vector."""
        v = C.c_void_p(self.L.Rf_allocVector(STRSXP, len(items)))
        for i, x in enumerate(items):
            self.L.SET_STRING_ELT(v, i, C.c_void_p(self.L.Rf_mkChar(x.encode())))
        return v

    def list(self, items, names=None):
        v = C.c_void_p(self.L.Rf_allocVector(VECSXP, len(items)))
        for i, x in enumerate(items):
            self.L.SET_VECTOR_ELT(v, i, x)
        if names is not None:
            arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
            self.L.mini_r_set_names(v, len(names), arr)
        return v

    def named(self, d: dict):
        return self.list(list(d.values()), list(d.keys()))

    # ---- objects out
    def value(self, x):
        x = C.c_void_p(x) if not isinstance(x, C.c_void_p) else x
        if x.value == self.nil.value:
            return None
        t, n = self.L.TYPEOF(x), self.L.Rf_xlength(x)
        if t == VECSXP:
            items = [self.value(self.L.VECTOR_ELT(x, i)) for i in range(n)]
            nm = C.c_void_p(self.L.Rf_getAttrib(x, self.names_symbol))
            if nm.value != self.nil.value:
                return {self.L.CHAR(self.L.STRING_ELT(nm, i)).decode(): items[i] for i in range(n)}
            return items
        if t == STRSXP:
            return [self.L.CHAR(self.L.STRING_ELT(x, i)).decode() for i in range(n)]
        dt = {LGLSXP: np.int32, INTSXP: np.int32, REALSXP: np.float64, RAWSXP: np.uint8}.get(t)
        if dt is None:
            raise TypeError(f"mini_r: type {t}")
        a = np.ctypeslib.as_array(C.cast(self.L.mini_r_data(x), C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(max(n, 1),))[:n].copy()
        dim = C.c_void_p(self.L.Rf_getAttrib(x, self.dim_symbol))
        if dim.value != self.nil.value:
            a = a.reshape((self.L.Rf_nrows(x), self.L.Rf_ncols(x)), order="F")
        return a

    # ---- .Call
    def dotcall(self, name: str, *args):
        arr = (C.c_void_p * max(len(args), 1))(*args)
        out = self.L.mini_r_dotcall(name.encode(), len(args), arr)
        # the emulated gctorture (tests/c/mini_r.c): an object used after a collection would have freed it, or a PROTECT stack left
        # unbalanced, fails the call that did it -- whatever the test was looking at
        if self.L.mini_r_gc_violations() or self.L.mini_r_protect_imbalance():
            raise AssertionError("R memory protocol: " + self.L.mini_r_gc_report().decode())
        if not out:
            raise RError(self.L.mini_r_last_error().decode())
        return self.value(out)

    def arity(self, name: str) -> int:
        return self.L.mini_r_arity(name.encode())

    def load_unif(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        self.L.mini_r_load_unif(u.ctypes.data_as(C.POINTER(C.c_double)), u.size)

    def reset(self):
        self.L.mini_r_reset()

    # ---- the reference's objects
    def sample_reads(self, s):
        """``sampleReads`` (copied-from-stitch.cpp:153-160): per read list(J, wif, bq, u), everything 0-based as STITCH keeps it."""
        out = []
        ptr = np.asarray(s.read_ptr)
        for r in range(s.nReads):
            a, b = int(ptr[r]), int(ptr[r + 1])
            out.append(self.list([self.integer([b - a - 1]), self.integer([int(s.wif[r])]), self.integer(np.asarray(s.bq[a:b]).reshape(-1, 1)),
                                  self.integer(np.asarray(s.u[a:b]).reshape(-1, 1))]))
        return self.list(out)

    def panel_objects(self, panel, **more):
        d = dict(hapMatcherR=self.raw(panel.hapMatcherR), distinctHapsB=self.integer(panel.distinctHapsB),
                 distinctHapsIE=self.real(panel.distinctHapsIE), eMatDH_special_matrix_helper=self.integer(panel.eMatDH_special_matrix_helper),
                 eMatDH_special_matrix=self.integer(panel.eMatDH_special_matrix),
                 rhb_t=self.integer(panel.rhb_t if panel.rhb_t is not None else np.zeros((1, 1), dtype=np.int32)),
                 transMatRate_t=self.real(panel.transMatRate_t), ref_error=self.real([panel.ref_error]))
        d.update(more)
        return self.named(d)
