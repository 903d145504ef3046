"""The C-ABI library loads and exports every symbol include/quilt_amd.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from quilt_amd import native


def _declared_functions():
    text = open(native.HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qa_[a-zA-Z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(native.LIB_PATH):
        native.build()
    return native.lib()


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/quilt_amd.h but not exported"
    assert lib.qa_abi_version() == 1


def test_no_cpu_fallback_without_a_device(lib):
    if lib.qa_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    h = C.c_void_p()
    d = native.PanelDesc()
    assert lib.qa_panel_create(C.byref(d), C.byref(h)) == native.QA_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.qa_last_error()
    gl = np.ones((2, 4))
    assert lib.qa_Rcpp_haploid_dosage_versus_refs(None, native.ptr(gl), None, None, None, None, None, None, None,
                                                  None, None, None, None, C.c_int64(0)) == native.QA_ERR_NO_DEVICE
    assert lib.qa_gibbs_batch(None, None, 1, None, None, None, None, None, None, None, None, None, None, None, None,
                              None, None, None, None) == native.QA_ERR_NO_DEVICE
    with pytest.raises(native.QuiltAmdError):
        native.check(lib.qa_set_device(0))


def test_product_path_never_imports_the_oracle():
    """Nothing under quilt_amd/ may import, link or call the CPU oracle."""
    root = os.path.dirname(native.CSRC)
    pat = re.compile(r"(^\s*(import|from)\s+oracle\b|liboracle|\bqo_[a-z]|#include\s+\".*oracle)", re.M)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), f"{f} references the oracle"
