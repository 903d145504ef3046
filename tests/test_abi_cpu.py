"""The C-ABI library loads and exports every symbol include/*.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from quilt_amd import native


def _declared_functions():
    import glob
    names = set()
    for header in glob.glob(os.path.join(os.path.dirname(native.HEADER), "*.h")):   # quilt_amd.h, quilt_amd_io.h
        text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
        names |= set(re.findall(r"\b(qa_[a-zA-Z0-9_]+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(native.LIB_PATH):
        native.build()
    return native.lib()


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    assert lib.qa_abi_version() == 5


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


def test_nipt_block_table_host_logic_matches_oracle():
    """The library's host-side block definition (gibbs_blocks.hpp: smoothing, quantile, peak picking,
    make_gibbs_considers) against the oracle's restatement of gibbs-nipt-block.cpp:366-523 / :1307-1553 on random inputs
    -- no device needed."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    from quilt_amd import native
    lib = native.lib()
    rng = np.random.default_rng(11)
    for trial in range(40):
        G = int(rng.integers(12, 300))
        L_grid = np.cumsum(rng.integers(200, 3000, size=G)).astype(np.int32)
        rate2 = rng.random(G - 1) * rng.choice([0.005, 0.05, 0.5])
        for _ in range(int(rng.integers(0, 6))):
            at = int(rng.integers(0, G - 1))
            rate2[max(at - 2, 0):at + 3] = rng.random() * 1.5
        rate2[-1] = 0
        R = int(rng.integers(5, 400))
        wif = np.sort(rng.integers(0, G, size=R)).astype(np.int32)
        radius = int(rng.choice([500, 5000, 50000]))
        q = float(rng.choice([0.9, 0.95]))
        blocked_ref = O.define_blocked_grids(rate2, L_grid, radius, q)
        ref = O.make_gibbs_considers(blocked_ref, wif)
        arrs = [np.zeros(G, dtype=np.int32) for _ in range(6)]
        n = C.c_int32()
        native.check(lib.qa_nipt_block_table(native.ptr(rate2), native.ptr(L_grid), C.c_int32(G), C.c_int32(radius),
                                             C.c_double(q), native.ptr(wif), C.c_int32(R), *[native.ptr(a) for a in arrs],
                                             C.byref(n)))
        assert np.array_equal(arrs[0], blocked_ref)
        assert n.value == ref["n_blocks"]
        for a, name in zip(arrs[1:5], ("consider_grid_start_0_based", "consider_grid_end_0_based",
                                       "consider_reads_start_0_based", "consider_reads_end_0_based")):
            assert np.array_equal(a[:n.value], ref[name]), name
        assert np.array_equal(arrs[5], ref["consider_grid_where_0_based"])
