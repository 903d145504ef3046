"""shim/quilt_amd_shim.c loaded under the test runtime of R's C API subset (tests/c/mini_r.c) on a machine without a GPU: the
table R_init_quilt_amd_shim registers (names and arities the reference's own, tests/golden/callentries.json), `.Call`'s arity
check, and that a compute routine's refusal -- no device here -- arrives as an R error with the library's text."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def R():
    from tests.mini_r import R as Runtime
    r = Runtime()
    yield r
    r.reset()


def test_registered_routines_have_the_references_arities(R):
    entries = json.load(open(os.path.join(ROOT, "tests", "golden", "callentries.json")))
    table = {name: e["n_registered"] for name, e in entries.items()}
    for name in ("_QUILT_rcpp_forwardBackwardGibbsNIPT", "_QUILT_Rcpp_haploid_dosage_versus_refs", "_QUILT_Rcpp_make_gl_bound",
                 "_QUILT_rcpp_make_eMatRead_t"):
        assert R.arity(name) == table[name], name
    assert R.arity("qa_impute_sample_range") == 6 and R.arity("qa_shim_release") == 0 and R.arity("_QUILT_not_there") == -1


def test_make_gl_bound_through_dotcall(R):
    """_QUILT_Rcpp_make_gl_bound(gl, minGLValue, to_fix): host arithmetic, so it runs here -- in place, as the reference's
    (reference-single.cpp:68-94): the columns listed in to_fix are scaled so that their larger entry is 1 and floored."""
    from tests.mini_r import RError
    gl = np.array([[1e-30, 0.2, 0.5], [1e-12, 0.4, 1e-40]])
    x = R.real(gl)
    assert R.dotcall("_QUILT_Rcpp_make_gl_bound", x, R.real([1e-10]), R.integer([0, 2])) is None
    got = R.value(x)
    assert np.array_equal(got[:, 1], gl[:, 1])
    for j in (0, 2):
        col = gl[:, j] / gl[:, j].max()
        assert np.allclose(got[:, j], np.maximum(col, 1e-10), rtol=0, atol=0)
    with pytest.raises(RError, match="Incorrect number of arguments"):
        R.dotcall("_QUILT_Rcpp_make_gl_bound", x, R.real([1e-10]))


def test_compute_routine_without_a_device_is_an_r_error(R):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: tests/test_shim_gpu.py runs the routine")
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.mini_r import RError
    panel = make_synthetic_panel(K=64, nSNPs=320, seed=2)
    s = make_synthetic_sample(panel, seed=3, n_reads=40)
    prm = R.named(dict(nGibbsSamples=R.integer([2])))
    with pytest.raises(RError, match="(?i)device"):
        R.dotcall("qa_impute_sample_range", R.list([R.sample_reads(s)]), R.panel_objects(panel), prm, R.real([0.0]), R.integer([1]), R.nil)


def test_the_runtime_catches_the_two_classic_r_memory_mistakes(R):
    """tests/c/mini_r.c emulates R's gctorture(TRUE): inside a `.Call` every allocation collects what the call made and left
    unreachable from the PROTECT stack, any later use of such an object is a violation, and the stack must be empty at return --
    so every shim test also exercises PROTECT discipline (tests/mini_r.py fails the call that breaks it).  The emulation itself:
    an object used across an allocation it was not protected for is caught (bit 0), a routine that returns with a PROTECT too many
    is caught (bit 1), the correct forms raise nothing (bit 2 stays clear)."""
    assert R.L.mini_r_gc_selftest() == 3
    assert R.L.mini_r_gc_violations() == 0 and R.L.mini_r_protect_imbalance() == 0
    assert R.arity("qa_impute_bam_range") == 6
