"""The drop-in boundary as code (no GPU): the R shim type-checks against R's API declarations with the reference's
registered names and arities, and the plain-C harness links against the library and gets QA_ERR_NO_DEVICE."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_r_shim_type_checks():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "shim"), "check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


def test_r_shim_registers_the_reference_arities():
    """RcppExports.cpp:1703-1782 registers _QUILT_rcpp_forwardBackwardGibbsNIPT with 63 arguments,
    _QUILT_Rcpp_haploid_dosage_versus_refs with 38, _QUILT_Rcpp_make_gl_bound with 3; the shim's entry points take exactly
    as many SEXPs (no extra panel-handle argument) and say so in its CallEntries table."""
    import json
    src = open(os.path.join(ROOT, "shim", "quilt_amd_shim.c")).read()
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "callentries.json")))   # from RcppExports.cpp (make_callentries.py)
    for name, n in (("_QUILT_rcpp_forwardBackwardGibbsNIPT", 63), ("_QUILT_Rcpp_haploid_dosage_versus_refs", 38),
                    ("_QUILT_Rcpp_make_gl_bound", 3)):
        assert ref[name]["n_registered"] == n
        m = re.search(r"SEXP " + name + r"\(([^)]*)\)\s*\{", src, re.S)
        assert m, name
        args = [a[:-4] for a in re.findall(r"\bSEXP\s+(\w+)", m.group(1))]
        assert args == ref[name]["args"], name   # the same arguments in the same order
        assert re.search(r'\{"' + name + r'", \(DL_FUNC\)&' + name + r", " + str(n) + r"\}", src)


def test_c_harness_builds_and_reports_no_device_here():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([os.path.join(ROOT, "tests", "c", "c_harness")], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "HARNESS_OK" in run.stdout or "NO_DEVICE" in run.stdout
    assert "HOST_FORMATS_OK" in run.stdout   # include/quilt_amd_io.h from plain C: needs no device
