"""The drop-in boundary as code (no GPU): the R shim type-checks against R's API declarations with the reference's
registered names and arities, and the plain-C harness links against the library and gets QA_ERR_NO_DEVICE."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_r_shim_type_checks():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "shim"), "check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


ENTRIES = (("_QUILT_rcpp_forwardBackwardGibbsNIPT", 63), ("_QUILT_Rcpp_haploid_dosage_versus_refs", 38),
           ("_QUILT_Rcpp_make_gl_bound", 3), ("_QUILT_rcpp_make_eMatRead_t", 15))


def test_r_shim_registers_the_reference_arities():
    """RcppExports.cpp:1703-1782 registers _QUILT_rcpp_forwardBackwardGibbsNIPT with 63 arguments,
    _QUILT_Rcpp_haploid_dosage_versus_refs with 38, _QUILT_Rcpp_make_gl_bound with 3, _QUILT_rcpp_make_eMatRead_t with 15; the
    shim's functions take exactly as many SEXPs, in the reference's order (no extra panel-handle argument), and its table
    registers them under the reference's names."""
    import json
    src = open(os.path.join(ROOT, "shim", "quilt_amd_shim.c")).read()
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "callentries.json")))   # from RcppExports.cpp (make_callentries.py)
    for name, n in ENTRIES:
        assert ref[name]["n_registered"] == n
        m = re.search(r"SEXP qa" + name + r"\(([^)]*)\)\s*\{", src, re.S)
        assert m, name
        args = [a[:-4] for a in re.findall(r"\bSEXP\s+(\w+)", m.group(1))]
        assert args == ref[name]["args"], name   # the same arguments in the same order
        assert re.search(r'\{"' + name + r'", \(DL_FUNC\)&qa' + name + r", " + str(n) + r"\}", src)


def test_r_shim_defines_no_symbol_of_rcppexports(tmp_path):
    """The shim is compiled INTO QUILT.so next to RcppExports.cpp, which defines `_QUILT_<fn>` for every export: the shim
    must not define any of them (a duplicate-symbol link error), only qa_QUILT_<fn>, qa_shim_* and its own init."""
    obj = str(tmp_path / "shim.o")
    out = subprocess.run(["gcc", "-std=c11", "-c", os.path.join(ROOT, "shim", "quilt_amd_shim.c"), "-o", obj],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    nm = subprocess.run(["nm", "--defined-only", "-g", obj], capture_output=True, text=True).stdout
    defined = [ln.split()[-1] for ln in nm.splitlines() if ln.strip()]
    assert defined, nm
    assert not [d for d in defined if d.startswith("_QUILT_")], defined
    allowed = {"qa" + n for n, _ in ENTRIES} | {"qa_shim_release", "R_init_quilt_amd_shim", "qa_impute_sample_range"}
    assert set(defined) == allowed, sorted(set(defined) ^ allowed)


def test_quilt_src_patch():
    """shim/QUILT-src.patch (what a maintainer applies to QUILT/src): its CallEntries rows carry the reference's names and
    arities (tests/golden/callentries.json) and point at functions the shim defines; with the reference beside the repository
    the committed patch is what shim/make_patch.py generates and it applies to the reference's files."""
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "callentries.json")))
    patch = open(os.path.join(ROOT, "shim", "QUILT-src.patch")).read()
    plus = re.findall(r'^\+\s*\{"(_QUILT_\w+)", \(DL_FUNC\) &(\w+), (\d+)\},', patch, re.M)
    minus = re.findall(r'^-\s*\{"(_QUILT_\w+)", \(DL_FUNC\) &(\w+), (\d+)\},', patch, re.M)
    assert sorted(p[0] for p in plus) == sorted(n for n, _ in ENTRIES) == sorted(m[0] for m in minus)
    for name, fn, n in plus:
        assert fn == "qa" + name and int(n) == ref[name]["n_registered"]
        decl = re.search(r'^\+extern "C" SEXP ' + fn + r"\(([^)]*)\);", patch, re.M)
        assert decl and decl.group(1).count("SEXP") == int(n)
    assert "-lquilt_amd" in patch and "-DQA_HAVE_R" in patch and "quilt_amd_shim.c" in patch
    if os.path.exists("/root/reference/QUILT/src/RcppExports.cpp"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_patch", os.path.join(ROOT, "shim", "make_patch.py"))
        mp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mp)
        assert mp.make("/root/reference") == patch, "shim/QUILT-src.patch is stale: python shim/make_patch.py"
        # after the patch no row of the table names a Rcpp wrapper the shim replaces, every other row is untouched
        text = mp.patched_rcppexports(open("/root/reference/QUILT/src/RcppExports.cpp").read())
        rows = re.findall(r'\{"(_QUILT_\w+)", \(DL_FUNC\) &(\w+), (\d+)\},', text)
        assert len(rows) > 60
        for name, fn, _ in rows:
            assert fn == ("qa" + name if name in dict(ENTRIES) else name)


def test_c_harness_builds_and_reports_no_device_here():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([os.path.join(ROOT, "tests", "c", "c_harness")], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "HARNESS_OK" in run.stdout or "NO_DEVICE" in run.stdout
    assert "HOST_FORMATS_OK" in run.stdout   # include/quilt_amd_io.h from plain C: needs no device


def test_device_gate_admission_rules():
    """Device phases (include/quilt_amd.h, qa_panel_set_exclusive): Gibbs launches that fit run together, the queue is first
    come first served with nobody overtaking a waiting launch set, express holds (the msPBWT search) go first."""
    import ctypes
    from quilt_amd import native
    lib = native.lib()
    lib.qa_gate_selftest.restype = ctypes.c_int
    lib.qa_last_error.restype = ctypes.c_char_p
    assert lib.qa_gate_selftest() == 0, lib.qa_last_error()
