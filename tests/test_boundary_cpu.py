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
    allowed = {"qa" + n for n, _ in ENTRIES} | {"qa_shim_release", "R_init_quilt_amd_shim", "qa_impute_sample_range", "qa_impute_bam_range_call"}
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


# ---------------------------------------------------------------------------------------------------------------------------
# shim/QUILT-R.patch: the R side of the fast path (one .Call per sample range) as a patch against QUILT/R/quilt.R + a new file
# ---------------------------------------------------------------------------------------------------------------------------
def _r_strip(text):
    """R source with comments and string literals blanked out (enough for bracket balance and identifier scans)."""
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch == "#":
            while i < n and text[i] != "\n":
                i += 1
            continue
        if ch in "\"'":
            q = ch
            i += 1
            while i < n and text[i] != q:
                i += 2 if text[i] == "\\" else 1
            i += 1
            out.append('""')
            continue
        out.append(ch)
        i += 1
    return "".join(out)


def _balanced(text):
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for ch in _r_strip(text):
        if ch in "([{":
            stack.append(ch)
        elif ch in pairs:
            if not stack or stack.pop() != pairs[ch]:
                return False
    return not stack


def _formals(text, fn):
    m = re.search(re.escape(fn) + r" <- function\(", text)
    assert m, fn
    depth, i = 1, m.end()
    while depth:
        depth += {"(": 1, ")": -1}.get(text[i], 0)
        i += 1
    body = _r_strip(text[m.end():i - 1])
    return [a.split("=")[0].strip() for a in re.split(r",(?![^()]*\))", body) if a.strip()]


def test_r_patch_for_the_fast_path_is_current_and_consistent():
    """shim/QUILT-R.patch = what shim/make_patch.py generates from the reference's quilt.R; the range call in it passes exactly the
    arguments quilt_amd_impute_sample_range / quilt_amd_range_is_covered declare, every value it passes is a name QUILT() has in
    scope at that point, every reference function quilt-amd.R calls exists in the reference's R sources, and brackets balance
    (R itself is not in this image: the patch cannot be executed here)."""
    patch = open(os.path.join(ROOT, "shim", "QUILT-R.patch")).read()
    amd = open(os.path.join(ROOT, "shim", "quilt-amd.R")).read()
    assert _balanced(amd)
    added = "\n".join(l[1:] for l in patch.split("--- a/QUILT/R/quilt-amd.R")[0].splitlines() if l.startswith("+") and not l.startswith("+++"))
    assert _balanced(added.rsplit("\n", 1)[0])   # (the last added row replaces one that opens the same call: `... get_and_impute_one_sample(`)
    assert added.endswith("get_and_impute_one_sample(")
    assert "get_and_impute_one_sample(" in added and "amd_results[[iSample - sampleRange[1] + 1]]" in added   # the fallback stays
    for fn in ("quilt_amd_impute_sample_range", "quilt_amd_range_is_covered"):
        formals = _formals(amd, fn)
        call = re.search(re.escape(fn) + r"\((.*?)\n\s*\)\)? \{?\n|" + re.escape(fn) + r"\((.*?)\n\s*\)\n", added, re.S)
        assert call, fn
        passed = re.findall(r"(\w+) = ", (call.group(1) or call.group(2)))
        assert set(passed) <= set(formals), (fn, set(passed) - set(formals))
        required = [f for f in formals if not re.search(r"\b" + f + r" = ", amd[amd.index(fn + " <- function("):amd.index(") {", amd.index(fn + " <- function("))])]
        assert set(required) <= set(passed), (fn, set(required) - set(passed))
    if not os.path.exists("/root/reference/QUILT/R/quilt.R"):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_patch", os.path.join(ROOT, "shim", "make_patch.py"))
    mp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp)
    assert mp.make_R("/root/reference") == patch, "shim/QUILT-R.patch is stale: python shim/make_patch.py"
    quilt = open("/root/reference/QUILT/R/quilt.R").read()
    patched = mp.patched_quilt_R(quilt)
    assert _balanced(patched) == _balanced(quilt)
    # every value handed to the two functions is in scope in QUILT() before the loop: an argument of QUILT() or assigned above
    head = _r_strip(quilt[:quilt.index(mp.R_ANCHOR_LOOP)])
    # (the prepared reference's objects -- grid, L, pos, rhb_t, ... -- arrive through load(): they are in scope where the reference's
    # own call of get_and_impute_one_sample, right below, passes them under the same names)
    ref_call = _r_strip(quilt[quilt.index(mp.R_ANCHOR_CALL):quilt.index("if (out[[\"sample_was_imputed\"]])")])
    passed_by_reference = set(re.findall(r"\w+ = +([A-Za-z_][\w.]*)", ref_call))
    for call in re.findall(r"quilt_amd_\w+\((.*?)\n\s*\)\)? ?\{?\n", added, re.S):
        for name, value in re.findall(r"(\w+) = ([A-Za-z_][\w.]*)", call):
            if value in ("TRUE", "FALSE", "NULL"):
                continue
            assert value in passed_by_reference or re.search(r"\b" + re.escape(value) + r"\s*(=|<-)", head), \
                f"{value} (passed as {name}) is not defined in QUILT() before the loop"
    # the reference functions the new file relies on
    ref_R = "\n".join(open(os.path.join("/root/reference/QUILT/R", f)).read() for f in os.listdir("/root/reference/QUILT/R") if f.endswith(".R"))
    for fn in ("loadBamAndConvert", "file_sampleReads", "removeTmpSamplesFile", "snap_sampleReads_to_grid", "print_message",
               "increment2N", "get_max_gen_rapid", "STITCH::rcpp_make_column_of_vcf", "STITCH::convertScaledBQtoProbs"):
        assert fn + "(" in amd and fn + "(" in ref_R, fn
    # the names the shim's routine reads from its lists are the names quilt-amd.R fills
    shim = open(os.path.join(ROOT, "shim", "quilt_amd_shim.c")).read()
    for key in re.findall(r"(\w+) = ", _r_strip(amd[amd.index("panel_objects <- list("):amd.index("print_message(paste0(\"Imputing samples")])):
        if key in ("params", "panel_objects", "all_reads", "list", "c"):
            continue
        assert f'"{key}"' in shim, f"quilt-amd.R passes {key}, which shim/quilt_amd_shim.c never reads"
