"""Writes shim/QUILT-src.patch: the change a QUILT maintainer applies to QUILT/src to route the four production `.Call`
entries of the hot path to libquilt_amd.  Run in a checkout that has the reference beside it:

    python shim/make_patch.py /root/reference        # (re)writes shim/QUILT-src.patch

What the patch does, and nothing else:
  * QUILT/src/RcppExports.cpp -- in `CallEntries[]` (:1703-1777) the rows of _QUILT_rcpp_make_eMatRead_t,
    _QUILT_Rcpp_make_gl_bound, _QUILT_rcpp_forwardBackwardGibbsNIPT and _QUILT_Rcpp_haploid_dosage_versus_refs name the shim's
    functions (qa_QUILT_<fn>, same arities) instead of the Rcpp wrappers; four `extern "C"` declarations are added above the
    table.  The Rcpp wrappers stay defined (no duplicate symbol: the shim's functions have other names) and unregistered.
    `Rcpp::compileAttributes()` regenerates this file: re-apply the patch afterwards.
  * the same table gets one NEW row, qa_impute_sample_range (6 arguments): the loop over a core's sample range as one call.
  * QUILT/src/Makevars -- the include path of include/quilt_amd.h, -DQA_HAVE_R (the shim then includes R's own headers) and
    the link line for libquilt_amd.so (QUILT_AMD = the root of this repository).
  * QUILT/src/quilt_amd_shim.c -- added by copying shim/quilt_amd_shim.c (R compiles every .c in src/); the patch carries a
    one-line stub that includes it from $(QUILT_AMD) so that the file is not duplicated.
The diff is written with zero lines of context: it holds the changed rows only, none of the reference's other text.
"""
import difflib
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ENTRIES = {"_QUILT_rcpp_make_eMatRead_t": 15, "_QUILT_Rcpp_make_gl_bound": 3, "_QUILT_rcpp_forwardBackwardGibbsNIPT": 63,
           "_QUILT_Rcpp_haploid_dosage_versus_refs": 38}


# routines the reference does not have: the loop over a core's sample range as one call (quilt.R:688-996 -> qa_impute_samples)
EXTRA = {"qa_impute_sample_range": 6}


def patched_rcppexports(text):
    lines = text.split("\n")
    out, seen = [], set()
    in_table = False
    for ln in lines:
        if ln.startswith("static const R_CallMethodDef CallEntries[]"):
            out.append("// libquilt_amd: the hot path's entry points (quilt_amd_shim.c), registered below under the reference's names")
            for name, n in ENTRIES.items():
                out.append('extern "C" SEXP qa%s(%s);' % (name, ", ".join(["SEXP"] * n)))
            for name, n in EXTRA.items():
                out.append('extern "C" SEXP %s(%s);' % (name, ", ".join(["SEXP"] * n)))
            in_table = True
        m = re.match(r'\s*\{"(_QUILT_\w+)", \(DL_FUNC\) &(_QUILT_\w+), (\d+)\},', ln)
        if m and m.group(1) in ENTRIES:
            assert m.group(1) == m.group(2) and int(m.group(3)) == ENTRIES[m.group(1)], ln
            ln = '    {"%s", (DL_FUNC) &qa%s, %d},' % (m.group(1), m.group(1), ENTRIES[m.group(1)])
            seen.add(m.group(1))
        if in_table and re.match(r"\s*\{NULL, NULL, 0\}", ln):   # new routines of the shim: rows of their own before the terminator
            for name, n in EXTRA.items():
                out.append('    {"%s", (DL_FUNC) &%s, %d},' % (name, name, n))
            in_table = False
        out.append(ln)
    assert seen == set(ENTRIES), "CallEntries rows not found: %s" % (set(ENTRIES) - seen)
    return "\n".join(out)


def patched_makevars(text):
    add = ["# libquilt_amd (MI355X): set QUILT_AMD to the root of the quilt_amd repository",
           "PKG_CPPFLAGS += -I$(QUILT_AMD)/include -DQA_HAVE_R -DQA_INSIDE_QUILT_SO",
           "PKG_LIBS += -L$(QUILT_AMD)/quilt_amd/csrc -lquilt_amd -Wl,-rpath,$(QUILT_AMD)/quilt_amd/csrc"]
    return text.rstrip("\n") + "\n" + "\n".join(add) + "\n"


def udiff(a, b, path):
    return "".join(difflib.unified_diff(a.splitlines(True), b.splitlines(True), "a/" + path, "b/" + path, n=0))


def make(ref_root):
    src = os.path.join(ref_root, "QUILT", "src")
    rc = open(os.path.join(src, "RcppExports.cpp")).read()
    mk = open(os.path.join(src, "Makevars")).read()
    stub = '/* the R side of the libquilt_amd boundary: compiled into QUILT.so */\n#include "../../../quilt_amd/shim/quilt_amd_shim.c"   /* adjust to $(QUILT_AMD)/shim/quilt_amd_shim.c, or copy the file here */\n'
    return (udiff(rc, patched_rcppexports(rc), "QUILT/src/RcppExports.cpp") + udiff(mk, patched_makevars(mk), "QUILT/src/Makevars") +
            udiff("", stub, "QUILT/src/quilt_amd_shim.c"))


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    open(os.path.join(HERE, "QUILT-src.patch"), "w").write(make(ref))
    print(open(os.path.join(HERE, "QUILT-src.patch")).read())
