"""Generates tests/golden/callentries.json: names and argument lists of the `.Call` entries the hot path needs, read from
the reference's registration file (QUILT/src/RcppExports.cpp: RcppExport prototypes and CallEntries[], :1703-1777).
Run in the build container (needs /root/reference); the JSON (data: names and counts) is what the tests use."""
import json
import os
import re

SRC = "/root/reference/QUILT/src/RcppExports.cpp"
WANT = ["_QUILT_rcpp_forwardBackwardGibbsNIPT", "_QUILT_Rcpp_haploid_dosage_versus_refs", "_QUILT_Rcpp_make_gl_bound",
        "_QUILT_rcpp_make_eMatRead_t"]

text = open(SRC).read()
out = {}
for name in WANT:
    m = re.search(r"RcppExport SEXP " + name + r"\(([^)]*)\)", text)
    args = [a.strip().split()[-1] for a in m.group(1).split(",")]
    reg = re.search(r'\{"' + name + r'", \(DL_FUNC\) &' + name + r", (\d+)\}", text)
    out[name] = dict(n_registered=int(reg.group(1)), args=[a[:-4] if a.endswith("SEXP") else a for a in args])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "callentries.json"), "w"), indent=1)
print({k: v["n_registered"] for k, v in out.items()})
