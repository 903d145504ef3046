"""ONE table of the modes of the per-sample loop, from which the tests that hold its two statements together are generated.

The loop exists twice -- quilt_amd/driver.py (Python: the tested statement, runs on any backend) and csrc/impute.cpp (C++: the
product, qa_impute_samples) -- and the reference has it a third time, in R (functions.R:3-1500).  A mode built into one of them
only would go unnoticed by tests written mode by mode; here every parameter of the loop is CLASSIFIED, the cases are the
product of the switches plus one row per branch a number selects, and tests/test_mode_matrix_cpu.py (oracle backend) and
tests/test_mode_matrix_gpu.py (device) run every case through both loops and require the same bytes.

A new field of DriverParams fails `test_every_parameter_of_the_loop_is_classified` until it is entered below.
"""
from dataclasses import fields

# switches: a value selects a code path of its own; the cases are their PRODUCT
SWITCHES = {
    "method": ("diploid", "nipt"),
    "use_mspbwt": (False, True),
    "impute_rare_common": (False, True),
}

# numbers that select a branch at some value: one case per (base mode, setting) beyond the defaults of BASE
BRANCHES = [
    ("complete_lists", dict(K_top_matches=1)),                       # functions.R:2276-2302 (no mspbwt only)
    ("one_seek_iteration", dict(n_seek_its=1)),                      # no re-selection at all
    ("no_burn_in", dict(n_seek_its=3, n_burn_in_seek_its=0)),        # every round accumulates
    ("panel_smaller_than_Ksubset", dict(Ksubset=600, Knew=600)),     # quilt.R:453-463 (K = 300 here)
    ("Knew_below_Ksubset", dict(Ksubset=64, Knew=40)),               # keeps some of the old small panel (no mspbwt only)
    ("two_gibbs_sample_its", dict(n_gibbs_sample_its=2)),
    ("no_block_passes", dict(small_ref_panel_block_gibbs_iterations=())),
    ("thin_half", dict(heuristic_match_thin=0.5)),
]
BRANCH_BASES = [dict(method="diploid"), dict(method="nipt"), dict(method="diploid", use_mspbwt=True)]
NOT_WITH_MSPBWT = {"complete_lists", "Knew_below_Ksubset", "thin_half"}   # (the full-panel pass they steer is not run in that mode)

# plain numbers: passed through, exercised at the non-default values of BASE
KNOBS = {"nGibbsSamples", "n_seek_its", "n_burn_in_seek_its", "Ksubset", "Knew", "K_top_matches", "heuristic_match_thin",
         "small_ref_panel_gibbs_iterations", "n_gibbs_sample_its", "small_ref_panel_block_gibbs_iterations",
         "maxDifferenceBetweenReads", "minGLValue", "Jmax", "seed", "shuffle_bin_radius", "mspbwtL", "mspbwtM", "mspbwt_nindices"}

# parameters of the Python loop only, and why the native loop has no counterpart
PYTHON_ONLY = {
    "diploid_block_gibbs": "names the one behaviour both loops have (the reference's no-op); no second value exists",
    "mspbwt_search": "'exhaustive' is this library's own device search, an alternative to the reference's query; qa_impute_samples "
                     "runs the reference's ('scan') and quilt_amd.impute.make_params refuses the other",
    "mspbwt_max_matches": "a knob of the 'exhaustive' search only",
}

BASE = dict(nGibbsSamples=2, n_seek_its=2, Ksubset=64, Knew=64, seed=9, small_ref_panel_gibbs_iterations=4,
            small_ref_panel_block_gibbs_iterations=(2,), mspbwt_nindices=2, shuffle_bin_radius=4000, Jmax=5000)


def classified():
    return set(SWITCHES) | KNOBS | set(PYTHON_ONLY)


def driver_fields():
    from quilt_amd.driver import DriverParams
    return {f.name for f in fields(DriverParams)}


def cases():
    """[(name, DriverParams keyword arguments)]: the product of the switches, then the branch rows."""
    out = []
    for method in SWITCHES["method"]:
        for mspbwt in SWITCHES["use_mspbwt"]:
            for rc in SWITCHES["impute_rare_common"]:
                name = method + ("+mspbwt" if mspbwt else "") + ("+rare_common" if rc else "")
                out.append((name, dict(BASE, method=method, use_mspbwt=mspbwt, impute_rare_common=rc)))
    for base in BRANCH_BASES:
        for bname, kw in BRANCHES:
            if base.get("use_mspbwt") and bname in NOT_WITH_MSPBWT:
                continue
            if base.get("use_mspbwt") and "Knew" in kw and kw["Knew"] != kw["Ksubset"]:
                continue
            tag = base["method"] + ("+mspbwt" if base.get("use_mspbwt") else "")
            out.append((f"{tag}:{bname}", dict(BASE, **base, **kw)))
    return out


def make_case(kw, n_samples=3, K=300, nSNPs=640, reads=150, seed0=50):
    """(panel, rare_common or None, samples, DriverParams) of a case, small enough for the oracle."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample, make_synthetic_sample_rare_common
    panel = make_synthetic_panel(K=K, nSNPs=nSNPs, seed=5)
    P = DriverParams(**kw)
    nipt = P.method == "nipt"
    rc = make_rare_common(panel, 3) if P.impute_rare_common else None
    samples = []
    for i in range(n_samples):
        ff = (0.1 + 0.06 * i) if nipt else 0.0
        if rc is not None:
            s = make_synthetic_sample_rare_common(panel, rc, seed0 + i, n_reads=reads)[0]
            if nipt:
                s.ff = ff
        else:
            s = make_synthetic_sample(panel, seed=seed0 + i, n_reads=reads, ff=ff)
        samples.append(s)
    return panel, rc, samples, P
