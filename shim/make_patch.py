"""Writes shim/QUILT-src.patch: the change a QUILT maintainer applies to QUILT/src to route the four production `.Call`
entries of the hot path to libquilt_amd.  Run in a checkout that has the reference beside it:

    python shim/make_patch.py /root/reference        # (re)writes shim/QUILT-src.patch

What the patch does, and nothing else:
  * QUILT/src/RcppExports.cpp -- in `CallEntries[]` (:1703-1777) the rows of _QUILT_rcpp_make_eMatRead_t,
    _QUILT_Rcpp_make_gl_bound, _QUILT_rcpp_forwardBackwardGibbsNIPT and _QUILT_Rcpp_haploid_dosage_versus_refs name the shim's
    functions (qa_QUILT_<fn>, same arities) instead of the Rcpp wrappers; four `extern "C"` declarations are added above the
    table.  The Rcpp wrappers stay defined (no duplicate symbol: the shim's functions have other names) and unregistered.
    `Rcpp::compileAttributes()` regenerates this file: re-apply the patch afterwards.
  * the same table gets two NEW rows: qa_impute_sample_range (6 arguments: the loop over a core's sample range as one call, reads
    loaded by R) and qa_impute_bam_range (6 arguments: the same from BAM paths to VCF columns, I/O on native host threads).
  * QUILT/src/Makevars -- the include path of include/quilt_amd.h, -DQA_HAVE_R (the shim then includes R's own headers) and
    the link line for libquilt_amd.so (QUILT_AMD = the root of this repository).
  * QUILT/src/quilt_amd_shim.c -- added by copying shim/quilt_amd_shim.c (R compiles every .c in src/); the patch carries a
    one-line stub that includes it from $(QUILT_AMD) so that the file is not duplicated.
The diff is written with zero lines of context: it holds the changed rows only, none of the reference's other text.

It also writes shim/QUILT-R.patch: the R side of the FAST path.
  * QUILT/R/quilt.R -- inside the mclapply body (:692-990), in front of the loop over a core's samples (:832), the whole range
    goes through ONE `.Call("qa_impute_sample_range", ...)` (quilt_amd_impute_sample_range, new file below) whenever the run
    asks for nothing the range call does not cover (quilt_amd_range_is_covered: no plots, no HLA run, no phasefile / genfile
    truth, no per-read outputs, ...); the loop then takes each sample's result from that call instead of calling
    get_and_impute_one_sample (:835).  Otherwise -- and always with QUILT_AMD_RANGE=0 -- the unpatched loop runs, through the
    four per-call entries of QUILT-src.patch.
  * QUILT/R/quilt-amd.R -- new: shim/quilt-amd.R (loading by the reference's own loader, the call, the per-sample VCF column
    and counts by the reference's own functions).
"""
import difflib
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ENTRIES = {"_QUILT_rcpp_make_eMatRead_t": 15, "_QUILT_Rcpp_make_gl_bound": 3, "_QUILT_rcpp_forwardBackwardGibbsNIPT": 63,
           "_QUILT_Rcpp_haploid_dosage_versus_refs": 38}


# routines the reference does not have: the loop over a core's sample range as one call (quilt.R:688-996 -> qa_impute_samples)
EXTRA = {"qa_impute_sample_range": 6, "qa_impute_bam_range_call": 6}
# (name registered with R -> the C function, where they differ: the C ABI already has a qa_impute_bam_range)
EXTRA_REGISTERED_AS = {"qa_impute_bam_range_call": "qa_impute_bam_range"}


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
                out.append('    {"%s", (DL_FUNC) &%s, %d},' % (EXTRA_REGISTERED_AS.get(name, name), name, n))
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


R_ANCHOR_LOOP = "        for(iSample in sampleRange[1]:sampleRange[2]) {\n"
R_ANCHOR_CALL = "            out <- get_and_impute_one_sample(\n"
R_RANGE_CALL = '''        ## libquilt_amd: the whole sample range as ONE call (quilt-amd.R) when the run asks for nothing else
        amd_results <- NULL
        if (quilt_amd_range_is_covered(
            method = method, make_plots = make_plots, make_plots_block_gibbs = make_plots_block_gibbs, hla_run = hla_run,
            have_truth_haplotypes = have_truth_haplotypes, have_truth_genotypes = have_truth_genotypes,
            record_interim_dosages = record_interim_dosages, output_read_label_prob = output_read_label_prob,
            record_read_label_usage = record_read_label_usage, plot_per_sample_likelihoods = plot_per_sample_likelihoods,
            plot_p1 = plot_p1, make_heuristic_plot = make_heuristic_plot,
            estimate_bq_using_truth_read_labels = estimate_bq_using_truth_read_labels, addOptimalHapsToVCF = addOptimalHapsToVCF,
            use_splitreadgl = use_splitreadgl, small_ref_panel_skip_equally_likely_reads = small_ref_panel_skip_equally_likely_reads,
            shard_check_every_pair = shard_check_every_pair, use_hapMatcherR = use_hapMatcherR,
            calculate_gamma_on_the_fly = calculate_gamma_on_the_fly, RData_objects_to_save = RData_objects_to_save
        )) {
            amd_results <- quilt_amd_impute_sample_range(
                sampleRange = sampleRange, n_handles = 3L, device = iCore - 1L,
                rhb_t = rhb_t, hapMatcherR = hapMatcherR, distinctHapsB = distinctHapsB, distinctHapsIE = distinctHapsIE,
                eMatDH_special_matrix_helper = eMatDH_special_matrix_helper, eMatDH_special_matrix = eMatDH_special_matrix,
                use_eMatDH_special_symbols = use_eMatDH_special_symbols, small_transMatRate_tc_H = small_transMatRate_tc_H,
                ref_error = ref_error, L_grid = L_grid,
                method = method, nGibbsSamples = nGibbsSamples, n_seek_its = n_seek_its, n_burn_in_seek_its = n_burn_in_seek_its,
                Ksubset = Ksubset, Knew = Knew, K_top_matches = K_top_matches, heuristic_match_thin = heuristic_match_thin,
                small_ref_panel_gibbs_iterations = small_ref_panel_gibbs_iterations,
                small_ref_panel_block_gibbs_iterations = small_ref_panel_block_gibbs_iterations,
                maxDifferenceBetweenReads = maxDifferenceBetweenReads, minGLValue = minGLValue,
                shuffle_bin_radius = shuffle_bin_radius, seed = seed, ff_values = ff_values,
                use_mspbwt = use_mspbwt, mspbwtL = mspbwtL, mspbwtM = mspbwtM, mspbwt_nindices = mspbwt_nindices,
                impute_rare_common = impute_rare_common, special_rare_common_objects = special_rare_common_objects, pos_all = pos_all,
                L = L, pos = pos, grid = grid, bam_files = bam_files, cram_files = cram_files, reference = reference,
                iSizeUpperLimit = iSizeUpperLimit, bqFilter = bqFilter, useSoftClippedBases = useSoftClippedBases, chr = chr,
                sampleNames = sampleNames, downsampleToCov = downsampleToCov, tempdir = tempdir, regionName = regionName,
                chrStart = chrStart, chrEnd = chrEnd, use_bx_tag = use_bx_tag, bxTagUpperLimit = bxTagUpperLimit,
                minimum_number_of_sample_reads = minimum_number_of_sample_reads,
                output_gt_phased_genotypes = output_gt_phased_genotypes
            )
            ## native I/O (quilt-amd.R, form (a)): the range's four count arrays come back summed (in sample order) -- added here
            ## once; the per-sample entries the loop below adds are exact zeros then
            amd_counts <- attr(amd_results, "quilt_amd_counts")
            if (!is.null(amd_counts)) {
                infoCount <- infoCount + amd_counts[["infoCount"]]
                afCount <- afCount + amd_counts[["afCount"]]
                hweCount <- hweCount + amd_counts[["hweCount"]]
                alleleCount <- alleleCount + amd_counts[["alleleCount"]]
            }
        }

'''
R_NEW_CALL = ("            out <- if (!is.null(amd_results)) amd_results[[iSample - sampleRange[1] + 1]] else get_and_impute_one_sample(\n")


def patched_quilt_R(text):
    assert text.count(R_ANCHOR_LOOP) == 1 and text.count(R_ANCHOR_CALL) == 1, "QUILT/R/quilt.R: the loop over a core's samples was not found"
    assert text.index(R_ANCHOR_LOOP) < text.index(R_ANCHOR_CALL)
    text = text.replace(R_ANCHOR_LOOP, R_RANGE_CALL + R_ANCHOR_LOOP)
    return text.replace(R_ANCHOR_CALL, R_NEW_CALL)


def make_R(ref_root):
    q = open(os.path.join(ref_root, "QUILT", "R", "quilt.R")).read()
    new_file = open(os.path.join(HERE, "quilt-amd.R")).read()
    return udiff(q, patched_quilt_R(q), "QUILT/R/quilt.R") + udiff("", new_file, "QUILT/R/quilt-amd.R")


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    open(os.path.join(HERE, "QUILT-src.patch"), "w").write(make(ref))
    print(open(os.path.join(HERE, "QUILT-src.patch")).read())
    open(os.path.join(HERE, "QUILT-R.patch"), "w").write(make_R(ref))
    print("shim/QUILT-R.patch written (%d lines)" % len(open(os.path.join(HERE, "QUILT-R.patch")).read().splitlines()))
