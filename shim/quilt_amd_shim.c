/*
 * quilt_amd_shim.c -- the R side of the drop-in boundary: a shared object whose registered `.Call` routines have the
 * reference's own names and arities and forward to libquilt_amd.so (include/quilt_amd.h).
 *
 * What it replaces.  QUILT's R code reaches its native hot path through `.Call('_QUILT_<fn>', PACKAGE = 'QUILT', ...)`
 * (QUILT/R/RcppExports.R); the routines are registered in QUILT/src/RcppExports.cpp:1703-1782 (`CallEntries[]`,
 * `R_init_QUILT`).  This file registers, under the same symbol names and with the same number of arguments:
 *
 *   _QUILT_Rcpp_haploid_dosage_versus_refs   38 arguments   RcppExports.cpp:1580-1625   -> qa_Rcpp_haploid_dosage_versus_refs
 *   _QUILT_rcpp_forwardBackwardGibbsNIPT     63 arguments   RcppExports.cpp:966-1038    -> qa_gibbs_batch(_rare_common)
 *   _QUILT_Rcpp_make_gl_bound                 3 arguments   RcppExports.cpp:1263-1273   -> qa_Rcpp_make_gl_bound
 *   _QUILT_rcpp_make_eMatRead_t              15 arguments   RcppExports.cpp:15-38       -> qa_rcpp_make_eMatRead_t
 *
 * The C functions are called qa_QUILT_<fn> -- NOT _QUILT_<fn>: RcppExports.cpp defines those symbols itself, and this
 * file is compiled INTO QUILT.so (shim/QUILT-src.patch adds it to QUILT/src and points the four rows of RcppExports.cpp's
 * CallEntries[] at the functions here; R_init_QUILT then registers them under the reference's names, so every
 * `.Call('_QUILT_<fn>', PACKAGE = 'QUILT', ...)` of the unmodified R code lands here).  It can also be built as a DLL of its
 * own (R_init_quilt_amd_shim below) and swapped in with `assignInNamespace` (INTEGRATION.md shows both); no R function
 * changes its signature.  The prepared panel is uploaded on first use and cached in this file (keyed by the identity of the R
 * objects -- data pointer and dimensions of distinctHapsB / hapMatcherR -- plus ref_error and a checksum over transMatRate and
 * samples of both tables, so that a different panel that lands on a recycled address is not mistaken for the cached one), so
 * the 38- and 63-argument entries keep their arity; `qa_shim_release()` (0 arguments) drops the cache.  R owns every buffer it passes: results are copied back into
 * them before returning (SURVEY.md 8(b) "Ownership").
 *
 * Random numbers.  The reference draws inside the native call from R's generator (`Rcpp::runif`, `Rcpp::sample`:
 * gibbs-nipt.cpp:2845-2848, 3013-3018; gibbs-nipt-block.cpp:2054) under `Rcpp::RNGScope` (RcppExports.cpp:971).  The
 * shim draws the same uniforms, in the same order, with unif_rand() (`sample(nReads, 1)` = one unif_rand() scaled by nReads,
 * as Rcpp's sugar EmpiricalSample does) between GetRNGstate() / PutRNGstate() and passes them down; the device never generates R-incompatible numbers on this
 * path.  The shard pass draws nGrids - 1 uniforms per block iteration: what gibbs-nipt-block.cpp:2054 draws with
 * shard_check_every_pair = TRUE, the production value (quilt.R:178); FALSE is rejected.  Where the reference's draw count depends on
 * intermediate results (`Rcpp::sample(1:3, 1, prob)` per read whose class leaves a choice, in rcpp_sample_H_using_H_class, NIPT
 * only) the stream cannot be pre-drawn: since round 6 the library asks for those uniforms WHEN the reference draws them
 * (qa_gibbs_opts_t.draw_uniforms -> draw_uniforms_from_R below: runif_proposed / runif_block / runif_total before a block pass,
 * one unif_rand() per drawing read after its relabelling), so the NIPT entry consumes R's generator in the reference's order and
 * number as well (tests/test_shim_gpu.py counts them against the oracle run on the same stream).
 *
 * Type-checked without R by `make -C shim check` against shim/qa_r_api.h (declarations only), and EXECUTED without R by
 * tests/test_shim_gpu.py / tests/test_shim_cpu.py under tests/c/mini_r.c, a test runtime behind the same declarations (objects
 * with type, length, names and dim; `.Call` by registered name with R's arity check; Rf_error as an exception; unif_rand() from a
 * loaded sequence): every routine below is called with R-shaped arguments and compared with the Python mirror's call.
 */
#ifdef QA_HAVE_R
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Random.h>
#include <R_ext/Rdynload.h>
#else
#include "qa_r_api.h"
#endif

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../include/quilt_amd.h"
#include "../include/quilt_amd_io.h"   /* qa_impute_bam_range */

/* ---- small helpers ---------------------------------------------------------------------------------------------- */

static SEXP list_get(SEXP list, const char *name) {
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    for (R_xlen_t i = 0; i < Rf_xlength(list); i++)
        if (names != R_NilValue && strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
    return R_NilValue;
}
static int flag(SEXP param_list, const char *name, int dflt) {
    SEXP v = list_get(param_list, name);
    return v == R_NilValue ? dflt : Rf_asLogical(v);
}
static void check_status(int st, const char *what) {
    if (st < 0) Rf_error("%s: libquilt_amd status %d: %s", what, st, qa_last_error());
}
static SEXP named_list(int n, const char **names) {
    SEXP out = PROTECT(Rf_allocVector(VECSXP, n));
    SEXP nm = PROTECT(Rf_allocVector(STRSXP, n));
    for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

/* ---- the prepared panel, uploaded once per process ---------------------------------------------------------------- */

static struct {
    qa_panel_t *panel;
    qa_rare_common_t *rc;
    const void *key_B, *key_hm, *key_rare;
    int K, G, T, nMaxDH;
    double ref_error;
    uint64_t checksum;
} g_cache;

/* FNV-1a over the transition rates and evenly spaced samples of the two panel tables: cheap next to any call it guards */
static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static uint64_t panel_checksum(const void *hm, size_t hm_bytes, const int *B, size_t nB, const double *tm, size_t n_tm) {
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, tm, n_tm * sizeof(double));
    const size_t step_h = hm_bytes / 4096 + 1, step_B = nB / 4096 + 1;
    for (size_t i = 0; i < hm_bytes; i += step_h) h = fnv(h, (const unsigned char *)hm + i, 1);
    for (size_t i = 0; i < nB; i += step_B) h = fnv(h, B + i, sizeof(int));
    return h;
}

static void cache_drop(void) {
    if (g_cache.rc) qa_rare_common_destroy(g_cache.rc);
    if (g_cache.panel) qa_panel_destroy(g_cache.panel);
    memset(&g_cache, 0, sizeof g_cache);
}

/* transMatRate: 2 x (G - 1) doubles (column-major), the first slice of transMatRate_tc_H or transMatRate_t itself */
static qa_panel_t *panel_for(SEXP hapMatcher, SEXP hapMatcherR, int use_hapMatcherR, SEXP distinctHapsB, SEXP distinctHapsIE,
                             SEXP special_helper, SEXP special_matrix, SEXP special_grid_which, SEXP rhb_t, double ref_error,
                             int use_eMatDH_special_symbols, const double *transMatRate) {
    const int nMaxDH = Rf_nrows(distinctHapsB), G = Rf_ncols(distinctHapsB), T = Rf_ncols(distinctHapsIE);
    SEXP hm = use_hapMatcherR ? hapMatcherR : hapMatcher;
    const int K = Rf_nrows(hm);
    const void *kB = INTEGER(distinctHapsB), *kh = use_hapMatcherR ? (const void *)RAW(hapMatcherR) : (const void *)INTEGER(hapMatcher);
    const uint64_t sum = panel_checksum(kh, (size_t)K * G * (use_hapMatcherR ? 1 : sizeof(int)), INTEGER(distinctHapsB),
                                        (size_t)nMaxDH * G, transMatRate, 2 * (size_t)(G > 1 ? G - 1 : 0));
    if (g_cache.panel && g_cache.key_B == kB && g_cache.key_hm == kh && g_cache.K == K && g_cache.G == G && g_cache.T == T &&
        g_cache.nMaxDH == nMaxDH && g_cache.ref_error == ref_error && g_cache.checksum == sum)
        return g_cache.panel;
    cache_drop();
    qa_panel_desc_t d;
    memset(&d, 0, sizeof d);
    d.K = K; d.nGrids = G; d.nSNPs = T; d.nMaxDH = nMaxDH;
    d.hapMatcherR = use_hapMatcherR ? RAW(hapMatcherR) : NULL;
    d.hapMatcher = use_hapMatcherR ? NULL : INTEGER(hapMatcher);
    const int have_rhb = Rf_nrows(rhb_t) == K && Rf_ncols(rhb_t) == G;   /* 1 x 1 in msPBWT mode (quilt.R:551-563) */
    d.rhb_t = have_rhb ? INTEGER(rhb_t) : NULL;
    d.distinctHapsB = INTEGER(distinctHapsB);
    d.distinctHapsIE = REAL(distinctHapsIE);
    int *which = NULL;
    if (special_grid_which != R_NilValue) {
        d.eMatDH_special_grid_which = INTEGER(special_grid_which);
    } else {   /* the Gibbs entry does not receive it: a grid holds specials iff its helper row is set (first row > 0) */
        which = (int *)calloc((size_t)G, sizeof(int));
        int n = 0;
        for (int g = 0; g < G; g++)
            if (Rf_nrows(special_helper) == G && INTEGER(special_helper)[g] > 0) which[g] = ++n;
        d.eMatDH_special_grid_which = which;
    }
    d.eMatDH_special_matrix_helper = Rf_nrows(special_helper) == G ? INTEGER(special_helper) : NULL;
    d.eMatDH_special_matrix = INTEGER(special_matrix);
    d.eMatDH_special_matrix_nrow = Rf_nrows(special_matrix);
    d.use_eMatDH_special_symbols = use_eMatDH_special_symbols || !have_rhb;
    d.transMatRate_t = transMatRate;
    d.ref_error = ref_error;
    qa_panel_t *p = NULL;
    const int st = qa_panel_create(&d, &p);
    free(which);
    check_status(st, "qa_panel_create");
    {   /* QUILT_AMD_SUM_ORDER (Sys.setenv before the first call; read when the panel is uploaded): 1 / 2 = the VALIDATION MODE of
         * the full-panel passes (include/quilt_amd.h, qa_panel_set_sum_order) for the per-call entries as well */
        const char *e = getenv("QUILT_AMD_SUM_ORDER");
        if (e && (e[0] == '1' || e[0] == '2') && e[1] == 0) check_status(qa_panel_set_sum_order(p, e[0] - '0'), "qa_panel_set_sum_order");
    }
    g_cache.panel = p; g_cache.key_B = kB; g_cache.key_hm = kh;
    g_cache.K = K; g_cache.G = G; g_cache.T = T; g_cache.nMaxDH = nMaxDH;
    g_cache.ref_error = ref_error; g_cache.checksum = sum;
    return p;
}

SEXP qa_shim_release(void) {
    cache_drop();
    (void)qa_impute_release_buffers();   /* pinned transfer buffers of qa_impute_samples, if any are still kept */
    return R_NilValue;
}

/* ---- _QUILT_Rcpp_make_gl_bound(gl, minGLValue, to_fix)  (reference-single.cpp:68-94; to_fix 0-based) --------------- */

SEXP qa_QUILT_Rcpp_make_gl_bound(SEXP glSEXP, SEXP minGLValueSEXP, SEXP to_fixSEXP) {
    check_status(qa_Rcpp_make_gl_bound(REAL(glSEXP), Rf_asReal(minGLValueSEXP), INTEGER(to_fixSEXP), Rf_length(to_fixSEXP)),
                 "qa_Rcpp_make_gl_bound");
    return R_NilValue;
}

/* ---- _QUILT_Rcpp_haploid_dosage_versus_refs: the 38 arguments of RcppExports.cpp:1582, in order ------------------------- */

SEXP qa_QUILT_Rcpp_haploid_dosage_versus_refs(
    SEXP glSEXP, SEXP arma_alphaHat_tSEXP, SEXP eigen_alphaHat_tSEXP, SEXP betaHat_tSEXP, SEXP cSEXP, SEXP gamma_tSEXP,
    SEXP gammaSmall_tSEXP, SEXP best_haps_stuff_listSEXP, SEXP dosageSEXP, SEXP transMatRate_tSEXP, SEXP rhb_tSEXP,
    SEXP ref_errorSEXP, SEXP use_eMatDHSEXP, SEXP distinctHapsBSEXP, SEXP distinctHapsIESEXP,
    SEXP eMatDH_special_matrix_helperSEXP, SEXP eMatDH_special_matrixSEXP, SEXP use_eMatDH_special_symbolsSEXP,
    SEXP hapMatcherSEXP, SEXP hapMatcherRSEXP, SEXP use_hapMatcherRSEXP, SEXP gammaSmall_cols_to_getSEXP,
    SEXP eMatDH_special_grid_whichSEXP, SEXP eMatDH_special_values_listSEXP, SEXP K_top_matchesSEXP, SEXP suppressOutputSEXP,
    SEXP min_emission_prob_normalization_thresholdSEXP, SEXP return_betaHat_tSEXP, SEXP return_dosageSEXP,
    SEXP return_gamma_tSEXP, SEXP return_gammaSmall_tSEXP, SEXP get_best_haps_from_thinned_sitesSEXP, SEXP is_version_2SEXP,
    SEXP is_version_3SEXP, SEXP return_extraSEXP, SEXP always_normalizeSEXP, SEXP use_eigenSEXP, SEXP normalize_emissionsSEXP) {
    (void)eMatDH_special_values_listSEXP; (void)is_version_2SEXP; (void)return_extraSEXP; (void)use_eigenSEXP;
    if (!Rf_asLogical(use_eMatDHSEXP)) Rf_error("quilt_amd: use_eMatDH = FALSE is not supported (the production path uses TRUE)");
    qa_panel_t *panel = panel_for(hapMatcherSEXP, hapMatcherRSEXP, Rf_asLogical(use_hapMatcherRSEXP), distinctHapsBSEXP,
                                  distinctHapsIESEXP, eMatDH_special_matrix_helperSEXP, eMatDH_special_matrixSEXP,
                                  eMatDH_special_grid_whichSEXP, rhb_tSEXP, Rf_asReal(ref_errorSEXP),
                                  Rf_asLogical(use_eMatDH_special_symbolsSEXP), REAL(transMatRate_tSEXP));
    qa_fullpass_opts_t o;
    memset(&o, 0, sizeof o);
    o.K_top_matches = Rf_asInteger(K_top_matchesSEXP);
    o.return_betaHat_t = Rf_asLogical(return_betaHat_tSEXP);
    o.return_dosage = Rf_asLogical(return_dosageSEXP);
    o.return_gamma_t = Rf_asLogical(return_gamma_tSEXP);
    o.return_gammaSmall_t = Rf_asLogical(return_gammaSmall_tSEXP);
    o.get_best_haps_from_thinned_sites = Rf_asLogical(get_best_haps_from_thinned_sitesSEXP);
    o.always_normalize = Rf_asLogical(always_normalizeSEXP);
    o.normalize_emissions = Rf_asLogical(normalize_emissionsSEXP);
    o.min_emission_prob_normalization_threshold = Rf_asReal(min_emission_prob_normalization_thresholdSEXP);
    o.suppressOutput = Rf_asInteger(suppressOutputSEXP);
    /* version 3 writes alpha through the Eigen map, versions 1 / 2 through the arma matrix: both alias R's matrix */
    SEXP alphaSEXP = Rf_asLogical(is_version_3SEXP) ? eigen_alphaHat_tSEXP : arma_alphaHat_tSEXP;
    const int G = g_cache.G, K = g_cache.K;
    double *alpha = (Rf_nrows(alphaSEXP) == K && Rf_ncols(alphaSEXP) == G) ? REAL(alphaSEXP) : NULL;
    const int *cols = INTEGER(gammaSmall_cols_to_getSEXP);
    int n_thin = 0;
    for (int g = 0; g < G; g++) if (cols[g] + 1 > n_thin) n_thin = cols[g] + 1;
    int32_t *bptr = (int32_t *)calloc((size_t)n_thin + 1, sizeof(int32_t));
    int64_t cap = 64 * (int64_t)(n_thin > 0 ? n_thin : 1);
    int32_t *bidx = NULL;
    double *bval = NULL;
    int st = QA_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        bidx = (int32_t *)realloc(bidx, sizeof(int32_t) * (size_t)cap);
        bval = (double *)realloc(bval, sizeof(double) * (size_t)cap);
        st = qa_Rcpp_haploid_dosage_versus_refs(panel, REAL(glSEXP), cols, &o, alpha,
                                                o.return_betaHat_t ? REAL(betaHat_tSEXP) : NULL, REAL(cSEXP),
                                                o.return_gamma_t ? REAL(gamma_tSEXP) : NULL,
                                                o.return_gammaSmall_t ? REAL(gammaSmall_tSEXP) : NULL,
                                                o.return_dosage ? REAL(dosageSEXP) : NULL, bptr, bidx, bval, cap);
        if (st != QA_ERR_CAPACITY) break;
        cap = bptr[n_thin];   /* needed sizes are in bptr */
    }
    if (st >= 0 && o.get_best_haps_from_thinned_sites) {
        /* best_haps_stuff_list[[i]] <- list(top_matches = <0-based k>, top_matches_values = <gamma>) (reference-single.cpp:2024-2030) */
        static const char *nm[2] = {"top_matches", "top_matches_values"};
        for (int i = 0; i < n_thin && i < Rf_length(best_haps_stuff_listSEXP); i++) {
            const int n = bptr[i + 1] - bptr[i];
            SEXP e = PROTECT(named_list(2, nm));
            SEXP tm = PROTECT(Rf_allocVector(INTSXP, n)), tv = PROTECT(Rf_allocVector(REALSXP, n));
            memcpy(INTEGER(tm), bidx + bptr[i], sizeof(int) * (size_t)n);
            memcpy(REAL(tv), bval + bptr[i], sizeof(double) * (size_t)n);
            SET_VECTOR_ELT(e, 0, tm);
            SET_VECTOR_ELT(e, 1, tv);
            SET_VECTOR_ELT(best_haps_stuff_listSEXP, i, e);
            UNPROTECT(3);
        }
    }
    free(bptr); free(bidx); free(bval);
    check_status(st, "qa_Rcpp_haploid_dosage_versus_refs");
    return R_NilValue;
}

/* ---- _QUILT_rcpp_forwardBackwardGibbsNIPT: the 63 arguments of RcppExports.cpp:968, in order -------------------------- */

/* calculate_likelihoods_values + add_to_per_it_likelihoods (gibbs-nipt.cpp:1463-1621) from the per-sweep record */
/* qa_gibbs_opts_t.draw_uniforms for the 63-argument entry (NIPT): the block passes' uniforms drawn from R's generator WHEN the
 * reference draws them.  At a block iteration the reference draws runif_proposed (6 x nReads, unused by block_approach 6),
 * runif_block (nReads) and runif_total (nReads, read only by the total relabelling, which is off) -- gibbs-nipt.cpp:3013-3017 --
 * and then, inside rcpp_sample_H_using_H_class (gibbs-nipt-block.cpp:213-246), one uniform per Rcpp::sample(1:3, 1, prob), i.e.
 * per read whose class leaves a choice, in read order.  The library asks for the first kind before the pass (what = 0) and for
 * the second after its relabelling (what = 1), from the thread that made the .Call. */
static void draw_uniforms_from_R(void *ctx, int32_t chain, int32_t pass, int32_t what, int32_t n, double *out) {
    (void)ctx; (void)chain; (void)pass;
    if (what == 0) {
        for (long i = 0; i < 6L * n; i++) (void)unif_rand();   /* :3013-3015 runif_proposed */
        for (int r = 0; r < n; r++) out[r] = unif_rand();     /* :3016 runif_block */
        for (int r = 0; r < n; r++) (void)unif_rand();        /* :3017 runif_total */
    } else {
        for (int r = 0; r < n; r++) out[r] = unif_rand();     /* Rcpp::sample(one_through_3, 1, false, probs): one unif_rand() each */
    }
}

static double lgamma1(double x) { return lgamma(x + 1.0); }
static void fill_per_it_row(double *m, int nrow, int row, const double *rec, double ff, int it, double p_H_class) {
    const double prior[3] = {0.5, (1 - ff) / 2, ff / 2};
    const double d1 = rec[0], d2 = rec[1], d3 = rec[2], rc[3] = {rec[3], rec[4], rec[5]};
    double dH = 0, set = lgamma1(rc[0] + rc[1] + rc[2]);
    for (int h = 0; h < 3; h++) {
        if (rc[h] > 0) dH += rc[h] * log(prior[h]);
        if (prior[h] > 0) set += rc[h] * log(prior[h]) - lgamma1(rc[h]);
    }
    const double v[13] = {1, 1, it + 1, 1, d1, d2, d3, d1 + d2 + d3, dH, d1 + d2 + d3 + dH, set, 0, p_H_class};
    for (int j = 0; j < 13; j++) m[(size_t)j * nrow + row] = v[j];
}
static double log_p_H_class(const int *H_class, int n, double ff) {   /* rcpp_get_log_p_H_class, gibbs-nipt-block.cpp:146-164 */
    const double vals[8] = {0, log(0.5), log(0.5 - ff * 0.5), log(ff * 0.5), log(1.0 - ff * 0.5), log(0.5 + ff * 0.5), log(0.5), 0};
    double out = 0;
    for (int i = 0; i < n; i++) out += vals[H_class[i] & 7];
    return out;
}

SEXP qa_QUILT_rcpp_forwardBackwardGibbsNIPT(
    SEXP sampleReadsSEXP, SEXP eMatRead_tSEXP, SEXP priorCurrent_mSEXP, SEXP alphaMatCurrent_tcSEXP, SEXP eHapsCurrent_tcSEXP,
    SEXP transMatRate_tc_HSEXP, SEXP ffSEXP, SEXP blocks_for_outputSEXP, SEXP alphaHat_t1SEXP, SEXP betaHat_t1SEXP,
    SEXP alphaHat_t2SEXP, SEXP betaHat_t2SEXP, SEXP alphaHat_t3SEXP, SEXP betaHat_t3SEXP, SEXP eMatGrid_t1SEXP,
    SEXP eMatGrid_t2SEXP, SEXP eMatGrid_t3SEXP, SEXP gammaMT_t_localSEXP, SEXP gammaMU_t_localSEXP, SEXP gammaP_t_localSEXP,
    SEXP hapSum_tcSEXP, SEXP hapMatcherSEXP, SEXP hapMatcherRSEXP, SEXP use_hapMatcherRSEXP, SEXP distinctHapsBSEXP,
    SEXP distinctHapsIESEXP, SEXP eMatDH_special_matrix_helperSEXP, SEXP eMatDH_special_matrixSEXP, SEXP rhb_tSEXP,
    SEXP ref_errorSEXP, SEXP which_haps_to_useSEXP, SEXP wif0SEXP, SEXP grid_has_readSEXP, SEXP L_gridSEXP, SEXP smooth_cmSEXP,
    SEXP param_listSEXP, SEXP skip_read_iterationSEXP, SEXP Jmax_localSEXP, SEXP maxDifferenceBetweenReadsSEXP,
    SEXP maxEmissionMatrixDifferenceSEXP, SEXP run_fb_grid_offsetSEXP, SEXP gridSEXP, SEXP snp_start_1_basedSEXP,
    SEXP snp_end_1_basedSEXP, SEXP generate_fb_snp_offsetsSEXP, SEXP suppressOutputSEXP, SEXP n_gibbs_startsSEXP,
    SEXP n_gibbs_sample_itsSEXP, SEXP n_gibbs_burn_in_itsSEXP, SEXP double_list_of_starting_read_labelsSEXP,
    SEXP seed_vectorSEXP, SEXP prev_list_of_alphaBetaBlocksSEXP, SEXP i_snp_block_for_alpha_betaSEXP,
    SEXP do_block_resamplingSEXP, SEXP artificial_relabelSEXP, SEXP class_sum_cutoffSEXP, SEXP shuffle_bin_radiusSEXP,
    SEXP block_gibbs_iterationsSEXP, SEXP block_gibbs_quantile_probSEXP, SEXP rare_per_hap_infoSEXP,
    SEXP common_snp_indexSEXP, SEXP snp_is_commonSEXP, SEXP rare_per_snp_infoSEXP) {
    /* arguments the production caller holds constant or that only the unused code paths read */
    (void)eMatRead_tSEXP; (void)priorCurrent_mSEXP; (void)alphaMatCurrent_tcSEXP; (void)eHapsCurrent_tcSEXP;
    (void)blocks_for_outputSEXP; (void)alphaHat_t3SEXP; (void)betaHat_t3SEXP; (void)eMatGrid_t3SEXP; (void)gammaMT_t_localSEXP;
    (void)gammaMU_t_localSEXP; (void)gammaP_t_localSEXP; (void)hapSum_tcSEXP; (void)grid_has_readSEXP; (void)smooth_cmSEXP;
    (void)skip_read_iterationSEXP; (void)maxEmissionMatrixDifferenceSEXP; (void)run_fb_grid_offsetSEXP; (void)gridSEXP;
    (void)snp_start_1_basedSEXP; (void)snp_end_1_basedSEXP; (void)suppressOutputSEXP; (void)seed_vectorSEXP;
    (void)prev_list_of_alphaBetaBlocksSEXP; (void)i_snp_block_for_alpha_betaSEXP; (void)do_block_resamplingSEXP;
    (void)artificial_relabelSEXP; (void)common_snp_indexSEXP; (void)rare_per_snp_infoSEXP;
    SEXP pl = param_listSEXP;
    if (Rf_asInteger(n_gibbs_startsSEXP) != 1 || flag(pl, "run_fb_subset", 0) || Rf_asLogical(generate_fb_snp_offsetsSEXP) ||
        !flag(pl, "use_starting_read_labels", 1) || flag(pl, "pass_in_eMatRead_t", 0) || flag(pl, "use_small_eHapsCurrent_tc", 0))
        Rf_error("quilt_amd: only the production form of rcpp_forwardBackwardGibbsNIPT is supported (n_gibbs_starts = 1, "
                 "run_fb_subset = FALSE, use_starting_read_labels = TRUE, pass_in_eMatRead_t = FALSE, packed panel)");
    if (!flag(pl, "shard_check_every_pair", 1))
        Rf_error("quilt_amd: shard_check_every_pair = FALSE is not supported (QUILT() passes TRUE, quilt.R:178)");
    const double ff = Rf_asReal(ffSEXP);
    const int rare_common = flag(pl, "make_eMatRead_t_rare_common", 0);
    const int G_panel = Rf_ncols(distinctHapsBSEXP);
    /* transMatRate_tc_H: 2 x (G - 1) x S; with rare + common it is the all-SNP grid's and the panel's own is not passed:
     * the panel handle must then exist already (created by a full-panel call or an earlier common-SNP Gibbs call) */
    const int G = Rf_nrows(transMatRate_tc_HSEXP) == 2 ? (int)(Rf_xlength(transMatRate_tc_HSEXP) / 2) + 1 : G_panel;
    if (rare_common && !g_cache.panel) Rf_error("quilt_amd: the rare + common call needs the panel of an earlier common-SNP call");
    qa_panel_t *panel = rare_common ? g_cache.panel :
        panel_for(hapMatcherSEXP, hapMatcherRSEXP, Rf_asLogical(use_hapMatcherRSEXP), distinctHapsBSEXP, distinctHapsIESEXP,
                  eMatDH_special_matrix_helperSEXP, eMatDH_special_matrixSEXP, R_NilValue, rhb_tSEXP, Rf_asReal(ref_errorSEXP),
                  flag(pl, "use_eMatDH_special_symbols", 0), REAL(transMatRate_tc_HSEXP));
    const int T = rare_common ? Rf_length(snp_is_commonSEXP) : g_cache.T;
    if (rare_common && (!g_cache.rc || g_cache.key_rare != (const void *)LOGICAL(snp_is_commonSEXP))) {
        /* rare_per_hap_info: list over the K haplotypes of 1-based all-SNP indices (rare_common.R:222-247) */
        if (g_cache.rc) { qa_rare_common_destroy(g_cache.rc); g_cache.rc = NULL; }
        const int K = g_cache.K;
        int64_t *rptr = (int64_t *)calloc((size_t)K + 1, sizeof(int64_t));
        for (int k = 0; k < K; k++) rptr[k + 1] = rptr[k] + Rf_length(VECTOR_ELT(rare_per_hap_infoSEXP, k));
        int32_t *rsnp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rptr[K] > 0 ? rptr[K] : 1));
        for (int k = 0; k < K; k++)
            memcpy(rsnp + rptr[k], INTEGER(VECTOR_ELT(rare_per_hap_infoSEXP, k)), sizeof(int) * (size_t)(rptr[k + 1] - rptr[k]));
        uint8_t *isc = (uint8_t *)malloc((size_t)T);
        for (int t = 0; t < T; t++) isc[t] = LOGICAL(snp_is_commonSEXP)[t] != 0;
        const int st = qa_rare_common_create(panel, T, isc, rptr, rsnp, REAL(transMatRate_tc_HSEXP), &g_cache.rc);
        free(rptr); free(rsnp); free(isc);
        check_status(st, "qa_rare_common_create");
        g_cache.key_rare = LOGICAL(snp_is_commonSEXP);
    }

    /* ---- sampleReads -> CSR (sampleReads[[r]] = list(J, wif, bq matrix, u matrix): copied-from-stitch.cpp:153-160) */
    const int R = Rf_length(sampleReadsSEXP), Ks = Rf_length(which_haps_to_useSEXP);
    int32_t *read_ptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)R + 1));
    if (!read_ptr) Rf_error("quilt_amd: rcpp_make_eMatRead_t: out of memory");
    read_ptr[0] = 0;
    for (int r = 0; r < R; r++) read_ptr[r + 1] = read_ptr[r] + Rf_length(VECTOR_ELT(VECTOR_ELT(sampleReadsSEXP, r), 3));
    const int nB = read_ptr[R];
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nB > 0 ? nB : 1)), *bq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nB > 0 ? nB : 1));
    if (!u || !bq) {
        free(read_ptr); free(u); free(bq);
        Rf_error("quilt_amd: rcpp_make_eMatRead_t: out of memory");
    }
    for (int r = 0; r < R; r++) {
        SEXP rd = VECTOR_ELT(sampleReadsSEXP, r);
        const int n = read_ptr[r + 1] - read_ptr[r];
        memcpy(bq + read_ptr[r], INTEGER(VECTOR_ELT(rd, 2)), sizeof(int) * (size_t)n);
        memcpy(u + read_ptr[r], INTEGER(VECTOR_ELT(rd, 3)), sizeof(int) * (size_t)n);
    }
    int32_t read_off[2] = {0, R};

    /* ---- options */
    const int n_burn = Rf_asInteger(n_gibbs_burn_in_itsSEXP), n_samp = Rf_asInteger(n_gibbs_sample_itsSEXP), n_its = n_burn + n_samp;
    const int nb = Rf_length(block_gibbs_iterationsSEXP);
    qa_gibbs_opts_t o;
    memset(&o, 0, sizeof o);
    o.Ks = Ks; o.ff = ff; o.sample_is_diploid = flag(pl, "sample_is_diploid", ff == 0);
    o.Jmax = Rf_asInteger(Jmax_localSEXP);
    o.maxDifferenceBetweenReads = Rf_asReal(maxDifferenceBetweenReadsSEXP);
    o.rescale_eMatRead_t = flag(pl, "rescale_eMatRead_t", 1);
    o.n_gibbs_burn_in_its = n_burn; o.n_gibbs_sample_its = n_samp;
    o.block_gibbs_iterations = INTEGER(block_gibbs_iterationsSEXP); o.n_block_gibbs_iterations = nb;
    o.perform_block_gibbs = flag(pl, "perform_block_gibbs", 0);
    o.do_shard_block_gibbs = flag(pl, "do_shard_block_gibbs", 1);
    o.gibbs_initialize_iteratively = flag(pl, "gibbs_initialize_iteratively", 0);
    o.disable_read_category_usage = flag(pl, "disable_read_category_usage", 0);
    o.class_sum_cutoff = Rf_asReal(class_sum_cutoffSEXP);
    o.L_grid = INTEGER(L_gridSEXP); o.shuffle_bin_radius = Rf_asInteger(shuffle_bin_radiusSEXP);
    o.block_gibbs_quantile_prob = Rf_asReal(block_gibbs_quantile_probSEXP);
    double *per_it = (double *)calloc((size_t)(n_its > 0 ? n_its : 1) * 8, sizeof(double));
    o.per_it_out = per_it;

    /* ---- the reference's draws, in its order (RcppExports.cpp:971 RNGScope) */
    const size_t n_pass_unif = ff != 0 ? (size_t)nb * 2 * (size_t)R : (size_t)nb * (size_t)(G - 1);
    double *runif_reads = (double *)malloc(sizeof(double) * ((size_t)R * (size_t)n_its + 1));
    double *runif_pass = (double *)calloc(n_pass_unif + 1, sizeof(double));   /* (NIPT: filled by the library through draw_uniforms) */
    int32_t first_read = 0;
    GetRNGstate();
    for (size_t i = 0; i < (size_t)R * (size_t)n_its; i++) runif_reads[i] = unif_rand();         /* gibbs-nipt.cpp:2845 */
    if (!flag(pl, "gibbs_initialize_at_first_read", 0) && R > 0) {
        /* :2846-2848 Rcpp::sample(nReads, 1) - 1.  Rcpp's sugar (sugar/functions/sample.h, EmpiricalSample, size < 2) draws
         * static_cast<int>(n * unif_rand() + 1): ONE unif_rand(), no rejection loop (R_unif_index under sample.kind =
         * "Rejection" would consume a data-dependent number of draws and return another value). */
        first_read = (int32_t)(unif_rand() * (double)R);
        if (first_read >= R) first_read = R - 1;
    }
    if (o.perform_block_gibbs && ff != 0) {
        /* NIPT: the block passes' draws depend on the passes' own results (one Rcpp::sample per read whose class leaves a choice),
         * so they are made when the reference makes them, through the library's callback; the generator stays open until the call
         * returns (PutRNGstate below the call) */
        o.draw_uniforms = draw_uniforms_from_R;
        o.draw_uniforms_ctx = NULL;
    }
    if (o.perform_block_gibbs && ff == 0) {
        for (int ib = 0; ib < nb; ib++) {
            for (size_t i = 0; i < 6 * (size_t)R; i++) (void)unif_rand();                        /* :3013-3015 runif_proposed (unused by approach 6) */
            {
                for (size_t i = 0; i < 2 * (size_t)R; i++) (void)unif_rand();                    /* :3016-3017, no effect for diploid samples */
                if (o.do_shard_block_gibbs)
                    for (int g = 0; g < G - 1; g++) runif_pass[(size_t)ib * (G - 1) + g] = unif_rand();   /* gibbs-nipt-block.cpp:2054 */
            }
        }
    }

    /* ---- starting labels in, ending labels out */
    SEXP start = VECTOR_ELT(VECTOR_ELT(double_list_of_starting_read_labelsSEXP, 0), 0);
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1)), *H_class = (int32_t *)calloc((size_t)(R > 0 ? R : 1), sizeof(int32_t));
    memcpy(H, INTEGER(start), sizeof(int) * (size_t)R);
    const int want_hap = flag(pl, "return_hapProbs", 0), want_gen = flag(pl, "return_genProbs", 0);
    SEXP hap = PROTECT(Rf_allocMatrix(REALSXP, 3, T)), gm = PROTECT(Rf_allocMatrix(REALSXP, 3, T)), gf = PROTECT(Rf_allocMatrix(REALSXP, 3, T));
    double *state = (double *)malloc(sizeof(double) * ((size_t)6 * Ks * G + (size_t)3 * G));
    int32_t underflow = 0;
    const int st = rare_common
        ? qa_gibbs_batch_rare_common(panel, g_cache.rc, &o, 1, INTEGER(which_haps_to_useSEXP), read_off, read_ptr, u, bq,
                                     INTEGER(wif0SEXP), runif_reads, &first_read, runif_pass, H, H_class,
                                     (want_hap || want_gen) ? REAL(hap) : NULL, want_gen ? REAL(gm) : NULL,
                                     want_gen ? REAL(gf) : NULL, &underflow, state, NULL, NULL)
        : qa_gibbs_batch(panel, &o, 1, INTEGER(which_haps_to_useSEXP), read_off, read_ptr, u, bq, INTEGER(wif0SEXP), runif_reads,
                         &first_read, runif_pass, H, H_class, (want_hap || want_gen) ? REAL(hap) : NULL,
                         want_gen ? REAL(gm) : NULL, want_gen ? REAL(gf) : NULL, &underflow, state, NULL, NULL);
    PutRNGstate();   /* (after the call: with NIPT block passes the library draws through draw_uniforms_from_R while it runs) */
    SEXP out = R_NilValue;
    if (st >= 0 && !underflow) {
        /* the matrices the reference mutates in place (pass_in_alphaBeta = TRUE: R's buffers, quilt.R:731-745) */
        const size_t m = (size_t)Ks * G;
        SEXP dst[6] = {alphaHat_t1SEXP, alphaHat_t2SEXP, betaHat_t1SEXP, betaHat_t2SEXP, eMatGrid_t1SEXP, eMatGrid_t2SEXP};
        for (int i = 0; i < 6; i++)
            if (Rf_nrows(dst[i]) == Ks && Rf_ncols(dst[i]) == G) memcpy(REAL(dst[i]), state + (size_t)i * m, sizeof(double) * m);
        /* the result list, names as gibbs-nipt.cpp:3217-3306 */
        const char *nm[8];
        int n = 0;
        nm[n++] = "underflow_problem";
        if (want_gen) { nm[n++] = "genProbsM_t"; nm[n++] = "genProbsF_t"; }
        if (want_hap) nm[n++] = "hapProbs_t";
        nm[n++] = "H"; nm[n++] = "double_list_of_ending_read_labels"; nm[n++] = "per_it_likelihoods"; nm[n++] = "H_class";
        out = PROTECT(named_list(n, nm));
        int at = 0;
        SET_VECTOR_ELT(out, at++, Rf_ScalarLogical(0));
        if (want_gen) { SET_VECTOR_ELT(out, at++, gm); SET_VECTOR_ELT(out, at++, gf); }
        if (want_hap) SET_VECTOR_ELT(out, at++, hap);
        SEXP Hs = PROTECT(Rf_allocVector(INTSXP, R)), Hc = PROTECT(Rf_allocVector(INTSXP, R));
        memcpy(INTEGER(Hs), H, sizeof(int) * (size_t)R);
        memcpy(INTEGER(Hc), H_class, sizeof(int) * (size_t)R);
        SET_VECTOR_ELT(out, at++, Hs);
        SEXP l1 = PROTECT(Rf_allocVector(VECSXP, 1)), l2 = PROTECT(Rf_allocVector(VECSXP, 1));   /* [[s]][[i_gibbs_sampling]] */
        SET_VECTOR_ELT(l2, 0, Hs);
        SET_VECTOR_ELT(l1, 0, l2);
        SET_VECTOR_ELT(out, at++, l1);
        /* per_it_likelihoods: one row per sweep, the 13 columns of gibbs-nipt.cpp:2767-2768 */
        static const char *cn[13] = {"s", "i_samp", "i_it", "i_result_it", "p_O1_given_H1_L", "p_O2_given_H2_L", "p_O3_given_H3_L",
                                     "p_O_given_H_L", "p_H_given_L", "p_O_H_given_L_up_to_C", "p_set_H_given_L", "relabel",
                                     "p_H_class_given_L"};
        SEXP pit = PROTECT(Rf_allocMatrix(REALSXP, n_its, 13));
        for (int it = 0; it < n_its; it++)   /* H_class is recorded by the last sweep only: NA before */
            fill_per_it_row(REAL(pit), n_its, it, per_it + (size_t)it * 8, ff, it,
                            it == n_its - 1 ? log_p_H_class(H_class, R, ff) : NA_REAL);
        SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2)), cns = PROTECT(Rf_allocVector(STRSXP, 13));
        for (int j = 0; j < 13; j++) SET_STRING_ELT(cns, j, Rf_mkChar(cn[j]));
        SET_VECTOR_ELT(dn, 0, R_NilValue);
        SET_VECTOR_ELT(dn, 1, cns);
        Rf_setAttrib(pit, R_DimNamesSymbol, dn);
        SET_VECTOR_ELT(out, at++, pit);
        SET_VECTOR_ELT(out, at++, Hc);
        UNPROTECT(8);
    } else if (st >= 0) {
        /* gibbs-nipt.cpp:2959-2969: only `underflow_problem = TRUE`; impute_one_sample retries (functions.R:2704-2715) */
        static const char *nm[1] = {"underflow_problem"};
        out = PROTECT(named_list(1, nm));
        SET_VECTOR_ELT(out, 0, Rf_ScalarLogical(1));
        UNPROTECT(1);
    }
    free(read_ptr); free(u); free(bq); free(per_it); free(runif_reads); free(runif_pass); free(H); free(H_class); free(state);
    UNPROTECT(3);
    check_status(st, "qa_gibbs_batch");
    return out;
}

/* ---- _QUILT_rcpp_make_eMatRead_t: the 15 arguments of RcppExports.cpp:17, in order ----------------------------------
 * Read likelihoods against the K rows of eHapsCurrent_tc[, , s + 1] (copied-from-stitch.cpp:115-229), written into the
 * caller's K x nReads eMatRead_t; production caller: calculate_eMatRead_t_vs_haplotypes (functions.R:2975-3020, K = 2 or 3).
 * Needs the device context of a panel handle: the calls that precede it in the driver loop have created one. */
#ifdef QA_INSIDE_QUILT_SO
/* Compiled into QUILT.so (shim/QUILT-src.patch): the package's own Rcpp wrapper is still there and takes every call this
 * entry does not cover -- the other callers of rcpp_make_eMatRead_t (K = KL rows: functions.R:2903, reference-single.R:545,
 * gibbs-nipt.R:126/1634, the test drivers), pseudo-haploid mode, and calls made before any panel was uploaded. */
extern SEXP _QUILT_rcpp_make_eMatRead_t(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
#define QA_EMATREAD_NOT_COVERED(msg)                                                                                          \
    return _QUILT_rcpp_make_eMatRead_t(eMatRead_tSEXP, sampleReadsSEXP, eHapsCurrent_tcSEXP, sSEXP, maxDifferenceBetweenReadsSEXP, \
                                       JmaxSEXP, eMatHapOri_tSEXP, pRgivenH1SEXP, pRgivenH2SEXP, prevSEXP, suppressOutputSEXP,  \
                                       prev_sectionSEXP, next_sectionSEXP, run_pseudo_haploidSEXP, rescale_eMatRead_tSEXP)
#else
#define QA_EMATREAD_NOT_COVERED(msg) Rf_error("quilt_amd: rcpp_make_eMatRead_t: " msg)
#endif
#define QA_EMATREAD_MAX_K 3   /* the device form serves the driver loop's caller (K = 2 or 3 sampled haplotypes) */

SEXP qa_QUILT_rcpp_make_eMatRead_t(SEXP eMatRead_tSEXP, SEXP sampleReadsSEXP, SEXP eHapsCurrent_tcSEXP, SEXP sSEXP,
                                   SEXP maxDifferenceBetweenReadsSEXP, SEXP JmaxSEXP, SEXP eMatHapOri_tSEXP, SEXP pRgivenH1SEXP,
                                   SEXP pRgivenH2SEXP, SEXP prevSEXP, SEXP suppressOutputSEXP, SEXP prev_sectionSEXP,
                                   SEXP next_sectionSEXP, SEXP run_pseudo_haploidSEXP, SEXP rescale_eMatRead_tSEXP) {
    (void)eMatHapOri_tSEXP; (void)pRgivenH1SEXP; (void)pRgivenH2SEXP; (void)prevSEXP; (void)suppressOutputSEXP;
    (void)prev_sectionSEXP; (void)next_sectionSEXP;
    if (Rf_asLogical(run_pseudo_haploidSEXP)) QA_EMATREAD_NOT_COVERED("run_pseudo_haploid = TRUE is not supported");
    if (!g_cache.panel) QA_EMATREAD_NOT_COVERED("needs the panel of an earlier full-panel or Gibbs call");
    const int K = Rf_nrows(eMatRead_tSEXP), R = Rf_length(sampleReadsSEXP), s = Rf_asInteger(sSEXP);
    if (K > QA_EMATREAD_MAX_K) QA_EMATREAD_NOT_COVERED("more than 3 haplotype rows");
    SEXP dim = Rf_getAttrib(eHapsCurrent_tcSEXP, R_DimSymbol);
    if (Rf_length(dim) != 3 || INTEGER(dim)[0] != K || s < 0 || s >= INTEGER(dim)[2] || Rf_ncols(eMatRead_tSEXP) != R)
        Rf_error("quilt_amd: rcpp_make_eMatRead_t: eHapsCurrent_tc must be K x nSNPs x S with K = nrow(eMatRead_t), 0 <= s < S");
    const int T = INTEGER(dim)[1];
    int32_t *read_ptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)R + 1));
    read_ptr[0] = 0;
    for (int r = 0; r < R; r++) read_ptr[r + 1] = read_ptr[r] + Rf_length(VECTOR_ELT(VECTOR_ELT(sampleReadsSEXP, r), 3));
    const int nB = read_ptr[R];
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nB > 0 ? nB : 1)), *bq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nB > 0 ? nB : 1));
    for (int r = 0; r < R; r++) {
        SEXP rd = VECTOR_ELT(sampleReadsSEXP, r);
        const int n = read_ptr[r + 1] - read_ptr[r];
        memcpy(bq + read_ptr[r], INTEGER(VECTOR_ELT(rd, 2)), sizeof(int) * (size_t)n);
        memcpy(u + read_ptr[r], INTEGER(VECTOR_ELT(rd, 3)), sizeof(int) * (size_t)n);
    }
    const int32_t read_off[2] = {0, R};
    const int st = qa_rcpp_make_eMatRead_t_nsnps(g_cache.panel, T, 1, K, REAL(eHapsCurrent_tcSEXP) + (size_t)s * K * T, read_off,
                                                 read_ptr, u, bq, Rf_asReal(maxDifferenceBetweenReadsSEXP), Rf_asInteger(JmaxSEXP),
                                                 Rf_asLogical(rescale_eMatRead_tSEXP), REAL(eMatRead_tSEXP));
    free(read_ptr); free(u); free(bq);
    check_status(st, "qa_rcpp_make_eMatRead_t");
    return R_NilValue;
}


/* ---- qa_impute_sample_range: the body of QUILT()'s loop over a core's samples as ONE `.Call` -----------------------------
 * QUILT/R/quilt.R:688-996 runs `for(iSample in sampleRange[1]:sampleRange[2]) get_and_impute_one_sample(...)` inside
 * mclapply; with the GPU library the whole range goes to qa_impute_samples (include/quilt_amd.h; csrc/impute.cpp: the loop of
 * functions.R:3-1500 in host C++ over the batched kernels), n_handles host threads sharing the device.  R-side (INTEGRATION.md
 * 4a has the replacement of the mclapply body):
 *     out <- .Call("qa_impute_sample_range", list_of_sampleReads, panel_objects, params, sample_offset, n_handles,
 *                  list_of_allSNP_sampleReads)
 *   list_of_sampleReads   one sampleReads (list of list(J, wif, bq, u), copied-from-stitch.cpp:153-160) per sample of the range
 *   panel_objects         named list: hapMatcherR, distinctHapsB, distinctHapsIE, eMatDH_special_matrix_helper,
 *                         eMatDH_special_matrix, rhb_t, transMatRate_t (2 x (nGrids - 1)), ref_error, use_eMatDH_special_symbols;
 *                         method = "nipt": L_grid (nGrids);
 *                         impute_rare_common = TRUE: rare_common = list(snp_is_common, rare_per_hap_info, transMatRate_t
 *                         (2 x (nGrids_all - 1)), L_grid) of special_rare_common_objects (prepare_reference_functions.R:172-247)
 *   params                named list of QUILT()'s arguments the path sees (missing entries = the reference's defaults):
 *                         nGibbsSamples, n_seek_its, n_burn_in_seek_its, Ksubset, Knew, K_top_matches, heuristic_match_thin,
 *                         small_ref_panel_gibbs_iterations, small_ref_panel_block_gibbs_iterations (0-based), maxDifferenceBetweenReads,
 *                         minGLValue, Jmax, seed (a non-negative whole number below 2^53), samples_per_launch_set, device (0-based GPU of
 *                         this worker, taken modulo the number of devices: mclapply's iCore - 1; absent: the current device);
 *                         use_mspbwt, mspbwtL, mspbwtM, mspbwt_nindices;
 *                         impute_rare_common; method ("diploid" / "nipt"), ff (one fetal fraction per sample), shuffle_bin_radius
 *   sample_offset         0-based index of the range's first sample among ALL samples (keys the random streams: a sample's
 *                         result does not depend on the range it lands in)
 *   list_of_allSNP_sampleReads   impute_rare_common = TRUE: allSNP_sampleReads per sample (functions.R:162-172: u over all SNPs,
 *                         wif on the all-SNP grid); NULL otherwise
 * Returns list(dosage = nSNPs x n, gp_t = 3 nSNPs x n (per sample 3 x nSNPs, row-major as the library writes it),
 * phasing_haps = 2 nSNPs x n (nipt: 3 nSNPs x n), read_labels = list of integer vectors, nDosage, stats[, fet_dosage, fet_gp_t]);
 * nSNPs = all SNPs with impute_rare_common.  Draws: the library's counter streams (R's stream cannot be handed to 2 048 chains
 * advancing in lock-step); `seed` plays set.seed's part.  use_mspbwt: the panel's msPBWT indices are built here by
 * qa_mspbwt_create (the `ms_indices` the reference loads are the mspbwt package's own structures). */

static double num_or(SEXP list, const char *name, double dflt) {
    SEXP v = list_get(list, name);
    return (v == R_NilValue || Rf_length(v) < 1) ? dflt : Rf_asReal(v);
}

/* a list of sampleReads in the flattened form of include/quilt_amd.h (read_off n + 1; read_ptr R + 1 per sample; bases back to back) */
typedef struct { int32_t *read_off, *read_ptr, *wif, *u, *bq; } flat_reads_t;
static void flat_reads_free(flat_reads_t *f) { free(f->read_off); free(f->read_ptr); free(f->wif); free(f->u); free(f->bq); }
/* Every R-level check of a list of sampleReads, BEFORE anything is allocated: an R error longjmps out of the routine, so it must
 * not be raised (by INTEGER() on a REALSXP, say) while panel handles, the msPBWT index or malloc'd buffers are live. */
static void validate_reads_list(SEXP readsListSEXP, const char *what) {
    if (TYPEOF(readsListSEXP) != VECSXP) Rf_error("quilt_amd: qa_impute_sample_range: %s must be a list of sampleReads", what);
    const int n = Rf_length(readsListSEXP);
    for (int i = 0; i < n; i++) {
        SEXP sr = VECTOR_ELT(readsListSEXP, i);
        if (TYPEOF(sr) != VECSXP) Rf_error("quilt_amd: qa_impute_sample_range: %s[[%d]] is not a list of reads", what, i + 1);
        const int R = Rf_length(sr);
        for (int r = 0; r < R; r++) {
            SEXP rd = VECTOR_ELT(sr, r);
            if (TYPEOF(rd) != VECSXP || Rf_length(rd) < 4)
                Rf_error("quilt_amd: qa_impute_sample_range: %s[[%d]][[%d]] is not list(J, wif, bq, u)", what, i + 1, r + 1);
            SEXP bq = VECTOR_ELT(rd, 2), u = VECTOR_ELT(rd, 3), wif = VECTOR_ELT(rd, 1);
            if (TYPEOF(bq) != INTSXP || TYPEOF(u) != INTSXP || Rf_length(bq) != Rf_length(u))
                Rf_error("quilt_amd: qa_impute_sample_range: %s[[%d]][[%d]]: bq and u must be integer vectors of one length "
                         "(sampleReads as loadBamAndConvert writes them)", what, i + 1, r + 1);
            if ((TYPEOF(wif) != INTSXP && TYPEOF(wif) != REALSXP) || Rf_length(wif) < 1)
                Rf_error("quilt_amd: qa_impute_sample_range: %s[[%d]][[%d]]: the central grid (element 2) must be a number", what, i + 1, r + 1);
        }
    }
}

/* returns 0, or -1 when an allocation failed (everything it allocated is then freed again).  The list has been validated. */
static int flatten_reads(SEXP readsListSEXP, flat_reads_t *f) {
    const int n = Rf_length(readsListSEXP);
    memset(f, 0, sizeof *f);
    f->read_off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    if (!f->read_off) return -1;
    f->read_off[0] = 0;
    size_t nbases = 0;
    for (int i = 0; i < n; i++) {
        SEXP sr = VECTOR_ELT(readsListSEXP, i);
        const int R = Rf_length(sr);
        f->read_off[i + 1] = f->read_off[i] + R;
        for (int r = 0; r < R; r++) nbases += (size_t)Rf_length(VECTOR_ELT(VECTOR_ELT(sr, r), 3));
    }
    const int totR = f->read_off[n];
    f->read_ptr = (int32_t *)malloc(sizeof(int32_t) * ((size_t)totR + (size_t)n + 1));
    f->wif = (int32_t *)malloc(sizeof(int32_t) * (size_t)(totR > 0 ? totR : 1));
    f->u = (int32_t *)malloc(sizeof(int32_t) * (nbases > 0 ? nbases : 1));
    f->bq = (int32_t *)malloc(sizeof(int32_t) * (nbases > 0 ? nbases : 1));
    if (!f->read_ptr || !f->wif || !f->u || !f->bq) {
        flat_reads_free(f);
        memset(f, 0, sizeof *f);
        return -1;
    }
    size_t at = 0;
    for (int i = 0; i < n; i++) {
        SEXP sr = VECTOR_ELT(readsListSEXP, i);
        const int R = Rf_length(sr);
        int32_t *rp = f->read_ptr + f->read_off[i] + i;
        rp[0] = 0;
        for (int r = 0; r < R; r++) {
            SEXP rd = VECTOR_ELT(sr, r);
            const int nb = Rf_length(VECTOR_ELT(rd, 3));
            f->wif[f->read_off[i] + r] = Rf_asInteger(VECTOR_ELT(rd, 1));
            memcpy(f->bq + at, INTEGER(VECTOR_ELT(rd, 2)), sizeof(int) * (size_t)nb);
            memcpy(f->u + at, INTEGER(VECTOR_ELT(rd, 3)), sizeof(int) * (size_t)nb);
            at += (size_t)nb;
            rp[r + 1] = rp[r] + nb;
        }
    }
    return 0;
}

/* What the two range routines share: the panel handles (one per host thread), the parameter struct with its msPBWT index,
 * all-SNP handles and NIPT block, from the R objects `panel_objects` and `params`. */
typedef struct {
    int K, G, T, T_out, n_handles, nipt, rare;
    qa_panel_t *handles[16];
    int made;
    qa_impute_params_t ip;
    qa_mspbwt_t *index;
    qa_impute_rare_common_t rcq;
    qa_rare_common_t *rcs[16];
    int n_rc;
    int64_t *rare_ptr;
    int32_t *rare_snp, *L_grid_all, *L_grid;
    uint8_t *is_common;
    qa_impute_nipt_t nq;
    char msg[512];
} range_ctx_t;

/* Every check that can raise an R error comes first: nothing is allocated yet, nothing can leak (an R error longjmps). */
static void range_validate(SEXP panelSEXP, SEXP paramsSEXP, const char *who) {
    SEXP hapMatcherR = list_get(panelSEXP, "hapMatcherR"), distinctHapsB = list_get(panelSEXP, "distinctHapsB");
    SEXP distinctHapsIE = list_get(panelSEXP, "distinctHapsIE"), tm = list_get(panelSEXP, "transMatRate_t");
    SEXP helper = list_get(panelSEXP, "eMatDH_special_matrix_helper"), spmat = list_get(panelSEXP, "eMatDH_special_matrix");
    SEXP rhb_t = list_get(panelSEXP, "rhb_t");
    if (hapMatcherR == R_NilValue || distinctHapsB == R_NilValue || distinctHapsIE == R_NilValue || tm == R_NilValue ||
        helper == R_NilValue || spmat == R_NilValue)
        Rf_error("quilt_amd: %s: panel_objects needs hapMatcherR, distinctHapsB, distinctHapsIE, "
                 "eMatDH_special_matrix_helper, eMatDH_special_matrix, transMatRate_t", who);
    if (TYPEOF(hapMatcherR) != RAWSXP || TYPEOF(distinctHapsB) != INTSXP || TYPEOF(distinctHapsIE) != REALSXP || TYPEOF(tm) != REALSXP ||
        TYPEOF(helper) != INTSXP || TYPEOF(spmat) != INTSXP || (rhb_t != R_NilValue && TYPEOF(rhb_t) != INTSXP))
        Rf_error("quilt_amd: %s: panel_objects: hapMatcherR must be raw, distinctHapsB / eMatDH_special_matrix(_helper) / "
                 "rhb_t integer, distinctHapsIE / transMatRate_t numeric", who);
    const int G = Rf_ncols(hapMatcherR);
    if (Rf_length(tm) != 2 * (G > 1 ? G - 1 : 0))
        Rf_error("quilt_amd: %s: panel_objects$transMatRate_t must be 2 x (nGrids - 1)", who);
    SEXP rc0 = list_get(panelSEXP, "rare_common");
    SEXP rph0 = rc0 == R_NilValue ? R_NilValue : list_get(rc0, "rare_per_hap_info");
    if (rph0 != R_NilValue) {
        if (TYPEOF(rph0) != VECSXP) Rf_error("quilt_amd: %s: rare_common$rare_per_hap_info must be a list", who);
        for (int k = 0; k < Rf_length(rph0); k++)
            if (TYPEOF(VECTOR_ELT(rph0, k)) != INTSXP && Rf_length(VECTOR_ELT(rph0, k)) > 0)
                Rf_error("quilt_amd: %s: rare_common$rare_per_hap_info[[%d]] must be an integer vector", who, k + 1);
        SEXP sic0 = list_get(rc0, "snp_is_common"), tma0 = list_get(rc0, "transMatRate_t");
        if ((sic0 != R_NilValue && TYPEOF(sic0) != LGLSXP) || (tma0 != R_NilValue && TYPEOF(tma0) != REALSXP))
            Rf_error("quilt_amd: %s: rare_common$snp_is_common must be logical, $transMatRate_t numeric", who);
    }
    {   /* grid positions (method = "nipt": the block definition): numbers, wherever they are given */
        SEXP lg = list_get(panelSEXP, "L_grid"), lga = rc0 == R_NilValue ? R_NilValue : list_get(rc0, "L_grid");
        if ((lg != R_NilValue && TYPEOF(lg) != INTSXP && TYPEOF(lg) != REALSXP) || (lga != R_NilValue && TYPEOF(lga) != INTSXP && TYPEOF(lga) != REALSXP))
            Rf_error("quilt_amd: %s: L_grid must be numeric", who);
        SEXP ff = list_get(paramsSEXP, "ff");
        if (ff != R_NilValue && TYPEOF(ff) != REALSXP) Rf_error("quilt_amd: %s: params$ff must be numeric (as.numeric)", who);
    }
    const double seed_d = num_or(paramsSEXP, "seed", 1);
    if (!(seed_d >= 0) || seed_d > 9007199254740992.0 /* 2^53 */ || seed_d != floor(seed_d))   /* (NA / NaN fail the first test) */
        Rf_error("quilt_amd: %s: params$seed must be a non-negative whole number below 2^53", who);
    const double so = num_or(paramsSEXP, "sum_order", 0);
    if (!(so == 0 || so == 1 || so == 2)) Rf_error("quilt_amd: %s: params$sum_order must be 0, 1 or 2 (include/quilt_amd.h: qa_panel_set_sum_order)", who);
    SEXP blocks = list_get(paramsSEXP, "small_ref_panel_block_gibbs_iterations");
    if (blocks != R_NilValue && TYPEOF(blocks) != INTSXP)
        Rf_error("quilt_amd: %s: params$small_ref_panel_block_gibbs_iterations must be an integer vector (0-based sweeps)", who);
    /* which GPU: params$device (0-based; taken modulo the number of devices, so that mclapply's iCore - 1 can be passed as it
     * is); absent: the process's current device.  Forked workers must make their first HIP call after the fork -- this one. */
    const double dev_d = num_or(paramsSEXP, "device", -1);
    if (dev_d >= 0) {
        const int n_dev = qa_device_count();
        if (n_dev < 1) Rf_error("quilt_amd: %s: no gfx950 device (libquilt_amd has no CPU fallback)", who);
        check_status(qa_set_device((int)dev_d % n_dev), "qa_set_device");
    }
}

static void range_teardown(range_ctx_t *cx) {
    for (int i = 0; i < cx->n_rc; i++) if (cx->rcs[i]) qa_rare_common_destroy(cx->rcs[i]);
    if (cx->index) qa_mspbwt_destroy(cx->index);
    free(cx->rare_ptr); free(cx->rare_snp); free(cx->is_common); free(cx->L_grid); free(cx->L_grid_all);
    for (int i = 0; i < cx->made; i++) if (cx->handles[i]) qa_panel_destroy(cx->handles[i]);
    cx->n_rc = 0; cx->made = 0; cx->index = NULL;
    cx->rare_ptr = NULL; cx->rare_snp = NULL; cx->is_common = NULL; cx->L_grid = NULL; cx->L_grid_all = NULL;
}

/* Raises no R error: returns a QA status with the text in cx->msg; on failure everything made so far is torn down again.
 * `n_sample`: what params$ff must have one entry of (method = "nipt"); the fetus' outputs and ff are the caller's to set. */
static int range_setup(range_ctx_t *cx, SEXP panelSEXP, SEXP paramsSEXP, int n_handles, int n_sample) {
    memset(cx, 0, sizeof *cx);
    SEXP hapMatcherR = list_get(panelSEXP, "hapMatcherR"), distinctHapsB = list_get(panelSEXP, "distinctHapsB");
    SEXP distinctHapsIE = list_get(panelSEXP, "distinctHapsIE"), tm = list_get(panelSEXP, "transMatRate_t");
    SEXP helper = list_get(panelSEXP, "eMatDH_special_matrix_helper"), spmat = list_get(panelSEXP, "eMatDH_special_matrix");
    SEXP rhb_t = list_get(panelSEXP, "rhb_t");
    const int K = cx->K = Rf_nrows(hapMatcherR), G = cx->G = Rf_ncols(hapMatcherR), T = cx->T = Rf_ncols(distinctHapsIE);
    cx->T_out = T;
    if (n_handles < 1) n_handles = 1;
    if (n_handles > 16) n_handles = 16;
    cx->n_handles = n_handles;
    /* one handle per host thread: replicas of the panel on this process's device, taking the device in turn */
    qa_panel_desc_t d;
    memset(&d, 0, sizeof d);
    d.K = K; d.nGrids = G; d.nSNPs = T; d.nMaxDH = Rf_nrows(distinctHapsB);
    d.hapMatcherR = RAW(hapMatcherR);
    const int have_rhb = rhb_t != R_NilValue && Rf_nrows(rhb_t) == K && Rf_ncols(rhb_t) == G;
    d.rhb_t = have_rhb ? INTEGER(rhb_t) : NULL;
    d.distinctHapsB = INTEGER(distinctHapsB);
    d.distinctHapsIE = REAL(distinctHapsIE);
    int *which = (int *)calloc((size_t)(G > 0 ? G : 1), sizeof(int));
    if (!which) { snprintf(cx->msg, sizeof cx->msg, "out of memory"); return QA_ERR_INVALID; }
    int nsp = 0;
    for (int g = 0; g < G; g++)
        if (Rf_nrows(helper) == G && INTEGER(helper)[g] > 0) which[g] = ++nsp;
    d.eMatDH_special_grid_which = which;
    d.eMatDH_special_matrix_helper = Rf_nrows(helper) == G ? INTEGER(helper) : NULL;
    d.eMatDH_special_matrix = INTEGER(spmat);
    d.eMatDH_special_matrix_nrow = Rf_nrows(spmat);
    d.use_eMatDH_special_symbols = !have_rhb || (int)num_or(panelSEXP, "use_eMatDH_special_symbols", 0);
    d.transMatRate_t = REAL(tm);
    d.ref_error = num_or(panelSEXP, "ref_error", 1e-3);
    /* params$sum_order (quilt-amd.R takes it from QUILT_AMD_SUM_ORDER): 0 = production kernels; 1 = VALIDATION MODE, every K-wide
     * sum of the full-panel passes in the order the reference's code adds it (bit-identical to the CPU package's lists, 20-50x
     * slower passes); 2 = the same with grid 0's Armadillo sum read left to right (include/quilt_amd.h) */
    const int sum_order = (int)num_or(paramsSEXP, "sum_order", 0);
    int st = QA_OK;
    for (; cx->made < n_handles && st == QA_OK; cx->made++) {
        cx->handles[cx->made] = NULL;
        st = qa_panel_create(&d, &cx->handles[cx->made]);
        if (st == QA_OK) st = qa_panel_set_device_share(cx->handles[cx->made], n_handles);
        if (st == QA_OK && n_handles > 1) st = qa_panel_set_exclusive(cx->handles[cx->made], 1);
        if (st == QA_OK) st = qa_panel_set_dosage_precision(cx->handles[cx->made], 64);   /* the reference computes in double */
        if (st == QA_OK && sum_order) st = qa_panel_set_sum_order(cx->handles[cx->made], sum_order);
    }
    free(which);
    if (st != QA_OK) {
        snprintf(cx->msg, sizeof cx->msg, "qa_panel_create: %s", qa_last_error());
        range_teardown(cx);
        return st;
    }
    qa_impute_params_t *ip = &cx->ip;
    qa_impute_params_default(ip);
    ip->nGibbsSamples = (int)num_or(paramsSEXP, "nGibbsSamples", ip->nGibbsSamples);
    ip->n_seek_its = (int)num_or(paramsSEXP, "n_seek_its", ip->n_seek_its);
    ip->n_burn_in_seek_its = (int)num_or(paramsSEXP, "n_burn_in_seek_its", -1);   /* NA -> n_seek_its - 1 (quilt.R:248-250) */
    ip->Ksubset = (int)num_or(paramsSEXP, "Ksubset", ip->Ksubset);
    ip->Knew = (int)num_or(paramsSEXP, "Knew", ip->Knew);
    ip->K_top_matches = (int)num_or(paramsSEXP, "K_top_matches", ip->K_top_matches);
    ip->heuristic_match_thin = num_or(paramsSEXP, "heuristic_match_thin", ip->heuristic_match_thin);
    ip->small_ref_panel_gibbs_iterations = (int)num_or(paramsSEXP, "small_ref_panel_gibbs_iterations", ip->small_ref_panel_gibbs_iterations);
    SEXP blocks = list_get(paramsSEXP, "small_ref_panel_block_gibbs_iterations");
    if (blocks != R_NilValue && TYPEOF(blocks) == INTSXP) {
        ip->small_ref_panel_block_gibbs_iterations = INTEGER(blocks);
        ip->n_block_gibbs_iterations = Rf_length(blocks);
    }
    ip->maxDifferenceBetweenReads = num_or(paramsSEXP, "maxDifferenceBetweenReads", ip->maxDifferenceBetweenReads);
    ip->minGLValue = num_or(paramsSEXP, "minGLValue", ip->minGLValue);
    ip->Jmax = (int)num_or(paramsSEXP, "Jmax", ip->Jmax);
    ip->seed = (uint64_t)num_or(paramsSEXP, "seed", 1);   /* (range-checked by range_validate) */
    ip->samples_per_launch_set = (int)num_or(paramsSEXP, "samples_per_launch_set", 0);
    /* use_mspbwt = TRUE (QUILT2's default; mspbwt.R:225-474): the panel's indices, built here */
    if (flag(paramsSEXP, "use_mspbwt", 0)) {
        ip->use_mspbwt = 1;
        ip->mspbwtL = (int)num_or(paramsSEXP, "mspbwtL", ip->mspbwtL);
        ip->mspbwtM = (int)num_or(paramsSEXP, "mspbwtM", ip->mspbwtM);
        cx->index = qa_mspbwt_create(K, G, RAW(hapMatcherR), d.nMaxDH, INTEGER(distinctHapsB), (int)num_or(paramsSEXP, "mspbwt_nindices", 4));
        if (!cx->index) { st = QA_ERR_INVALID; snprintf(cx->msg, sizeof cx->msg, "qa_mspbwt_create: %s", qa_last_error()); }
        ip->mspbwt_index = cx->index;
    }
    /* impute_rare_common = TRUE (functions.R:1042-1123): one all-SNP handle per panel handle (the all-SNP reads are the caller's) */
    cx->rare = flag(paramsSEXP, "impute_rare_common", 0);
    if (st == QA_OK && cx->rare) {
        SEXP rc = list_get(panelSEXP, "rare_common");
        SEXP sic = rc == R_NilValue ? R_NilValue : list_get(rc, "snp_is_common"), rph = rc == R_NilValue ? R_NilValue : list_get(rc, "rare_per_hap_info");
        SEXP tma = rc == R_NilValue ? R_NilValue : list_get(rc, "transMatRate_t"), lga = rc == R_NilValue ? R_NilValue : list_get(rc, "L_grid");
        if (sic == R_NilValue || rph == R_NilValue || tma == R_NilValue || Rf_length(rph) != K) {
            st = QA_ERR_INVALID;
            snprintf(cx->msg, sizeof cx->msg, "impute_rare_common: panel_objects$rare_common needs snp_is_common, rare_per_hap_info (one entry per "
                                              "haplotype), transMatRate_t");
        } else {
            const int T_out = cx->T_out = Rf_length(sic);
            cx->is_common = (uint8_t *)malloc((size_t)(T_out > 0 ? T_out : 1));
            cx->rare_ptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)K + 1));
            if (cx->is_common && cx->rare_ptr) {
                for (int t = 0; t < T_out; t++) cx->is_common[t] = LOGICAL(sic)[t] ? 1 : 0;
                cx->rare_ptr[0] = 0;
                for (int k = 0; k < K; k++) cx->rare_ptr[k + 1] = cx->rare_ptr[k] + Rf_length(VECTOR_ELT(rph, k));
                cx->rare_snp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cx->rare_ptr[K] > 0 ? cx->rare_ptr[K] : 1));
            }
            if (!cx->is_common || !cx->rare_ptr || !cx->rare_snp) {
                st = QA_ERR_INVALID;
                snprintf(cx->msg, sizeof cx->msg, "out of memory preparing the rare + common inputs");
            }
            for (int k = 0; st == QA_OK && k < K; k++)
                if (cx->rare_ptr[k + 1] > cx->rare_ptr[k])
                    memcpy(cx->rare_snp + cx->rare_ptr[k], INTEGER(VECTOR_ELT(rph, k)), sizeof(int) * (size_t)(cx->rare_ptr[k + 1] - cx->rare_ptr[k]));
            for (; cx->n_rc < n_handles && st == QA_OK; cx->n_rc++) {
                cx->rcs[cx->n_rc] = NULL;
                st = qa_rare_common_create(cx->handles[cx->n_rc], T_out, cx->is_common, cx->rare_ptr, cx->rare_snp, REAL(tma), &cx->rcs[cx->n_rc]);
            }
            if (st != QA_OK && !cx->msg[0]) snprintf(cx->msg, sizeof cx->msg, "qa_rare_common_create: %s", qa_last_error());
            cx->rcq.handles = (const qa_rare_common_t *const *)cx->rcs;
            cx->rcq.nSNPs_all = T_out;
            cx->rcq.nGrids_all = (T_out + 31) / 32;
            cx->rcq.snp_is_common = cx->is_common;
            if (st == QA_OK && lga != R_NilValue && Rf_length(lga) == cx->rcq.nGrids_all) {
                cx->L_grid_all = (int32_t *)malloc(sizeof(int32_t) * (size_t)cx->rcq.nGrids_all);
                for (int g = 0; cx->L_grid_all && g < cx->rcq.nGrids_all; g++)
                    cx->L_grid_all[g] = TYPEOF(lga) == INTSXP ? INTEGER(lga)[g] : (int32_t)REAL(lga)[g];
                cx->rcq.L_grid_all = cx->L_grid_all;
            }
            ip->rare_common = &cx->rcq;
        }
    }
    /* method = "nipt" (functions.R:586, :1009-1016, :1218-1231): one fetal fraction per sample, the block definition's grid */
    SEXP methodSEXP = list_get(paramsSEXP, "method");
    cx->nipt = methodSEXP != R_NilValue && TYPEOF(methodSEXP) == STRSXP && Rf_length(methodSEXP) == 1 &&
               strcmp(CHAR(STRING_ELT(methodSEXP, 0)), "nipt") == 0;
    if (st == QA_OK && cx->nipt) {
        SEXP ff = list_get(paramsSEXP, "ff"), lg = list_get(panelSEXP, "L_grid");
        if (ff == R_NilValue || TYPEOF(ff) != REALSXP || Rf_length(ff) != n_sample || lg == R_NilValue || Rf_length(lg) != G) {
            st = QA_ERR_INVALID;
            snprintf(cx->msg, sizeof cx->msg, "method = \"nipt\": params$ff (one per sample, numeric) and panel_objects$L_grid (nGrids) are needed");
        } else {
            cx->L_grid = (int32_t *)malloc(sizeof(int32_t) * (size_t)G);
            if (!cx->L_grid) { st = QA_ERR_INVALID; snprintf(cx->msg, sizeof cx->msg, "out of memory"); }
            for (int g = 0; cx->L_grid && g < G; g++) cx->L_grid[g] = TYPEOF(lg) == INTSXP ? INTEGER(lg)[g] : (int32_t)REAL(lg)[g];
            cx->nq.ff = REAL(ff);
            cx->nq.L_grid = cx->L_grid;
            cx->nq.shuffle_bin_radius = (int)num_or(paramsSEXP, "shuffle_bin_radius", 5000);
            ip->nipt = &cx->nq;
        }
    }
    if (st != QA_OK) range_teardown(cx);
    return st;
}

/* sample_offset: ONE number (sample i of the call is global sample offset + i) or one 0-based global index PER SAMPLE -- the
 * form quilt-amd.R uses, so that a sample skipped for too few reads does not shift the streams of the samples behind it */
static int64_t *sample_index_of(SEXP sample_offsetSEXP, int n) {
    if (Rf_length(sample_offsetSEXP) != n || n <= 1) return NULL;   /* (n == 1: offset and index are the same thing) */
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int i = 0; idx && i < n; i++)
        idx[i] = TYPEOF(sample_offsetSEXP) == INTSXP ? (int64_t)INTEGER(sample_offsetSEXP)[i] : (int64_t)REAL(sample_offsetSEXP)[i];
    return idx;
}

SEXP qa_impute_sample_range(SEXP readsListSEXP, SEXP panelSEXP, SEXP paramsSEXP, SEXP sample_offsetSEXP, SEXP n_handlesSEXP,
                            SEXP allReadsListSEXP) {
    static const char *who = "qa_impute_sample_range";
    const int n = Rf_length(readsListSEXP);
    validate_reads_list(readsListSEXP, "list_of_sampleReads");
    if (allReadsListSEXP != R_NilValue) validate_reads_list(allReadsListSEXP, "list_of_allSNP_sampleReads");
    range_validate(panelSEXP, paramsSEXP, who);
    if ((TYPEOF(sample_offsetSEXP) != REALSXP && TYPEOF(sample_offsetSEXP) != INTSXP) ||
        (Rf_length(sample_offsetSEXP) != 1 && Rf_length(sample_offsetSEXP) != n))
        Rf_error("quilt_amd: %s: sample_offset must be one number or one global index per sample", who);
    if (flag(paramsSEXP, "impute_rare_common", 0) && (allReadsListSEXP == R_NilValue || Rf_length(allReadsListSEXP) != n))
        Rf_error("quilt_amd: %s: impute_rare_common needs one allSNP_sampleReads per sample", who);
    range_ctx_t cx;
    int st = range_setup(&cx, panelSEXP, paramsSEXP, Rf_asInteger(n_handlesSEXP), n);
    if (st != QA_OK) Rf_error("quilt_amd: %s: %s", who, cx.msg);
    const int T_out = cx.T_out, nL = cx.nipt ? 3 : 2;
    /* flatten the range's sampleReads */
    flat_reads_t fr, fa;
    memset(&fa, 0, sizeof fa);
    if (flatten_reads(readsListSEXP, &fr) != 0 || (cx.rare && flatten_reads(allReadsListSEXP, &fa) != 0)) {
        if (fr.read_off) flat_reads_free(&fr);
        range_teardown(&cx);
        Rf_error("quilt_amd: %s: out of memory flattening the sampleReads", who);
    }
    if (cx.rare) { cx.rcq.read_off = fa.read_off; cx.rcq.read_ptr = fa.read_ptr; cx.rcq.u = fa.u; cx.rcq.bq = fa.bq; cx.rcq.wif = fa.wif; }
    const int32_t *read_off = fr.read_off;
    const int totR = read_off[n];
    int n_prot = 0;
    SEXP fet_dosage = R_NilValue, fet_gp_t = R_NilValue;
    if (cx.nipt) {
        fet_dosage = PROTECT(Rf_allocMatrix(REALSXP, T_out, n));
        fet_gp_t = PROTECT(Rf_allocMatrix(REALSXP, 3 * T_out, n));
        n_prot = 2;
        cx.nq.fet_dosage = REAL(fet_dosage);
        cx.nq.fet_gp_t = REAL(fet_gp_t);
    }
    SEXP dosage = PROTECT(Rf_allocMatrix(REALSXP, T_out, n)), gp_t = PROTECT(Rf_allocMatrix(REALSXP, 3 * T_out, n));
    SEXP haps = PROTECT(Rf_allocMatrix(REALSXP, nL * T_out, n)), nDosage = PROTECT(Rf_allocVector(INTSXP, n));
    SEXP stats = PROTECT(Rf_allocVector(REALSXP, 11));
    int32_t *labels = (int32_t *)malloc(sizeof(int32_t) * (size_t)(totR > 0 ? totR : 1));
    int64_t *sidx = sample_index_of(sample_offsetSEXP, n);
    char msg[512];
    msg[0] = 0;
    if (!labels || (Rf_length(sample_offsetSEXP) == n && n > 1 && !sidx)) { st = QA_ERR_INVALID; snprintf(msg, sizeof msg, "out of memory"); }
    int64_t st64[11] = {0};
    if (st == QA_OK) {
        cx.ip.sample_index = sidx;
        st = qa_impute_samples(cx.handles, cx.n_handles, &cx.ip, n, sidx ? 0 : (int64_t)Rf_asReal(sample_offsetSEXP), read_off, fr.read_ptr, fr.u,
                               fr.bq, fr.wif, REAL(dosage), REAL(gp_t), REAL(haps), labels, INTEGER(nDosage), st64);
        if (st != QA_OK) snprintf(msg, sizeof msg, "qa_impute_samples: %s", qa_last_error());
    }
    range_teardown(&cx);
    free(sidx);
    if (fa.read_off) flat_reads_free(&fa);
    SEXP lab = PROTECT(Rf_allocVector(VECSXP, n));
    if (st == QA_OK)
        for (int i = 0; i < n; i++) {
            const int R = read_off[i + 1] - read_off[i];
            SEXP v = PROTECT(Rf_allocVector(INTSXP, R));
            memcpy(INTEGER(v), labels + read_off[i], sizeof(int) * (size_t)R);
            SET_VECTOR_ELT(lab, i, v);
            UNPROTECT(1);
        }
    flat_reads_free(&fr);
    free(labels);
    if (st != QA_OK) {
        UNPROTECT(6 + n_prot);
        Rf_error("quilt_amd: %s: %s", who, msg);
    }
    for (int i = 0; i < 11; i++) REAL(stats)[i] = (double)st64[i];
    const char *names[] = {"dosage", "gp_t", "phasing_haps", "read_labels", "nDosage", "stats", "fet_dosage", "fet_gp_t"};
    SEXP out = PROTECT(named_list(cx.nipt ? 8 : 6, names));
    SET_VECTOR_ELT(out, 0, dosage); SET_VECTOR_ELT(out, 1, gp_t); SET_VECTOR_ELT(out, 2, haps);
    SET_VECTOR_ELT(out, 3, lab); SET_VECTOR_ELT(out, 4, nDosage); SET_VECTOR_ELT(out, 5, stats);
    if (cx.nipt) { SET_VECTOR_ELT(out, 6, fet_dosage); SET_VECTOR_ELT(out, 7, fet_gp_t); }
    UNPROTECT(7 + n_prot);
    return out;
}

/* ---- qa_impute_bam_range: the same loop body from BAM PATHS to VCF COLUMNS, natively ------------------------------------------
 * qa_impute_sample_range above leaves both ends of get_and_impute_one_sample to R: STITCH's loadBamAndConvert + load() +
 * snap_sampleReads_to_grid per sample in front of the call (functions.R:243-298) and the per-sample column / count code behind
 * it (functions.R:1380-1463) -- about a second per sample on the R worker that owns a GPU which imputes ~40 per second.  This
 * routine is the whole body of the loop over a core's samples (quilt.R:832-982) behind one `.Call`:
 *     out <- .Call("qa_impute_bam_range", bam_files, sites, panel_objects, params, sample_index, n_handles)
 *   bam_files       character vector: the range's BAM files (bamlist order)
 *   sites           named list: chr; L (pos[, 2]), ref, alt (pos[, 3:4] as character vectors of single letters), grid (0-based);
 *                   impute_rare_common: L_all / ref_all / alt_all / grid_all (pos_all, special_rare_common_objects$grid);
 *                   loader options bqFilter, iSizeUpperLimit, useSoftClippedBases, downsampleToCov, chrStart, chrEnd (the
 *                   window of functions.R:262-263); minimum_number_of_sample_reads, output_gt_phased_genotypes, n_io_threads
 *   panel_objects, params, n_handles   as for qa_impute_sample_range (params$ff: one per FILE)
 *   sample_index    0-based global index of every file's sample (iSample - 1)
 * Returns list(sample_was_imputed (logical), n_reads (integer), per_sample_vcf_col (list: character vector per imputed sample,
 * NULL otherwise), read_labels (list), infoCount (nSNPs x 2), afCount, hweCount (nSNPs x 3), alleleCount (nSNPs x 2): the
 * range's sums in sample order as quilt.R:955-961 forms them; seconds (load, impute, format, total), stats).  The loader is
 * csrc/hostio.cpp's (include/quilt_amd_io.h says where it is unpinned against STITCH: CRAM is refused); quilt-amd.R calls this
 * routine only for the options it implements and falls back to the R loader otherwise. */
static const char *one_string(SEXP list, const char *name) {
    SEXP v = list_get(list, name);
    return (v != R_NilValue && TYPEOF(v) == STRSXP && Rf_length(v) == 1) ? CHAR(STRING_ELT(v, 0)) : NULL;
}
/* a character vector of single letters -> n bytes; NULL when an entry is not one letter (nothing allocated then) */
static char *letters_of(SEXP v, int n) {
    if (v == R_NilValue || TYPEOF(v) != STRSXP || Rf_length(v) != n) return NULL;
    for (int i = 0; i < n; i++) if (strlen(CHAR(STRING_ELT(v, i))) != 1) return NULL;
    char *out = (char *)malloc((size_t)(n > 0 ? n : 1));
    for (int i = 0; out && i < n; i++) out[i] = CHAR(STRING_ELT(v, i))[0];
    return out;
}
static void range_result_finalizer(SEXP p) {
    qa_bam_range_result_t *r = (qa_bam_range_result_t *)R_ExternalPtrAddr(p);
    if (r) qa_bam_range_destroy(r);
    R_ClearExternalPtr(p);
}
static int32_t *ints_of(SEXP v, int n) {
    if (v == R_NilValue || (TYPEOF(v) != INTSXP && TYPEOF(v) != REALSXP) || Rf_length(v) != n) return NULL;
    int32_t *out = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; out && i < n; i++) out[i] = TYPEOF(v) == INTSXP ? INTEGER(v)[i] : (int32_t)REAL(v)[i];
    return out;
}

SEXP qa_impute_bam_range_call(SEXP bamFilesSEXP, SEXP sitesSEXP, SEXP panelSEXP, SEXP paramsSEXP, SEXP sample_indexSEXP, SEXP n_handlesSEXP) {
    static const char *who = "qa_impute_bam_range";
    if (TYPEOF(bamFilesSEXP) != STRSXP) Rf_error("quilt_amd: %s: bam_files must be a character vector", who);
    const int n = Rf_length(bamFilesSEXP);
    if ((TYPEOF(sample_indexSEXP) != REALSXP && TYPEOF(sample_indexSEXP) != INTSXP) || Rf_length(sample_indexSEXP) != n)
        Rf_error("quilt_amd: %s: sample_index must hold one 0-based global index per file", who);
    range_validate(panelSEXP, paramsSEXP, who);
    const char *chr = one_string(sitesSEXP, "chr");
    SEXP Lx = list_get(sitesSEXP, "L");
    if (!chr || Lx == R_NilValue) Rf_error("quilt_amd: %s: sites needs chr, L, ref, alt, grid", who);
    const int T = Rf_length(Lx);
    if (T != Rf_ncols(list_get(panelSEXP, "distinctHapsIE"))) Rf_error("quilt_amd: %s: sites$L must have one entry per SNP of the panel", who);
    const int rare = flag(paramsSEXP, "impute_rare_common", 0);
    const int Ta = rare ? Rf_length(list_get(sitesSEXP, "L_all")) : 0;
    /* the sites as plain arrays (malloc: freed below on every path; R errors are raised only before or after) */
    int32_t *L = ints_of(Lx, T), *grid = ints_of(list_get(sitesSEXP, "grid"), T);
    char *ref = letters_of(list_get(sitesSEXP, "ref"), T), *alt = letters_of(list_get(sitesSEXP, "alt"), T);
    int32_t *La = rare ? ints_of(list_get(sitesSEXP, "L_all"), Ta) : NULL, *grida = rare ? ints_of(list_get(sitesSEXP, "grid_all"), Ta) : NULL;
    char *refa = rare ? letters_of(list_get(sitesSEXP, "ref_all"), Ta) : NULL, *alta = rare ? letters_of(list_get(sitesSEXP, "alt_all"), Ta) : NULL;
    const char **paths = (const char **)malloc(sizeof(char *) * (size_t)(n > 0 ? n : 1));
    int64_t *sidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    const int sites_ok = L && grid && ref && alt && paths && sidx && (!rare || (La && grida && refa && alta));
    range_ctx_t cx;
    memset(&cx, 0, sizeof cx);
    int st = QA_OK;
    char msg[512];
    msg[0] = 0;
    if (!sites_ok) {
        st = QA_ERR_INVALID;
        snprintf(msg, sizeof msg, "sites: L / grid numeric, ref / alt character vectors of single letters, all of the panel's length%s",
                 rare ? " (and L_all / grid_all / ref_all / alt_all over all SNPs)" : "");
    }
    if (st == QA_OK) {
        st = range_setup(&cx, panelSEXP, paramsSEXP, Rf_asInteger(n_handlesSEXP), n);
        if (st != QA_OK) snprintf(msg, sizeof msg, "%s", cx.msg);
    }
    qa_bam_range_result_t *res = NULL;
    if (st == QA_OK) {
        for (int i = 0; i < n; i++) {
            paths[i] = CHAR(STRING_ELT(bamFilesSEXP, i));
            sidx[i] = TYPEOF(sample_indexSEXP) == INTSXP ? (int64_t)INTEGER(sample_indexSEXP)[i] : (int64_t)REAL(sample_indexSEXP)[i];
        }
        qa_bam_range_io_t io;
        memset(&io, 0, sizeof io);
        io.chr = chr; io.nSNPs = T; io.L = L; io.ref = ref; io.alt = alt; io.grid = grid;
        io.nSNPs_all = Ta; io.L_all = La; io.ref_all = refa; io.alt_all = alta; io.grid_all = grida;
        qa_bam_opts_default(&io.bam);
        io.bam.bqFilter = (int)num_or(sitesSEXP, "bqFilter", io.bam.bqFilter);
        {
            const double isz = num_or(sitesSEXP, "iSizeUpperLimit", io.bam.iSizeUpperLimit);
            io.bam.iSizeUpperLimit = isz > 2147483647.0 ? 2147483647 : (int)isz;
        }
        io.bam.useSoftClippedBases = flag(sitesSEXP, "useSoftClippedBases", 0);
        io.bam.downsampleToCov = (int)num_or(sitesSEXP, "downsampleToCov", io.bam.downsampleToCov);
        io.bam.chrStart = (int)num_or(sitesSEXP, "chrStart", 0);
        io.bam.chrEnd = (int)num_or(sitesSEXP, "chrEnd", 0);
        io.minimum_number_of_sample_reads = (int)num_or(sitesSEXP, "minimum_number_of_sample_reads", 2);
        io.output_gt_phased_genotypes = flag(sitesSEXP, "output_gt_phased_genotypes", 1);
        io.n_io_threads = (int)num_or(sitesSEXP, "n_io_threads", 0);
        io.discard_sample_arrays = 1;   /* (columns, labels and counts go back to R: the per-SNP numbers behind them are not kept) */
        st = qa_impute_bam_range(cx.handles, cx.n_handles, &cx.ip, &io, n, paths, sidx, cx.nipt ? cx.nq.ff : NULL, &res);
        if (st != QA_OK) snprintf(msg, sizeof msg, "%s", qa_last_error());
        range_teardown(&cx);
    }
    free(L); free(grid); free(ref); free(alt); free(La); free(grida); free(refa); free(alta); free((void *)paths); free(sidx);
    if (st != QA_OK) {
        if (res) qa_bam_range_destroy(res);
        Rf_error("quilt_amd: %s: %s", who, msg);
    }
    /* results as R objects.  (An allocation failure inside R longjmps past the end of this routine: the result is handed to an
     * external pointer with a finalizer first, so that the collector frees it in that case; it is freed explicitly below otherwise.) */
    SEXP guard = PROTECT(R_MakeExternalPtr(res, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(guard, range_result_finalizer, TRUE);
    const int T_out = qa_bam_range_n_snps(res);
    const char *names[] = {"sample_was_imputed", "n_reads", "per_sample_vcf_col", "read_labels", "infoCount", "afCount", "hweCount",
                           "alleleCount", "seconds", "stats"};
    SEXP out = PROTECT(named_list(10, names));
    SEXP imputed = PROTECT(Rf_allocVector(LGLSXP, n)), n_reads = PROTECT(Rf_allocVector(INTSXP, n));
    SEXP cols = PROTECT(Rf_allocVector(VECSXP, n)), labs = PROTECT(Rf_allocVector(VECSXP, n));
    SET_VECTOR_ELT(out, 0, imputed); SET_VECTOR_ELT(out, 1, n_reads); SET_VECTOR_ELT(out, 2, cols); SET_VECTOR_ELT(out, 3, labs);
    for (int i = 0; i < n; i++) {
        LOGICAL(imputed)[i] = qa_bam_range_imputed(res, i);
        INTEGER(n_reads)[i] = qa_bam_range_n_reads(res, i);
        const char *buf = NULL;
        const int64_t *off = NULL;
        qa_bam_range_column(res, i, &buf, &off);
        if (!buf) continue;
        SEXP col = PROTECT(Rf_allocVector(STRSXP, T_out));
        for (int t = 0; t < T_out; t++) SET_STRING_ELT(col, t, Rf_mkChar(buf + off[t]));
        SET_VECTOR_ELT(cols, i, col);
        UNPROTECT(1);
        const int32_t *rl = NULL;
        int32_t nl = 0;
        qa_bam_range_sample(res, i, NULL, NULL, NULL, NULL, NULL, &rl, &nl, NULL);
        SEXP v = PROTECT(Rf_allocVector(INTSXP, nl));
        if (nl > 0) memcpy(INTEGER(v), rl, sizeof(int) * (size_t)nl);
        SET_VECTOR_ELT(labs, i, v);
        UNPROTECT(1);
    }
    SEXP info = PROTECT(Rf_allocMatrix(REALSXP, T_out, 2)), af = PROTECT(Rf_allocVector(REALSXP, T_out));
    SEXP hwe = PROTECT(Rf_allocMatrix(REALSXP, T_out, 3)), ac = PROTECT(Rf_allocMatrix(REALSXP, T_out, 2));
    qa_bam_range_counts(res, REAL(info), REAL(af), REAL(hwe), REAL(ac));
    SET_VECTOR_ELT(out, 4, info); SET_VECTOR_ELT(out, 5, af); SET_VECTOR_ELT(out, 6, hwe); SET_VECTOR_ELT(out, 7, ac);
    SEXP sec = PROTECT(Rf_allocVector(REALSXP, 4)), stats = PROTECT(Rf_allocVector(REALSXP, 11));
    int64_t st64[11] = {0};
    qa_bam_range_timings(res, REAL(sec), st64, NULL);
    for (int i = 0; i < 11; i++) REAL(stats)[i] = (double)st64[i];
    SET_VECTOR_ELT(out, 8, sec); SET_VECTOR_ELT(out, 9, stats);
    range_result_finalizer(guard);
    UNPROTECT(12);
    return out;
}

/* ---- registration (RcppExports.cpp:1703-1782) ----------------------------------------------------------------------
 * Inside QUILT.so (shim/QUILT-src.patch): RcppExports.cpp's own CallEntries[] names these functions in the four rows, and
 * R_init_QUILT registers them.  As a DLL of its own: R_init_quilt_amd_shim. */

static const R_CallMethodDef CallEntries[] = {
    {"_QUILT_rcpp_forwardBackwardGibbsNIPT", (DL_FUNC)&qa_QUILT_rcpp_forwardBackwardGibbsNIPT, 63},
    {"_QUILT_Rcpp_haploid_dosage_versus_refs", (DL_FUNC)&qa_QUILT_Rcpp_haploid_dosage_versus_refs, 38},
    {"_QUILT_Rcpp_make_gl_bound", (DL_FUNC)&qa_QUILT_Rcpp_make_gl_bound, 3},
    {"_QUILT_rcpp_make_eMatRead_t", (DL_FUNC)&qa_QUILT_rcpp_make_eMatRead_t, 15},
    {"qa_shim_release", (DL_FUNC)&qa_shim_release, 0},
    {"qa_impute_sample_range", (DL_FUNC)&qa_impute_sample_range, 6},
    {"qa_impute_bam_range", (DL_FUNC)&qa_impute_bam_range_call, 6},
    {NULL, NULL, 0}};

void R_init_quilt_amd_shim(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
