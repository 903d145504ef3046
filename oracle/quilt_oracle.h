/*
 * quilt_oracle.h -- CPU oracle for the QUILT hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C fp64 restatement of the reference's Rcpp kernels, written
 * from reading the reference; every function cites the reference file:line it
 * follows.  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg have something to check the HIP path against.  Nothing in
 * the product path (quilt_amd/) may import, link or call it.
 *
 * PARITY PINNING: the reference (R + Rcpp + RcppArmadillo/RcppEigen + STITCH)
 * cannot be built or run in this image and stores no golden vectors.  The
 * oracle is pinned against (i) the RNG-independent known-answer tests the
 * reference's testthat suite holds (binary searches, quantile, H_class
 * log-probabilities, label / class swap table, make_gibbs_considers' defining
 * properties, gl bounding rule, top-K picker definition; not the rlcM table of
 * block_approach 4, which production does not use) and (ii) the structural
 * invariants those tests assert, plus, for the
 * rare + common forms, equality with the dense computation on haplotypes
 * expanded over all SNPs.  For the bulk forward/backward and Gibbs
 * arithmetic that is all that exists: "parity unpinned" beyond those.
 *
 * All matrices are column-major (R layout).  Indices are 0-based unless a
 * parameter name ends in "_1based".
 */
#ifndef QUILT_ORACLE_H
#define QUILT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Armadillo's sum() ------------------------------------------------------
 *
 * Wherever the reference writes sum(x) on a dense Armadillo vector (arma::colvec / arma::rowvec, a .col() view of an
 * arma::mat) or on an element-wise expression of such vectors (x % y, x / y, log(x)), Armadillo does NOT add left to right.
 * For a vector argument sum() is accu() (armadillo_bits/fn_sum.hpp: "resolves_to_vector" -> accu(X)), and
 *   - accu() of a Mat / subview_col is arrayops::accumulate(mem, n_elem) (armadillo_bits/arrayops_meat.hpp),
 *   - accu() of an element-wise expression is accu_proxy_linear(P)            (armadillo_bits/fn_accu.hpp),
 * both of which, when the translation unit is NOT compiled with -ffast-math (__FINITE_MATH_ONLY__ unset: QUILT/src/Makevars
 * has plain -O3) and without OpenMP (Makevars has no -fopenmp, so arma_config::openmp is false), run TWO accumulators:
 *
 *     acc1 = acc2 = 0;
 *     for (i = 0, j = 1; j < n; i += 2, j += 2) { acc1 += x[i]; acc2 += x[j]; }
 *     if (i < n) acc1 += x[i];                 // odd tail
 *     return acc1 + acc2;
 *
 * i.e. the even-indexed elements are added in order into one chain, the odd-indexed ones into another, and the two chains
 * meet once.  The compiler may not re-associate either chain, so this IS the arithmetic of those call sites.  Armadillo is
 * not in this image: the form above is restated from the library's published source (the routine has had this shape since
 * Armadillo 3.x through 14.x), not observed by running it.  Because that cannot be checked here, the left-to-right form
 * that rounds 1-5 of this oracle used stays available behind qo_set_sum_order(1); a maintainer with R can tell which is
 * right from ONE number printed at full precision -- c(0) of Rcpp_haploid_dosage_versus_refs (reference-single.cpp:2347)
 * for any gl with K >= 3: print it with sprintf("%a") and compare with the oracle's two settings.
 *
 * Sites restated with QO_ARMA_SUM (every other accumulation of the reference on this path is an explicit C++ loop --
 * run_total / sum_e_times_b / matched_gammas / minus_log_c*_sum / logC_inside, or Rcpp sugar's sum() on a NumericVector,
 * e.g. choice_probs -- and stays left to right):
 *   reference-single.cpp:2347                                             sum(alphaHat_t_col), grid 0 of the full pass
 *   copied-from-stitch.cpp:367, 383, 405, 432, 435                        haploid forward / backward
 *   gibbs-nipt.cpp:644, 655, 688, 700, 724, 854-857, 909-912, 971-974, 1270-1287   the sampler's K-wide sums
 *   gibbs-nipt-block.cpp:353-362, 669, 915-920, 1226, 1245, 1819-1821, 2123-2131, 2235-2238   block / shard passes
 * The R twins (oracle/rtwin.py) restate R code, whose sum() is neither: R accumulates left to right in a long double
 * (src/main/summary.c: rsum, LDOUBLE) and rounds once at the end; rtwin.py says so where it sums.
 */
void qo_set_sum_order(int left_to_right);   /* 0 (default): Armadillo's two accumulators; 1: left to right (rounds 1-5) */
int qo_get_sum_order(void);
extern int qo_sum_left_to_right;

/* out = sum over I = 0 .. n-1 of EXPR (an expression in I), in the order Armadillo's accu() adds it */
#define QO_ARMA_SUM(out, n, I, EXPR)                                                   \
    do {                                                                               \
        double qo_acc1_ = 0, qo_acc2_ = 0;                                             \
        int I = 0;                                                                     \
        const int qo_n_ = (n);                                                         \
        if (qo_sum_left_to_right) {                                                    \
            for (; I < qo_n_; I++) qo_acc1_ += (EXPR);                                 \
        } else {                                                                       \
            while (I + 1 < qo_n_) {                                                    \
                qo_acc1_ += (EXPR);                                                    \
                I++;                                                                   \
                qo_acc2_ += (EXPR);                                                    \
                I++;                                                                   \
            }                                                                          \
            if (I < qo_n_) qo_acc1_ += (EXPR);                                         \
        }                                                                              \
        (out) = qo_acc1_ + qo_acc2_;                                                   \
    } while (0)

/* ---- small helpers ------------------------------------------------------ */

/* reference-single.cpp:68-94 */
void qo_make_gl_bound(double *gl, double minGLValue, const int *to_fix, int n_to_fix);

/* reference-single.R:19-42 (+ STITCH convertScaledBQtoProbs, restated from
 * copied-from-stitch.cpp:166-175).  u is 0-based here. */
void qo_make_gl_from_u_bq(const int *u, const int *bq, int n, int nSNPs,
                          double minGLValue, double *gl /* 2 x nSNPs */);

/* reference-single.cpp:272-329 */
void qo_build_eMatDH(const int32_t *distinctHapsB, const double *gl, int nMaxDH,
                     int nGrids, int nSNPs, double ref_error, int add_zero_row,
                     double *eMatDH /* (nMaxDH + add_zero_row) x nGrids */);

/* gibbs-small.cpp:26-59 ; returns 0-based position */
int qo_simple_binary_search(int val, const int32_t *vec, int n);

/* gibbs-small.cpp:69-105 ; mat is nrow x 2 column-major, s1/e1 1-based rows */
int qo_simple_binary_matrix_search(int val, const int32_t *mat, int nrow, int s1, int e1);

/* reference-single.cpp:100-108 */
void qo_nth_partial_sort(const double *x, int n, int nth, double *y);

/* reference-single.cpp:129-194.  Returns the number of matches written.
 * top_idx/top_val need capacity K. */
int qo_get_top_K_or_more_matches_while_building_gamma(
    const double *alpha_col, const double *beta_col, double *gamma_col, int K,
    int K_top_matches, double special_multiplication_value, int32_t *top_idx,
    double *top_val);

/* ---- full-panel haploid forward/backward -------------------------------- */

typedef struct {
    /* panel (read-only) */
    int K, nGrids, nSNPs, nMaxDH;
    const int32_t *rhb_t;            /* K x nGrids, may be NULL when special symbols are used */
    const int32_t *hapMatcher;       /* K x nGrids int32, or NULL */
    const uint8_t *hapMatcherR;      /* K x nGrids uint8, or NULL */
    const int32_t *distinctHapsB;    /* nMaxDH x nGrids */
    const double *distinctHapsIE;    /* nMaxDH x nSNPs */
    const int32_t *eMatDH_special_grid_which;      /* nGrids ; 0 = none, else 1-based list id */
    const int32_t *special_values_ptr;             /* CSR offsets over lists (n_lists + 1) */
    const int32_t *special_values;                 /* concatenated 0-based k */
    const int32_t *eMatDH_special_matrix_helper;   /* nGrids x 2, 1-based first/last row */
    const int32_t *eMatDH_special_matrix;          /* nrow x 2: col0 = k (0-based), col1 = word */
    int eMatDH_special_matrix_nrow;
    int use_eMatDH_special_symbols;
    const double *transMatRate_t;    /* 2 x (nGrids-1) */
    double ref_error;
} qo_panel_t;

typedef struct {
    int K_top_matches;
    double min_emission_prob_normalization_threshold;
    int return_betaHat_t, return_dosage, return_gamma_t, return_gammaSmall_t;
    int get_best_haps_from_thinned_sites;
    int always_normalize, normalize_emissions;
} qo_fullpass_opts_t;

/*
 * reference-single.cpp:2189-2413 (orchestration), :878-1131 (forward, v3 ==
 * v2 arithmetic), :1781-2179 (backward, v3 == v2 arithmetic), use_eMatDH=TRUE.
 *
 * alphaHat_t: K x nGrids (always full size; when only_store_alpha_at_gamma_small
 *             applies, only column 0 and the thinned columns are written, as in
 *             the reference).
 * best_ptr (n_thin + 1), best_idx / best_val (capacity best_cap): CSR form of
 *             best_haps_stuff_list, entry i = thinned column i.
 * Returns 0, or -1 if best_cap was too small (best_ptr still holds the sizes).
 */
int qo_haploid_dosage_versus_refs(
    const qo_panel_t *p, const qo_fullpass_opts_t *o, const double *gl /* 2 x nSNPs */,
    const int32_t *gammaSmall_cols_to_get /* nGrids, -1 or 0-based col */,
    double *alphaHat_t, double *betaHat_t, double *c, double *gamma_t,
    double *gammaSmall_t /* K x n_thin */, double *dosage /* nSNPs */,
    int32_t *best_ptr, int32_t *best_idx, double *best_val, int64_t best_cap);


/* ---- small-panel Gibbs read-label sampler -------------------------------- */

/* gibbs-small.cpp:116-265.  eMatRead_t (Ks x nReads) must be pre-filled (normally 1). */
void qo_make_eMatRead_t_for_gibbs_using_objects(
    const qo_panel_t *p, const int32_t *which_haps_to_use_1based, int Ks, int nReads,
    const int32_t *read_ptr, const int32_t *u, const int32_t *bq, int rescale_eMatRead_t, int Jmax,
    double maxDifferenceBetweenReads, double *eMatRead_t);

/* gibbs-nipt.cpp:338-382 */
void qo_evaluate_read_variability(const double *eMatRead_t, int Ks, int nReads,
                                  int32_t *number_of_non_1_reads, int32_t *indices_of_non_1_reads,
                                  int32_t *read_category);

/* The "special_rare_common_objects" a rare/common call needs (prepare_reference_functions.R:172-247,
 * rare_common.R:222-223): SNPs of the panel tables are the common ones; rare SNPs are held per haplotype. */
typedef struct qo_rare_common {
    int nSNPs_all, nGrids_all;
    const uint8_t *snp_is_common;       /* nSNPs_all */
    const int32_t *common_snp_index;    /* nSNPs_all: 1-based index among the common SNPs, 0 for a rare SNP */
    const int64_t *rare_ptr;            /* K + 1: CSR over rare_per_hap_info */
    const int32_t *rare_snp_1based;     /* all-SNP 1-based indices of the rare SNPs each haplotype carries the alt of */
    const double *transMatRate_t_all;   /* 2 x (nGrids_all - 1) */
} qo_rare_common_t;

/* gibbs-small.cpp:270-460 (Rcpp_make_eMatRead_t_for_final_rare_common_gibbs_using_objects); rare_per_snp_info is
 * built from rare_per_hap_info and which_haps_to_use as rare_common.R:313-322 does. */
void qo_make_eMatRead_t_rare_common(
    const qo_panel_t *p, const qo_rare_common_t *rc, const int32_t *which_haps_to_use_1based, int Ks, int nReads,
    const int32_t *read_ptr, const int32_t *u, const int32_t *bq, int rescale_eMatRead_t, int Jmax,
    double maxDifferenceBetweenReads, double *eMatRead_t);

typedef struct {
    int Ks, nReads;
    const int32_t *which_haps_to_use_1based; /* Ks */
    const int32_t *read_ptr, *u, *bq, *wif;  /* flattened sampleReads (a0 in SURVEY.md 8(a)) */
    const uint8_t *grid_has_read;            /* nGrids */
    double ff;
    int Jmax;
    double maxDifferenceBetweenReads;
    int n_gibbs_burn_in_its, n_gibbs_sample_its;
    const int32_t *block_gibbs_iterations;
    int n_block_gibbs_iterations;
    int perform_block_gibbs, do_shard_block_gibbs;
    int gibbs_initialize_iteratively, sample_is_diploid, disable_read_category_usage, rescale_eMatRead_t;
    double class_sum_cutoff;
    /* the uniforms the reference draws from R's RNG, as inputs */
    const double *runif_reads;   /* nReads * n_its   (gibbs-nipt.cpp:2845) */
    int first_read;              /* 0-based          (gibbs-nipt.cpp:2846-2848) */
    const double *runif_shard;   /* n_block_its * (nGrids - 1)  (gibbs-nipt-block.cpp:2054) */
    /* NULL, or the final all-SNP ("rare + common") Gibbs of QUILT2 (make_eMatRead_t_rare_common = TRUE,
     * rare_common.R:109-420): the sampler then runs on rc->nGrids_all grids of rc->nSNPs_all SNPs, reads index
     * all SNPs, and hapProbs_t / genProbs*_t are 3 x rc->nSNPs_all. */
    const struct qo_rare_common *rc;
    /* NIPT block Gibbs (ff > 0 with perform_block_gibbs): what Rcpp_define_blocked_snps_using_gamma_on_the_fly needs
     * beyond the state (gibbs-nipt.cpp:3008), and the uniforms of every block pass (gibbs-nipt.cpp:3016,
     * gibbs-nipt-block.cpp:226-243), n_block_gibbs_iterations x nReads each */
    const int32_t *L_grid;
    int shuffle_bin_radius;
    double block_gibbs_quantile_prob;
    const double *runif_block, *runif_resample;
    /* NULL, or the block passes' uniforms as ONE stream in the order the reference consumes R's generator at a block iteration:
     * nReads of runif_block (gibbs-nipt.cpp:3016), then one uniform for every read whose class leaves a choice, in read order
     * (Rcpp::sample(1:3, 1, prob) inside rcpp_sample_H_using_H_class, gibbs-nipt-block.cpp:213-246) -- a count that depends on
     * the pass's own result.  runif_block / runif_resample are then not read; *runif_stream_used comes back with the number
     * consumed.  (The reference's unused draws around them -- runif_proposed, runif_total -- are the caller's to skip.) */
    const double *runif_stream;
    int64_t *runif_stream_used;
} qo_gibbs_args_t;

/* pieces of the NIPT block Gibbs with known answers in the reference's tests (test-unit-gibbs-block-nipt.R) */
double qo_simple_quantile(const double *x, int n, double q);
void qo_make_smoothed_rate(const double *sigma_rate, const int32_t *L_grid, int nGrids, int shuffle_bin_radius,
                           double *smoothed_rate);
void qo_define_blocked_grids(const double *rate2, const int32_t *L_grid, int nGrids, int shuffle_bin_radius,
                             double block_gibbs_quantile_prob, int32_t *blocked_grid);
int qo_make_gibbs_considers(const int32_t *blocked_grid, int nGrids, const int32_t *wif0, int nReads,
                            int32_t *grid_start, int32_t *grid_end, int32_t *reads_start, int32_t *reads_end,
                            int32_t *grid_where);
double qo_get_log_p_H_class2(int n1, int n2, int n3, int n4, int n5, int n6, double ff);
void qo_zero_based_swap(int ir_chosen, int swap[8]);
int qo_sample3(const double probs[3], double u);

/* rcpp_forwardBackwardGibbsNIPT (gibbs-nipt.cpp:2395-3307), production argument values.
 * H (1-based labels) is updated in place; alphaHat_t/betaHat_t/eMatGrid_t are Ks x nGrids each,
 * c_out nGrids each, eMatRead_t Ks x nReads (output), hapProbs_t / genProbs* 3 x nSNPs.
 * Returns 0, 1 (underflow_problem) or -2 (unsupported mode). */
int qo_gibbs(const qo_panel_t *p, const qo_gibbs_args_t *a, int32_t *H, int32_t *H_class,
             double *alphaHat_t[3], double *betaHat_t[3], double *eMatGrid_t[3], double *c_out[3],
             double *eMatRead_t, int32_t *read_category_out, double *hapProbs_t, double *genProbsM_t,
             double *genProbsF_t);

#ifdef __cplusplus
}
#endif
#endif
