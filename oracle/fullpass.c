/*
 * oracle/fullpass.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see quilt_oracle.h).
 *
 * fp64 restatement of the full-panel haploid Li-Stephens forward/backward of
 * QUILT/src/reference-single.cpp, in the reference's arithmetic order
 * (version 3 == version 2 arithmetic), use_eMatDH = TRUE.
 *
 * ORDER OF THE K-WIDE SUMS, site by site.  run_total (:1002-1075), sum_e_times_b (:1899-1955), matched_gammas and the
 * dosage sums (:2083-2139) are explicit C++ loops in the reference: the special haplotypes in list order, then
 * k = 0 .. K-1, left to right -- restated as such.  c(0) = 1 / sum(alphaHat_t_col) (:2347) is Armadillo's sum() of an
 * arma::colvec, i.e. arrayops::accumulate: even-indexed elements into one accumulator, odd-indexed ones into a second,
 * acc1 + acc2 at the end (quilt_oracle.h says which Armadillo routine, why that cannot be observed in this image, and how
 * a maintainer with R settles it from one printed c(0)).  Rounds 1-5 added that sum left to right as well and called it
 * "the reference's order": it was this file's order.  qo_set_sum_order(1) restores it.
 *
 * PIN: the reference cannot be built or run in this container (no R, Rcpp, Armadillo, Eigen) and its tests hold no golden
 * vectors, so this file is pinned by (1) oracle/rtwin.py -- an independent NumPy restatement of the reference's R twin of
 * this code (QUILT/R/reference-single.R:94-372), cross-checked by tests/golden/make_golden_rtwin.py and on every CPU run by
 * tests/test_rtwin_cpu.py against fixtures generated from the R-twin side (dosage 1e-12, gamma 1e-9, best-haplotype lists
 * identical); (2) the RNG-independent known answers of the reference's testthat suite (tests/golden/known_answers.json);
 * (3) the invariants that suite asserts (tests/test_oracle_cpu.py).  Not an execution of the reference: parity claims
 * resting on this oracle are "unpinned against a reference run" (DESIGN.md section 3).
 */
#include "quilt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* reference-single.cpp:68-94: rescale a (ref, alt) likelihood pair so that
 * its larger member is 1 and its smaller member is at least minGLValue. */
int qo_sum_left_to_right = 0;
void qo_set_sum_order(int left_to_right) { qo_sum_left_to_right = left_to_right ? 1 : 0; }
int qo_get_sum_order(void) { return qo_sum_left_to_right; }

void qo_make_gl_bound(double *gl, double minGLValue, const int *to_fix, int n_to_fix)
{
    for (int i = 0; i < n_to_fix; i++) {
        double *p = gl + 2 * (size_t)to_fix[i];
        double a = p[0], b = p[1];
        if (a > b) {
            b = b / a;
            a = 1;
            if (b < minGLValue) b = minGLValue;
        } else {
            a = a / b;
            b = 1;
            if (a < minGLValue) a = minGLValue;
        }
        p[0] = a;
        p[1] = b;
    }
}

/* reference-single.R:19-42.  Base-quality convention restated from
 * copied-from-stitch.cpp:166-175 / gibbs-small.cpp:172-181: bq < 0 means the
 * read shows REF (pR = 1-eps, pA = eps/3), bq > 0 means ALT. */
void qo_make_gl_from_u_bq(const int *u, const int *bq, int n, int nSNPs,
                          double minGLValue, double *gl)
{
    for (int i = 0; i < 2 * nSNPs; i++) gl[i] = 1.0;
    if (n == 0) return;
    for (int i = 0; i < n; i++) {
        double pR, pA;
        if (bq[i] < 0) {
            double eps = pow(10.0, (double)bq[i] / 10.0);
            pR = 1 - eps;
            pA = eps / 3;
        } else if (bq[i] > 0) {
            double eps = pow(10.0, -(double)bq[i] / 10.0);
            pR = eps / 3;
            pA = 1 - eps;
        } else {
            /* the caller filters bq == 0 (functions.R:2018-2020) */
            continue;
        }
        gl[2 * (size_t)u[i] + 0] *= pR;
        gl[2 * (size_t)u[i] + 1] *= pA;
    }
    if (minGLValue > 0) {
        for (int s = 0; s < nSNPs; s++) {
            if (gl[2 * (size_t)s] < minGLValue || gl[2 * (size_t)s + 1] < minGLValue) {
                int idx = s;
                qo_make_gl_bound(gl, minGLValue, &idx, 1);
            }
        }
    }
}

/* emission of one 32-SNP word against a slice of gl (reference-single.cpp:307-322) */
static double word_emission(uint32_t w, const double *gl_local, int nSNPsLocal,
                            double ref_error, double ref_one_minus_error)
{
    double prob = 1;
    for (int b = 0; b < nSNPsLocal; b++) {
        double dR = gl_local[2 * b], dA = gl_local[2 * b + 1];
        if (w & (1u << b)) {
            prob *= (dR * ref_error + dA * ref_one_minus_error);
        } else {
            prob *= (dR * ref_one_minus_error + dA * ref_error);
        }
    }
    return prob;
}

/* reference-single.cpp:272-329 */
void qo_build_eMatDH(const int32_t *distinctHapsB, const double *gl, int nMaxDH,
                     int nGrids, int nSNPs, double ref_error, int add_zero_row,
                     double *eMatDH)
{
    const double ref_one_minus_error = 1 - ref_error;
    const int kbump = add_zero_row ? 1 : 0;
    const int nrow = nMaxDH + kbump;
    for (int g = 0; g < nGrids; g++) {
        int s = 32 * g, e = 32 * (g + 1) - 1;
        if (e > nSNPs - 1) e = nSNPs - 1;
        int nLocal = e - s + 1;
        double *col = eMatDH + (size_t)nrow * g;
        if (add_zero_row) col[0] = 1;
        for (int k = 0; k < nMaxDH; k++) {
            uint32_t w = (uint32_t)distinctHapsB[(size_t)nMaxDH * g + k];
            col[kbump + k] = word_emission(w, gl + 2 * (size_t)s, nLocal, ref_error, ref_one_minus_error);
        }
        if (add_zero_row) {
            double m = col[0]; /* == 1 at this point, part of the min as in arma::min(col) */
            for (int k = 1; k < nrow; k++) if (col[k] < m) m = col[k];
            col[0] = m;
        }
    }
}

/* gibbs-small.cpp:26-59 */
int qo_simple_binary_search(int val, const int32_t *vec, int nori)
{
    if (nori == 1) return 0;
    int n = nori;
    int i = n / 2;
    n = n / 4;
    for (;;) {
        if (vec[i] == val) return i;
        if (vec[i] < val) i += n; else i -= n;
        n = n / 2;
        if (n < 1) n = 1;
        if (i < 0) i = 0;
        if (i > nori - 1) i = nori - 1;
    }
}

/* gibbs-small.cpp:69-105.  Quirks kept: a one-row range returns the integer
 * 0 rather than the stored word; gives up after 100 probes and returns
 * mat(s1, 1). */
int qo_simple_binary_matrix_search(int val, const int32_t *mat, int nrow, int s1, int e1)
{
    int nori = e1 - s1 + 1;
    if (nori == 1) return 0;
    int n = nori;
    int i = n / 2;
    n = n / 4;
    int c = 0;
    while (c < 100) {
        c++;
        int32_t key = mat[s1 - 1 + i];
        if (key == val) return mat[(size_t)nrow + s1 - 1 + i];
        if (key < val) i += n; else i -= n;
        n = n / 2;
        if (n < 1) n = 1;
        if (i < 0) i = 0;
        if (i > nori - 1) i = nori - 1;
    }
    return mat[(size_t)nrow + s1];
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* reference-single.cpp:100-108: the first nth entries of y are the nth
 * smallest of x in ascending order; the rest is unspecified (here: sorted). */
void qo_nth_partial_sort(const double *x, int n, int nth, double *y)
{
    (void)nth;
    memcpy(y, x, sizeof(double) * (size_t)n);
    qsort(y, (size_t)n, sizeof(double), cmp_double);
}

/* reference-single.cpp:129-194 */
int qo_get_top_K_or_more_matches_while_building_gamma(
    const double *alpha_col, const double *beta_col, double *gamma_col, int K,
    int K_top_matches, double special_multiplication_value, int32_t *top_idx,
    double *top_val)
{
    double *top = (double *)calloc((size_t)K_top_matches, sizeof(double)); /* ascending */
    for (int k = 0; k < K; k++) {
        double g = alpha_col[k] * beta_col[k];
        gamma_col[k] = g;
        if (g == top[0]) {
            /* counted only */
        } else if (g > top[0]) {
            int beats = 0;
            for (int j = 0; j < K_top_matches; j++) if (g > top[j]) beats = j;
            for (int i = 0; i < beats; i++) top[i] = top[i + 1];
            top[beats] = g;
        }
    }
    int count = 0;
    for (int k = 0; k < K; k++) {
        if (gamma_col[k] >= top[0]) {
            top_idx[count] = k;
            top_val[count] = gamma_col[k] * special_multiplication_value;
            count++;
        }
    }
    free(top);
    return count;
}

/* ------------------------------------------------------------------------ */

static inline int dh_at(const qo_panel_t *p, int k, int g)
{
    if (p->hapMatcherR) return p->hapMatcherR[(size_t)p->K * g + k];
    return p->hapMatcher[(size_t)p->K * g + k];
}

/* list of special haplotypes at grid g (reference-single.cpp:1002-1014) */
static int special_list(const qo_panel_t *p, int g, const int32_t **list)
{
    int which = p->eMatDH_special_grid_which[g];
    if (which <= 0) { *list = NULL; return 0; }
    if (p->use_eMatDH_special_symbols) {
        int s1 = p->eMatDH_special_matrix_helper[g];
        int e1 = p->eMatDH_special_matrix_helper[(size_t)p->nGrids + g];
        *list = p->eMatDH_special_matrix + (s1 - 1);
        return e1 - s1 + 1;
    }
    *list = p->special_values + p->special_values_ptr[which - 1];
    return p->special_values_ptr[which] - p->special_values_ptr[which - 1];
}

static uint32_t special_word(const qo_panel_t *p, int k, int g)
{
    if (p->use_eMatDH_special_symbols) {
        int s1 = p->eMatDH_special_matrix_helper[g];
        int e1 = p->eMatDH_special_matrix_helper[(size_t)p->nGrids + g];
        return (uint32_t)qo_simple_binary_matrix_search(k, p->eMatDH_special_matrix,
                                                        p->eMatDH_special_matrix_nrow, s1, e1);
    }
    return (uint32_t)p->rhb_t[(size_t)p->K * g + k];
}

static void grid_bounds(int g, int nSNPs, int *s, int *nLocal)
{
    int e = 32 * (g + 1) - 1;
    *s = 32 * g;
    if (e > nSNPs - 1) e = nSNPs - 1;
    *nLocal = e - *s + 1;
}

static int grid_has_variant(const double *gl, int s, int nLocal)
{
    for (int i = 0; i < nLocal; i++) {
        if (gl[2 * (size_t)(s + i)] != 1 || gl[2 * (size_t)(s + i) + 1] != 1) return 1;
    }
    return 0;
}

/* load a column of eMatDH, apply normalize_emissions, return emission_max
 * (reference-single.cpp:983-990 / :1886-1893) */
static double load_emission_col(const double *eMatDH, int nrow, int g, int normalize_emissions,
                                double *col, double prev_emission_max)
{
    double emission_max = prev_emission_max;
    memcpy(col, eMatDH + (size_t)nrow * g, sizeof(double) * (size_t)nrow);
    if (normalize_emissions) {
        emission_max = col[0];
        for (int i = 1; i < nrow; i++) if (col[i] > emission_max) emission_max = col[i];
        if (emission_max < 1) {
            double f = 1 / emission_max;
            for (int i = 0; i < nrow; i++) col[i] *= f;
        }
    }
    return emission_max;
}

int qo_haploid_dosage_versus_refs(
    const qo_panel_t *p, const qo_fullpass_opts_t *o, const double *gl,
    const int32_t *gammaSmall_cols_to_get, double *alphaHat_t, double *betaHat_t, double *c,
    double *gamma_t, double *gammaSmall_t, double *dosage, int32_t *best_ptr,
    int32_t *best_idx, double *best_val, int64_t best_cap)
{
    const int K = p->K, nGrids = p->nGrids, nSNPs = p->nSNPs, nMaxDH = p->nMaxDH;
    const int nrow = nMaxDH + 1;
    const double ref_error = p->ref_error, ref_one_minus_error = 1 - ref_error;
    const double double_K = (double)K;
    const double one_over_K = 1 / (double)K;
    const double *tm = p->transMatRate_t;
    int status = 0;

    /* reference-single.cpp:2263-2268 */
    int only_store_alpha_at_gamma_small =
        ((o->get_best_haps_from_thinned_sites || o->return_gammaSmall_t) && !o->return_gamma_t &&
         !o->return_dosage && !o->return_betaHat_t);

    double *eMatDH = (double *)malloc(sizeof(double) * (size_t)nrow * nGrids);
    qo_build_eMatDH(p->distinctHapsB, gl, nMaxDH, nGrids, nSNPs, ref_error, 1, eMatDH);

    double *alpha_col = (double *)malloc(sizeof(double) * K);
    double *tmp_col = (double *)malloc(sizeof(double) * K);
    double *beta_col = (double *)malloc(sizeof(double) * K);
    double *gamma_col = (double *)calloc((size_t)K, sizeof(double));
    double *e_times_b = (double *)malloc(sizeof(double) * K);
    double *ecol = (double *)malloc(sizeof(double) * nrow);
    double *matched = (double *)malloc(sizeof(double) * nrow);
    int32_t *tk_idx = (int32_t *)malloc(sizeof(int32_t) * K);
    double *tk_val = (double *)malloc(sizeof(double) * K);

    /* ---- alpha at grid 0 (reference-single.cpp:2292-2354) ---- */
    {
        int s, nLocal;
        grid_bounds(0, nSNPs, &s, &nLocal);
        double sum = 0;
        for (int k = 0; k < K; k++) {
            int dh = dh_at(p, k, 0);
            double prob;
            if (dh > 0) {
                prob = eMatDH[dh];
            } else {
                uint32_t w;
                if (p->use_eMatDH_special_symbols) {
                    /* s1 = e1 = 0 when grid 0 has no specials (cannot happen with dh == 0) */
                    w = special_word(p, k, 0);
                } else {
                    w = (uint32_t)p->rhb_t[k];
                }
                prob = word_emission(w, gl + 2 * (size_t)s, nLocal, ref_error, ref_one_minus_error);
            }
            alpha_col[k] = prob * one_over_K;
        }
        /* c(0) = 1 / sum(alphaHat_t_col) (:2347): an arma::colvec, i.e. arrayops::accumulate's two accumulators
         * (quilt_oracle.h, "Armadillo's sum()") -- the one sum of the full pass that is not an explicit loop */
        QO_ARMA_SUM(sum, K, k, alpha_col[k]);
        c[0] = 1 / sum;
        for (int k = 0; k < K; k++) alphaHat_t[k] = alpha_col[k] * c[0];
    }

    /* ---- forward (reference-single.cpp:935-1129) ---- */
    {
        double running_min_emission_prob = 1, min_emission_prob = 1;
        double prev_sum = 1, emission_max = 1, run_total = 0;
        for (int g = 1; g < nGrids; g++) {
            c[g] = 1;
            double jump_prob = tm[2 * (size_t)(g - 1) + 1] / double_K;
            double jump_prob_plus = o->always_normalize ? jump_prob : jump_prob * prev_sum;
            double not_jump_prob = tm[2 * (size_t)(g - 1)];
            double jpp_div = jump_prob_plus / not_jump_prob;
            int s, nLocal;
            grid_bounds(g, nSNPs, &s, &nLocal);
            int has_variant = grid_has_variant(gl, s, nLocal);
            if (g == 1) {
                has_variant = 1;
                for (int k = 0; k < K; k++) alpha_col[k] = alphaHat_t[k];
            }
            int store = 1;
            if (only_store_alpha_at_gamma_small && gammaSmall_cols_to_get[g] < 0) store = 0;
            if (has_variant) {
                emission_max = load_emission_col(eMatDH, nrow, g, o->normalize_emissions, ecol, emission_max);
                min_emission_prob = ecol[0];
                for (int i = 1; i < nrow; i++) if (ecol[i] < min_emission_prob) min_emission_prob = ecol[i];
                run_total = 0;
                const int32_t *list;
                int nsp = special_list(p, g, &list);
                for (int i = 0; i < nsp; i++) {
                    int k = list[i];
                    uint32_t w = special_word(p, k, g);
                    double prob = word_emission(w, gl + 2 * (size_t)s, nLocal, ref_error, ref_one_minus_error);
                    prob *= (1 / emission_max);
                    tmp_col[k] = (jpp_div + alpha_col[k]) * prob;
                    run_total += tmp_col[k];
                    if (prob < min_emission_prob) min_emission_prob = prob;
                }
                ecol[0] = 0;
                for (int k = 0; k < K; k++) {
                    alpha_col[k] = (jpp_div + alpha_col[k]) * ecol[dh_at(p, k, g)];
                    run_total += alpha_col[k];
                }
                for (int i = 0; i < nsp; i++) {
                    int k = list[i];
                    run_total -= alpha_col[k];
                    alpha_col[k] = tmp_col[k];
                }
                running_min_emission_prob *= min_emission_prob;
            } else {
                for (int k = 0; k < K; k++) alpha_col[k] = jpp_div + alpha_col[k];
                run_total = prev_sum / not_jump_prob;
            }
            c[g] /= not_jump_prob;
            if (o->always_normalize ||
                running_min_emission_prob < o->min_emission_prob_normalization_threshold ||
                g == nGrids - 1) {
                double x = 1 / run_total;
                for (int k = 0; k < K; k++) alpha_col[k] *= x;
                c[g] /= run_total;
                run_total = 1;
                running_min_emission_prob = 1;
            }
            prev_sum = run_total;
            if (store) memcpy(alphaHat_t + (size_t)K * g, alpha_col, sizeof(double) * K);
        }
    }

    /* ---- backward (reference-single.cpp:1854-2177) ---- */
    {
        double not_jump_prob = 1, jump_prob = 0;
        double B_prev = 1, B_prev_star = 1, emission_max = 1, sum_e_times_b = 0;
        int64_t best_n = 0;
        int n_thin = 0;
        for (int g = 0; g < nGrids; g++) if (gammaSmall_cols_to_get[g] >= 0) n_thin++;
        if (best_ptr) for (int i = 0; i <= n_thin; i++) best_ptr[i] = 0;
        for (int g = nGrids - 1; g >= 0; --g) {
            if (g == nGrids - 1) {
                for (int k = 0; k < K; k++) beta_col[k] = 1 / not_jump_prob;
                B_prev_star = K * c[g] * not_jump_prob;
            } else {
                jump_prob = tm[2 * (size_t)g + 1] / double_K;
                not_jump_prob = tm[2 * (size_t)g];
                int s, nLocal;
                grid_bounds(g + 1, nSNPs, &s, &nLocal);
                int has_variant = grid_has_variant(gl, s, nLocal);
                if (has_variant) {
                    emission_max = load_emission_col(eMatDH, nrow, g + 1, o->normalize_emissions, ecol, emission_max);
                    sum_e_times_b = 0;
                    const int32_t *list;
                    int nsp = special_list(p, g + 1, &list);
                    for (int i = 0; i < nsp; i++) {
                        int k = list[i];
                        uint32_t w = special_word(p, k, g + 1);
                        double prob = word_emission(w, gl + 2 * (size_t)s, nLocal, ref_error, ref_one_minus_error);
                        prob *= 1 / emission_max;
                        tmp_col[k] = beta_col[k] * prob;
                        sum_e_times_b += tmp_col[k];
                    }
                    ecol[0] = 0;
                    for (int k = 0; k < K; k++) {
                        e_times_b[k] = beta_col[k] * ecol[dh_at(p, k, g + 1)];
                        sum_e_times_b += e_times_b[k];
                    }
                    for (int i = 0; i < nsp; i++) e_times_b[list[i]] = tmp_col[list[i]];
                    double val = jump_prob / not_jump_prob * sum_e_times_b;
                    for (int k = 0; k < K; k++) beta_col[k] = e_times_b[k] + val;
                    B_prev = sum_e_times_b;
                    B_prev_star = c[g] * B_prev;
                } else {
                    double val = jump_prob / not_jump_prob * B_prev_star;
                    for (int k = 0; k < K; k++) beta_col[k] = beta_col[k] + val;
                    B_prev = B_prev_star;
                    B_prev_star = c[g] * B_prev;
                }
            }
            /* gamma / top-K (reference-single.cpp:2014-2060) */
            int calc_small = (o->return_gammaSmall_t && gammaSmall_cols_to_get[g] >= 0);
            const double *acol = alphaHat_t + (size_t)K * g;
            if (o->get_best_haps_from_thinned_sites && gammaSmall_cols_to_get[g] >= 0) {
                int n = qo_get_top_K_or_more_matches_while_building_gamma(
                    acol, beta_col, gamma_col, K, o->K_top_matches, not_jump_prob, tk_idx, tk_val);
                int col = gammaSmall_cols_to_get[g];
                if (best_ptr) best_ptr[col + 1] = n; /* sizes first; prefix-summed after the loop */
                /* grids are visited in descending order, i.e. descending column:
                 * fill the output from the back, compact to the front afterwards */
                if (best_idx && best_val) {
                    if (best_n + n <= best_cap) {
                        int64_t off = best_cap - best_n - n;
                        for (int i = 0; i < n; i++) {
                            best_idx[off + i] = tk_idx[i];
                            best_val[off + i] = tk_val[i];
                        }
                    } else {
                        status = -1;
                    }
                    best_n += n;
                }
            } else if (o->return_dosage || o->return_gamma_t || calc_small) {
                for (int k = 0; k < K; k++) gamma_col[k] = acol[k] * beta_col[k];
            }
            if (o->return_dosage) {
                int s, nLocal;
                grid_bounds(g, nSNPs, &s, &nLocal);
                double dosageL[32];
                for (int b = 0; b < 32; b++) dosageL[b] = 0;
                for (int i = 0; i < nrow; i++) matched[i] = 0;
                for (int k = 0; k < K; k++) matched[dh_at(p, k, g)] += gamma_col[k];
                for (int i = 0; i < nrow; i++) matched[i] *= not_jump_prob;
                const int32_t *list;
                int nsp = special_list(p, g, &list);
                for (int i = 0; i < nsp; i++) {
                    int k = list[i];
                    uint32_t w = special_word(p, k, g);
                    double gk = gamma_col[k] * not_jump_prob;
                    for (int b = 0; b < nLocal; b++) {
                        if (w & (1u << b)) dosageL[b] += gk * ref_one_minus_error;
                        else dosageL[b] += gk * ref_error;
                    }
                }
                for (int b = 0; b < nLocal; b++) {
                    for (int dh = 0; dh < nMaxDH; dh++) {
                        dosageL[b] += p->distinctHapsIE[(size_t)nMaxDH * (s + b) + dh] * matched[dh + 1];
                    }
                    dosage[s + b] = dosageL[b];
                }
            }
            {
                double x = c[g] * not_jump_prob;
                for (int k = 0; k < K; k++) beta_col[k] *= x;
            }
            if (o->return_betaHat_t) memcpy(betaHat_t + (size_t)K * g, beta_col, sizeof(double) * K);
            if (o->return_gamma_t) {
                for (int k = 0; k < K; k++) gamma_col[k] *= not_jump_prob;
                memcpy(gamma_t + (size_t)K * g, gamma_col, sizeof(double) * K);
            }
            if (calc_small) {
                memcpy(gammaSmall_t + (size_t)K * gammaSmall_cols_to_get[g], gamma_col, sizeof(double) * K);
            }
        }
        /* compact the best-haps lists (filled from the back in descending column
         * order, so they are already in ascending column order) to the front */
        if (best_ptr && best_idx && best_val && status == 0) {
            int64_t off = best_cap - best_n;
            if (off > 0) {
                memmove(best_idx, best_idx + off, sizeof(int32_t) * (size_t)best_n);
                memmove(best_val, best_val + off, sizeof(double) * (size_t)best_n);
            }
        }
        if (best_ptr) {
            /* sizes -> offsets */
            int32_t acc = 0;
            for (int i = 0; i < n_thin; i++) {
                int32_t n = best_ptr[i + 1];
                best_ptr[i + 1] = acc + n;
                acc += n;
            }
            best_ptr[0] = 0;
        }
    }

    free(eMatDH); free(alpha_col); free(tmp_col); free(beta_col); free(gamma_col);
    free(e_times_b); free(ecol); free(matched); free(tk_idx); free(tk_val);
    return status;
}
