/*
 * oracle/gibbs.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see quilt_oracle.h).
 *
 * fp64 restatement, in the reference's operation order (its Ks-wide sums are Armadillo sum() calls: two accumulators, even
 * and odd elements -- quilt_oracle.h lists the sites and states what is assumed about Armadillo, which is not in this
 * image; qo_set_sum_order(1) restores the left-to-right sums of rounds 1-5), of the small-panel Gibbs
 * read-label sampler: QUILT/src/gibbs-nipt.cpp (rcpp_forwardBackwardGibbsNIPT and
 * its helpers), QUILT/src/gibbs-small.cpp (packed-panel emissions and
 * hapProbs/genProbs), QUILT/src/copied-from-stitch.cpp (haploid forward/backward)
 * and the shard resampler of QUILT/src/gibbs-nipt-block.cpp, for the production
 * argument values: S = 1, n_gibbs_starts = 1, priorCurrent_m and
 * alphaMatCurrent_tc constant 1/Ks, use_small_eHapsCurrent_tc = FALSE,
 * calculate_gamma_on_the_fly = TRUE, pass_in_alphaBeta = TRUE, record_read_set = TRUE.
 *
 * R's RNG cannot be reproduced here; every uniform the reference draws
 * (gibbs-nipt.cpp:2845-2848, gibbs-nipt-block.cpp:2054) is an INPUT.
 *
 * Block Gibbs, diploid (ff = 0, sample_is_diploid): Rcpp_block_gibbs_resampler
 * (gibbs-nipt-block.cpp:1636-1967) cannot relabel in this mode: c3 is all zero
 * (gibbs-nipt.cpp:2678 and the !sample_is_diploid guards), so logC_after(2) =
 * sum(log(c3)) = -inf and "logC_after(2) -= log(c3(g))" is NaN
 * (gibbs-nipt-block.cpp:1819-1821, :1896-1898); every choice_log_probs entry is
 * NaN (:661-675), every "chance < cumsum" test fails, ir_chosen stays 0 (:741-752)
 * and the "No change warranted" branch is taken (:830).  What remains is the
 * final backward re-run (:1947-1954), which reproduces the beta the sweep already
 * holds bit for bit.  The oracle therefore treats the diploid block resampler as
 * the identity and runs the shard resampler (:1975-2355), which is active.
 * Block Gibbs for ff > 0 (NIPT): block definition (:311-523), make_gibbs_considers (:1307-1553) and the block
 * resampler proper (:590-949, :1122-1292, :1636-1967) for block_approach = 6, restated below.
 *
 * PIN: the reference cannot be built or run in this container and its tests hold no golden vectors.  This file is pinned by
 * (1) oracle/rtwin.py, an independent NumPy restatement of the reference's R sampler (QUILT/R/gibbs-nipt.R:508-997 with
 * make_eMatRead_t / evaluate_read_variability / make_rlc): same uniforms in, read labels and H_class identical, alpha / beta
 * / eMatGrid to 1e-8 (tests/golden/make_golden_rtwin.py; tests/test_rtwin_cpu.py on every CPU run, fixtures generated from
 * the R-twin side) -- the R twin has no block / shard pass, so those rest on (2) the known answers and (3) the invariants
 * of the reference's tests (tests/golden/known_answers.json, tests/test_oracle_cpu.py: state after a pass == from-scratch
 * forward-backward given the labels, the label / class swap table, get_log_p_H_class2, make_gibbs_considers' definition).
 * Not an execution of the reference: "unpinned against a reference run" (DESIGN.md section 3).
 */
#define _GNU_SOURCE /* qsort_r */
#include "quilt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- packed-panel read emissions (gibbs-small.cpp:116-265) ----------------- */

static inline int panel_code(const qo_panel_t *p, int k, int g)
{
    if (p->hapMatcherR) return p->hapMatcherR[(size_t)p->K * g + k];
    return p->hapMatcher[(size_t)p->K * g + k];
}

static uint32_t panel_special_word(const qo_panel_t *p, int k, int g)
{
    if (p->use_eMatDH_special_symbols) {
        int s1 = p->eMatDH_special_matrix_helper[g];
        int e1 = p->eMatDH_special_matrix_helper[(size_t)p->nGrids + g];
        return (uint32_t)qo_simple_binary_matrix_search(k, p->eMatDH_special_matrix,
                                                        p->eMatDH_special_matrix_nrow, s1, e1);
    }
    return (uint32_t)p->rhb_t[(size_t)p->K * g + k];
}

void qo_make_eMatRead_t_for_gibbs_using_objects(
    const qo_panel_t *p, const int32_t *which_haps_to_use_1based, int Ks, int nReads,
    const int32_t *read_ptr, const int32_t *u, const int32_t *bq, int rescale_eMatRead_t, int Jmax,
    double maxDifferenceBetweenReads, double *eMatRead_t /* Ks x nReads, pre-filled (normally 1) */)
{
    double pR = 1, pA = 1; /* NB: carried over between reads when bq == 0 (:139-140) */
    const double d2 = 1 / maxDifferenceBetweenReads;
    int *codes = (int *)malloc(sizeof(int) * (size_t)Ks);
    for (int r = 0; r < nReads; r++) {
        const int32_t *ru = u + read_ptr[r], *rbq = bq + read_ptr[r];
        int J = read_ptr[r + 1] - read_ptr[r] - 1;
        double *col = eMatRead_t + (size_t)Ks * r;
        int g_prev = ru[0] / 32;
        for (int k = 0; k < Ks; k++) codes[k] = panel_code(p, which_haps_to_use_1based[k] - 1, g_prev);
        if (J >= Jmax) J = Jmax;
        for (int j = 0; j <= J; j++) {
            if (rbq[j] < 0) {
                double eps = pow(10, (double)rbq[j] / 10);
                pR = 1 - eps;
                pA = eps / 3;
            }
            if (rbq[j] > 0) {
                double eps = pow(10, -(double)rbq[j] / 10);
                pR = eps / 3;
                pA = 1 - eps;
            }
            int g = ru[j] / 32;
            if (g != g_prev)
                for (int k = 0; k < Ks; k++) codes[k] = panel_code(p, which_haps_to_use_1based[k] - 1, g);
            g_prev = g;
            for (int k = 0; k < Ks; k++) {
                double e;
                if (codes[k] > 0) {
                    e = p->distinctHapsIE[(size_t)p->nMaxDH * ru[j] + (codes[k] - 1)];
                } else {
                    uint32_t w = panel_special_word(p, which_haps_to_use_1based[k] - 1, g);
                    e = ((w >> (ru[j] % 32)) & 1u) ? 1 - p->ref_error : p->ref_error;
                }
                col[k] *= (e * pA + (1 - e) * pR);
            }
        }
        if (rescale_eMatRead_t) {
            double x = 0;
            for (int k = 0; k < Ks; k++) if (col[k] > x) x = col[k];
            double d1 = 1 / x;
            if (isinf(x) || x == 0 || isinf(d1)) { /* the NaN comparisons of :244 are never true */
                for (int k = 0; k < Ks; k++) col[k] = 1;
            } else {
                for (int k = 0; k < Ks; k++) {
                    col[k] *= d1;
                    if (col[k] < d2) col[k] = d2;
                }
            }
        }
    }
    free(codes);
}

/* gibbs-nipt.cpp:338-382 */
void qo_evaluate_read_variability(const double *eMatRead_t, int Ks, int nReads,
                                  int32_t *number_of_non_1_reads, int32_t *indices_of_non_1_reads,
                                  int32_t *read_category)
{
    const double thresh = 1 - pow(10, -12);
    const int thresh2 = (int)(Ks * 0.20);
    for (int r = 0; r < nReads; r++) {
        const double *col = eMatRead_t + (size_t)Ks * r;
        int c = 0, more_than_two = 0;
        double val = -1;
        for (int k = 0; k < Ks; k++) {
            if (col[k] < thresh) {
                indices_of_non_1_reads[(size_t)Ks * r + c] = k;
                c++;
                if (val == -1) val = col[k];
                else if (val != col[k]) more_than_two = 1;
            }
        }
        number_of_non_1_reads[r] = c;
        if (c == 0) read_category[r] = 1;
        else if (!more_than_two) read_category[r] = 2;
        else if (c < thresh2) read_category[r] = 3;
        else read_category[r] = 0;
    }
}

/* ---- haploid forward / backward (copied-from-stitch.cpp) ------------------- */

/* sum() of an arma column (a .col() view or a colvec): arrayops::accumulate, two accumulators (quilt_oracle.h) */
static double col_sum(const double *x, int n)
{
    double s;
    QO_ARMA_SUM(s, n, i, x[i]);
    return s;
}

/* Rcpp_run_forward_haploid, copied-from-stitch.cpp:340-387, prior = alphaMat = 1/Ks */
static void run_forward_haploid(double *alpha, double *c, const double *eMatGrid, const double *tm,
                                int Ks, int G, int initialize_only)
{
    const double prior = 1.0 / Ks;
    for (int k = 0; k < Ks; k++) alpha[k] = prior * eMatGrid[k];
    c[0] = 1 / col_sum(alpha, Ks);
    for (int k = 0; k < Ks; k++) alpha[k] = alpha[k] * c[0];
    if (initialize_only) return;
    for (int g = 1; g < G; g++) {
        const double s0 = tm[2 * (size_t)(g - 1)], s1 = tm[2 * (size_t)(g - 1) + 1];
        double *a = alpha + (size_t)Ks * g;
        const double *ap = alpha + (size_t)Ks * (g - 1), *e = eMatGrid + (size_t)Ks * g;
        for (int k = 0; k < Ks; k++) a[k] = e[k] * (s0 * ap[k] + s1 * prior);
        c[g] = 1 / col_sum(a, Ks);
        for (int k = 0; k < Ks; k++) a[k] *= c[g];
    }
}

/* Rcpp_run_backward_haploid, copied-from-stitch.cpp:392-409 (alphaMat = 1/Ks) */
static void run_backward_haploid(double *beta, const double *c, const double *eMatGrid, const double *tm,
                                 int Ks, int G, double *etb)
{
    const double am = 1.0 / Ks;
    for (int g = G - 2; g >= 0; --g) {
        const double *e = eMatGrid + (size_t)Ks * (g + 1), *bn = beta + (size_t)Ks * (g + 1);
        double *b = beta + (size_t)Ks * g;
        double s;
        for (int k = 0; k < Ks; k++) etb[k] = e[k] * bn[k];
        QO_ARMA_SUM(s, Ks, k, am * etb[k]);   /* sum(alphaMatCurrent_tc.slice(s).col(iGrid) % e_times_b), :405 */
        double x = tm[2 * (size_t)g + 1] * s;
        for (int k = 0; k < Ks; k++) b[k] = c[g] * (x + tm[2 * (size_t)g] * etb[k]);
    }
}

/* Rcpp_run_backward_haploid_QUILT_faster, copied-from-stitch.cpp:417-440 */
static void run_backward_haploid_faster(double *beta, const double *c, const double *eMatGrid,
                                        const double *tm, const uint8_t *grid_has_read, int Ks, int G,
                                        double *etb)
{
    const double one_over_K = 1 / (double)Ks;
    for (int g = G - 2; g >= 0; --g) {
        const double *bn = beta + (size_t)Ks * (g + 1);
        double *b = beta + (size_t)Ks * g;
        if (grid_has_read[g + 1]) {
            const double *e = eMatGrid + (size_t)Ks * (g + 1);
            for (int k = 0; k < Ks; k++) etb[k] = e[k] * bn[k];
            double x = tm[2 * (size_t)g + 1] * col_sum(etb, Ks) * one_over_K;
            for (int k = 0; k < Ks; k++) b[k] = c[g] * (x + tm[2 * (size_t)g] * etb[k]);
        } else {
            double x = tm[2 * (size_t)g + 1] * col_sum(bn, Ks) * one_over_K;
            for (int k = 0; k < Ks; k++) b[k] = c[g] * (x + tm[2 * (size_t)g] * bn[k]);
        }
    }
}

/* rcpp_alpha_forward_one_QUILT_faster, gibbs-nipt.cpp:671-707 (normalize = true) */
static void alpha_forward_one_faster(int g, int Ks, double *alpha, const double *tm, const double *eMatGrid,
                                     double *c, const uint8_t *grid_has_read)
{
    const double one_over_K = 1 / (double)Ks;
    const double *ap = alpha + (size_t)Ks * (g - 1);
    double *a = alpha + (size_t)Ks * g;
    double alphaConst = tm[2 * (size_t)(g - 1) + 1] * col_sum(ap, Ks);
    double x = tm[2 * (size_t)(g - 1)];
    double c2 = c[g];
    if (grid_has_read[g]) {
        const double *e = eMatGrid + (size_t)Ks * g;
        for (int k = 0; k < Ks; k++) a[k] = e[k] * (x * ap[k] + alphaConst * one_over_K);
    } else {
        for (int k = 0; k < Ks; k++) a[k] = (x * ap[k] + alphaConst * one_over_K);
    }
    double aa = 1 / (c2 * col_sum(a, Ks));
    c[g] *= aa;
    aa *= c2;
    for (int k = 0; k < Ks; k++) a[k] *= aa;
}

/* rcpp_alpha_forward_one, gibbs-nipt.cpp:627-657 (normalize = true, alphaMat = 1/Ks) */
static void alpha_forward_one(int g, int Ks, double *alpha, const double *tm, const double *eMatGrid, double *c)
{
    const double am = 1.0 / Ks;
    const double *ap = alpha + (size_t)Ks * (g - 1), *e = eMatGrid + (size_t)Ks * g;
    double *a = alpha + (size_t)Ks * g;
    double alphaConst = tm[2 * (size_t)(g - 1) + 1] * col_sum(ap, Ks);
    double x = tm[2 * (size_t)(g - 1)];
    double c2 = c[g];
    for (int k = 0; k < Ks; k++) a[k] = c2 * e[k] * (x * ap[k] + alphaConst * am);
    double aa = 1 / col_sum(a, Ks);
    c[g] *= aa;
    for (int k = 0; k < Ks; k++) a[k] *= aa;
}

/* ---- one Gibbs sweep (gibbs-nipt.cpp:1756-1956 + :733-1295) ---------------- */

typedef struct {
    int Ks, G, R, nH;               /* nH = 2 (sample_is_diploid) or 3 */
    double *alpha[3], *beta[3], *eg[3], *c[3];
    const double *eMatRead;
    const int32_t *wif;
    const uint8_t *grid_has_read;
    const double *tm;
    int32_t *H, *H_class;
    const int32_t *read_category, *n_non1, *idx_non1;
    double prior_probs[3];
    double rlc[7][3];
    double class_sum_cutoff;
    int sample_is_diploid;
} sweep_t;

static void sample_reads_in_grid(sweep_t *S, int *iRead_io, int g, int *done_reads, int *read_wif,
                                 int iteration, const double *runif_reads, int init_iteratively, int first_read,
                                 double *am /* Ks x nH */, double *bm, double *ab, double *pC, double *pA1,
                                 double *pA2)
{
    const int Ks = S->Ks, nH = S->nH, R = S->R;
    int iRead = *iRead_io;
    int h_rC = 0, h_rA1 = 1, h_rA2 = 2, h_rN = 0;
    int this_grid_has_at_least_one_read = 0, at_least_one_read_has_changed = 0;
    int normal = 0, ginit = 0, pass = 0;
    while (!*done_reads && *read_wif == g) {
        if (!S->sample_is_diploid || S->read_category[iRead] != 1) {
            if (!init_iteratively) {
                normal = 1;
            } else if (iRead < first_read && iteration == 0) {
                pass = 1;
            } else if (first_read <= iRead && iteration == 0) {
                pass = 0; ginit = 1;
            } else if (iRead < first_read && iteration == 1) {
                pass = 0; ginit = 1;
            } else {
                ginit = 0; normal = 1;
            }
            if (!this_grid_has_at_least_one_read) {
                for (int h = 0; h < 3; h++) pC[h] = pA1[h] = pA2[h] = 1;
                for (int h = 0; h < nH; h++) {
                    memcpy(am + (size_t)Ks * h, S->alpha[h] + (size_t)Ks * g, sizeof(double) * Ks);
                    memcpy(bm + (size_t)Ks * h, S->beta[h] + (size_t)Ks * g, sizeof(double) * Ks);
                }
                for (int i = 0; i < Ks * nH; i++) ab[i] = am[i] * bm[i];
                for (int h = 0; h < nH; h++) pC[h] = col_sum(ab + (size_t)Ks * h, Ks);
                this_grid_has_at_least_one_read = 1;
            }
            const double *er = S->eMatRead + (size_t)Ks * iRead;
            if (normal) {
                h_rC = S->H[iRead] - 1;
                if (h_rC == 0) { h_rA1 = 1; h_rA2 = 2; }
                else if (h_rC == 1) { h_rA1 = 0; h_rA2 = 2; }
                else { h_rA1 = 0; h_rA2 = 1; }
                for (int h = 0; h < 3; h++) pA1[h] = pA2[h] = pC[h];
                const int cat = S->read_category[iRead];
                double *abC = ab + (size_t)Ks * h_rC, *abA1 = ab + (size_t)Ks * h_rA1;
                double *abA2 = (nH == 3) ? ab + (size_t)Ks * h_rA2 : NULL;
                if (cat == 0) {
                    /* sum(ab_m.col(h) / eMatRead_t_col), sum(ab_m.col(h) % eMatRead_t_col) (:909-912): accu_proxy_linear */
                    double s1, s2, s3;
                    QO_ARMA_SUM(s1, Ks, k, abC[k] / er[k]);
                    QO_ARMA_SUM(s2, Ks, k, abA1[k] * er[k]);
                    pA1[h_rC] = s1;
                    pA1[h_rA1] = s2;
                    if (!S->sample_is_diploid) {
                        QO_ARMA_SUM(s3, Ks, k, abA2[k] * er[k]);
                        pA2[h_rA2] = s3;
                    }
                } else if (cat == 2) {
                    double v1 = 0, v2 = 0, v3 = 0;
                    int k = 0;
                    const int32_t *idx = S->idx_non1 + (size_t)Ks * iRead;
                    for (int ik = 0; ik < S->n_non1[iRead]; ik++) {
                        k = idx[ik];
                        v1 += abC[k];
                        v2 += abA1[k];
                        if (!S->sample_is_diploid) v3 += abA2[k];
                    }
                    pA1[h_rC] += v1 * (1 / er[k] - 1);   /* k = last listed index (:926-927) */
                    pA1[h_rA1] += v2 * (er[k] - 1);
                    if (!S->sample_is_diploid) pA2[h_rA2] += v3 * (er[k] - 1);
                } else if (cat == 3) {
                    const int32_t *idx = S->idx_non1 + (size_t)Ks * iRead;
                    for (int ik = 0; ik < S->n_non1[iRead]; ik++) {
                        int k = idx[ik];
                        pA1[h_rC] += abC[k] * (1 / er[k] - 1);
                        pA1[h_rA1] += abA1[k] * (er[k] - 1);
                        if (!S->sample_is_diploid) pA2[h_rA2] += abA2[k] * (er[k] - 1);
                    }
                }
                /* cat == 1 (only reachable when not diploid): nothing changes */
                pA2[h_rA1] = pC[h_rA1];
                pA2[h_rC] = pA1[h_rC];
            } else if (ginit) {
                h_rC = 0; h_rA1 = 1; h_rA2 = 2;
                for (int h = 0; h < 3; h++) pA1[h] = pA2[h] = pC[h];
                double s1, s2, s3;   /* (:971-974) */
                QO_ARMA_SUM(s1, Ks, k, ab[k] * er[k]);
                QO_ARMA_SUM(s2, Ks, k, ab[(size_t)Ks + k] * er[k]);
                pC[h_rC] = s1;
                pA1[h_rA1] = s2;
                if (!S->sample_is_diploid) {
                    QO_ARMA_SUM(s3, Ks, k, ab[(size_t)2 * Ks + k] * er[k]);
                    pA2[h_rA2] = s3;
                }
            } else {
                for (int h = 0; h < 3; h++) pA1[h] = pA2[h] = pC[h];
            }
            double prod_pC = (pC[0] * pC[1] * pC[2]) * S->prior_probs[h_rC];
            double prod_pA1 = (pA1[0] * pA1[1] * pA1[2]) * S->prior_probs[h_rA1];
            double prod_pA2 = (pA2[0] * pA2[1] * pA2[2]) * S->prior_probs[h_rA2];
            double denom = prod_pC + prod_pA1 + prod_pA2;
            double norm_pC = prod_pC / denom, norm_pA1 = prod_pA1 / denom, norm_pA2 = prod_pA2 / denom;
            double chance = runif_reads[(size_t)R * iteration + iRead];
            double cs[3] = {0, 0, 0};
            cs[h_rC] = norm_pC;
            cs[h_rA1] = norm_pA1;
            cs[h_rA2] = norm_pA2;
            cs[1] += cs[0];
            cs[2] += cs[1];
            h_rN = 0;
            for (int i = 2; i >= 0; i--) if (chance < cs[i]) h_rN = i;
            if (((h_rN != h_rC) || ginit) && !pass) {
                at_least_one_read_has_changed = 1;
                S->H[iRead] = h_rN + 1;
                if (normal) {
                    double *x = am + (size_t)Ks * h_rC, *y = ab + (size_t)Ks * h_rC;
                    for (int k = 0; k < Ks; k++) x[k] /= er[k];
                    for (int k = 0; k < Ks; k++) y[k] /= er[k];
                }
                {
                    double *x = am + (size_t)Ks * h_rN, *y = ab + (size_t)Ks * h_rN;
                    for (int k = 0; k < Ks; k++) x[k] *= er[k];
                    for (int k = 0; k < Ks; k++) y[k] *= er[k];
                }
                if (normal && (h_rC < 2 || !S->sample_is_diploid)) {
                    double *e = S->eg[h_rC] + (size_t)Ks * g;
                    for (int k = 0; k < Ks; k++) e[k] /= er[k];
                }
                if (h_rN < 2 || !S->sample_is_diploid) {
                    double *e = S->eg[h_rN] + (size_t)Ks * g;
                    for (int k = 0; k < Ks; k++) e[k] *= er[k];
                }
                if (normal) {
                    /* the A1 move goes to the lower of the two other labels, A2 to the higher (:1103-1117) */
                    const double *src = (h_rN == h_rA1) ? pA1 : pA2;
                    for (int i = 0; i < 3; i++) pC[i] = src[i];
                } else if (ginit) {
                    if (h_rN == 1) for (int i = 0; i < 3; i++) pC[i] = pA1[i];
                    if (h_rN == 2) for (int i = 0; i < 3; i++) pC[i] = pA2[i];
                }
            }
            /* record_read_set (:1142-1165) */
            {
                double x[3];
                x[h_rC] = norm_pC;
                x[h_rA1] = norm_pA1;
                x[h_rA2] = norm_pA2;
                double local_min = 2;
                int which = 8;
                for (int i = 0; i < 7; i++) {
                    double y = fabs(S->rlc[i][0] - x[0]) + fabs(S->rlc[i][1] - x[1]) + fabs(S->rlc[i][2] - x[2]);
                    if (y < local_min) { local_min = y; which = i; }
                }
                S->H_class[iRead] = (local_min < S->class_sum_cutoff) ? which + 1 : 0;
            }
        }
        iRead++;
        if (R - 1 < iRead) { *done_reads = 1; *read_wif = -1; }
        else *read_wif = S->wif[iRead];
    }
    if (at_least_one_read_has_changed) {
        for (int h = 0; h < nH; h++) {
            double *a = S->alpha[h] + (size_t)Ks * g;
            memcpy(a, am + (size_t)Ks * h, sizeof(double) * Ks);
            double alphaConst = 1 / col_sum(am + (size_t)Ks * h, Ks);
            S->c[h][g] *= alphaConst;
            for (int k = 0; k < Ks; k++) a[k] *= alphaConst;
        }
    }
    *iRead_io = iRead;
}

static void gibbs_iterate(sweep_t *S, int iteration, const double *runif_reads, int init_iteratively,
                          int first_read, double *work)
{
    const int Ks = S->Ks, G = S->G, nH = S->nH;
    double *am = work, *bm = work + (size_t)3 * Ks, *ab = work + (size_t)6 * Ks, *etb = work + (size_t)9 * Ks;
    double pC[3] = {1, 1, 1}, pA1[3] = {1, 1, 1}, pA2[3] = {1, 1, 1};
    int done_reads = 0, iRead = -1, read_wif = -1;
    const double prior = 1.0 / Ks;
    for (int g = 0; g < G; g++) {
        if (g > 0) {
            for (int h = 0; h < nH; h++) alpha_forward_one_faster(g, Ks, S->alpha[h], S->tm, S->eg[h], S->c[h], S->grid_has_read);
        } else {
            /* rcpp_reinitialize_in_iterations (:712-727) */
            for (int h = 0; h < nH; h++) {
                double *a = S->alpha[h];
                for (int k = 0; k < Ks; k++) a[k] = prior * S->eg[h][k];
                S->c[h][0] = 1 / col_sum(a, Ks);
                for (int k = 0; k < Ks; k++) a[k] *= S->c[h][0];
            }
        }
        iRead++;
        if (!done_reads) {
            if (iRead < S->R) read_wif = S->wif[iRead];
            else { done_reads = 1; read_wif = -1; }
        } else {
            read_wif = -1;
        }
        if (read_wif == g)
            sample_reads_in_grid(S, &iRead, g, &done_reads, &read_wif, iteration, runif_reads, init_iteratively,
                                 first_read, am, bm, ab, pC, pA1, pA2);
        iRead = iRead - 1;
    }
    for (int h = 0; h < nH; h++) {
        double *b = S->beta[h] + (size_t)Ks * (G - 1);
        for (int k = 0; k < Ks; k++) b[k] = S->c[h][G - 1];
        run_backward_haploid_faster(S->beta[h], S->c[h], S->eg[h], S->tm, S->grid_has_read, Ks, G, etb);
    }
}

/* Rcpp_shard_block_gibbs_resampler (gibbs-nipt-block.cpp:1975-2355), ff == 0,
 * shard_check_every_pair = TRUE. */
static void shard_block_gibbs_diploid(sweep_t *S, const double *runif_block /* G - 1 */, double *work)
{
    const int Ks = S->Ks, G = S->G, R = S->R;
    double *etb = work;
    double mlc1 = 0, mlc2 = 0, mloc1 = 0, mloc2 = 0;
    for (int g = 0; g < G; g++) {
        mloc1 -= log(S->c[0][g]);
        mloc2 -= log(S->c[1][g]);
    }
    int in_flip_mode = 0, iRead = 0;
    const double prior = 1.0 / Ks;
    for (int g = 0; g < G; g++) {
        double oc1 = S->c[0][g], oc2 = S->c[1][g];
        double *a1 = S->alpha[0] + (size_t)Ks * g, *a2 = S->alpha[1] + (size_t)Ks * g;
        double *e1 = S->eg[0] + (size_t)Ks * g, *e2 = S->eg[1] + (size_t)Ks * g;
        if (g == 0) {
            for (int k = 0; k < Ks; k++) a1[k] = prior * e1[k];
            S->c[0][0] = 1 / col_sum(a1, Ks);
            for (int k = 0; k < Ks; k++) a1[k] *= S->c[0][0];
            for (int k = 0; k < Ks; k++) a2[k] = prior * e2[k];
            S->c[1][0] = 1 / col_sum(a2, Ks);
            for (int k = 0; k < Ks; k++) a2[k] *= S->c[1][0];
        } else {
            if (in_flip_mode) {
                for (int k = 0; k < Ks; k++) { double t = e1[k]; e1[k] = e2[k]; e2[k] = t; }
            }
            alpha_forward_one(g, Ks, S->alpha[0], S->tm, S->eg[0], S->c[0]);
            alpha_forward_one(g, Ks, S->alpha[1], S->tm, S->eg[1], S->c[1]);
        }
        mlc1 -= log(S->c[0][g]);
        mlc2 -= log(S->c[1][g]);
        int done_reads = 0;
        while (!done_reads) {
            if (iRead > R - 1) {
                done_reads = 1;
            } else {
                if (S->wif[iRead] == g) {
                    if (in_flip_mode) S->H[iRead] = 3 - S->H[iRead];
                    iRead++;
                }
                if (iRead > R - 1) done_reads = 1;
                else if (S->wif[iRead] > g) done_reads = 1;
            }
        }
        if (g < G - 1) {
            const double *b1 = S->beta[0] + (size_t)Ks * g, *b2 = S->beta[1] + (size_t)Ks * g;
            double s11, s22, s21, s12;   /* sum(alphaHat_t?.col(iGrid) % betaHat_t?.col(iGrid)), :2235-2238 */
            QO_ARMA_SUM(s11, Ks, k, a1[k] * b1[k]);
            QO_ARMA_SUM(s22, Ks, k, a2[k] * b2[k]);
            QO_ARMA_SUM(s21, Ks, k, a2[k] * b1[k]);
            QO_ARMA_SUM(s12, Ks, k, a1[k] * b2[k]);
            double pA1 = mlc1 + mloc1 + log(s11);
            double pA2 = mlc2 + mloc2 + log(s22);
            double pB1 = mlc2 + mloc1 + log(s21);
            double pB2 = mlc1 + mloc2 + log(s12);
            double diff = pB1 + pB2 - pA1 - pA2;
            double probs1 = 1, probs2 = exp(diff);
            double ps = probs1 + probs2;
            probs1 /= ps;
            in_flip_mode = runif_block[g] > probs1;
        }
        mloc1 += log(oc1);
        mloc2 += log(oc2);
    }
    for (int h = 0; h < 2; h++) {
        double *b = S->beta[h] + (size_t)Ks * (G - 1);
        for (int k = 0; k < Ks; k++) b[k] = S->c[h][G - 1];
        run_backward_haploid(S->beta[h], S->c[h], S->eg[h], S->tm, Ks, G, etb);
    }
}

/* rcpp_calculate_gibbs_small_genProbs_and_hapProbs_using_binary_objects
 * (gibbs-small.cpp:472-635), calculate_gamma_on_the_fly = TRUE */
static void calc_hapProbs(const qo_panel_t *p, const int32_t *which_1based, sweep_t *S, double *genProbsM,
                          double *genProbsF, double *hapProbs)
{
    const int Ks = S->Ks, G = S->G, T = p->nSNPs;
    const double eps = p->ref_error, ome = 1 - eps;
    double *gam = (double *)calloc((size_t)3 * Ks, sizeof(double));
    for (int g = 0; g < G; g++) {
        int s = 32 * g, e = 32 * (g + 1) - 1;
        if (e > T - 1) e = T - 1;
        int nLocal = e - s + 1;
        double g0[32] = {0}, g1[32] = {0}, g2[32] = {0}, h0[32] = {0}, h1[32] = {0}, h2[32] = {0};
        for (int h = 0; h < S->nH; h++) {
            double x = 1 / S->c[h][g];
            const double *a = S->alpha[h] + (size_t)Ks * g, *b = S->beta[h] + (size_t)Ks * g;
            for (int k = 0; k < Ks; k++) gam[(size_t)Ks * h + k] = (a[k] * b[k]) * x;
        }
        for (int k = 0; k < Ks; k++) {
            double gk0 = gam[k], gk1 = gam[(size_t)Ks + k], gk2 = gam[(size_t)2 * Ks + k];
            int kk = panel_code(p, which_1based[k] - 1, g);
            uint32_t w = kk > 0 ? (uint32_t)p->distinctHapsB[(size_t)p->nMaxDH * g + (kk - 1)]
                                : panel_special_word(p, which_1based[k] - 1, g);
            for (int b = 0; b < nLocal; b++, w >>= 1) {
                if ((w & 1u) == 0) { h0[b] += gk0; h1[b] += gk1; h2[b] += gk2; }
                else { g0[b] += gk0; g1[b] += gk1; g2[b] += gk2; }
            }
        }
        for (int b = 0; b < nLocal; b++) {
            g0[b] = g0[b] * ome + h0[b] * eps;
            g1[b] = g1[b] * ome + h1[b] * eps;
            g2[b] = g2[b] * ome + h2[b] * eps;
        }
        for (int b = 0; b < nLocal; b++) {
            double *gm = genProbsM + 3 * (size_t)(s + b), *gf = genProbsF + 3 * (size_t)(s + b);
            double *hp = hapProbs + 3 * (size_t)(s + b);
            gm[0] = (1 - g0[b]) * (1 - g1[b]);
            gm[1] = (g0[b] * (1 - g1[b]) + (1 - g0[b]) * g1[b]);
            gm[2] = g0[b] * g1[b];
            gf[0] = (1 - g0[b]) * (1 - g2[b]);
            gf[1] = (g0[b] * (1 - g2[b]) + (1 - g0[b]) * g2[b]);
            gf[2] = g0[b] * g2[b];
            hp[0] = g0[b]; hp[1] = g1[b]; hp[2] = g2[b];
        }
    }
    free(gam);
}

/* ---- NIPT block Gibbs (gibbs-nipt-block.cpp), ff > 0, block_approach = 6 ---------------------- */

/* rcpp_simple_quantile (gibbs-nipt-block.cpp:81-85): x[order(x)][int(n * q)] */
static int cmp_idx_asc(const void *a, const void *b, void *x)
{
    const double *v = (const double *)x;
    int i = *(const int *)a, j = *(const int *)b;
    if (v[i] < v[j]) return -1;
    if (v[i] > v[j]) return 1;
    return i - j;
}
static void sort_index(const double *x, int n, int *idx, int descending)
{
    for (int i = 0; i < n; i++) idx[i] = i;
    qsort_r(idx, (size_t)n, sizeof(int), cmp_idx_asc, (void *)x);
    if (descending) { /* ties keep increasing index (arma::sort_index is not stable; ties are measure zero here) */
        int *tmp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        int o = 0;
        for (int hi = n; hi > 0;) {
            int lo = hi - 1;
            while (lo > 0 && x[idx[lo - 1]] == x[idx[hi - 1]]) lo--;
            for (int i = lo; i < hi; i++) tmp[o++] = idx[i];
            hi = lo;
        }
        memcpy(idx, tmp, sizeof(int) * (size_t)n);
        free(tmp);
    }
}
double qo_simple_quantile(const double *x, int n, double q)
{
    int *idx = (int *)malloc(sizeof(int) * (size_t)n);
    sort_index(x, n, idx, 0);
    double v = x[idx[(int)(n * q)]];
    free(idx);
    return v;
}

/* rcpp_make_smoothed_rate (copied-from-stitch.cpp:446-518) */
void qo_make_smoothed_rate(const double *sigma_rate, const int32_t *L_grid, int nGrids, int shuffle_bin_radius,
                           double *smoothed_rate /* nGrids - 1 */)
{
    for (int g = 0; g < nGrids - 1; g++) {
        int focal_point = (L_grid[g] + L_grid[g + 1]) / 2;
        int left = g, bp_remaining = shuffle_bin_radius, bp_prev = focal_point, bp_to_add;
        double total = 0, acc = 0;
        while ((0 < bp_remaining) & (0 <= left)) {
            bp_to_add = bp_prev - L_grid[left];
            if ((bp_remaining - bp_to_add) < 0) { bp_to_add = bp_remaining; bp_remaining = 0; }
            else bp_remaining = bp_remaining - bp_to_add;
            acc = acc + bp_to_add * sigma_rate[left];
            total += bp_to_add;
            bp_prev = L_grid[left];
            left = left - 1;
        }
        int right = g + 1;
        bp_remaining = shuffle_bin_radius;
        bp_prev = focal_point;
        while ((0 < bp_remaining) & (right < nGrids)) {
            bp_to_add = L_grid[right] - bp_prev;
            if ((bp_remaining - bp_to_add) < 0) { bp_to_add = bp_remaining; bp_remaining = 0; }
            else bp_remaining = bp_remaining - bp_to_add;
            acc = acc + bp_to_add * sigma_rate[right - 1];
            total += bp_to_add;
            bp_prev = L_grid[right];
            right = right + 1;
        }
        smoothed_rate[g] = acc / total;
    }
}

/* rcpp_determine_where_to_stop (copied-from-stitch.cpp:522-567) */
static int determine_where_to_stop(const double *smoothed_rate, const uint8_t *available, int snp_best, double thresh,
                                   int nGrids, int is_left)
{
    const int mult = is_left ? 1 : -1;
    int snp_consider = snp_best;
    double val_cur, val_prev = smoothed_rate[snp_best];
    int snp_min = snp_consider;
    double val_min = smoothed_rate[snp_min];
    int c = 1, are_done = 0;
    while (!are_done) {
        snp_consider = snp_consider + (-1) * mult;
        val_cur = smoothed_rate[snp_consider];
        if (5 <= c) val_prev = smoothed_rate[snp_consider + 5 * mult];
        c += 1;
        if (val_cur < val_min) { snp_min = snp_consider; val_min = val_cur; }
        if ((snp_consider <= 2) | ((nGrids - 3) <= snp_consider)) are_done = 1;
        else if (!available[snp_consider + (-1) * mult]) are_done = 1;
        else if ((3 * val_min) < val_cur) are_done = 1;
        else if ((val_cur < thresh) & (val_prev < val_cur)) are_done = 1;
    }
    return snp_min;
}

/* Rcpp_define_blocked_snps_using_gamma_on_the_fly (gibbs-nipt-block.cpp:311-523) from rate2 on, at the level of grids:
 * blocked_snps[i] = blocked_grid[grid[i]] and every 32-SNP grid holds a SNP, so nothing is lost.  (The reference
 * multiplies rate2 by smooth_cm AFTER smoothed_rate was computed from it, :373-381: no effect.) */
void qo_define_blocked_grids(const double *rate2 /* nGrids - 1 */, const int32_t *L_grid, int nGrids,
                             int shuffle_bin_radius, double block_gibbs_quantile_prob, int32_t *blocked_grid /* nGrids */)
{
    const int n = nGrids - 1;
    double *sm = (double *)malloc(sizeof(double) * (size_t)n);
    uint8_t *available = (uint8_t *)calloc((size_t)n, 1);
    int *best = (int *)malloc(sizeof(int) * (size_t)n);
    int *to_keep = (int *)malloc(sizeof(int) * (size_t)(n + 3));
    int n_keep = 0;
    for (int g = 0; g < nGrids; g++) blocked_grid[g] = 0;
    qo_make_smoothed_rate(rate2, L_grid, nGrids, shuffle_bin_radius, sm);
    double break_thresh = 1;
    double d = qo_simple_quantile(sm, n, block_gibbs_quantile_prob);
    if (d < break_thresh) break_thresh = d;
    int nAvailable = 0;
    for (int i = 0; i < n; i++) {
        available[i] = 0; /* (< 0.01 and NA leave it FALSE) */
        if (break_thresh < sm[i]) available[i] = 1;
        nAvailable += available[i];
    }
    if (nAvailable == 0) goto done;
    sort_index(sm, n, best, 1);
    for (int iBest = 0; iBest < nAvailable; iBest++) {
        if (!available[best[iBest]]) continue;
        const int snp_best = best[iBest];
        const int a = snp_best - 1 > 0 ? snp_best - 1 : 0;
        const int b = snp_best + 1 < nGrids - 2 ? snp_best + 1 : nGrids - 2;
        int cnt = 0;
        for (int j = a; j <= b; j++) cnt += available[j];
        if (cnt == 3) {
            int left = determine_where_to_stop(sm, available, snp_best, break_thresh, nGrids, 1);
            int right = determine_where_to_stop(sm, available, snp_best, break_thresh, nGrids, 0);
            for (int j = left; j <= right; j++) available[j] = 0;
        } else {
            for (int j = a; j <= b; j++) available[j] = 0;
        }
        to_keep[n_keep++] = snp_best + 1;
    }
    {
        int mn = to_keep[0], mx = to_keep[0];
        for (int i = 1; i < n_keep; i++) { if (to_keep[i] < mn) mn = to_keep[i]; if (to_keep[i] > mx) mx = to_keep[i]; }
        if (mn != 0) to_keep[n_keep++] = 0;
        if (mx != nGrids - 1) to_keep[n_keep++] = nGrids - 1;
        for (int i = 1; i < n_keep; i++) { /* sort ascending */
            int v = to_keep[i], j = i - 1;
            while (j >= 0 && to_keep[j] > v) { to_keep[j + 1] = to_keep[j]; j--; }
            to_keep[j + 1] = v;
        }
        for (int i = 0; i < n_keep - 1; i++)
            for (int j = to_keep[i]; j <= to_keep[i + 1]; j++) blocked_grid[j] = i;
    }
done:
    free(sm); free(available); free(best); free(to_keep);
}

static double ceiling_point5(double x) { return ((double)(int)x < x) ? x + 0.5 : x; }

/* Rcpp_make_gibbs_considers (gibbs-nipt-block.cpp:1307-1553), do_removal = TRUE, at the level of grids (see above).
 * Arrays need capacity nGrids; returns n_blocks. */
int qo_make_gibbs_considers(const int32_t *blocked_grid_in, int nGrids, const int32_t *wif0, int nReads,
                            int32_t *grid_start, int32_t *grid_end, int32_t *reads_start, int32_t *reads_end,
                            int32_t *grid_where /* nGrids */)
{
    int n_blocks = blocked_grid_in[nGrids - 1] + 1;
    int iBlock = 0, start = 0;
    for (int g = 0; g < nGrids; g++) {
        int record = (g == nGrids - 1) || (blocked_grid_in[g] < blocked_grid_in[g + 1]);
        if (record) { grid_start[iBlock] = start; grid_end[iBlock] = g; start = g + 1; iBlock++; }
    }
    int32_t *blocked_grid = (int32_t *)calloc((size_t)nGrids, sizeof(int32_t));
    for (int b = 0; b < n_blocks; b++)
        for (int i = grid_start[b]; i <= grid_end[b]; i++) blocked_grid[i] = b;
    for (int b = 0; b < n_blocks; b++) reads_start[b] = reads_end[b] = -1;
    if (nReads > 0) {
        int previous_block_first_iRead = 0;
        int previous_block = blocked_grid[wif0[0]];
        for (int this_iRead = 1; this_iRead < nReads; this_iRead++) {
            int this_block = blocked_grid[wif0[this_iRead]];
            if (this_iRead == nReads - 1) {
                reads_start[this_block] = previous_block_first_iRead;
                reads_end[this_block] = this_iRead;
            } else if (previous_block < this_block) {
                reads_start[previous_block] = previous_block_first_iRead;
                reads_end[previous_block] = this_iRead - 1;
                previous_block_first_iRead = this_iRead;
                previous_block = blocked_grid[wif0[this_iRead]];
            }
        }
    }
    int n_to_remove = 0;
    int *w = (int *)malloc(sizeof(int) * (size_t)(n_blocks > 0 ? n_blocks : 1));
    for (int b = 0; b < n_blocks; b++) if (reads_start[b] == -1) w[n_to_remove++] = b;
    if (n_to_remove > 0 && n_to_remove < n_blocks) {
        int jBefore = 0;
        for (int jNow = 0; jNow < n_to_remove; jNow++) {
            int todo;
            if (jNow == n_to_remove - 1) todo = 1;
            else if ((w[jNow + 1] - w[jNow]) == 1) { todo = 0; jBefore -= 1; }
            else todo = 1;
            if (todo) {
                int s1 = w[jBefore], e1 = w[jNow];
                double x = ceiling_point5(0.5 * (double)(grid_start[s1] + grid_end[e1]));
                if (s1 == 0) { s1 = 1; x = 0; }
                if (e1 == n_blocks - 1) { e1 = e1 - 1; x = grid_end[n_blocks - 1]; }
                grid_start[e1 + 1] = (int32_t)x;
                grid_end[s1 - 1] = (int32_t)(x - 1);
                jBefore = jNow;
            }
            jBefore += 1;
        }
        int o = 0;
        for (int b = 0; b < n_blocks; b++) {
            if (reads_start[b] == -1) continue;
            reads_start[o] = reads_start[b]; reads_end[o] = reads_end[b];
            grid_start[o] = grid_start[b]; grid_end[o] = grid_end[b];
            o++;
        }
        n_blocks = o;
    }
    for (int g = 0; g < nGrids; g++) grid_where[g] = -1;
    for (int b = 0; b < n_blocks; b++) grid_where[grid_end[b]] = b;
    free(w); free(blocked_grid);
    return n_blocks;
}

/* rcpp_get_log_p_H_class2 (gibbs-nipt-block.cpp:170-207) */
double qo_get_log_p_H_class2(int n1, int n2, int n3, int n4, int n5, int n6, double ff)
{
    if (ff == 0)
        return 0 + n1 * log(0.5) + n2 * log(0.5 - ff * 0.5) + n3 * log(0.001) + n4 * log(1 - ff * 0.5) +
               n5 * log(1 * 0.5 + ff * 0.5) + n6 * log(1 * 0.5);
    if (ff == 1)
        return 0 + n1 * log(0.5) + n2 * log(0.001) + n3 * log(ff * 0.5) + n4 * log(1 - ff * 0.5) +
               n5 * log(1 * 0.5 + ff * 0.5) + n6 * log(1 * 0.5);
    return 0 + n1 * log(0.5) + n2 * log(0.5 - ff * 0.5) + n3 * log(ff * 0.5) + n4 * log(1 - ff * 0.5) +
           n5 * log(1 * 0.5 + ff * 0.5) + n6 * log(1 * 0.5);
}

static const int kRR[6][3] = {{1, 2, 3}, {1, 3, 2}, {2, 1, 3}, {2, 3, 1}, {3, 1, 2}, {3, 2, 1}};   /* :1755-1761 */
static const int kRX[6][3] = {{1, 2, 3}, {1, 3, 2}, {2, 1, 3}, {3, 1, 2}, {2, 3, 1}, {3, 2, 1}};   /* :752-758 */

/* the label / class relabelling of choice ir (0-based), :760-769 */
void qo_zero_based_swap(int ir_chosen, int swap[8])
{
    swap[0] = 0;
    swap[1] = kRX[ir_chosen][0]; swap[2] = kRX[ir_chosen][1]; swap[3] = kRX[ir_chosen][2];
    swap[4] = 7 - kRX[ir_chosen][2]; swap[5] = 7 - kRX[ir_chosen][1]; swap[6] = 7 - kRX[ir_chosen][0];
    swap[7] = 7;
}

/* Rcpp::sample(1:3, 1, FALSE, probs) given its one uniform: probabilities normalised, sorted in decreasing order
 * (Rf_revsort), first j with u <= cumulative mass (Rcpp sugar sample.h, SampleReplace for size 1). */
int qo_sample3(const double probs[3], double u)
{
    double p[3] = {probs[0], probs[1], probs[2]};
    int perm[3] = {1, 2, 3};
    double sum = p[0] + p[1] + p[2];
    for (int i = 0; i < 3; i++) p[i] /= sum;
    for (int i = 1; i < 3; i++) { /* stable insertion sort, descending */
        double v = p[i]; int q = perm[i], j = i - 1;
        while (j >= 0 && p[j] < v) { p[j + 1] = p[j]; perm[j + 1] = perm[j]; j--; }
        p[j + 1] = v; perm[j + 1] = q;
    }
    p[1] += p[0]; p[2] += p[1];
    int j;
    for (j = 0; j < 2; j++) if (u <= p[j]) break;
    return perm[j];
}

/*
 * Rcpp_block_gibbs_resampler (gibbs-nipt-block.cpp:1636-1967) with Rcpp_gibbs_block_forward_one (:1122-1253),
 * Rcpp_consider_block_relabelling (:590-949), Rcpp_reset_local_variables (:1257-1292) and
 * rcpp_sample_H_using_H_class (:213-246), for the production arguments: ff > 0, block_approach = 6,
 * consider_total_relabelling = FALSE, resample_H_using_H_class = TRUE.
 *   runif_block     one uniform per block (the reference draws nReads of them and uses the first n_blocks, :3016)
 *   runif_resample  one uniform per read, used by the reads whose class leaves a choice (:226-243)
 */
static void block_gibbs_resampler_nipt(sweep_t *S, double ff, const int32_t *blocked_grid, const double *runif_block,
                                       const double *runif_resample, const double **draw_next /* NULL, or the sequential stream */)
{
    const int Ks = S->Ks, G = S->G, R = S->R;
    const double one_over_K = 1 / (double)Ks, prior = 1.0 / Ks;
    int32_t *gs = (int32_t *)malloc(sizeof(int32_t) * (size_t)G * 5);
    int32_t *ge = gs + G, *rs = ge + G, *re = rs + G, *where = re + G;
    const int n_blocks = qo_make_gibbs_considers(blocked_grid, G, S->wif, R, gs, ge, rs, re, where);
    double *alphaStore = (double *)calloc((size_t)18 * Ks, sizeof(double));   /* [ir][h][k] */
    double *log_cStore = (double *)calloc((size_t)18 * G, sizeof(double));    /* [ir][h][g] */
    double *eLocal = (double *)malloc(sizeof(double) * (size_t)3 * Ks);
    double logC_before[3] = {0, 0, 0}, logC_after[3] = {0, 0, 0};
    /* logC_after(h) = sum(log(c_h)) (:1819-1821): c_h is an arma::rowvec, log() an element-wise expression */
    for (int h = 0; h < 3; h++) QO_ARMA_SUM(logC_after[h], G, g, log(S->c[h][g]));
    double sum_H[3] = {0, 0, 0};
    for (int r = 0; r < R; r++) sum_H[S->H[r] - 1] += 1;
    int ever_changed = 0;
#define AS(ir, h) (alphaStore + ((size_t)(ir) * 3 + (h)) * Ks)
#define LC(ir, h, g) log_cStore[((size_t)(ir) * 3 + (h)) * G + (g)]
    for (int g = 0; g < G; g++) {
        for (int i = 0; i < 3; i++) memcpy(eLocal + (size_t)i * Ks, S->eg[i] + (size_t)Ks * g, sizeof(double) * Ks);
        /* Rcpp_gibbs_block_forward_one, block_approach = 6 */
        for (int ir = 0; ir < 6; ir++)
            for (int i = 0; i < 3; i++) {
                const int h = kRR[ir][i] - 1;
                double *a = AS(ir, h);
                const double *e = eLocal + (size_t)i * Ks;
                if (g == 0) {
                    for (int k = 0; k < Ks; k++) a[k] = prior * e[k];
                } else {
                    const double t0 = S->tm[2 * (size_t)(g - 1)], t1 = S->tm[2 * (size_t)(g - 1) + 1];
                    for (int k = 0; k < Ks; k++) a[k] = e[k] * (t0 * a[k] + t1 * one_over_K);
                }
                double d = 1 / col_sum(a, Ks);
                LC(ir, h, g) = log(d);
                for (int k = 0; k < Ks; k++) a[k] = d * a[k];
            }
        if (where[g] > -1) {
            const int iBlock = where[g];
            const int grid_start = gs[iBlock], grid_end = ge[iBlock], read_start = rs[iBlock], read_end = re[iBlock];
            /* ---- Rcpp_consider_block_relabelling ---- */
            double Pm[6][3], P[6] = {0, 0, 0, 0, 0, 0};
            for (int ir = 0; ir < 6; ir++)
                for (int i = 0; i < 3; i++) {
                    double logC_inside = 0;
                    for (int g2 = grid_start; g2 <= grid_end; g2++) logC_inside += LC(ir, i, g2);
                    const double *a = AS(ir, i), *b = S->beta[i] + (size_t)Ks * g;
                    double dot;   /* sum(alphaStore.slice(ir).col(i) % betaHatLocal.col(i)), :669 */
                    QO_ARMA_SUM(dot, Ks, k, a[k] * b[k]);
                    Pm[ir][i] = log(dot) + -logC_before[i] + -logC_inside + -logC_after[i];
                    P[ir] += Pm[ir][i];
                }
            int ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = read_start; r <= read_end; r++) ns[S->H_class[r]]++;
            double clp[6], cp[6];
            for (int ir = 0; ir < 6; ir++) {
                const int *q = kRR[ir];
                clp[ir] = qo_get_log_p_H_class2(ns[q[0]], ns[q[1]], ns[q[2]], ns[7 - q[2]], ns[7 - q[1]], ns[7 - q[0]], ff) + P[ir];
            }
            double mx = clp[0];
            for (int ir = 1; ir < 6; ir++) if (clp[ir] > mx) mx = clp[ir];
            double a = -mx, tot = 0;
            for (int ir = 0; ir < 6; ir++) {
                clp[ir] += a;
                if (clp[ir] < -100) clp[ir] = -100;
                cp[ir] = exp(clp[ir]);
                tot += cp[ir];
            }
            double d = 1 / tot;
            for (int ir = 0; ir < 6; ir++) cp[ir] *= d;
            const double chance = runif_block[iBlock];
            double cum[6];
            cum[0] = cp[0];
            for (int ir = 1; ir < 6; ir++) cum[ir] = cp[ir] + cum[ir - 1];
            int ir_chosen = 0;
            for (int ir = 5; ir >= 0; ir--) if (chance < cum[ir]) ir_chosen = ir;
            int swap[8];
            qo_zero_based_swap(ir_chosen, swap);
            if ((ever_changed == 1) | (ir_chosen != 0)) {
                ever_changed = 1;
                int iRead = read_start;
                int wif_read = S->wif[iRead];
                for (int g2 = grid_start; g2 <= grid_end; g2++) {
                    for (size_t i = 0; i < (size_t)3 * Ks; i++) eLocal[i] = 1;
                    while ((iRead <= (R - 1)) & (wif_read < g2)) {
                        iRead += 1;
                        if (iRead < (R - 1)) wif_read = S->wif[iRead];
                    }
                    while ((iRead <= (R - 1)) & (wif_read == g2)) {
                        const int h = swap[S->H[iRead]] - 1;
                        const double *er = S->eMatRead + (size_t)Ks * iRead;
                        double *e = eLocal + (size_t)h * Ks;
                        for (int k = 0; k < Ks; k++) e[k] *= er[k];
                        iRead += 1;
                        if (iRead <= (R - 1)) wif_read = S->wif[iRead];
                    }
                    for (int h = 0; h < 3; h++) {
                        double *e = S->eg[h] + (size_t)Ks * g2, *al = S->alpha[h] + (size_t)Ks * g2;
                        memcpy(e, eLocal + (size_t)h * Ks, sizeof(double) * Ks);
                        if (g2 == 0) {
                            for (int k = 0; k < Ks; k++) al[k] = prior * e[k];
                        } else {
                            const double t0 = S->tm[2 * (size_t)(g2 - 1)], t1 = S->tm[2 * (size_t)(g2 - 1) + 1];
                            const double *ap = S->alpha[h] + (size_t)Ks * (g2 - 1);
                            for (int k = 0; k < Ks; k++) al[k] = e[k] * (t0 * ap[k] + t1 * prior);
                        }
                        S->c[h][g2] = 1 / col_sum(al, Ks);
                        for (int k = 0; k < Ks; k++) al[k] *= S->c[h][g2];
                    }
                }
                for (int r = read_start; r <= read_end; r++) {
                    const int lost = S->H[r] - 1, gained = swap[S->H[r]] - 1;
                    S->H_class[r] = swap[S->H_class[r]];
                    S->H[r] = gained + 1;
                    sum_H[gained] += 1.0;
                    sum_H[lost] -= 1.0;
                }
            }
            /* Rcpp_reset_local_variables */
            if ((iBlock + 1) < n_blocks)
                for (int ir = 0; ir < 6; ir++)
                    for (int h = 0; h < 3; h++) {
                        memcpy(AS(ir, h), S->alpha[h] + (size_t)Ks * g, sizeof(double) * Ks);
                        LC(ir, h, g) = log(S->c[h][g]);
                    }
            for (int g2 = grid_start; g2 <= grid_end; g2++)
                for (int h = 0; h < 3; h++) logC_before[h] += log(S->c[h][g2]);
        }
        for (int h = 0; h < 3; h++) logC_after[h] -= log(S->c[h][g]);
    }
#undef AS
#undef LC
    /* resample H given its class, rebuild everything (:1898-1934) */
    {
        const double probs07[3] = {0.5, 0.5 - ff * 0.5, ff * 0.5}, probs4[3] = {0.5, 0.5 - 0.5 * ff, 0};
        const double probs5[3] = {0.5, 0, 0.5 * ff}, probs6[3] = {0, 0.5 - ff * 0.5, ff * 0.5};
        for (int r = 0; r < R; r++) {
            const int hc = S->H_class[r];
            if (hc >= 1 && hc <= 3) { S->H[r] = hc; continue; }
            /* (only these reads draw: with a sequential stream the uniform is the NEXT one, as Rcpp::sample takes it from R) */
            const double u = draw_next ? *(*draw_next)++ : runif_resample[r];
            if (hc == 0 || hc == 7) S->H[r] = qo_sample3(probs07, u);
            else if (hc == 4) S->H[r] = qo_sample3(probs4, u);
            else if (hc == 5) S->H[r] = qo_sample3(probs5, u);
            else S->H[r] = qo_sample3(probs6, u);
        }
        for (int h = 0; h < 3; h++) for (size_t i = 0; i < (size_t)Ks * G; i++) S->eg[h][i] = 1;
        for (int r = 0; r < R; r++) {
            double *e = S->eg[S->H[r] - 1] + (size_t)Ks * S->wif[r];
            const double *er = S->eMatRead + (size_t)Ks * r;
            for (int k = 0; k < Ks; k++) e[k] *= er[k];
        }
        for (int h = 0; h < 3; h++) run_forward_haploid(S->alpha[h], S->c[h], S->eg[h], S->tm, Ks, G, 0);
    }
    /* re-run backward (:1939-1954): the QUILT_faster form overwrites the generic one's result */
    for (int h = 0; h < 3; h++) {
        double *b = S->beta[h] + (size_t)Ks * (G - 1);
        for (int k = 0; k < Ks; k++) b[k] = S->c[h][G - 1];
        run_backward_haploid_faster(S->beta[h], S->c[h], S->eg[h], S->tm, S->grid_has_read, Ks, G, eLocal);
    }
    (void)sum_H;
    free(gs); free(alphaStore); free(log_cStore); free(eLocal);
}

/* rate2 of Rcpp_define_blocked_snps_using_gamma_on_the_fly (:347-363) */
static void block_rate2(const sweep_t *S, double ff, double *rate2 /* G - 1 */)
{
    const int Ks = S->Ks, G = S->G;
    for (int g = 0; g < G - 1; g++) rate2[g] = 0;
    for (int h = 0; h < (ff > 0 ? 3 : 2); h++)
        for (int g = 0; g < G - 2; g++) {
            const double d = S->tm[2 * (size_t)g];
            const double *a = S->alpha[h] + (size_t)Ks * g, *b = S->beta[h] + (size_t)Ks * (g + 1);
            const double *e = S->eg[h] + (size_t)Ks * (g + 1);
            double s;   /* sum(alphaHat_t.col(iGrid) % betaHat_t.col(iGrid + 1) % eMatGrid_t.col(iGrid + 1)), :353-362 */
            QO_ARMA_SUM(s, Ks, k, a[k] * b[k] * e[k]);
            rate2[g] += 1 - d * s;
        }
}

/* ---- rare + common SNPs: the final all-SNP Gibbs of QUILT2 (rare_common.R:109-420) ------------- */

/* rare_per_snp_info (rare_common.R:313-322) for one which_haps_to_use: per all-SNP index the small-panel rows
 * (0-based here, in increasing order, as the R loop appends them) whose haplotype carries the alt.  An empty list
 * here is R's lone -1 ("k_with_alt.length() == 1"). */
typedef struct {
    int64_t *ptr;
    int32_t *k;
} rare_snp_lists_t;

static rare_snp_lists_t build_rare_per_snp(const qo_rare_common_t *rc, const int32_t *which_1based, int Ks)
{
    rare_snp_lists_t L;
    const int T = rc->nSNPs_all;
    L.ptr = (int64_t *)calloc((size_t)T + 1, sizeof(int64_t));
    for (int k = 0; k < Ks; k++) {
        const int hap = which_1based[k] - 1;
        for (int64_t i = rc->rare_ptr[hap]; i < rc->rare_ptr[hap + 1]; i++) L.ptr[rc->rare_snp_1based[i]]++;
    }
    for (int t = 0; t < T; t++) L.ptr[t + 1] += L.ptr[t];
    L.k = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L.ptr[T] > 0 ? L.ptr[T] : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(T > 0 ? T : 1));
    for (int t = 0; t < T; t++) cur[t] = L.ptr[t];
    for (int k = 0; k < Ks; k++) {
        const int hap = which_1based[k] - 1;
        for (int64_t i = rc->rare_ptr[hap]; i < rc->rare_ptr[hap + 1]; i++) L.k[cur[rc->rare_snp_1based[i] - 1]++] = k;
    }
    free(cur);
    return L;
}

/* Rcpp_make_eMatRead_t_for_final_rare_common_gibbs_using_objects (gibbs-small.cpp:270-460) */
void qo_make_eMatRead_t_rare_common(
    const qo_panel_t *p, const qo_rare_common_t *rc, const int32_t *which_haps_to_use_1based, int Ks, int nReads,
    const int32_t *read_ptr, const int32_t *u, const int32_t *bq, int rescale_eMatRead_t, int Jmax,
    double maxDifferenceBetweenReads, double *eMatRead_t /* Ks x nReads, pre-filled with 1 */)
{
    double pR = 1, pA = 1; /* carried over between reads when bq == 0 (:293-294) */
    const double d2 = 1 / maxDifferenceBetweenReads;
    const double ref_error = p->ref_error, one_minus_ref_error = 1 - ref_error;
    rare_snp_lists_t L = build_rare_per_snp(rc, which_haps_to_use_1based, Ks);
    int *codes = (int *)malloc(sizeof(int) * (size_t)Ks);
    for (int r = 0; r < nReads; r++) {
        const int32_t *ru = u + read_ptr[r], *rbq = bq + read_ptr[r];
        int J = read_ptr[r + 1] - read_ptr[r] - 1;
        double *col = eMatRead_t + (size_t)Ks * r;
        int g_prev = -1;
        if (J >= Jmax) J = Jmax;
        for (int j = 0; j <= J; j++) {
            if (rbq[j] < 0) {
                double eps = pow(10, (double)rbq[j] / 10);
                pR = 1 - eps;
                pA = eps / 3;
            }
            if (rbq[j] > 0) {
                double eps = pow(10, -(double)rbq[j] / 10);
                pR = eps / 3;
                pA = 1 - eps;
            }
            const int snp = ru[j];
            if (rc->snp_is_common[snp]) {
                const int uc = rc->common_snp_index[snp] - 1; /* index among the common SNPs (:347) */
                const int g = uc / 32;
                if (g != g_prev)
                    for (int k = 0; k < Ks; k++) codes[k] = panel_code(p, which_haps_to_use_1based[k] - 1, g);
                g_prev = g;
                for (int k = 0; k < Ks; k++) {
                    double e;
                    if (codes[k] > 0) {
                        e = p->distinctHapsIE[(size_t)p->nMaxDH * uc + (codes[k] - 1)];
                    } else {
                        uint32_t w = panel_special_word(p, which_haps_to_use_1based[k] - 1, g);
                        e = ((w >> (uc % 32)) & 1u) ? 1 - ref_error : ref_error;
                    }
                    col[k] *= (e * pA + (1 - e) * pR);
                }
            } else {
                const int64_t a0 = L.ptr[snp], a1 = L.ptr[snp + 1];
                if (a0 == a1) { /* no selected haplotype carries the alt (:385-389) */
                    if (!rescale_eMatRead_t) {
                        double xe1 = ref_error * pA + one_minus_ref_error * pR;
                        for (int k = 0; k < Ks; k++) col[k] *= xe1;
                    }
                } else { /* everyone as ref, then the carriers re-done (:390-400) */
                    double xe1 = ref_error * pA + one_minus_ref_error * pR;
                    double xe2 = one_minus_ref_error * pA + ref_error * pR;
                    for (int k = 0; k < Ks; k++) col[k] *= xe1;
                    for (int64_t i = a0; i < a1; i++) col[L.k[i]] *= xe2 / xe1;
                }
            }
        }
        if (rescale_eMatRead_t) {
            double x = 0;
            for (int k = 0; k < Ks; k++) if (col[k] > x) x = col[k];
            double d1 = 1 / x;
            if (isinf(x) || x == 0 || isinf(d1)) {
                for (int k = 0; k < Ks; k++) col[k] = 1;
            } else {
                for (int k = 0; k < Ks; k++) {
                    col[k] *= d1;
                    if (col[k] < d2) col[k] = d2;
                }
            }
        }
    }
    free(codes); free(L.ptr); free(L.k);
}

/* rcpp_calculate_genProbs_and_hapProbs_final_rare_common (gibbs-small.cpp:711-867); the outputs start at 0 */
static void calc_hapProbs_rare_common(const qo_panel_t *p, const qo_rare_common_t *rc, const int32_t *which_1based,
                                      sweep_t *S, double *genProbsM, double *genProbsF, double *hapProbs)
{
    const int Ks = S->Ks, T = rc->nSNPs_all, nH = S->nH;
    const double ref_error = p->ref_error, one_minus_2_times_ref_error = 1 - 2 * ref_error;
    rare_snp_lists_t L = build_rare_per_snp(rc, which_1based, Ks);
    double *gam = (double *)calloc((size_t)3 * Ks, sizeof(double));
    for (size_t i = 0; i < (size_t)3 * T; i++) hapProbs[i] = genProbsM[i] = genProbsF[i] = 0;
    int g_prev = -1;
    for (int snp = 0; snp < T; snp++) {
        const int g = snp / 32;
        double *hp = hapProbs + 3 * (size_t)snp;
        if (g != g_prev) {
            for (int h = 0; h < nH; h++) {
                double x = 1 / S->c[h][g];
                const double *a = S->alpha[h] + (size_t)Ks * g, *b = S->beta[h] + (size_t)Ks * g;
                for (int k = 0; k < Ks; k++) gam[(size_t)Ks * h + k] = (a[k] * b[k]) * x;
            }
            g_prev = g;
        }
        if (rc->snp_is_common[snp]) {
            const int cs = rc->common_snp_index[snp] - 1, cg = cs / 32;
            for (int k = 0; k < Ks; k++) {
                const int kk = panel_code(p, which_1based[k] - 1, cg);
                double d;
                if (kk > 0) {
                    d = p->distinctHapsIE[(size_t)p->nMaxDH * cs + (kk - 1)];
                } else {
                    uint32_t w = panel_special_word(p, which_1based[k] - 1, cg);
                    d = ((w >> (cs % 32)) & 1u) ? 1 - ref_error : ref_error;
                }
                for (int h = 0; h < nH; h++) hp[h] += gam[(size_t)Ks * h + k] * d;
            }
        } else {
            const int64_t a0 = L.ptr[snp], a1 = L.ptr[snp + 1];
            if (a0 == a1) {
                for (int h = 0; h < nH; h++) hp[h] = ref_error;
            } else {
                for (int k = 0; k < Ks; k++)
                    for (int h = 0; h < nH; h++) hp[h] += gam[(size_t)Ks * h + k] * ref_error;
                for (int64_t i = a0; i < a1; i++)
                    for (int h = 0; h < nH; h++) hp[h] += gam[(size_t)Ks * h + L.k[i]] * one_minus_2_times_ref_error;
            }
        }
        {
            double h1 = hp[0], h2 = hp[1];
            double *gm = genProbsM + 3 * (size_t)snp;
            gm[0] = (1 - h1) * (1 - h2);
            gm[1] = h1 * (1 - h2) + h2 * (1 - h1);
            gm[2] = h1 * h2;
        }
        if (nH == 3) {
            double h1 = hp[0], h2 = hp[2];
            double *gf = genProbsF + 3 * (size_t)snp;
            gf[0] = (1 - h1) * (1 - h2);
            gf[1] = h1 * (1 - h2) + h2 * (1 - h1);
            gf[2] = h1 * h2;
        }
    }
    free(gam); free(L.ptr); free(L.k);
}

/*
 * rcpp_forwardBackwardGibbsNIPT (gibbs-nipt.cpp:2395-3307), production path.
 * Returns 0 ok, 1 underflow_problem (:2959-2969), -2 unsupported (NIPT block Gibbs).
 */
int qo_gibbs(const qo_panel_t *p, const qo_gibbs_args_t *a, int32_t *H, int32_t *H_class,
             double *alphaHat_t[3], double *betaHat_t[3], double *eMatGrid_t[3], double *c_out[3],
             double *eMatRead_t, int32_t *read_category_out, double *hapProbs_t, double *genProbsM_t,
             double *genProbsF_t)
{
    const qo_rare_common_t *rc = a->rc;
    const int Ks = a->Ks, G = rc ? rc->nGrids_all : p->nGrids, R = a->nReads, T = rc ? rc->nSNPs_all : p->nSNPs;
    const int nH = a->sample_is_diploid ? 2 : 3;
    const int n_its = a->n_gibbs_burn_in_its + a->n_gibbs_sample_its;
    sweep_t S;
    memset(&S, 0, sizeof S);
    S.Ks = Ks; S.G = G; S.R = R; S.nH = nH;
    S.eMatRead = eMatRead_t; S.wif = a->wif; S.grid_has_read = a->grid_has_read;
    S.tm = rc ? rc->transMatRate_t_all : p->transMatRate_t;
    S.H = H; S.H_class = H_class; S.sample_is_diploid = a->sample_is_diploid;
    S.class_sum_cutoff = a->class_sum_cutoff;
    const double ff = a->ff;
    S.prior_probs[0] = 0.5; S.prior_probs[1] = (1 - ff) * 0.5; S.prior_probs[2] = ff * 0.5;
    {
        const double *pp = S.prior_probs;
        double r[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1},
                          {pp[0] / (pp[0] + pp[1]), pp[1] / (pp[0] + pp[1]), 0},
                          {pp[0] / (pp[0] + pp[2]), 0, pp[2] / (pp[0] + pp[2])},
                          {0, pp[1] / (pp[1] + pp[2]), pp[2] / (pp[1] + pp[2])},
                          {pp[0], pp[1], pp[2]}};
        memcpy(S.rlc, r, sizeof r);
    }
    double *c3_local = NULL;
    for (int h = 0; h < 3; h++) {
        S.alpha[h] = alphaHat_t[h]; S.beta[h] = betaHat_t[h]; S.eg[h] = eMatGrid_t[h]; S.c[h] = c_out[h];
    }
    for (int h = 0; h < 3; h++) for (int g = 0; g < G; g++) S.c[h][g] = 0; /* arma::zeros (:2676-2678) */
    for (int i = 0; i < R; i++) H_class[i] = 0;

    /* emissions (pass_in_eMatRead_t = FALSE: ones, then multiply) */
    for (size_t i = 0; i < (size_t)Ks * R; i++) eMatRead_t[i] = 1;
    if (rc) /* make_eMatRead_t_rare_common (gibbs-nipt.cpp:2805-2806) */
        qo_make_eMatRead_t_rare_common(p, rc, a->which_haps_to_use_1based, Ks, R, a->read_ptr, a->u, a->bq,
                                       a->rescale_eMatRead_t, a->Jmax, a->maxDifferenceBetweenReads, eMatRead_t);
    else
        qo_make_eMatRead_t_for_gibbs_using_objects(p, a->which_haps_to_use_1based, Ks, R, a->read_ptr, a->u, a->bq,
                                                   a->rescale_eMatRead_t, a->Jmax, a->maxDifferenceBetweenReads,
                                                   eMatRead_t);
    int32_t *n_non1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
    int32_t *idx_non1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)Ks * (R > 0 ? R : 1));
    int32_t *cat = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
    qo_evaluate_read_variability(eMatRead_t, Ks, R, n_non1, idx_non1, cat);
    if (a->disable_read_category_usage) for (int i = 0; i < R; i++) cat[i] = 0;
    if (read_category_out) memcpy(read_category_out, cat, sizeof(int32_t) * (size_t)R);
    S.read_category = cat; S.n_non1 = n_non1; S.idx_non1 = idx_non1;

    double *work = (double *)malloc(sizeof(double) * (size_t)10 * Ks);

    /* rcpp_gibbs_nipt_initialize (:1629-1750) */
    for (int h = 0; h < 3; h++) for (size_t i = 0; i < (size_t)Ks * G; i++) S.eg[h][i] = 1;
    if (!a->gibbs_initialize_iteratively) {
        for (int r = 0; r < R; r++) { /* rcpp_make_eMatGrid_t (copied-from-stitch.cpp:262-281), bound = false */
            int h = H[r] - 1;
            double *e = S.eg[h] + (size_t)Ks * a->wif[r];
            const double *er = eMatRead_t + (size_t)Ks * r;
            for (int k = 0; k < Ks; k++) e[k] *= er[k];
        }
        for (int h = 0; h < nH; h++) { /* rcpp_initialize_gibbs_forward_backward (:453-487) */
            run_forward_haploid(S.alpha[h], S.c[h], S.eg[h], S.tm, Ks, G, 0);
            double *b = S.beta[h] + (size_t)Ks * (G - 1);
            for (int k = 0; k < Ks; k++) b[k] = S.c[h][G - 1];
            run_backward_haploid(S.beta[h], S.c[h], S.eg[h], S.tm, Ks, G, work);
        }
    } else {
        for (int h = 0; h < nH; h++) {
            for (size_t i = 0; i < (size_t)Ks * G; i++) { S.alpha[h][i] = 1; S.beta[h][i] = 1; }
            for (int g = 0; g < G; g++) S.c[h][g] = 1;
            run_forward_haploid(S.alpha[h], S.c[h], S.eg[h], S.tm, Ks, G, 1);
        }
    }

    int status = 0;
    int shard_it = 0;
    const double *stream_at = a->runif_stream;
    for (int it = 0; it < n_its; it++) {
        gibbs_iterate(&S, it, a->runif_reads, a->gibbs_initialize_iteratively, a->first_read, work);
        /* underflow check (:2959-2969): sum(c3) is only looked at when ff == 0 */
        for (int h = 0; h < 3; h++) {
            if (h == 2 && ff != 0) continue;
            double s = 0;
            for (int g = 0; g < G; g++) s += S.c[h][g];
            if (!isfinite(s)) status = 1;
        }
        if (status == 1) break;
        int to_block = 0;
        if (a->perform_block_gibbs)
            for (int i = 0; i < a->n_block_gibbs_iterations; i++) if (a->block_gibbs_iterations[i] == it) to_block = 1;
        if (to_block && !(a->sample_is_diploid && ff == 0)) {
            /* NIPT (gibbs-nipt.cpp:3003-3021): define the blocks from the current state, then the block resampler;
             * do_shard_block_gibbs is FALSE for ff > 0 (functions.R:2552-2556) */
            if (!a->L_grid || (!a->runif_stream && (!a->runif_block || !a->runif_resample))) { status = -2; break; }
            double *rate2 = (double *)malloc(sizeof(double) * (size_t)(G > 1 ? G - 1 : 1));
            int32_t *blocked = (int32_t *)malloc(sizeof(int32_t) * (size_t)G);
            block_rate2(&S, ff, rate2);
            qo_define_blocked_grids(rate2, a->L_grid, G, a->shuffle_bin_radius, a->block_gibbs_quantile_prob, blocked);
            if (a->runif_stream) {
                const double *blk = stream_at;   /* this pass's runif_block, then its draws */
                stream_at += R;
                block_gibbs_resampler_nipt(&S, ff, blocked, blk, NULL, &stream_at);
            } else {
                block_gibbs_resampler_nipt(&S, ff, blocked, a->runif_block + (size_t)shard_it * R,
                                           a->runif_resample + (size_t)shard_it * R, NULL);
            }
            shard_it++;
            free(rate2); free(blocked);
        } else if (to_block) {
            /* diploid: Rcpp_block_gibbs_resampler is the identity (see header) */
            if (a->do_shard_block_gibbs) {
                shard_block_gibbs_diploid(&S, a->runif_shard + (size_t)shard_it * (G - 1), work);
                shard_it++;
            }
        }
        if (it + 1 > a->n_gibbs_burn_in_its) {
            if (rc) /* gibbs-nipt.cpp:2305-2318 */
                calc_hapProbs_rare_common(p, rc, a->which_haps_to_use_1based, &S, genProbsM_t, genProbsF_t, hapProbs_t);
            else
                calc_hapProbs(p, a->which_haps_to_use_1based, &S, genProbsM_t, genProbsF_t, hapProbs_t);
        }
    }
    (void)T; (void)c3_local;
    free(work); free(n_non1); free(idx_non1); free(cat);
    if (a->runif_stream && a->runif_stream_used) *a->runif_stream_used = (int64_t)(stream_at - a->runif_stream);
    return status;
}

/* rcpp_make_eMatRead_t (copied-from-stitch.cpp:115-229) for dense haplotype dosages, run_pseudo_haploid
 * = false.  eHaps is K x nSNPs (column-major); eMatRead_t K x nReads must be pre-filled with 1. */
void qo_make_eMatRead_t_dense(const double *eHaps, int K, int nReads, const int32_t *read_ptr, const int32_t *u,
                              const int32_t *bq, double maxDifferenceBetweenReads, int Jmax,
                              int rescale_eMatRead_t, double *eMatRead_t)
{
    double pR = 1, pA = 1;
    const double d2 = 1 / maxDifferenceBetweenReads;
    for (int r = 0; r < nReads; r++) {
        const int32_t *ru = u + read_ptr[r], *rbq = bq + read_ptr[r];
        int J = read_ptr[r + 1] - read_ptr[r] - 1;
        double *col = eMatRead_t + (size_t)K * r;
        if (J >= Jmax) J = Jmax;
        for (int j = 0; j <= J; j++) {
            if (rbq[j] < 0) { double eps = pow(10, (double)rbq[j] / 10); pR = 1 - eps; pA = eps / 3; }
            if (rbq[j] > 0) { double eps = pow(10, -(double)rbq[j] / 10); pR = eps / 3; pA = 1 - eps; }
            const double *e = eHaps + (size_t)K * ru[j];
            for (int k = 0; k < K; k++) col[k] *= (e[k] * pA + (1 - e[k]) * pR);
        }
        if (rescale_eMatRead_t) {
            double x = 0;
            for (int k = 0; k < K; k++) if (col[k] > x) x = col[k];
            double d1 = 1 / x;
            if (isinf(x) || x == 0 || isinf(d1)) {
                for (int k = 0; k < K; k++) col[k] = 1;
            } else {
                for (int k = 0; k < K; k++) { col[k] *= d1; if (col[k] < d2) col[k] = d2; }
            }
        }
    }
}
