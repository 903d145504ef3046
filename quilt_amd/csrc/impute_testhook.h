/* impute_testhook.h -- PRIVATE to the library's build and to tests/ (not part of the public ABI of include/quilt_amd.h).
 *
 * The per-sample driver loop of csrc/impute.cpp (qa_impute_samples) is written over a table of the batched entry points it
 * calls; the product fills the table with the library's own functions.  This header lets tests/ run that very host code
 * WITHOUT a device over a checker's entry points (tests/native_driver_backend.py fills it with the CPU oracle's).  Nothing in
 * the product -- quilt_amd/, shim/, bench.py's timed region -- uses it.
 */
#ifndef QA_IMPUTE_TESTHOOK_H
#define QA_IMPUTE_TESTHOOK_H
#include "../../include/quilt_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/*
 * The same loop over a caller-supplied table of the batched entry points it calls (signatures = the qa_*
 * functions named in the comments, with an opaque handle in place of the panel).  The product table is the library's own
 * functions; tests/ pass a checker's (the CPU oracle) to run this very host code without a device.  Nothing in the
 * library calls it.
 */
typedef struct {
    int (*gibbs_batch)(void *handle, const qa_gibbs_opts_t *opts, int32_t n_chain, const int32_t *which_haps_to_use_1based,
                       const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *wif,
                       const double *runif_reads, const int32_t *first_read, const double *runif_shard, int32_t *H,
                       int32_t *H_class, double *hapProbs_t, double *genProbsM_t, double *genProbsF_t, int32_t *underflow_problem,
                       double *state_out, const uint64_t *seed_reads, const uint64_t *seed_shard);   /* qa_gibbs_batch */
    int (*fullpass_reads_select_batch)(void *handle, int32_t n_chain, int32_t n_label, int32_t n_sample, const int32_t *chain_sample,
                       const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *H,
                       const int32_t *want_dosage, const int32_t *want_top, const int32_t *gammaSmall_cols_to_get,
                       int32_t K_top_matches, double minGLValue, double *dosage, int32_t top_width, int32_t *top_idx,
                       float *top_val, int32_t *top_cnt, int32_t Ksubset, int32_t Knew, const int32_t *which_haps_to_use,
                       const uint64_t *seed_select, int32_t *which_next, int32_t *select_status);   /* qa_fullpass_reads_select_batch */
    int (*fullpass_batch)(void *handle, int32_t n_pass, const double *gl, const int32_t *want_dosage,
                       const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches, double *dosage, int32_t *best_ptr,
                       int32_t *best_idx, double *best_val, int64_t best_cap);                        /* qa_fullpass_batch */
    int (*make_eMatRead_t_hap_major)(void *handle, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                       const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                       double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t, double *eMatRead_t);
    int (*mspbwt_select_new_haps)(const qa_mspbwt_t *index, int32_t n_chain, int32_t n_label, const int32_t *Zs, int32_t L,
                       int32_t M, int32_t Knew, const uint64_t *seed, int32_t *out);                  /* qa_mspbwt_select_new_haps */
    int (*accumulate_dosage)(int32_t n_chain, int32_t n_label, int32_t nSNPs, const double *hap, const int32_t *chain_sample,
                       int32_t n_sample, double *dosage, double *gp_t, double *fet_dosage, double *fet_gp_t);
    int (*consensus_read_labels)(int32_t nReads, int32_t n, const int32_t *labels, const double *p, int32_t K, double minrp,
                       int32_t can_hap, int32_t *out);
    void *(*host_alloc)(size_t bytes);
    int (*host_free)(void *p);
    void (*bind_thread)(void *handle);   /* may be NULL */
    /* impute_rare_common only (may be NULL otherwise): qa_gibbs_batch_rare_common and qa_rcpp_make_eMatRead_t_nsnps */
    int (*gibbs_batch_rare_common)(void *handle, const void *rc, const qa_gibbs_opts_t *opts, int32_t n_chain,
                       const int32_t *which_haps_to_use_1based, const int32_t *read_off, const int32_t *read_ptr, const int32_t *u,
                       const int32_t *bq, const int32_t *wif, const double *runif_reads, const int32_t *first_read,
                       const double *runif_shard, int32_t *H, int32_t *H_class, double *hapProbs_t, double *genProbsM_t,
                       double *genProbsF_t, int32_t *underflow_problem, double *state_out, const uint64_t *seed_reads,
                       const uint64_t *seed_shard);
    int (*make_eMatRead_t_nsnps)(void *handle, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                       const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                       double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t, double *eMatRead_t);
    /* optional (may be NULL: the loop then spreads the haplotypes over all SNPs itself and calls make_eMatRead_t_nsnps):
     * qa_rcpp_make_eMatRead_t_rare_common */
    int (*make_eMatRead_t_rare_common)(void *handle, const void *rc, int32_t n_chain, int32_t n_sample, const int32_t *chain_sample,
                       int32_t K, const double *hap_common, const int32_t *read_off, const int32_t *read_ptr, const int32_t *u,
                       const int32_t *bq, double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                       double *eMatRead_t);
} qa_impute_backend_t;
int qa_impute_samples_backend(const qa_impute_backend_t *backend, void *const *handles, int32_t n_handles, int32_t K, int32_t nGrids,
                              int32_t nSNPs, const qa_impute_params_t *params, int32_t n_sample, int64_t sample_offset,
                              const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                              const int32_t *wif, double *dosage, double *gp_t, double *phasing_haps, int32_t *read_labels,
                              int32_t *nDosage, int64_t *stats);


/* qa_impute_bam_range (include/quilt_amd_io.h) with its imputation step on a caller's table: the loader, the bookkeeping of
 * kept / dropped samples and their global indices, the column formatting and the count arrays are the product's own code */
#include "../../include/quilt_amd_io.h"
int qa_impute_bam_range_backend(const qa_impute_backend_t *backend, void *const *handles, int32_t n_handles, int32_t K, int32_t nGrids,
                                const qa_impute_params_t *params, const qa_bam_range_io_t *io, int32_t n_sample, const char *const *bam_paths,
                                const int64_t *sample_index, const double *ff, qa_bam_range_result_t **out);

#ifdef __cplusplus
}
#endif
#endif
