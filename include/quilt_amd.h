/*
 * quilt_amd.h -- C ABI of the MI355X-native QUILT hot path (libquilt_amd.so).
 *
 * This is the drop-in boundary: the entry points below are what the reference's
 * R -> native `.Call` layer binds for its per-sample hot path
 * (QUILT/src/RcppExports.cpp:1703-1782 registers them; QUILT/R/RcppExports.R holds
 * the R stubs).  Each declaration cites the reference interface it replaces.
 * INTEGRATION.md shows the R-side shim a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every matrix is column-major (R layout);
 *   - indices are 0-based unless the parameter name says `_1based`;
 *   - host pointers unless the name starts with `d_`;
 *   - every function returns a status: QA_OK (0), QA_UNDERFLOW (1, soft failure the
 *     R driver retries on: functions.R:2704-2715), or a negative hard error whose
 *     text is available from qa_last_error();
 *   - the library never keeps a caller pointer past the call (R owns its buffers,
 *     SURVEY.md 8(b) "Ownership"); device mirrors live behind the opaque handles;
 *   - no CPU fallback exists: every compute entry point fails with QA_ERR_NO_DEVICE
 *     when no gfx950 device is usable.
 */
#ifndef QUILT_AMD_H
#define QUILT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QA_OK 0
#define QA_UNDERFLOW 1
#define QA_ERR_NO_DEVICE (-1)
#define QA_ERR_INVALID (-2)
#define QA_ERR_UNSUPPORTED (-3)
#define QA_ERR_HIP (-4)
#define QA_ERR_CAPACITY (-5)

/* ---- library ------------------------------------------------------------ */

/* ABI version of this header (bumped on any signature change). */
int qa_abi_version(void);
/* Text of the last error on the calling thread (never NULL). */
const char *qa_last_error(void);
/* Number of usable gfx950 devices (0 when none: compute entry points then fail). */
int qa_device_count(void);
/* Bind the calling thread (and the objects it creates) to a device.  Must be called
 * in the worker process itself: HIP contexts do not survive fork (quilt.R:692). */
int qa_set_device(int device);

/* Host buffers the device reads and writes directly (pinned, device-visible).  Every entry point accepts any host pointer;
 * a buffer that lies inside a qa_host_alloc region skips the library's pinned staging copy -- worth it for the large,
 * reused ones (a dosage round returns n_chain x n_label x nSNPs doubles: 1 GB at 1 024 passes x 64 000 SNPs, whose staged
 * copy into fresh pageable memory costs ten times the transfer).  R's own vectors (the reference's caller,
 * functions.R:2043-2068) cannot live there: the shim keeps the staged path.  qa_host_alloc returns NULL on failure
 * (qa_last_error); qa_host_free returns QA_ERR_INVALID for a pointer it did not hand out. */
void *qa_host_alloc(size_t bytes);
int qa_host_free(void *p);

/* Diagnostic: milliseconds (best of three) of one transfer of `bytes` between device memory and a host buffer.  mode 0: the
 * library's copy kernel on a pinned buffer (what a qa_host_alloc buffer gets), 1: hipMemcpyAsync on a pinned buffer,
 * 2: the staged path into pageable memory (what any other host pointer gets). */
int qa_selftest_copy_rate(int32_t to_device, size_t bytes, int32_t mode, double *ms);

/* Per-kernel accumulators since the last reset, measured with HIP events on the launch stream
 * (replaces print_times(), copied-from-stitch.cpp:31-45).  Kernels are numbered 0 .. qa_profile_count() - 1 and named by
 * qa_profile_name (k_emat, k_fwd, k_bwd, k_dosage, k_ematread, k_gibbs, k_happrobs, k_fwd64, k_bwd64, k_topk, ...).
 * alg_bytes = algorithmic HBM bytes of those launches.  qa_profile_get_work: units = work units (Gibbs:
 * read visits + grid steps over all chains), serial = summed length of the launches' serial chains (Gibbs: read visits +
 * grid steps of the longest chain of each launch) -- for rates such as microseconds per grid step. */
int qa_profile_reset(void);
int qa_profile_count(void);
const char *qa_profile_name(int32_t kernel);
int qa_profile_get(int32_t kernel, double *ms, int64_t *launches, double *alg_bytes);
int qa_profile_get_work(int32_t kernel, double *units, double *serial);
/* Workgroups of the kernel's launches since the reset (Gibbs: one per chain; 0 for kernels that do not record them). */
int qa_profile_get_workgroups(int32_t kernel, double *workgroups);
/* Device time during which at least one launch of the kernel was running (ms, union of the launch intervals over all
 * streams): with several host threads the per-launch times of concurrent launches overlap, and the aggregate rate of a
 * kernel is alg_bytes / busy time. */
int qa_profile_get_busy(int32_t kernel, double *busy_ms);

/* ---- prepared reference panel (upload once per process) ------------------ */

typedef struct qa_panel qa_panel_t;

/* The objects QUILT() loads from the prepared-reference RData
 * (QUILT/R/quilt.R:389; produced by quilt-prepare-reference.R:416-428). */
typedef struct {
    int32_t K, nGrids, nSNPs, nMaxDH;
    const uint8_t *hapMatcherR;   /* K x nGrids raw, or NULL                          */
    const int32_t *hapMatcher;    /* K x nGrids int, used when hapMatcherR is NULL    */
    const int32_t *rhb_t;         /* K x nGrids packed panel, or NULL (msPBWT mode)   */
    const int32_t *distinctHapsB; /* nMaxDH x nGrids                                  */
    const double *distinctHapsIE; /* nMaxDH x nSNPs, or NULL (derived from B, eps)    */
    const int32_t *eMatDH_special_grid_which;    /* nGrids                             */
    const int32_t *eMatDH_special_matrix_helper; /* nGrids x 2, 1-based first/last    */
    const int32_t *eMatDH_special_matrix;        /* nrow x 2: k (0-based), word       */
    int32_t eMatDH_special_matrix_nrow;
    int32_t use_eMatDH_special_symbols; /* 1: decode specials from the matrix (with the
                                           reference's search, gibbs-small.cpp:69-105);
                                           0: from rhb_t                               */
    const double *transMatRate_t; /* 2 x (nGrids-1): sigma, 1 - sigma                  */
    double ref_error;
} qa_panel_desc_t;

int qa_panel_create(const qa_panel_desc_t *desc, qa_panel_t **out);
void qa_panel_destroy(qa_panel_t *panel);

/* The panel handle straight from the packed panel: per-grid dictionary compression on the device.  Replaces
 * STITCH::make_rhb_t_equality (STITCH 1.8.4; call sites QUILT/R/quilt-prepare-reference.R:416-428, QUILT/R/quilt.R:551-563)
 * followed by qa_panel_create: per grid the distinct 32-bit words of rhb_t are ranked by descending frequency (ties:
 * ascending signed value -- the builder's choice: STITCH is not in the reference tree, so its tie rule is UNPINNED; any
 * rule gives tables the path accepts and the same imputation, pinned by the round-trip property of
 * test-unit-reference-single.R:210-309 only), the first nMaxDH (<= 255) get the 1-based codes of hapMatcherR / rows of distinctHapsB, every
 * other haplotype code 0 and an entry in the special tables.
 *   rhb_t   K x nGrids int32, column-major (the R matrix), bit b of word g = allele at SNP 32 g + b
 *   use_eMatDH_special_symbols   as in qa_panel_desc_t: decode special words the way the reference does without rhb_t
 * qa_panel_export_tables copies the tables back in the reference's layouts (hapMatcherR K x nGrids raw, distinctHapsB
 * nMaxDH x nGrids int32; specials as CSR over grids: special_off[nGrids + 1], ascending 0-based k and the word of each);
 * any output may be NULL. */
int qa_panel_create_from_rhb(const int32_t *rhb_t, int32_t K, int32_t nGrids, int32_t nSNPs, int32_t nMaxDH,
                             const double *transMatRate_t, double ref_error, int32_t use_eMatDH_special_symbols,
                             qa_panel_t **out);
/* The packed panel from the integer allele matrix: replaces STITCH::make_rhb_t_from_rhi_t (STITCH 1.8.4, un-vendored; call
 * site QUILT/R/test-drivers.R:394, the panel builder of the reference's own tests).  rhi_t K x nSNPs int32 column-major, entries 0 / 1 (non-zero counts as
 * 1); rhb_t out K x ceil(nSNPs / 32) int32 column-major, bit b of word g = allele at SNP 32 g + b, unused high bits of the
 * last grid 0.  Packed on the device in slabs of whole grids. */
int qa_make_rhb_t_from_rhi_t(const int32_t *rhi_t, int32_t K, int32_t nSNPs, int32_t *rhb_t);

/* The handle's dimensions (any output may be NULL), and binding the calling host thread to the handle's device (HIP's current
 * device is per thread: a thread that allocates qa_host_alloc buffers for a handle binds first; the compute entry points bind
 * by themselves). */
int qa_panel_get_dims(const qa_panel_t *panel, int32_t *K, int32_t *nGrids, int32_t *nSNPs);
int qa_panel_bind_thread(const qa_panel_t *panel);

int qa_panel_export_tables(qa_panel_t *panel, uint8_t *hapMatcherR, int32_t *distinctHapsB, int32_t *special_off,
                           int32_t *special_k, int32_t *special_word, int64_t special_cap);

/* Arithmetic of the state behind the best-haplotype lists (get_best_haps_from_thinned_sites).  64 (default): the
 * lists come from a forward/backward with fp64 state -- the reference computes in double, and which of several
 * nearly tied haplotypes make a list decides the next small panel, so membership and order must be the reference's.
 * 32: lists from the fp32-state pass that also produces the dosage (faster, within ~1e-6 of the reference's gamma, but
 * near-ties may be ordered differently).  Dosages always come from fp32 state (|error| ~1e-6). */
int qa_panel_set_ranking_precision(qa_panel_t *panel, int32_t bits);

/* Arithmetic of the state behind every other output of the full-panel pass (dosage, alphaHat_t, betaHat_t, gamma_t, c):
 * 32 (default): fp32 state with fp64 emissions and normalisers (|dosage error| ~1e-6 against the reference's doubles);
 * 64: fp64 state throughout, as the reference (reference-single.cpp:2189-2413) -- a verification mode, several times
 * slower (18 instead of 10 algorithmic bytes per cell and untuned kernels); a pass that wants dosage and lists then yields
 * both from the one fp64 state. */
int qa_panel_set_dosage_precision(qa_panel_t *panel, int32_t bits);

/* VALIDATION MODE.  reference_order = 1: every full-panel pass of this handle (lists, dosage, alphaHat_t / betaHat_t / gamma_t)
 * runs on kernels that form each K-wide sum in the order the reference's code adds it, by a single lane:
 *   - the forward column sum run_total (reference-single.cpp:1002-1075), the backward sum_e_times_b (:1899-1955) and
 *     matched_gammas / the dosage sums (:2083-2139) are explicit C++ loops: the grid's special haplotypes first, in list
 *     order, then k = 0 .. K-1 one after the other;
 *   - c(0) = 1 / sum(alphaHat_t_col) (:2347) is Armadillo's sum() of an arma::colvec, which without -ffast-math runs two
 *     accumulators (arrayops::accumulate: even k into one, odd k into the other, acc1 + acc2) -- restated from Armadillo's
 *     published source, which is not in the build image.  reference_order = 2 is the same mode with that one sum added
 *     left to right instead (what this library did before round 6): a maintainer with R decides between 1 and 2 by
 *     printing one c(0) at full precision (oracle/quilt_oracle.h, "Armadillo's sum()").
 * A floating-point sum in a prescribed order does not parallelise, so this mode is 20-50x slower than the default (block-wide
 * tree sums, whose last bits differ from either order's).  Its purpose: on panels with many identical or exactly tied haplotypes the last bits of
 * those sums decide which of the tied haplotypes make a best-haplotype list ("ties"); with this mode the
 * device reproduces the CPU path's lists, c, alpha / beta and dosage bit for bit, which proves the order of the sums to be the
 * only difference between the two.  K <= 57 344.  0 (default): the production kernels. */
int qa_panel_set_sum_order(qa_panel_t *panel, int32_t reference_order);

/* Tell the library that n_sharers panel handles (normally one per host thread, each with its own stream and arena)
 * work on this device at the same time: each then sizes its scratch for 1 / n_sharers of the free memory and its Gibbs
 * launches for 1 / n_sharers of the SIMDs.  Two host threads hide each other's host-side phases (marshalling, the R-level
 * logic between native calls) behind the other's kernels. */
int qa_panel_set_device_share(qa_panel_t *panel, int32_t n_sharers);

/* Device phases.  on = 1: the launch sets of this handle take the device in arrival order together with those of the other
 * handles that opted in (first come, first served; only the short haplotype searches of qa_find_good_matches go before the
 * queued launch sets).  The full-panel passes of a call (one workgroup per compute
 * unit) hold the device EXCLUSIVELY; the kernels of a Gibbs call hold one SIMD slot per wave (of 1 024), so Gibbs launches
 * that fit together run together -- the 128 phasing chains of one batch beside the 896 main chains of another -- and one
 * that does not fit waits for the phase to end.  All of them carve their scratch from ONE device-wide arena (allocated once,
 * at 88 % of the device's free memory; freed when the last handle that opted in is destroyed) instead of the handle's
 * 1 / n_sharers part, so a Gibbs launch carries up to one chain per SIMD and a full-panel launch one pass per
 * compute unit (256) whatever the number of host threads.  The host-side parts of the calls (validation, tables,
 * marshalling) stay outside the queue: with 3-4 host threads per device they overlap the other threads' device phases.
 * Both kinds of launch sets are HBM-bound when they fill the chip, so nothing is lost by not overlapping them; measured
 * gains in DESIGN.md 5.  Default off (a single handle has the device to itself anyway).  qa_gate_stats since the last
 * reset: [0] ms with at least one holder, [1] ms callers queued, [2] holds, [3] ms with an exclusive holder (full-panel launch
 * sets), [4] sum over the Gibbs holds of SIMD slots x ms, [5] Gibbs holds, [6] their SIMD slots in total. */
int qa_panel_set_exclusive(qa_panel_t *panel, int32_t on);
int qa_gate_stats(int32_t device, double out[7]);
int qa_gate_stats_reset(int32_t device);
/* Diagnostics: switch the gate's hold trace on / off and read it.  Returns the number of rows recorded so far (>= 0) and copies
 * up to cap_rows of them -- request, admit, kernels done, release [ms on one clock], SIMD slots (0: exclusive), thread --
 * then clears the trace (a call with rows == NULL and on != 0 only switches it on). */
int qa_gate_trace(int32_t device, int32_t on, double *rows, int32_t cap_rows);
/* Diagnostics: the gate's admission rules (co-running Gibbs launches, first come first served, express holds first) on a gate
 * of its own; no device is touched.  QA_OK, or QA_ERR_INVALID with the failed rule in qa_last_error(). */
int qa_gate_selftest(void);

/* With several handles sharing a device: confine this handle's Gibbs launches to CUs [index, index + 1) * n_CU / count
 * (a CU-masked HIP stream).  A Gibbs chain holds a SIMD's whole register file for ~0.7 s; spread over every CU, one
 * handle's chains leave no CU on which the other handle's full-panel workgroups (one wave per SIMD, most of the LDS) can
 * start.  The full-panel passes stay on the unmasked stream and use whatever is free.  count = 1 removes the mask.
 * (Measured on the headline workload with two host threads: 17.5 samples/s with the partition against 19.3 without --
 * the other thread's passes then run beside the chains at half rate instead of after them at full rate -- so the
 * driver leaves it off.) */
int qa_panel_set_cu_partition(qa_panel_t *panel, int32_t index, int32_t count);

/* With several handles sharing a device: run this handle's full-panel calls on a stream of the highest priority (its
 * Gibbs launches stay on the default one).  A Gibbs launch is one long wave per chain, a full-panel launch many short
 * workgroups that each need most of a compute unit; at equal priority the next Gibbs launch's waves take the SIMDs the
 * last one frees one by one and the full-panel workgroups of the other handles wait for a whole free CU.  With the
 * priority the dispatcher places pending full-panel workgroups first, so those calls run at their stand-alone rate between
 * Gibbs launches instead of beside them.  on = 0 restores the default stream.  (Calls are host-synchronous, so the two
 * streams of a handle never overlap.) */
int qa_panel_set_pass_priority(qa_panel_t *panel, int32_t on);

/* ---- full-panel haploid forward/backward -------------------------------- */

/* Flags of Rcpp_haploid_dosage_versus_refs (QUILT/src/reference-single.cpp:2214-2227). */
typedef struct {
    int32_t K_top_matches;
    int32_t return_betaHat_t, return_dosage, return_gamma_t, return_gammaSmall_t;
    int32_t get_best_haps_from_thinned_sites;
    int32_t always_normalize;     /* honoured by every pass with fp64 state (the ranking passes, the dosage passes of a handle
                                     with qa_panel_set_dosage_precision(64), the validation mode): they follow the reference's
                                     lazy normalisation, reference-single.cpp:1096-1107.  Passes with fp32 state always
                                     normalise per grid, which the reference proves equivalent for dosage, gamma and
                                     sum(log c) (test-unit-reference-single.R:588-642) */
    int32_t normalize_emissions;
    double min_emission_prob_normalization_threshold; /* honoured by the same passes: renormalise when the running product of the
                                     per-grid minimum emissions falls below it.  Reference default 1e-100 (:2216); a value that
                                     is not positive -- e.g. a zero-initialised struct -- is read as that default, because 0
                                     would mean "never renormalise" and underflow to NaN dosages over a long region */
    int32_t suppressOutput;
} qa_fullpass_opts_t;

/*
 * Replaces `_QUILT_Rcpp_haploid_dosage_versus_refs`
 * (QUILT/src/RcppExports.cpp:1580-1625; kernel QUILT/src/reference-single.cpp:2189-2413;
 * called from QUILT/R/functions.R:2034-2070) for use_eMatDH = TRUE.
 *
 * Like the reference it returns nothing and writes into the caller's buffers; any
 * output pointer may be NULL when the matching flag is 0.
 *
 * Panel size: dosage, c and best_haps_stuff_list are available for ANY K (up to K = 57 344 a pass keeps its whole state on
 * one compute unit -- seven chunk rows of 8 192 haplotypes in registers and LDS; beyond that the chunk rows past the seventh
 * stream their state through HBM: K = 65 536 reads and writes one row of eight per grid, K = 131 072 nine of sixteen, the
 * same arithmetic).  The K x nGrids outputs -- alphaHat_t, betaHat_t, gamma_t, gammaSmall_t -- come from kernels that hold
 * the state on chip and are limited to K <= 57 344: QA_ERR_UNSUPPORTED beyond (qa_last_error says so).  The batched entry
 * points below (dosage + lists only) have no limit but the device's memory (QA_ERR_CAPACITY).
 *   gl                      2 x nSNPs
 *   gammaSmall_cols_to_get  nGrids, -1 or the 0-based thinned column
 *   alphaHat_t              K x nGrids (NULL: not copied back).  Columns are normalised
 *                           to sum 1 (always_normalize semantics).  When only thinned
 *                           outputs are requested only column 0 and the thinned columns
 *                           are written (reference-single.cpp:2264-2268).
 *   betaHat_t, gamma_t      K x nGrids
 *   c                       nGrids
 *   gammaSmall_t            K x n_thin
 *   dosage                  nSNPs
 *   best_ptr/idx/val        CSR form of best_haps_stuff_list: entry i (thinned column
 *                           i) = top_matches (0-based, ascending k) and
 *                           top_matches_values.  best_cap = capacity of idx/val; on
 *                           QA_ERR_CAPACITY best_ptr holds the needed sizes.
 */
int qa_Rcpp_haploid_dosage_versus_refs(
    qa_panel_t *panel, const double *gl, const int32_t *gammaSmall_cols_to_get,
    const qa_fullpass_opts_t *opts, double *alphaHat_t, double *betaHat_t, double *c,
    double *gamma_t, double *gammaSmall_t, double *dosage, int32_t *best_ptr,
    int32_t *best_idx, double *best_val, int64_t best_cap);

/* Replaces `_QUILT_Rcpp_make_gl_bound` (RcppExports.cpp:1263-1273;
 * reference-single.cpp:68-94).  Host arithmetic on a 2 x nSNPs matrix (O(n_to_fix)). */
int qa_Rcpp_make_gl_bound(double *gl, double minGLValue, const int32_t *to_fix, int32_t n_to_fix);

/*
 * Batched form used by the per-sample driver: n_pass independent passes (one per
 * (sample, Gibbs chain, read label)) in one launch set.  gl is n_pass stacked 2 x nSNPs
 * matrices; want_dosage[i] selects a "dosage" pass (else a "thin" pass: only
 * best-haps at the thinned grids, functions.R:748).  Outputs are stacked per pass:
 * dosage n_pass x nSNPs (rows of skipped passes untouched); best_* as above with
 * n_pass * n_thin entries (pass-major).
 */
int qa_fullpass_batch(
    qa_panel_t *panel, int32_t n_pass, const double *gl, const int32_t *want_dosage,
    const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches, double *dosage,
    int32_t *best_ptr, int32_t *best_idx, double *best_val, int64_t best_cap);


/*
 * The full-panel step of the per-sample driver in one call: `impute_using_everything`
 * (QUILT/R/functions.R:1922-2157) for n_chain (sample, Gibbs chain) pairs.  For every chain and read label
 * it builds the label's genotype likelihoods from the sample's reads on the device
 * (make_gl_from_u_bq, QUILT/R/reference-single.R:19-42, + Rcpp_make_gl_bound), runs the full-panel
 * forward/backward (dosage pass when want_dosage[chain], else thin pass) and returns, per thinned grid, the
 * top matches already ordered as `everything_per_hap_rejig_haps` orders them (functions.R:2161-2170).
 *
 *   chain_sample      n_chain: sample of each chain (several chains share one sample's reads)
 *   read_off, read_ptr, u, bq   reads of the n_sample samples, laid out as in qa_gibbs_batch (per sample)
 *   H                 read labels (1-based) of the chains back to back, chain c holding R_{chain_sample[c]} labels
 *   want_top          n_chain: whether the chain's best-haplotype lists are wanted (NULL: all chains); the
 *                     driver skips them where the reference computes but never reads them (chains 1..nGibbsSamples
 *                     at the last seek iteration)
 *   dosage            n_chain x n_label x nSNPs (rows of thin passes untouched); may be NULL
 *   top_idx/top_val   n_chain x n_label x n_thin x top_width: 0-based haplotypes / values, best first, -1 padded
 *   top_cnt           n_chain x n_label x n_thin: full length of each list (> top_width means truncated)
 */
int qa_fullpass_reads_batch(qa_panel_t *panel, int32_t n_chain, int32_t n_label, int32_t n_sample,
                            const int32_t *chain_sample, const int32_t *read_off, const int32_t *read_ptr,
                            const int32_t *u, const int32_t *bq, const int32_t *H, const int32_t *want_dosage,
                            const int32_t *want_top, const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches,
                            double minGLValue,
                            double *dosage, int32_t top_width, int32_t *top_idx, float *top_val,
                            int32_t *top_cnt);

/*
 * qa_fullpass_reads_batch followed, on the device, by the re-selection of every chain's small panel:
 * `everything_per_hap_rejig_haps` + `everything_select_good_haps` (QUILT/R/functions.R:2161-2170, 2262-2310; SURVEY.md
 * 8(f) rank 2(a)).  The best-haplotype lists stay on the device (top_idx / top_val may be NULL: then they are not copied
 * back at all; top_cnt is always cheap); only the next which_haps_to_use crosses PCIe.
 *
 *   Ksubset, Knew         size of the small panel and how many of it are replaced per round (functions.R:2286-2295)
 *   which_haps_to_use     n_chain x Ksubset, 1-based: the chains' current small panels
 *   seed_select           n_chain keys of the library's counter stream (splitmix64, see qa_gibbs_batch seed_reads).  The
 *                         reference draws with R's sample(); here each draw is "the n smallest keys, in key order":
 *                         previously_selected_haplotypes = the Ksubset - Knew entries of which_haps_to_use with the
 *                         smallest keys [0, Ksubset); the rank that overshoots Knew keeps the candidates with the
 *                         smallest keys [2^20, 2^20 + n_candidates) -- restated on the host in quilt_amd/driver.py
 *   which_next            n_chain x Ksubset, 1-based: previously selected, then the Knew new haplotypes (rows of chains
 *                         without want_top are left untouched)
 *   select_status         n_chain: 0 selected; 1 the ranks up to K_top_matches did not yield Knew new haplotypes -- the
 *                         reference then uses every entry of the complete lists and finally a random draw from the panel
 *                         (functions.R:2278-2300), which the caller does on the host from untruncated lists (rare);
 *                         -1 chain without want_top
 */
int qa_fullpass_reads_select_batch(qa_panel_t *panel, int32_t n_chain, int32_t n_label, int32_t n_sample,
                                   const int32_t *chain_sample, const int32_t *read_off, const int32_t *read_ptr,
                                   const int32_t *u, const int32_t *bq, const int32_t *H, const int32_t *want_dosage,
                                   const int32_t *want_top, const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches,
                                   double minGLValue, double *dosage, int32_t top_width, int32_t *top_idx, float *top_val,
                                   int32_t *top_cnt, int32_t Ksubset, int32_t Knew, const int32_t *which_haps_to_use,
                                   const uint64_t *seed_select, int32_t *which_next, int32_t *select_status);

/*
 * The haplotype search of the msPBWT mode (use_mspbwt = TRUE): stands in for mspbwt::Rcpp_find_good_matches_without_a as
 * `select_new_haps_mspbwt_v3` calls it (QUILT/R/mspbwt.R:265-301).  The mspbwt package (a
 * positional-BWT index over the panel's per-grid symbols) is not in the reference tree; here the query is compared with
 * EVERY haplotype -- the symbol table streams at HBM rate, no index to build or hold.  The definition of what is
 * reported is this library's (UNPINNED against mspbwt; see csrc/match.hip):
 *   Zs            n_query x nGrids int32: the queries' packed words (rcpp_int_contract of the rounded haploid dosage)
 *   nindices      interleaved indices (mspbwt_nindices): index i covers grids i, i + nindices, ... (positions 0, 1, ...)
 *   min_len       minimum number of consecutive matching positions (mspbwtM)
 *   max_matches   how many matches per (query, index) come back at most
 *   match         n_query x nindices x max_matches x 3: (index0 = 0-based haplotype, start0 = 0-based first position,
 *                 len1 = positions) -- per haplotype its longest run of matching positions, the max_matches longest of
 *                 those, ties at the cut to the lower haplotype; written in haplotype order
 *   n_match       n_query x nindices: entries written
 */
int qa_find_good_matches(qa_panel_t *panel, int32_t n_query, const int32_t *Zs, int32_t nindices, int32_t min_len,
                         int32_t max_matches, int32_t *match, int32_t *n_match);

/*
 * The msPBWT indices of the panel and the neighbour scan that queries them (host code; csrc/mspbwt.cpp).  Stands in for
 * mspbwt::ms_BuildIndices_Algorithm5 (the `ms_indices` quilt-prepare-reference stores: all_symbols, usge_all, egs) and
 * mspbwt::Rcpp_find_good_matches_without_a(Z, all_symbols, usge_all, egs, pbwtL = mspbwtL, pbwtM = mspbwtM, ...) as
 * select_new_haps_mspbwt_v3 calls it (QUILT/R/mspbwt.R:297-310) -- the query of use_mspbwt = TRUE, QUILT2's default.  The
 * mspbwt package is not in the reference tree: the algorithm is the published one (positional prefix order, the query's
 * insertion point, mspbwtL haplotypes up and down per position, matches of at least mspbwtM positions), stated in
 * tests/mspbwt_scan.py; PARITY UNPINNED against the package itself.  qa_find_good_matches above (every haplotype's longest
 * run, on the device) remains as the exhaustive alternative.
 *   hapMatcherR     K x nGrids uint8, R's column-major layout (element (k, g) at k + K g); distinctHapsB nMaxDH x nGrids int32
 *   nindices        mspbwt_nindices: index i covers grids i, i + nindices, ...
 * Memory: 8 K nGrids bytes + K nGrids (0.9 GB for K = 50 000 x 2 000 grids), host.  NULL + qa_last_error() on failure. */
typedef struct qa_mspbwt qa_mspbwt_t;
qa_mspbwt_t *qa_mspbwt_create(int32_t K, int32_t nGrids, const uint8_t *hapMatcherR, int32_t nMaxDH, const int32_t *distinctHapsB,
                              int32_t nindices);
void qa_mspbwt_destroy(qa_mspbwt_t *index);
int64_t qa_mspbwt_bytes(const qa_mspbwt_t *index);
/* The scan for n_query packed haplotypes (Zs n_query x nGrids, rcpp_int_contract of the rounded haploid dosage), CSR output:
 * row_ptr [n_query x nindices + 1], rows (index0 = 0-based haplotype, start0 = 0-based first position, len1) in (haplotype,
 * start) order, one row per (haplotype, start) -- the longest (mspbwt.R:330-345 drops the others).  Returns the number of rows
 * found; when that exceeds cap_rows nothing beyond the capacity was written (row_ptr is complete): call again with room.
 * 1 <= L <= 64, M >= 1.  Negative: QA_ERR_*. */
int64_t qa_mspbwt_find_good_matches(const qa_mspbwt_t *index, int32_t n_query, const int32_t *Zs, int32_t L, int32_t M,
                                    int64_t *row_ptr, int32_t *rows, int64_t cap_rows);
/* The scan followed by select_new_haps_mspbwt_v3 (QUILT/R/mspbwt.R:225-474; qa_select_new_haps_mspbwt of quilt_amd_io.h) for
 * every chain of a round, threaded over chains; the match tables stay on the native side.  Zs n_chain x n_label x nGrids,
 * seed one selection-stream key per chain, out n_chain x Knew 1-based haplotypes. */
int qa_mspbwt_select_new_haps(const qa_mspbwt_t *index, int32_t n_chain, int32_t n_label, const int32_t *Zs, int32_t L, int32_t M,
                              int32_t Knew, const uint64_t *seed, int32_t *out);

/* Timing of the most recent full-pass launch set on this thread, measured with HIP
 * events on the launch stream (ms): [0] emission build, [1] forward, [2] backward,
 * [3] dosage mat-vec, [4] total device.  Replaces print_times()
 * (copied-from-stitch.cpp:31-45). */
int qa_last_fullpass_timing_ms(double out[5]);


/* ---- small-panel Gibbs read-label sampler -------------------------------- */

/* The scalar arguments and param_list flags of rcpp_forwardBackwardGibbsNIPT that the production
 * caller varies (QUILT/R/functions.R:2566-2678).  Everything the caller holds constant is fixed at
 * the reference's production value: S = 1, n_gibbs_starts = 1, priorCurrent_m and
 * alphaMatCurrent_tc = 1/Ks, use_small_eHapsCurrent_tc = FALSE, calculate_gamma_on_the_fly = TRUE,
 * pass_in_alphaBeta = TRUE, record_read_set = TRUE, shard_check_every_pair = TRUE,
 * haploid_gibbs_equal_weighting = TRUE, use_starting_read_labels = TRUE. */
typedef struct {
    int32_t Ks;                      /* length(which_haps_to_use) (Ksubset) */
    double ff;                       /* fetal fraction; 0 = diploid */
    int32_t sample_is_diploid;
    int32_t Jmax;                    /* Jmax_local */
    double maxDifferenceBetweenReads;
    int32_t rescale_eMatRead_t;
    int32_t n_gibbs_burn_in_its, n_gibbs_sample_its;
    const int32_t *block_gibbs_iterations; /* 0-based sweep numbers */
    int32_t n_block_gibbs_iterations;
    int32_t perform_block_gibbs, do_shard_block_gibbs;
    int32_t gibbs_initialize_iteratively;
    int32_t disable_read_category_usage;
    double class_sum_cutoff;
    /* NIPT block Gibbs only (ff > 0 with perform_block_gibbs): what Rcpp_define_blocked_snps_using_gamma_on_the_fly
     * takes beyond the state (QUILT/src/gibbs-nipt.cpp:3008): L_grid (nGrids grid positions, bp),
     * shuffle_bin_radius (quilt.R:134, 5000), block_gibbs_quantile_prob (functions.R:2393, 0.95) */
    const int32_t *L_grid;
    int32_t shuffle_bin_radius;
    double block_gibbs_quantile_prob;
    /* NULL, or (NIPT only, ff > 0) n_chain fetal fractions in (0, 1), one per chain, overriding ff: lets one launch set
     * carry samples with different fetal fractions (ff_values[iSample], functions.R:128) */
    const double *ff_chain;
    /* NULL, or an OUTPUT of n_chain x (n_gibbs_burn_in_its + n_gibbs_sample_its) x 8 doubles: what add_to_per_it_likelihoods
     * (QUILT/src/gibbs-nipt.cpp:1583-1621, calculate_likelihoods_values :1483-1519) needs after every sweep, per chain and
     * sweep: -sum(log c1), -sum(log c2), -sum(log c3) (0 for a label the mode does not have), the number of reads holding
     * label 1, 2, 3, and two reserved zeros.  The R shim turns a row into the 13 columns of per_it_likelihoods
     * (p_O_given_H_L, p_H_given_L, p_set_H_given_L ...); p_H_class_given_L needs H_class, which only the last sweep
     * records (it is overwritten by every sweep: gibbs-nipt.cpp:1142-1165). */
    double *per_it_out;
    /* use_mspbwt = TRUE: the call's haploid dosages rounded and packed on the device, n_chain x 3 x nGrids int32 --
     * rcpp_int_contract(round(hapProbs_t[h, ])) (QUILT/R/mspbwt.R:271-272: bit b of word g = (hapProbs_t[h, 32 g + b] > 0.5)),
     * the queries qa_find_good_matches takes.  With it hapProbs_t itself need not cross PCIe on the rounds whose dosages are
     * not accumulated.  NULL: not produced. */
    int32_t *hap_words_out;
    /* NULL, or an OUTPUT of n_chain x hap_major_labels x nSNPs doubles: the call's haploid dosages (rows of hapProbs_t) label by
     * label, the layout qa_fullpass_reads_batch returns dosages in and qa_accumulate_dosage / qa_rcpp_make_eMatRead_t_hap_major
     * take -- use_mspbwt = TRUE keeps the Gibbs call's dosages (functions.R:839-843), so a round's 1 GB goes from the device
     * into the caller's buffer in that form (directly when the buffer comes from qa_host_alloc) instead of being transposed
     * chain by chain on the host.  hap_major_labels: 2 or 3. */
    double *hap_major_out;
    int32_t hap_major_labels;
    /* NULL, or n_chain indices: reads_same_as[c] = the first chain of THIS call whose reads (read_ptr block, u, bq) are the same
     * as chain c's -- c itself for the first chain of a sample.  A batched driver's seven chains of one sample share their
     * reads: with this the bases of a chain that names an earlier one are neither read nor uploaded (its u / bq block may be
     * left unwritten, its read_ptr block and wif must still be there), which at 192 000 SNPs saves six of seven copies of
     * 1.6 MB per chain on the host and over PCIe.  Results do not depend on it. */
    const int32_t *reads_same_as;
    /* NULL, or (NIPT, ff > 0, explicit uniforms) a source of the block passes' uniforms IN THE ORDER THE REFERENCE DRAWS THEM
     * (ABI 5).  The reference draws, at every block iteration, runif_block (gibbs-nipt.cpp:3016) before the pass and then, inside
     * rcpp_sample_H_using_H_class (gibbs-nipt-block.cpp:213-246), ONE uniform (Rcpp::sample(1:3, 1, prob)) for every read whose
     * class leaves a choice -- a number that depends on the pass's own result, so it cannot be drawn ahead.  With this set the
     * library asks for them when the reference would draw them, from the calling thread, chain by chain:
     *     what = 0   before block pass `pass` of chain `chain`: n = nReads uniforms, the pass's runif_block
     *     what = 1   after its relabelling: n = the number of reads with H_class in {0, 4, 5, 6, 7}, in read order
     * and the per-read re-draw slots of runif_shard are not read.  An R caller draws them with unif_rand() (and burns the
     * reference's unused runif_proposed / runif_total around what = 0), which keeps set.seed's stream in step with the CPU package. */
    void (*draw_uniforms)(void *ctx, int32_t chain, int32_t pass, int32_t what, int32_t n, double *out);
    void *draw_uniforms_ctx;
} qa_gibbs_opts_t;

/*
 * Replaces `_QUILT_rcpp_forwardBackwardGibbsNIPT` (QUILT/src/RcppExports.cpp:966-1105; kernel
 * QUILT/src/gibbs-nipt.cpp:2395-3307; called from QUILT/R/functions.R:2614-2678), batched over
 * n_chain independent (sample, Gibbs chain) problems against the same panel.
 *
 *   which_haps_to_use_1based  n_chain x Ks (1-based panel rows, as in R)
 *   read_off                  n_chain + 1: reads of chain c are read_off[c] .. read_off[c+1]-1
 *   read_ptr                  per chain R_c + 1 offsets (starting at 0) into that chain's bases,
 *                             chain c's block starts at read_ptr[read_off[c] + c]
 *   u, bq                     bases of all chains back to back: 0-based SNP index and signed base
 *                             quality (sampleReads[[r]][[4]] and [[3]]); wif = [[2]] (0-based grid)
 *   runif_reads               per chain R_c * n_its uniforms, chain c at offset read_off[c] * n_its:
 *                             what `Rcpp::runif(nReads * n_gibbs_full_its)` returns (gibbs-nipt.cpp:2845)
 *   first_read                per chain `Rcpp::sample(nReads, 1) - 1` (gibbs-nipt.cpp:2846-2848)
 *   runif_shard               n_chain x n_block_gibbs_iterations x (nGrids - 1): the
 *                             `Rcpp::runif(n_blocks - 1)` of each shard pass (gibbs-nipt-block.cpp:2054).
 *                             NIPT (ff > 0; no shard pass there): the uniforms of the block passes instead -- per
 *                             chain n_block_gibbs_iterations x 2 x R_c at offset read_off[c] * n_block_gibbs_iterations
 *                             * 2: per pass R_c `runif_block` values (gibbs-nipt.cpp:3016; entry b decides block b),
 *                             then one uniform per read for the reads whose class leaves a choice in
 *                             rcpp_sample_H_using_H_class (gibbs-nipt-block.cpp:226-243).
 *                             Read only when a pass draws from it (perform_block_gibbs, n_block_gibbs_iterations > 0, and
 *                             ff > 0 or do_shard_block_gibbs); it must then hold all of the above.
 *   H                         in: starting read labels (double_list_of_starting_read_labels), 1-based;
 *                             out: ending labels (double_list_of_ending_read_labels)
 *   H_class                   out (may be NULL)
 *   hapProbs_t, genProbs*_t   out, per chain 3 x nSNPs (may be NULL)
 *   underflow_problem         out per chain (may be NULL); the call returns QA_UNDERFLOW if any is set
 *   seed_reads, seed_shard    NULL, or per chain the seed of a counter-based uniform stream that replaces
 *                             runif_reads / runif_shard (element i = splitmix64 finaliser of
 *                             seed + (i + 1) * 0x9E3779B97F4A7C15, top 53 bits / 2^53): lets a batched driver
 *                             avoid shipping R_c * n_its uniforms per chain.  With R in the loop the explicit
 *                             arrays keep `set.seed` semantics.
 *   state_out                 NULL, or (n_chain == 1 only) 6 * Ks * nGrids + 3 * nGrids doubles:
 *                             alphaHat_t1, alphaHat_t2, betaHat_t1, betaHat_t2, eMatGrid_t1, eMatGrid_t2,
 *                             c1, c2, c3 -- the matrices the reference mutates in place
 *   first_read                per chain, 0-based (`sample(nReads, 1) - 1`); a negative entry makes that chain start
 *                             from its starting labels even when gibbs_initialize_iteratively is set (lets one
 *                             launch mix first-round chains with later-round chains)
 */
int qa_gibbs_batch(qa_panel_t *panel, const qa_gibbs_opts_t *opts, int32_t n_chain,
                   const int32_t *which_haps_to_use_1based, const int32_t *read_off,
                   const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *wif,
                   const double *runif_reads, const int32_t *first_read, const double *runif_shard,
                   int32_t *H, int32_t *H_class, double *hapProbs_t, double *genProbsM_t,
                   double *genProbsF_t, int32_t *underflow_problem, double *state_out,
                   const uint64_t *seed_reads, const uint64_t *seed_shard);


/*
 * Replaces `_QUILT_rcpp_make_eMatRead_t` (QUILT/src/RcppExports.cpp:15-37; kernel
 * QUILT/src/copied-from-stitch.cpp:115-229) as the driver uses it: read likelihoods against K = 2-3
 * dense per-SNP haplotype dosages (calculate_eMatRead_t_vs_haplotypes, QUILT/R/functions.R:2975-3020),
 * batched over n_chain (sample, haplotype-set) problems.  eHaps: n_chain stacked K x nSNPs matrices
 * (eHapsCurrent_tc[, , 1]); reads as in qa_gibbs_batch; eMatRead_t: per chain K x R_c, chains
 * back to back.
 */
int qa_rcpp_make_eMatRead_t(qa_panel_t *panel, int32_t n_chain, int32_t K, const double *eHaps,
                            const int32_t *read_off, const int32_t *read_ptr, const int32_t *u,
                            const int32_t *bq, double maxDifferenceBetweenReads, int32_t Jmax,
                            int32_t rescale_eMatRead_t, double *eMatRead_t);

/*
 * Host-side part of the NIPT block definition, exported for testing without a device: from rate2 (nGrids - 1 values,
 * QUILT/src/gibbs-nipt-block.cpp:347-363) to blocked_grid (:366-523) and the block table of Rcpp_make_gibbs_considers
 * (:1307-1553).  Output arrays need nGrids entries; *n_blocks receives the number of blocks.
 */
int qa_nipt_block_table(const double *rate2, const int32_t *L_grid, int32_t nGrids, int32_t shuffle_bin_radius,
                        double block_gibbs_quantile_prob, const int32_t *wif0, int32_t nReads, int32_t *blocked_grid,
                        int32_t *grid_start, int32_t *grid_end, int32_t *reads_start, int32_t *reads_end,
                        int32_t *grid_where, int32_t *n_blocks);

/* As qa_rcpp_make_eMatRead_t with eHapsCurrent_tc of nSNPs columns rather than the panel's: the all-SNP read
 * likelihoods of get_initial_read_labels (QUILT/R/rare_common.R:61-107). */
int qa_rcpp_make_eMatRead_t_nsnps(qa_panel_t *panel, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                                  const int32_t *read_off, const int32_t *read_ptr, const int32_t *u,
                                  const int32_t *bq, double maxDifferenceBetweenReads, int32_t Jmax,
                                  int32_t rescale_eMatRead_t, double *eMatRead_t);
/* The same with eHaps laid out chain x haplotype x SNP (the layout qa_fullpass_reads_batch returns haploid dosages in): the
 * driver's read-confidence step (QUILT/R/functions.R:2975-3020) then needs no transposed copy. */
int qa_rcpp_make_eMatRead_t_hap_major(qa_panel_t *panel, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                                  const int32_t *read_off, const int32_t *read_ptr, const int32_t *u,
                                  const int32_t *bq, double maxDifferenceBetweenReads, int32_t Jmax,
                                  int32_t rescale_eMatRead_t, double *eMatRead_t);

/* ---- rare + common SNPs: the final all-SNP Gibbs of QUILT2 ------------------ */

/*
 * The all-SNP side of a panel prepared with impute_rare_common = TRUE: `special_rare_common_objects`
 * (QUILT/R/prepare_reference_functions.R:172-247) as far as the native call uses it.  The panel tables cover the
 * common SNPs; rare SNPs are held per haplotype.
 *   snp_is_common        nSNPs_all flags; sum = the panel's nSNPs (common_snp_index of QUILT/R/rare_common.R:222-223
 *                        is derived from it)
 *   rare_ptr, rare_snp   rare_per_hap_info as CSR over the K panel haplotypes: 1-based all-SNP indices of the rare
 *                        SNPs haplotype k carries the alt of, ascending
 *   transMatRate_t_all   2 x (nGrids_all - 1), column-major, of the all-SNP grid (32 SNPs per grid):
 *                        small_transMatRate_tc_H of special_rare_common_objects
 */
typedef struct qa_rare_common qa_rare_common_t;
int qa_rare_common_create(qa_panel_t *panel, int32_t nSNPs_all, const uint8_t *snp_is_common,
                          const int64_t *rare_ptr, const int32_t *rare_snp_1based,
                          const double *transMatRate_t_all, qa_rare_common_t **out);
void qa_rare_common_destroy(qa_rare_common_t *rc);

/* get_initial_read_labels' read likelihoods (QUILT/R/rare_common.R:61-107) for many chains WITHOUT the haplotypes spread over all
 * SNPs on the host: hap_common is [n_chain][K][nSNPs_common] (hap-major: the layout the driver holds the last seek iteration's
 * haploid dosages in); the kernel supplies the 0.5 of the rare SNPs through rare_common's all-SNP -> common index.  The all-SNP
 * reads are given once per SAMPLE (n_sample, flattened as for qa_gibbs_batch) with chain_sample[c] (0-based) naming a chain's
 * sample.  eMatRead_t: chain after chain, [reads of the chain's sample][K].  Same numbers as qa_rcpp_make_eMatRead_t_nsnps on the
 * expanded haplotypes. */
int qa_rcpp_make_eMatRead_t_rare_common(qa_panel_t *panel, const qa_rare_common_t *rare_common, int32_t n_chain, int32_t n_sample,
                                        const int32_t *chain_sample, int32_t K, const double *hap_common, const int32_t *read_off,
                                        const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                                        double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                                        double *eMatRead_t);

/*
 * `_QUILT_rcpp_forwardBackwardGibbsNIPT` called with make_eMatRead_t_rare_common = TRUE
 * (QUILT/R/rare_common.R:325-398 via impute_one_sample; QUILT/src/gibbs-nipt.cpp:2805-2806 and :2305-2318):
 * read emissions by Rcpp_make_eMatRead_t_for_final_rare_common_gibbs_using_objects
 * (QUILT/src/gibbs-small.cpp:270-460), hapProbs / genProbs by
 * rcpp_calculate_genProbs_and_hapProbs_final_rare_common (:711-867).  Arguments as qa_gibbs_batch, except that
 * u indexes ALL SNPs, wif is the read's grid among the nGrids_all all-SNP grids, runif_shard holds
 * (nGrids_all - 1) uniforms per pass, and hapProbs_t / genProbs*_t are per chain 3 x nSNPs_all.
 * rare_per_snp_info (QUILT/R/rare_common.R:313-322) is derived here from rare_per_hap_info and which_haps_to_use.
 */
int qa_gibbs_batch_rare_common(qa_panel_t *panel, const qa_rare_common_t *rc, const qa_gibbs_opts_t *opts,
                               int32_t n_chain, const int32_t *which_haps_to_use_1based, const int32_t *read_off,
                               const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *wif,
                               const double *runif_reads, const int32_t *first_read, const double *runif_shard,
                               int32_t *H, int32_t *H_class, double *hapProbs_t, double *genProbsM_t,
                               double *genProbsF_t, int32_t *underflow_problem, double *state_out,
                               const uint64_t *seed_reads, const uint64_t *seed_shard);

/* ---- the per-sample driver loop for a range of samples ---------------------- */

/*
 * impute_rare_common = TRUE (QUILT/R/quilt.R:180; functions.R:1042-1123, rare_common.R:61-420): every Gibbs sample ends with one
 * Gibbs call over ALL SNPs (impute_final_gibbs_with_rare_common), started from labels drawn from the all-SNP reads' likelihoods
 * against the latest haploid dosages (get_initial_read_labels); results then cover all SNPs.
 *   handles          one qa_rare_common_t per panel handle, in the panels' order (qa_rare_common_create)
 *   snp_is_common    nSNPs_all flags (the common SNPs, in order, are the panel's)
 *   read_off .. wif  allSNP_sampleReads of the same samples, flattened like the common-SNP reads (u: 0-based all-SNP index, wif:
 *                    0-based all-SNP grid)
 */
typedef struct {
    const qa_rare_common_t *const *handles;
    int32_t nSNPs_all, nGrids_all;
    const uint8_t *snp_is_common;
    const int32_t *read_off, *read_ptr, *u, *bq, *wif;
    const int32_t *L_grid_all;   /* method = "nipt" only: nGrids_all positions (bp) of the all-SNP grid, for its block Gibbs */
} qa_impute_rare_common_t;

/*
 * method = "nipt" (mother + fetus, three read labels; functions.R:586, :1009-1016, :1218-1231, :1788-1829, :3214-3287): what the
 * loop needs beyond the diploid case.  dosage / gp_t of qa_impute_samples are then the MOTHER's, phasing_haps has THREE rows per
 * sample (n_sample x 3 x nSNPs: maternal transmitted, maternal untransmitted, paternal transmitted, after recast_nipt_haps).
 *   ff                   n_sample fetal fractions in (0, 1) (ff_values[iSample], functions.R:128)
 *   L_grid               nGrids grid positions (bp), shuffle_bin_radius (quilt.R:134: 5000): the block definition of the block Gibbs
 *   fet_dosage, fet_gp_t OUT, n_sample x nSNPs and n_sample x 3 x nSNPs: the fetus' (maternal transmitted + paternal transmitted);
 *                        over nSNPs_all with impute_rare_common (starting labels of the all-SNP call by read grouping:
 *                        gibbs-nipt.R:1655-1849; rare_common->L_grid_all is then needed)
 */
typedef struct {
    const double *ff;
    const int32_t *L_grid;
    int32_t shuffle_bin_radius;
    double *fet_dosage, *fet_gp_t;
} qa_impute_nipt_t;

/*
 * The reads of the call's samples handed over ONE SAMPLE AT A TIME, when the loop reaches them (params->sample_source) -- the
 * reference's loop loads a sample's BAM inside the loop body too (functions.R:251-298: get_sampleReads_from_dir_for_sample is the
 * first thing get_and_impute_one_sample does), so a range's first samples are on the device while its last ones are still on disk.
 *
 *   acquire(ctx, s, view)   called by the host thread that takes the launch set holding sample s, before anything of s is read;
 *                           for every s of a set in ascending order, once per s; concurrently for samples of DIFFERENT sets.  It
 *                           blocks until the sample is there and fills *view.  Returns
 *                             QA_OK              view filled: n_reads >= 1 reads in the per-sample form of the flat arrays (read_ptr
 *                                                n_reads + 1 offsets from 0 into u / bq; wif n_reads), with impute_rare_common the
 *                                                all-SNP reads too, and read_labels = where the sample's n_reads labels go
 *                             QA_END_OF_SAMPLES  the range ends BEFORE s: n_sample was an upper bound (a caller that drops samples
 *                                                with too few reads, functions.R:274-287, learns the count while loading).  Every
 *                                                later s ends too.
 *                             < 0                failure: the call fails with that status (the callback's text via qa_last_error
 *                                                if it set one)
 *   The views' arrays must stay valid until on_samples_done has covered the sample, or the call has returned.
 *   What is indexed by the sample -- params->sample_index[s], nipt->ff[s] -- is read only AFTER acquire(s) returned: the caller
 *   may fill those entries as it learns them.
 * With a source the flat read arrays and read_labels of qa_impute_samples may be NULL (they are not read); everything else --
 * results, random streams, launch plan -- is the same as for the same samples handed over flat.
 */
#define QA_END_OF_SAMPLES 2
typedef struct {
    int32_t n_reads;
    const int32_t *read_ptr, *u, *bq, *wif;
    int32_t n_reads_all;                                  /* impute_rare_common only */
    const int32_t *read_ptr_all, *u_all, *bq_all, *wif_all;
    int32_t *read_labels;
} qa_sample_view_t;
typedef struct {
    int (*acquire)(void *ctx, int32_t s, qa_sample_view_t *view);
    void *ctx;
} qa_sample_source_t;

/*
 * Arguments of QUILT() the hot path sees (QUILT/R/quilt.R:97-186), as get_and_impute_one_sample receives them.
 * qa_impute_params_default fills in the reference's defaults.
 */
typedef struct {
    int32_t nGibbsSamples;                      /* 7 */
    int32_t n_seek_its;                         /* 3 */
    int32_t n_burn_in_seek_its;                 /* -1 = NA: n_seek_its - 1 (quilt.R:248-250) */
    int32_t Ksubset, Knew;                      /* 600, 600; a panel with K < Ksubset: one seek iteration on all of it (quilt.R:453-471) */
    int32_t K_top_matches;                      /* 5 */
    double heuristic_match_thin;                /* 0.1 */
    int32_t small_ref_panel_gibbs_iterations;   /* 20 */
    int32_t n_gibbs_sample_its;                 /* 1 */
    const int32_t *small_ref_panel_block_gibbs_iterations;   /* {3, 6, 9} (0-based sweeps), */
    int32_t n_block_gibbs_iterations;           /* 3 */
    double maxDifferenceBetweenReads;           /* 1e10 */
    double minGLValue;                          /* 1e-10 */
    int32_t Jmax;                               /* 10000 */
    uint64_t seed;                              /* keys every (sample, Gibbs sample) stream together with the GLOBAL sample index */
    int32_t use_mspbwt;                         /* 0: full-panel passes (impute_using_everything); 1: msPBWT mode */
    int32_t mspbwtL, mspbwtM;                   /* 3, 1 */
    const qa_mspbwt_t *mspbwt_index;            /* use_mspbwt: qa_mspbwt_create(..., nindices = mspbwt_nindices) */
    int32_t samples_per_launch_set;             /* samples whose chains advance in one launch set; 0 = 256 (2 048 Gibbs chains:
                                                   two per SIMD) */
    int32_t no_fused_tails;                     /* 1: the threads' last launch sets run their phasing rounds one after the other */
    const qa_impute_rare_common_t *rare_common; /* NULL, or impute_rare_common = TRUE: dosage / gp_t / phasing_haps then cover
                                                   nSNPs_all SNPs and nDosage counts the all-SNP rounds (functions.R:1305-1317) */
    const qa_impute_nipt_t *nipt;               /* NULL (method = "diploid"), or method = "nipt" */
    const int64_t *sample_index;                /* NULL: sample i of the call is global sample sample_offset + i.  Else n_sample
                                                   global indices (ABI 5): the call's samples need not be consecutive among ALL
                                                   samples -- a range from which samples with too few reads were dropped
                                                   (functions.R:274-287) keeps every remaining sample's own streams */
    void (*on_samples_done)(void *ctx, int32_t lo, int32_t hi);   /* NULL, or called from a host thread of the call when samples
                                                   [lo, hi) of the call are FINAL (every output row of theirs written): a caller that
                                                   formats or writes results can do so while later launch sets are still on the device
                                                   (qa_impute_bam_range formats VCF columns this way).  Must not call back into the
                                                   library's device entry points; may be called concurrently for disjoint ranges */
    void *on_samples_done_ctx;
    const qa_sample_source_t *sample_source;    /* NULL: every sample's reads are in the flat arrays of the call.  Else the reads are
                                                   handed over sample by sample WHEN THE LOOP REACHES THEM (qa_sample_source_t above):
                                                   the first launch set starts while later samples are still being read from disk */
} qa_impute_params_t;
int qa_impute_params_default(qa_impute_params_t *params);

/*
 * get_and_impute_one_sample (QUILT/R/functions.R:3-1500) for samples [0, n_sample) of the caller's range (method = "diploid";
 * params->nipt: method = "nipt"; params->rare_common: impute_rare_common = TRUE; params->use_mspbwt: use_mspbwt = TRUE):
 * the body of the reference's loop over a core's sample range (QUILT/R/quilt.R:688-996: `for(iSample in sampleRange[1]:
 * sampleRange[2])` inside mclapply) as ONE call -- (nGibbsSamples + 1) x n_seek_its rounds of [small-panel Gibbs -> full-panel
 * pass per read label -> new small panel] per sample, accumulation over the rounds past the burn-in (functions.R:999-1020),
 * read confidence and consensus labels (:1615-1660, :1680-1784), the phasing iteration and recast_haps (:3180-3209); host C++
 * in csrc/impute.cpp over the batched entry points above.  All chains of a launch set of samples advance in lock-step; launch
 * sets are pipelined (one set's phasing rounds share launches with the next set's main rounds); n_panels host threads, one
 * per handle (replicas of the panel on one device, qa_panel_set_exclusive(1) so that their launch sets take the device in
 * turn), hide each other's host phases.
 *
 *   panels            n_panels (1..16) handles of the SAME panel; 3 is what the headline workload wants
 *   sample_offset     global index of sample 0: (seed, sample_offset + i, Gibbs sample) keys the draws, so a sample gets
 *                     the same result whichever range, rank or launch set it lands in (params->sample_index, when set,
                     names each sample's global index itself and sample_offset is ignored)
 *   read_off          n_sample + 1: reads of sample i are read_off[i] .. read_off[i + 1] - 1 (every sample needs >= 1)
 *   read_ptr          per sample R_i + 1 offsets (starting at 0) into that sample's bases; sample i's block starts at
 *                     read_ptr[read_off[i] + i]
 *   u, bq             bases of all samples back to back (0-based SNP index, signed base quality); wif: per read, 0-based grid
 *                     -- the flattened sampleReads of qa_bam_load_sample_reads / qa_gibbs_batch
 *   dosage            out n_sample x nSNPs: mean over the counted rounds of hap1 + hap2
 *   gp_t              out n_sample x 3 x nSNPs: genotype posteriors
 *   phasing_haps      out n_sample x 2 x nSNPs: the phasing iteration's haplotypes after recast_haps (R's phasing_haps is
 *                     its transpose per sample)
 *   read_labels       out, read_off[n_sample] entries: the consensus labels the phasing iteration started from
 *   nDosage           out n_sample: rounds counted
 *                     (the output arrays need not be initialised: the accumulators' rows are zeroed when a sample's launch set
 *                     starts, everything else is written whole)
 *   Left-over launch sets (their number is not a multiple of n_panels) go whole to the first threads; a call with fewer sets
 *   than threads is cut across them.  Results do not depend on the plan.
 *   stats             NULL, or 11 counters: [0] underflow retries, [1] chains that needed complete best-haplotype lists,
 *                     [2] selections made on the device, [3] chains handed to qa_gibbs_batch, [4] Gibbs launch sets,
 *                     [5..10] ms summed over the host threads: Gibbs calls, full-panel calls, host, consensus, finish, accumulation
 * Random draws: R's stream cannot be reproduced without R; every draw the R code makes is defined on a counter stream
 * (quilt_amd/rng.py::ChainStream = csrc/impute.cpp).
 */
int qa_impute_samples(qa_panel_t *const *panels, int32_t n_panels, const qa_impute_params_t *params, int32_t n_sample,
                      int64_t sample_offset, const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                      const int32_t *wif, double *dosage, double *gp_t, double *phasing_haps, int32_t *read_labels,
                      int32_t *nDosage, int64_t *stats);

/* The host threads' marshalling and pinned transfer buffers are kept per panel handle between calls (a launch set of 2 048
 * chains moves ~3.5 GB through them): this frees them all.  Call it when no qa_impute_samples call is running. */
int qa_impute_release_buffers(void);
/* Diagnostic: the number of panel handles that currently keep such buffers (qa_panel_destroy drops its handle's). */
int qa_impute_kept_buffers(void);

#ifdef __cplusplus
}
#endif
#endif
