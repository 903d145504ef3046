/* quilt_amd_io.h -- host-side data formats either side of the hot path.
 *
 * Plain C ABI, no device work: these run on the host thread that feeds / drains the device workers of quilt_amd.h.
 *
 *   f3  BAM -> sampleReads       replaces STITCH::loadBamAndConvert + snap_sampleReads_to_grid as called from
 *                                QUILT/R/functions.R:243-298 (per-sample RData temp files and R lists of 4-element lists
 *                                become one flattened CSR, the layout qa_gibbs_batch / qa_fullpass_reads_batch take)
 *   f4  per-sample VCF column    replaces STITCH::rcpp_make_column_of_vcf and the paste0 assembly of
 *                                QUILT/R/functions.R:1408-1463, and the body writer QUILT/R/writers.R:80-128
 *                                (data.table::fwrite + bgzip become one BGZF stream written here)
 *
 * STITCH (1.8.4) is not vendored in the reference tree, so the parts that live in STITCH are restated from its documented
 * behaviour and from QUILT's call sites / parameter documentation (QUILT/R/quilt.R:30-56); where a choice is not visible
 * from QUILT (tie rules, the central SNP of an even-length read, which reads a coverage cap removes) the rule used here is
 * stated next to the function and is UNPINNED against STITCH.  The NIPT column and the header/INFO assembly are QUILT's
 * own R code and follow it exactly.
 */
#ifndef QUILT_AMD_IO_H
#define QUILT_AMD_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------------
 * f3: BAM -> sampleReads
 * ---------------------------------------------------------------------------------------------------------------------- */

typedef struct qa_sample_reads qa_sample_reads_t;   /* opaque; owned by the library */

typedef struct {
    int32_t bqFilter;              /* 17   minimum base quality of a SNP in a read; base quality is capped at the read's
                                    *      mapping quality, and a read whose mapping quality is below it is not used
                                    *      (QUILT/R/quilt.R:30) */
    int32_t iSizeUpperLimit;       /* 1e6  reads with |template length| above this are not used (quilt.R:42) */
    int32_t useSoftClippedBases;   /* 0    whether bases in soft-clipped parts count (quilt.R:31) */
    int32_t downsampleToCov;       /* 30   per-SNP coverage cap (quilt.R:54); <= 0: no cap */
    int32_t chrStart, chrEnd;      /* 1-based inclusive window alignments must overlap; 0, 0 = the whole chromosome
                                    *      (functions.R:262-263: regionStart - buffer .. regionEnd + buffer) */
    int32_t merge_mates;           /* 1    two alignments with the same query name become ONE read (they come from one
                                    *      molecule, hence one haplotype), bases in SNP order; a site both mates cover counts
                                    *      once: calls that agree keep the higher quality, calls that disagree drop the site */
    uint64_t seed;                 /* key of the counter stream used where the loader has to choose (coverage cap) */
} qa_bam_opts_t;

void qa_bam_opts_default(qa_bam_opts_t *opts);

/* Scan a (BGZF-compressed) BAM for alignments on `chr`, pile their bases onto the nSNPs sites and return the reads that
 * cover at least one site.
 *   L            nSNPs ascending 1-based positions (pos[, 2]); ref / alt: nSNPs bytes each, the two alleles of every site
 *                (pos[, 3:4]; sites are biallelic SNPs -- a base that is neither allele is skipped)
 *   grid         nSNPs 0-based grid index of every site (STITCH::assign_positions_to_grid; 32 SNPs per grid: site / 32)
 * A base enters with bq = +q when it shows the alternate allele and -q for the reference allele (the sign convention of
 * sampleReads[[r]][[3]], consumed at QUILT/src/gibbs-nipt.cpp:125-141), q = min(base quality, mapping quality), and is
 * dropped when q < bqFilter.  Unmapped, secondary, supplementary, duplicate and QC-fail alignments are skipped.
 * Unpinned-vs-STITCH rules: overlapping mates (above); the read's central SNP is its (n - 1) / 2-th site (lower median); the coverage cap visits sites
 * in ascending order and, at a site above the cap, drops the covering reads with the smallest counter-stream keys
 * (key = stream(seed, read index)) until the site is at the cap.
 * A coordinate-sorted file with a BAI index beside it (<file>.bai or <file without .bam>.bai) is entered at the linear index's
 * offset for the window's first 16 kb interval, tightened by the chunks of the bins that can hold an overlapping alignment
 * (SAM spec 5.1.3, 5.2, 5.3), instead of being scanned from the top; without a usable index the file is scanned
 * sequentially (and the scan stops once a sorted file has passed the window).
 * CRAM (`cramlist` + `reference`, QUILT/R/quilt.R:106-108) is not decoded: QA_ERR_UNSUPPORTED, and qa_last_error() holds the
 * `samtools view -b -T <reference.fa>` command that converts the file.
 * Reads come back ordered by the grid of their central SNP (stable), as snap_sampleReads_to_grid leaves them
 * (functions.R:295-298).  QA_ERR_INVALID for unreadable / malformed files and unknown chromosome names. */
int qa_bam_load_sample_reads(const char *bam_path, const char *chr, int32_t nSNPs, const int32_t *L, const char *ref,
                             const char *alt, const int32_t *grid, const qa_bam_opts_t *opts, qa_sample_reads_t **out);

int32_t qa_sample_reads_n_reads(const qa_sample_reads_t *s);
int64_t qa_sample_reads_n_bases(const qa_sample_reads_t *s);
/* counts for the log lines of functions.R:289-290 and the loader's filters: [0] alignments seen on chr, [1] used,
 * [2] dropped by mapping quality, [3] by insert size, [4] by flags, [5] reads removed by the coverage cap,
 * [6] mate pairs merged, [7] alignments without a site */
void qa_sample_reads_stats(const qa_sample_reads_t *s, int64_t stats[8]);
/* read_ptr[n_reads + 1], u / bq [n_bases] (0-based site, signed quality), wif [n_reads] (0-based grid of the central SNP),
 * central [n_reads] (0-based central site); any pointer may be NULL */
int qa_sample_reads_export(const qa_sample_reads_t *s, int32_t *read_ptr, int32_t *u, int32_t *bq, int32_t *wif,
                           int32_t *central);
void qa_sample_reads_destroy(qa_sample_reads_t *s);

/* ------------------------------------------------------------------------------------------------------------------------
 * f4: per-sample VCF column and the VCF body
 * ---------------------------------------------------------------------------------------------------------------------- */

/* One sample's column of the VCF body, method = "diploid" (functions.R:1420-1441), FORMAT GT:GP:DS:HD.
 *   gp_t           3 x nSNPs column-major genotype posteriors; phasing_haps nSNPs x 2 column-major haploid dosages
 *   phased_gt      output_gt_phased_genotypes: GT becomes round(hd1)|round(hd2) (R's round: half to even)
 * Entry t is "GT:gp0,gp1,gp2:ds:hd1,hd2" with three decimals (%.3f); unphased GT is the genotype whose posterior is >= 0.9,
 * "./." when none is (the threshold rule lives in STITCH: unpinned).  Entries are written back to back into buf, each
 * terminated by '\0'; off[t] is entry t's start, off[nSNPs] the bytes used.  QA_ERR_CAPACITY when cap is too small: *needed
 * is set and nothing else written (a second call with that capacity succeeds). */
int qa_vcf_column_diploid(int32_t nSNPs, const double *gp_t, const double *phasing_haps, int32_t phased_gt, char *buf,
                          int64_t cap, int64_t *off, int64_t *needed);

/* method = "nipt" (functions.R:1443-1459), FORMAT GT:MGP:MDS:FGP:FDS:
 *   "h1|h2|h3:m0,m1,m2:mds:f0,f1,f2:fds" where every number is R's paste0(round(x, 3)): three decimals, trailing zeros and a
 *   trailing point removed ("0.5", "1", "0").  phasing_haps nSNPs x 3 column-major. */
int qa_vcf_column_nipt(int32_t nSNPs, const double *mat_gp_t, const double *fet_gp_t, const double *phasing_haps,
                       const double *mat_dosage, const double *fet_dosage, char *buf, int64_t cap, int64_t *off,
                       int64_t *needed);

/* The column of a sample that was not imputed (fewer than minimum_number_of_sample_reads reads; functions.R:274-287):
 * every entry "./.:.,.,.:.:.,." */
const char *qa_vcf_missing_entry(void);

/* INFO strings from the cross-sample sums (writers.R:38-58, 73-80): per site
 *   "EAF=..;INFO_SCORE=..;HWE=..;ERC=..;EAC=..;PAF=.." with R's round(x, 5) / formatC(hwe, format = "e", digits = 2)
 *   eaf, info, hwe: nSNPs each; alleleCount nSNPs x 3 column-major (ref-count, total-count, frequency as in writers.R:76-78:
 *   ERC = alleleCount[, 1], EAC = alleleCount[, 2] - alleleCount[, 1], PAF = alleleCount[, 3]) */
int qa_vcf_info_column(int32_t nSNPs, const double *eaf, const double *info, const double *hwe, const double *alleleCount,
                       char *buf, int64_t cap, int64_t *off, int64_t *needed);

/* Exact Hardy-Weinberg p-value per site from the counts of most-likely genotypes (writers.R:58; replaces
 * STITCH::generate_hwe_on_counts -- the exact test of Wigginton et al. 2005; that STITCH uses this test is its documented
 * behaviour, unpinned).  counts nSNPs x 3 column-major (hom-ref, het, hom-alt), rounded to integers. */
int qa_hwe_exact(int32_t nSNPs, const double *counts, double *p_out);

/* Write the VCF body (writers.R:81-117): one line per site with keep[t] != 0 (inRegion2),
 *   chr \t pos \t . \t ref \t alt \t . \t PASS \t INFO \t FORMAT \t col_1 ... col_N \n
 * appended to `path`.  bgzf != 0: the bytes are written as BGZF blocks (what `bgzip` produces from the text, including the
 * end-of-file marker when finish != 0), so header and body can be appended in turn and `tabix` can index the result.
 *   chr            chromosome name; pos_bp nSNPs 1-based positions; ref / alt nSNPs bytes
 *   info / format  INFO entries (buf + off as produced by qa_vcf_info_column) and the FORMAT string
 *   cols / offs    N pointers to column buffers and their offset arrays (as produced by qa_vcf_column_*); a NULL column is
 *                  an unimputed sample
 */
int qa_vcf_write_body(const char *path, int32_t bgzf, int32_t finish, const char *chr, int32_t nSNPs, const int32_t *pos_bp,
                      const char *ref, const char *alt, const uint8_t *keep, const char *info, const int64_t *info_off,
                      const char *format, int32_t N, const char *const *cols, const int64_t *const *offs);
/* Append raw text (the VCF header of writers.R:1-36) through the same BGZF framing. */
int qa_vcf_write_text(const char *path, int32_t bgzf, int32_t truncate, const char *text, int64_t n);

/* ------------------------------------------------------------------------------------------------------------------------
 * f3 + the path + f4 for a whole sample range: BAM paths in, VCF columns and the range's count arrays out
 *
 * The body of QUILT()'s loop over a core's sample range (QUILT/R/quilt.R:832-982) with get_and_impute_one_sample's own I/O
 * either side of the imputation (functions.R:243-298 load, :1380-1463 counts and column) as ONE native call: the BAM files are
 * read on host threads in file order (qa_bam_load_sample_reads) BESIDE the imputation, the samples with at least
 * minimum_number_of_sample_reads reads go through ONE qa_impute_samples call (include/quilt_amd.h) that is handed each sample when
 * the launch set holding it is taken (qa_sample_source_t: the first launches start when their own files are read, not when the
 * last file of the range is), the columns of every finished launch set are formatted on host threads (qa_vcf_column_*) while
 * later sets are on the device, and the four per-SNP count arrays the loop keeps (quilt.R:955-961) are summed over the imputed
 * samples in sample order.  Serially in R these two ends cost about a second per sample; the device imputes ~40 samples per
 * second.
 * ---------------------------------------------------------------------------------------------------------------------- */
#include "quilt_amd.h"

typedef struct {
    const char *chr;
    int32_t nSNPs;                  /* the panel's (common) SNPs: L ascending 1-based positions, ref / alt one byte per site, */
    const int32_t *L;               /* grid the 0-based grid of every site -- the arguments of qa_bam_load_sample_reads       */
    const char *ref, *alt;
    const int32_t *grid;
    int32_t nSNPs_all;              /* impute_rare_common (params->rare_common set): the same four over ALL SNPs (pos_all,     */
    const int32_t *L_all;           /* special_rare_common_objects$grid; functions.R:132-172); else 0 / NULL                   */
    const char *ref_all, *alt_all;
    const int32_t *grid_all;
    qa_bam_opts_t bam;              /* loader options (qa_bam_opts_default) */
    int32_t minimum_number_of_sample_reads;   /* 2 (quilt.R:132): samples below it are not imputed (functions.R:274-287) */
    int32_t output_gt_phased_genotypes;       /* 1 (quilt.R:153) */
    int32_t n_io_threads;                     /* host threads for loading and for formatting (each); 0 = min(16, hardware threads) */
    int32_t discard_sample_arrays;            /* 1: a caller that wants the columns, labels and counts only (the R fast path): the pages
                                                 of a sample's dosage / gp_t / phasing_haps rows (48 bytes per SNP; 7.9 GB at 2 560
                                                 samples x 64 000 SNPs) go back to the system once its column is formatted, and
                                                 qa_bam_range_sample returns NULL for those arrays */
} qa_bam_range_io_t;

typedef struct qa_bam_range_result qa_bam_range_result_t;   /* opaque; owned by the library */

/*   panels, n_panels, params   as for qa_impute_samples.  params->sample_index, params->sample_source, the read arrays inside params->rare_common, and ff /
 *                              fet_dosage / fet_gp_t inside params->nipt are ignored: this call fills them for the samples it keeps
 *                              (params->rare_common: handles, nSNPs_all, nGrids_all, snp_is_common, L_grid_all; params->nipt:
 *                              L_grid, shuffle_bin_radius are the caller's)
 *   bam_paths                  n_sample files, one sample each (bamlist order)
 *   sample_index               n_sample GLOBAL 0-based indices (iSample - 1): every sample keeps its own random streams whichever
 *                              samples of the range are dropped for too few reads
 *   ff                         method = "nipt": n_sample fetal fractions (ff_values[iSample]); else NULL
 * A file that cannot be read fails the call (QA_ERR_INVALID / QA_ERR_UNSUPPORTED for CRAM, qa_last_error names the file). */
int qa_impute_bam_range(qa_panel_t *const *panels, int32_t n_panels, const qa_impute_params_t *params, const qa_bam_range_io_t *io,
                        int32_t n_sample, const char *const *bam_paths, const int64_t *sample_index, const double *ff,
                        qa_bam_range_result_t **out);

int32_t qa_bam_range_n_samples(const qa_bam_range_result_t *r);
int32_t qa_bam_range_n_snps(const qa_bam_range_result_t *r);              /* all SNPs with impute_rare_common */
int32_t qa_bam_range_imputed(const qa_bam_range_result_t *r, int32_t i); /* sample_was_imputed */
int32_t qa_bam_range_n_reads(const qa_bam_range_result_t *r, int32_t i); /* reads loaded (the number the log line of functions.R:280 prints) */
/* per_sample_vcf_col of sample i: entries back to back, NUL-terminated, off[t] = start of entry t, off[nSNPs] = bytes (the layout
 * of qa_vcf_column_* and qa_vcf_write_body); *buf = *off = NULL for a sample that was not imputed (qa_vcf_missing_entry) */
int qa_bam_range_column(const qa_bam_range_result_t *r, int32_t i, const char **buf, const int64_t **off);
/* the numbers behind the column, in qa_impute_samples' layouts (dosage nSNPs, gp_t 3 x nSNPs row by row, phasing_haps 2 (nipt: 3)
 * x nSNPs row by row; fet_*: nipt only), the consensus read labels and the rounds counted; any pointer may be NULL; NULL / 0 come
 * back for a sample that was not imputed */
int qa_bam_range_sample(const qa_bam_range_result_t *r, int32_t i, const double **dosage, const double **gp_t, const double **phasing_haps,
                        const double **fet_dosage, const double **fet_gp_t, const int32_t **read_labels, int32_t *n_labels, int32_t *nDosage);
/* the range's sums over its imputed samples, in sample order (quilt.R:955-961), column-major like R's arrays:
 * infoCount nSNPs x 2 (sum eij, sum fij - eij^2), afCount nSNPs (sum eij / 2), hweCount nSNPs x 3 (most likely genotype counts),
 * alleleCount nSNPs x 2 (pile-up: alt, ref + alt); any pointer may be NULL */
int qa_bam_range_counts(const qa_bam_range_result_t *r, double *infoCount, double *afCount, double *hweCount, double *alleleCount);
/* seconds: [0] when the last file was loaded (the loading runs beside the imputation), [1] qa_impute_samples, [2] the formatting
 * and counts left when it returned, [3] the whole call; impute_stats: qa_impute_samples' 11
 * counters; load_stats: qa_sample_reads_stats summed over the files */
void qa_bam_range_timings(const qa_bam_range_result_t *r, double seconds[4], int64_t impute_stats[11], int64_t load_stats[8]);
void qa_bam_range_destroy(qa_bam_range_result_t *r);

/* ------------------------------------------------------------------------------------------------------------------------
 * driver-side accumulation (QUILT/R/functions.R:999-1020): all chains of a round in one pass
 *   hap            n_chain x n_label x nSNPs haploid dosages of the round's full-panel passes (as qa_fullpass_reads_batch
 *                  returns them); chain_sample: the sample of each chain
 *   dosage, gp_t   n_sample x nSNPs and n_sample x 3 x nSNPs running sums: dosage += h1 + h2,
 *                  gp_t += rbind((1-h1)(1-h2), (1-h1) h2 + h1 (1-h2), h1 h2), chain by chain in order
 *   fet_*          NIPT (n_label = 3; both or neither): the same with (h1, h3) (functions.R:1009-1016)
 * ---------------------------------------------------------------------------------------------------------------------- */
int qa_accumulate_dosage(int32_t n_chain, int32_t n_label, int32_t nSNPs, const double *hap, const int32_t *chain_sample,
                         int32_t n_sample, double *dosage, double *gp_t, double *fet_dosage, double *fet_gp_t);

/* The sequential match weighting of select_new_haps_mspbwt_v3 (QUILT/R/mspbwt.R:418-427; msPBWT mode): matches in the given
 * order, weight[i] = (end1 - start1 + 1) / sum(cur_sum[start1..end1]) with cur_sum (all ones at first) incremented over each
 * match's span after it is weighted.  1-based inclusive coordinates. */
int qa_mspbwt_weights(int32_t n, const int64_t *start1, const int64_t *end1, double *weight);

/* select_new_haps_mspbwt_v3 (QUILT/R/mspbwt.R:303-474) for a batch of chains, from the match tables of qa_find_good_matches
 * (include/quilt_amd.h): match n_chain x n_label x nindices x max_matches x 3, n_match n_chain x n_label x nindices; seed: one
 * selection-stream key per chain (R's sample() = the smallest keys at offset 2^21, in key order); out n_chain x Knew, 1-based. */
int qa_select_new_haps_mspbwt(int32_t n_chain, int32_t n_label, int32_t nindices, int32_t max_matches, const int32_t *match,
                              const int32_t *n_match, int32_t Knew, int32_t Kfull, int32_t nGrids, const uint64_t *seed,
                              int32_t *out);

/* Read confidence and consensus read labels of one sample before its phasing pass (QUILT/R/functions.R:1615-1660, :1680-1784,
 * NIPT :1788-1829).  labels n x nReads (Gibbs-sample-major); p n x K x nReads: the reads' likelihoods against each Gibbs sample's
 * K haplotypes (K = 2, or 3 for NIPT: label 3 is folded into 2 for the consensus and put back); minrp 0.95; can_hap 1-based. */
int qa_consensus_read_labels(int32_t nReads, int32_t n, const int32_t *labels, const double *p, int32_t K, double minrp,
                             int32_t can_hap, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
