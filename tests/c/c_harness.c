/*
 * c_harness.c -- the C ABI called from plain C (no Python, no ctypes): qa_panel_create -> qa_gibbs_batch ->
 * qa_Rcpp_haploid_dosage_versus_refs on a small deterministic panel, the way the NR shim (shim/quilt_amd_shim.c) drives it.
 * Test infrastructure.  Exit code 0 and "HARNESS_OK" when every check holds on a gfx950 device; "NO_DEVICE" (exit 0) when
 * the library reports QA_ERR_NO_DEVICE (it has no CPU fallback); anything else is a failure.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/quilt_amd.h"
#include "../../include/quilt_amd_io.h"

#define NK 512
#define NT 256
#define NG 8
#define NMAXDH 16
#define NWORDS 8
#define NR 96
#define NKS 64

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s; }
#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL: %s (%s)\n", msg, qa_last_error()); return 1; } } while (0)

int main(void) {
    uint32_t seed = 12345;
    static uint8_t hm[NK * NG];
    static int32_t B[NMAXDH * NG], which_grid[NG], helper[2 * NG], spmat[2];
    static double tm[2 * (NG - 1)];
    for (int g = 0; g < NG; g++) {
        for (int d = 0; d < NMAXDH; d++) B[g * NMAXDH + d] = d < NWORDS ? (int32_t)lcg(&seed) : 0;
        for (int k = 0; k < NK; k++) hm[g * NK + k] = (uint8_t)(1 + lcg(&seed) % NWORDS);   /* column-major NK x NG */
    }
    for (int g = 0; g < NG - 1; g++) { tm[2 * g] = 0.99; tm[2 * g + 1] = 1 - 0.99; }
    qa_panel_desc_t d;
    memset(&d, 0, sizeof d);
    d.K = NK; d.nGrids = NG; d.nSNPs = NT; d.nMaxDH = NMAXDH;
    d.hapMatcherR = hm; d.distinctHapsB = B; d.eMatDH_special_grid_which = which_grid;
    d.eMatDH_special_matrix_helper = helper; d.eMatDH_special_matrix = spmat; d.eMatDH_special_matrix_nrow = 0;
    d.use_eMatDH_special_symbols = 1; d.transMatRate_t = tm; d.ref_error = 1e-3;
    printf("abi %d, devices %d\n", qa_abi_version(), qa_device_count());
    {   /* the host-side formats (include/quilt_amd_io.h) need no device: one VCF column, INFO strings, the consensus labels */
        const double gp[3 * 2] = {1.0, 0.0, 0.0, 0.05, 0.9, 0.05}, hd[2 * 2] = {0.0, 0.6, 0.0, 0.4};   /* 3 x 2; 2 x 2 col-major */
        char buf[256];
        int64_t off[3], need = 0;
        CHECK(qa_vcf_column_diploid(2, gp, hd, 0, buf, sizeof buf, off, &need) == QA_OK, "qa_vcf_column_diploid");
        CHECK(strcmp(buf + off[0], "0/0:1.000,0.000,0.000:0.000:0.000,0.000") == 0, "VCF entry 0");
        CHECK(strcmp(buf + off[1], "0/1:0.050,0.900,0.050:1.000:0.600,0.400") == 0, "VCF entry 1");
        CHECK(qa_vcf_column_diploid(2, gp, hd, 1, buf, 8, off, &need) == QA_ERR_CAPACITY && need > 8, "capacity protocol");
        CHECK(strcmp(qa_vcf_missing_entry(), "./.:.,.,.:.:.,.") == 0, "missing entry");
        const double counts[3] = {25, 50, 25};
        double pv = 0;
        CHECK(qa_hwe_exact(1, counts, &pv) == QA_OK && pv > 0.9 && pv <= 1.0, "qa_hwe_exact");
        int32_t lab[2 * 12], cons[12];
        double pr[2 * 2 * 12];
        for (int c = 0; c < 2; c++)
            for (int r = 0; r < 12; r++) {
                lab[c * 12 + r] = 1 + (r & 1);
                pr[(c * 2 + 0) * 12 + r] = (r & 1) ? 0.01 : 0.99;
                pr[(c * 2 + 1) * 12 + r] = (r & 1) ? 0.99 : 0.01;
            }
        CHECK(qa_consensus_read_labels(12, 2, lab, pr, 2, 0.95, 2, cons) == QA_OK, "qa_consensus_read_labels");
        for (int r = 0; r < 12; r++) CHECK(cons[r] == 1 + (r & 1), "agreeing Gibbs samples keep their labels");
        printf("HOST_FORMATS_OK\n");
    }
    qa_panel_t *panel = NULL;
    int st = qa_panel_create(&d, &panel);
    if (st == QA_ERR_NO_DEVICE) { printf("NO_DEVICE\n"); return 0; }
    CHECK(st == QA_OK && panel, "qa_panel_create");

    /* reads: 2 SNPs each, sorted by grid; truth = haplotypes 3 and 7 of the panel */
    static int32_t read_ptr[NR + 1], u[2 * NR], bq[2 * NR], wif[NR], H[NR], Hc[NR], which[NKS], first_read[1] = {5};
    int32_t read_off[2] = {0, NR};
    for (int r = 0; r < NR; r++) {
        const int s0 = (int)((long)r * (NT - 2) / NR);
        read_ptr[r] = 2 * r;
        wif[r] = s0 / 32;
        const int truth = (r & 1) ? 3 : 7;
        for (int j = 0; j < 2; j++) {
            const int snp = s0 + j, g = snp / 32;
            const uint32_t w = (uint32_t)B[g * NMAXDH + hm[g * NK + truth] - 1];
            u[2 * r + j] = snp;
            bq[2 * r + j] = ((w >> (snp % 32)) & 1u) ? 30 : -30;
        }
        H[r] = 1 + (int)(lcg(&seed) % 2);
    }
    read_ptr[NR] = 2 * NR;
    for (int k = 0; k < NKS; k++) which[k] = 1 + k * (NK / NKS);   /* 1-based, ascending; includes 1 + 0*8 ... */
    which[0] = 4; which[1] = 8;                                /* haplotypes 3 and 7 (0-based) are in the small panel */
    const int n_its = 21, nb = 3;
    static int32_t blocks[3] = {3, 6, 9};
    double *runif_reads = malloc(sizeof(double) * NR * n_its), *runif_shard = malloc(sizeof(double) * nb * (NG - 1));
    for (int i = 0; i < NR * n_its; i++) runif_reads[i] = (lcg(&seed) >> 8) / 16777216.0;
    for (int i = 0; i < nb * (NG - 1); i++) runif_shard[i] = (lcg(&seed) >> 8) / 16777216.0;
    qa_gibbs_opts_t o;
    memset(&o, 0, sizeof o);
    o.Ks = NKS; o.ff = 0; o.sample_is_diploid = 1; o.Jmax = 10000; o.maxDifferenceBetweenReads = 1e10; o.rescale_eMatRead_t = 1;
    o.n_gibbs_burn_in_its = 20; o.n_gibbs_sample_its = 1; o.block_gibbs_iterations = blocks; o.n_block_gibbs_iterations = nb;
    o.perform_block_gibbs = 1; o.do_shard_block_gibbs = 1; o.gibbs_initialize_iteratively = 1; o.class_sum_cutoff = 0.06;
    static double per_it[21 * 8];
    o.per_it_out = per_it;
    static double hap[3 * NT], gm[3 * NT], gf[3 * NT];
    int32_t underflow = 0;
    st = qa_gibbs_batch(panel, &o, 1, which, read_off, read_ptr, u, bq, wif, runif_reads, first_read, runif_shard, H, Hc, hap, gm,
                        gf, &underflow, NULL, NULL, NULL);
    CHECK(st == QA_OK && !underflow, "qa_gibbs_batch");
    int n1 = 0;
    for (int r = 0; r < NR; r++) { CHECK(H[r] == 1 || H[r] == 2, "labels are 1 or 2"); n1 += H[r] == 1; }
    /* the two truth haplotypes separate the reads: labels agree with the read parity up to the naming of the two labels */
    int agree = 0;
    for (int r = 0; r < NR; r++) agree += (H[r] == 1) == ((r & 1) == 0);
    if (agree < NR / 2) agree = NR - agree;
    CHECK(agree >= NR * 2 / 3, "Gibbs labels follow the two haplotypes (reads informative at ~3 in 4 SNP pairs)");
    for (int t = 0; t < NT; t++)
        for (int h = 0; h < 2; h++) CHECK(hap[3 * t + h] >= 0 && hap[3 * t + h] <= 1, "hapProbs in [0, 1]");
    CHECK(per_it[20 * 8 + 3] + per_it[20 * 8 + 4] == NR && per_it[20 * 8 + 3] == n1, "per-sweep label counts");
    CHECK(isfinite(per_it[20 * 8]) && isfinite(per_it[20 * 8 + 1]), "per-sweep -sum(log c)");
    /* uniform sources are validated (no uninitialised uniforms) */
    st = qa_gibbs_batch(panel, &o, 1, which, read_off, read_ptr, u, bq, wif, runif_reads, first_read, NULL, H, Hc, NULL, NULL, NULL,
                        &underflow, NULL, NULL, NULL);
    CHECK(st == QA_ERR_INVALID, "shard passes without uniforms are rejected");

    /* full-panel pass for label 1 */
    static double gl[2 * NT], dosage[NT], c[NG], best_val[4096];
    static int32_t cols[NG], best_ptr[NG + 1], best_idx[4096];
    for (int t = 0; t < 2 * NT; t++) gl[t] = 1.0;
    for (int r = 0; r < NR; r++) {
        if (H[r] != 1) continue;
        for (int j = 0; j < 2; j++) {
            const int t = u[2 * r + j], b = bq[2 * r + j];
            const double e = pow(10, -abs(b) / 10.0);
            gl[2 * t] *= b < 0 ? 1 - e : e / 3;
            gl[2 * t + 1] *= b < 0 ? e / 3 : 1 - e;
        }
    }
    int n_thin = 0;
    for (int g = 0; g < NG; g++) cols[g] = (g % 2) ? n_thin++ : -1;
    qa_fullpass_opts_t fo;
    memset(&fo, 0, sizeof fo);
    fo.K_top_matches = 5; fo.return_dosage = 1; fo.get_best_haps_from_thinned_sites = 1; fo.normalize_emissions = 1;
    fo.min_emission_prob_normalization_threshold = 1e-100;
    st = qa_Rcpp_haploid_dosage_versus_refs(panel, gl, cols, &fo, NULL, NULL, c, NULL, NULL, dosage, best_ptr, best_idx, best_val, 4096);
    CHECK(st == QA_OK, "qa_Rcpp_haploid_dosage_versus_refs");
    for (int t = 0; t < NT; t++) CHECK(dosage[t] >= -1e-6 && dosage[t] <= 1 + 1e-6, "dosage in [0, 1]");
    for (int g = 0; g < NG; g++) CHECK(isfinite(c[g]) && c[g] > 0, "c finite and positive");
    for (int i = 0; i < n_thin; i++) {
        CHECK(best_ptr[i + 1] - best_ptr[i] >= 5, "at least K_top_matches per list");
        for (int q = best_ptr[i] + 1; q < best_ptr[i + 1]; q++) CHECK(best_idx[q] > best_idx[q - 1], "lists ascend in k");
    }
    /* the truth haplotype of label 1's reads is a top match somewhere */
    const int want = (agree == 0 ? 0 : 1), truth1 = -1;
    (void)want; (void)truth1;
    int found = 0;
    for (int q = 0; q < best_ptr[n_thin]; q++) found |= best_idx[q] == 3 || best_idx[q] == 7;
    CHECK(found, "a truth haplotype is among the best matches");
    /* the same call with its dosage in a qa_host_alloc buffer (pinned: no staging copy): the same bytes */
    double *pinned = (double *)qa_host_alloc(sizeof(double) * NT);
    CHECK(pinned != NULL, "qa_host_alloc");
    st = qa_Rcpp_haploid_dosage_versus_refs(panel, gl, cols, &fo, NULL, NULL, c, NULL, NULL, pinned, best_ptr, best_idx, best_val, 4096);
    CHECK(st == QA_OK, "qa_Rcpp_haploid_dosage_versus_refs into a pinned buffer");
    CHECK(memcmp(pinned, dosage, sizeof(double) * NT) == 0, "pinned and staged dosages are identical");
    CHECK(qa_host_free(pinned) == QA_OK, "qa_host_free");
    CHECK(qa_host_free(dosage) == QA_ERR_INVALID, "qa_host_free rejects foreign pointers");
    {   /* the per-sample driver loop behind the ABI: two samples (the same reads twice: different global indices, so different
         * draws), two host threads = two handles taking the device in turn; then one thread: the same bytes */
        qa_panel_t *panel2 = NULL;
        CHECK(qa_panel_create(&d, &panel2) == QA_OK && panel2, "second handle");
        CHECK(qa_panel_set_device_share(panel, 2) == QA_OK && qa_panel_set_device_share(panel2, 2) == QA_OK, "device share");
        CHECK(qa_panel_set_exclusive(panel, 1) == QA_OK && qa_panel_set_exclusive(panel2, 1) == QA_OK, "device phases");
        qa_impute_params_t ip;
        CHECK(qa_impute_params_default(&ip) == QA_OK, "qa_impute_params_default");
        ip.nGibbsSamples = 3; ip.Ksubset = NKS; ip.Knew = NKS; ip.seed = 7; ip.samples_per_launch_set = 1;
        static int32_t i_read_off[3] = {0, NR, 2 * NR}, i_read_ptr[2 * (NR + 1)], i_u[4 * NR], i_bq[4 * NR], i_wif[2 * NR];
        for (int s2 = 0; s2 < 2; s2++) {
            memcpy(i_read_ptr + s2 * (NR + 1), read_ptr, sizeof read_ptr);
            memcpy(i_u + s2 * 2 * NR, u, sizeof u);
            memcpy(i_bq + s2 * 2 * NR, bq, sizeof bq);
            memcpy(i_wif + s2 * NR, wif, sizeof wif);
        }
        static double i_dos[2][2 * NT], i_gp[2][2 * 3 * NT], i_ph[2][2 * 2 * NT];
        static int32_t i_lab[2][2 * NR], i_nd[2][2];
        int64_t stats[11];
        qa_panel_t *both[2] = {panel, panel2};
        st = qa_impute_samples(both, 2, &ip, 2, 100, i_read_off, i_read_ptr, i_u, i_bq, i_wif, i_dos[0], i_gp[0], i_ph[0], i_lab[0],
                               i_nd[0], stats);
        CHECK(st == QA_OK, "qa_impute_samples (two host threads)");
        CHECK(stats[3] >= 2 * 4 * 3 && stats[4] >= 3, "chains handed to the Gibbs entry point");
        for (int s2 = 0; s2 < 2; s2++) {
            CHECK(i_nd[0][s2] == 3, "nGibbsSamples x (n_seek_its - n_burn_in_seek_its) rounds counted");
            for (int t = 0; t < NT; t++) {
                const double *g3 = i_gp[0] + (size_t)s2 * 3 * NT;
                CHECK(i_dos[0][s2 * NT + t] >= -1e-6 && i_dos[0][s2 * NT + t] <= 2 + 1e-6, "dosage in [0, 2]");
                CHECK(fabs(g3[t] + g3[NT + t] + g3[2 * NT + t] - 1) < 1e-6, "genotype posteriors sum to 1");
                CHECK(fabs(g3[NT + t] + 2 * g3[2 * NT + t] - i_dos[0][s2 * NT + t]) < 1e-6, "dosage = E[genotype]");
            }
            for (int r = 0; r < NR; r++) CHECK(i_lab[0][s2 * NR + r] == 1 || i_lab[0][s2 * NR + r] == 2, "consensus labels");
        }
        CHECK(memcmp(i_dos[0], i_dos[0] + NT, sizeof(double) * NT) != 0, "samples draw from their own streams (global index)");
        st = qa_impute_samples(both, 1, &ip, 2, 100, i_read_off, i_read_ptr, i_u, i_bq, i_wif, i_dos[1], i_gp[1], i_ph[1], i_lab[1],
                               i_nd[1], NULL);
        CHECK(st == QA_OK, "qa_impute_samples (one host thread)");
        CHECK(memcmp(i_dos[0], i_dos[1], sizeof i_dos[0]) == 0 && memcmp(i_gp[0], i_gp[1], sizeof i_gp[0]) == 0 &&
              memcmp(i_ph[0], i_ph[1], sizeof i_ph[0]) == 0 && memcmp(i_lab[0], i_lab[1], sizeof i_lab[0]) == 0,
              "results do not depend on the number of host threads");
        i_read_off[2] = NR;   /* a sample without reads */
        st = qa_impute_samples(both, 1, &ip, 2, 100, i_read_off, i_read_ptr, i_u, i_bq, i_wif, i_dos[1], i_gp[1], i_ph[1], i_lab[1],
                               i_nd[1], NULL);
        CHECK(st == QA_ERR_INVALID && strstr(qa_last_error(), "no reads"), "a sample without reads is refused");
        CHECK(qa_impute_release_buffers() == QA_OK, "qa_impute_release_buffers");
        qa_panel_destroy(panel2);
        printf("IMPUTE_SAMPLES_OK\n");
    }
    qa_panel_destroy(panel);
    free(runif_reads); free(runif_shard);
    printf("HARNESS_OK\n");
    return 0;
}
