// gibbs_dev.hpp -- device-side building blocks shared by the Gibbs kernels (gibbs.hip: diploid sampler; gibbs3.hip:
// the three-label sampler of NIPT mode): launch parameters, read emissions (k_ematread), the chain's register-resident
// column helpers, DPP reductions and the lane-held scalar streams.  Everything lives in an unnamed namespace: each
// translation unit gets its own copy.
#pragma once

#include "panel.hpp"

#include <cstdint>

namespace {

struct GibbsParams {
    // panel
    const uint8_t *hm;       // [G][Kp]
    const int32_t *B;        // [G][nMaxDH]
    const int32_t *sp_off;
    const int32_t *sp_k;
    const uint32_t *sp_word;
    const double *sigma;     // [G-1]
    int Kp, G, T, nMaxDH;
    double ref_error;
    // chain batch
    int C;                   // chains
    int Ks, Ksp, NE;         // Ksp = Ks rounded up to 64, NE = Ksp / 64
    const int32_t *which;    // [C][Ks] 0-based panel haplotype of each small-panel row
    // reads (per chain: offsets into the flattened arrays)
    const int32_t *read_off; // [C+1] read index offsets
    const int32_t *read_ptr; // [sum R + C] CSR over bases, per chain block starting at read_off[c] + c
    const int32_t *base_off; // [C+1] base index offsets
    const int32_t *u;        // SNP index per base
    const int32_t *bq;       // effective signed base quality per base (0 = factor 1)
    const int32_t *wif;      // [sum R] grid of each read
    const uint8_t *grid_has_read;  // [C][G]
    const double *pR_tab, *pA_tab;  // [2][256]: by |bq|, for bq < 0 (index 0) and bq > 0 (index 1)
    int Jmax;
    double inv_maxdiff;      // 1 / maxDifferenceBetweenReads
    int rescale;
    // sampler
    int n_its, n_burn_in;
    const int32_t *block_its;  // [n_block]
    int n_block;
    int do_shard;
    int init_iteratively;
    // labels: 2 (diploid) or 3 (NIPT: maternal transmitted, maternal untransmitted, paternal transmitted)
    int nH;
    // (the label priors (0.5, (1 - ff) / 2, ff / 2) and the read-label class prototypes, gibbs-nipt.cpp:2707-2729, are
    // derived in the kernels from the chain's fetal fraction: ff / ff_chain below)
    int disable_read_category_usage;
    double class_sum_cutoff;
    const double *runif_reads; // [C][R_c * n_its] at offset read_off[c] * n_its
    const int32_t *first_read; // [C]
    const double *runif_shard; // [C][n_block][G-1]
    const uint64_t *seed_reads, *seed_shard;  // [C] or null: uniforms from the counter-based stream instead
    // state (per chain)
    // Read emissions, compact form.  A read that covers n <= kMaxPatternBits informative SNPs takes at most 2^n distinct
    // values over the haplotypes -- one per allele pattern at those SNPs: per read a 64-entry table (er_tab; entry 63 is
    // 1, the padding rows' value) and per (read, row) a pattern byte (er_idx, laid out [read][thread][PADB] for the
    // geometry of the Gibbs launch, so that a thread fetches its rows' bytes with one load).  1.5 KB per read instead
    // of the 5 KB of a dense Ks-column of doubles, re-read by every sweep.  Reads with more SNPs keep a dense column in
    // eMatRead (dense_of[r] = its row, else -1).
    double *eMatRead;        // at eread_off[c] doubles: [n_dense_c][Ksp]
    const size_t *eread_off; // [C]
    uint8_t *er_idx;         // at eridx_off[c] bytes: [R_c][er_nt][er_padb]
    const size_t *eridx_off; // [C]
    double *er_tab;          // [totR][64]
    const int32_t *dense_of; // [totR]
    int er_nt, er_padb;
    uint8_t *is_cat1;        // [sum R]
    uint8_t *er_nent;        // [sum R] table entries a compact read refers to (2^informative SNPs; 64: unknown / all)
    double *alpha, *beta, *eg;  // [C][2][G][Ksp]
    double *cvec;            // [C][3][G]
    int32_t *H;              // [sum R] labels 1-based (in/out)
    int32_t *H_class;        // [sum R]
    int32_t *status;         // [C] 0 ok, 1 underflow
    // outputs
    double *hapProbs, *genProbsM, *genProbsF;  // [C][T][3]
    // rare + common SNPs (the final all-SNP Gibbs of QUILT2, rare_common.R:109-420): rc_common == null for an ordinary
    // call.  With it, T / G / sigma / wif / u refer to ALL SNPs, the panel tables to the common ones (rc_Gc grids).
    const int32_t *rc_common;    // [T] 0-based index among the common SNPs, -1 for a rare SNP
    const int64_t *rc_rare_ptr;  // [K + 1] CSR: rare SNPs each panel haplotype carries the alt of
    const int32_t *rc_rare_snp;  // 0-based all-SNP indices, ascending within a haplotype
    const uint32_t *rc_any;      // [C][rc_words] bit t: some selected haplotype of the chain carries the alt of SNP t
    int rc_words, rc_Gc;
    // the same information per chain as (SNP, row) pairs, sorted by SNP then row and indexed by all-SNP grid: k_happrobs_rc walks a
    // grid's handful of pairs instead of searching every selected haplotype's rare list for every rare SNP somebody carries
    const int32_t *rc_pair_off;  // [C][G + 1] offsets into the chain's pairs
    const size_t *rc_pair_base;  // [C] where the chain's pairs start
    const uint16_t *rc_pairs;    // (SNP & 31) << 10 | row (row < 1 024)
    // NIPT (three labels): a call is cut into segments of sweeps [it_begin, it_end) with a block-Gibbs pass between
    // them (gibbs3.hip); the state lives in HBM across the launches.  blk_*: the pass's block table per chain.
    int it_begin, it_end;
    int rebuild;                // this segment follows a block pass: eMatGrid, forward, backward from the labels first
    int lean3;                  // the three-label sampler's 256-register build (two chains per SIMD): launches of more than 1 024 chains
    const int32_t *blk_where;   // [C][G] consider_grid_where_0_based
    const int32_t *blk_tab;     // [C][4][G] per block: grid_start, grid_end, reads_start, reads_end
    const int32_t *blk_n;       // [C] n_blocks
    double *blk_rate2;          // [C][G] rate2 of the block definition (k_block_rate3)
    int blk_pass, blk_n_pass;   // this pass / passes per call (indexes the pass's uniforms)
    int defer_resample;         // k_block3 leaves rcpp_sample_H_using_H_class to k_resample3 (the caller draws its uniforms in between)
    double ff;
    const double *ff_chain;     // [C] or null: per-chain fetal fraction overriding ff
    double *per_it;             // [C][n_its][8] or null: per sweep -sum(log c_h) and the label counts (qa_gibbs_opts_t.per_it_out)
};

// does panel haplotype `hap` carry the alt allele of rare SNP `snp` (rare_per_hap_info)
__device__ __forceinline__ bool rare_has_alt(const GibbsParams &p, int hap, int snp) {
    int64_t lo = p.rc_rare_ptr[hap], hi = p.rc_rare_ptr[hap + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int v = p.rc_rare_snp[mid];
        if (v == snp) return true;
        if (v < snp) lo = mid + 1; else hi = mid;
    }
    return false;
}

// 64-lane sum of doubles by DPP row shifts / row broadcasts (no LDS traffic), result broadcast to every
// lane through a scalar register.  Deterministic order: within 16-lane rows, then rows 0+1, 2+3, then all.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum(double v) {
    v += dpp_get<0x111, 0xf>(v);  // row_shr:1
    v += dpp_get<0x112, 0xf>(v);  // row_shr:2
    v += dpp_get<0x114, 0xf>(v);  // row_shr:4
    v += dpp_get<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row total
    v += dpp_get<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// Two 64-lane sums for the price of little more than one: v_permlane32_swap puts a's upper half beside its lower half in
// lanes 0..31 and b's halves in lanes 32..63 (one add folds both), then one 32-lane butterfly serves both values
// (22 instructions instead of 40; checked by scripts/micro/permlane_sum2.hip).  Order: (l, l + 32) pairs, then 16-lane
// rows, then the two rows of each half.
__device__ __forceinline__ void wsum2(double &a, double &b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    double v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    v += dpp_get<0x111, 0xf>(v);  // row_shr:1
    v += dpp_get<0x112, 0xf>(v);  // row_shr:2
    v += dpp_get<0x114, 0xf>(v);  // row_shr:4
    v += dpp_get<0x118, 0xf>(v);  // row_shr:8
    v += dpp_get<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3 -> lane 31 holds sum(a), lane 63 sum(b)
    a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 31), __builtin_amdgcn_readlane(__double2loint(v), 31));
    b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// Three or four 64-lane sums in one 16-lane butterfly: v_permlane32_swap folds each value's halves (two values per register, as
// in wsum2), v_permlane16_swap then puts one value's two 16-lane rows side by side with another's (row 0: a, row 1: c, row 2: b,
// row 3: d after the add), and four row shifts finish all of them at once -- 27 / 29 instructions with the v_readlanes instead
// of 42 / 44 (scripts/micro/permlane_sum4.hip checks the lane map on the device).  Order: (l, l + 32) pairs, then (l, l + 16)
// pairs, then within 16-lane rows.
__device__ __forceinline__ double fold32(double a, double b) {   // lanes 0..31: a(l) + a(l + 32); lanes 32..63: b(l - 32) + b(l)
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold16_rows(double ab, double cd) {   // rows of 16 lanes hold the partial sums of a, c, b, d
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(ab), __double2loint(cd), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(ab), __double2hiint(cd), false, false);
    double v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    v += dpp_get<0x111, 0xf>(v);  // row_shr:1
    v += dpp_get<0x112, 0xf>(v);  // row_shr:2
    v += dpp_get<0x114, 0xf>(v);  // row_shr:4
    v += dpp_get<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row's total
    return v;
}
__device__ __forceinline__ double lane_of(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void wsum3(double &a, double &b, double &c) {
    const double v = fold16_rows(fold32(a, b), fold32(c, c));
    a = lane_of(v, 15); b = lane_of(v, 47); c = lane_of(v, 31);
}
__device__ __forceinline__ void wsum4(double &a, double &b, double &c, double &d) {
    const double v = fold16_rows(fold32(a, b), fold32(c, d));
    a = lane_of(v, 15); b = lane_of(v, 47); c = lane_of(v, 31); d = lane_of(v, 63);
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based uniform stream (splitmix64 finaliser of seed + (i + 1) * golden ratio), 53-bit mantissa in
// [0, 1): element i of stream `seed`.  The host restates it exactly (quilt_amd/rng.py).
__device__ __forceinline__ double stream_uniform(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// word of haplotype k (panel index) at grid g with code `code`
__device__ __forceinline__ uint32_t panel_word(const GibbsParams &p, int g, int k, int code) {
    if (code > 0) return (uint32_t)p.B[(size_t)g * p.nMaxDH + (code - 1)];
    int lo = p.sp_off[g], hi = p.sp_off[g + 1] - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (p.sp_k[mid] < k) lo = mid + 1; else hi = mid;
    }
    return p.sp_word[lo];
}

__device__ __forceinline__ double rl_f64(double v, int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rl_i32(int v, int j) { return __builtin_amdgcn_readlane(v, j); }

// ---------------------------------------------------------------------------------------------
// k_ematread: P(read r | small-panel haplotype k) (gibbs-small.cpp:148-263).  One wave per
// (block of kReadsPerWave consecutive reads, chain); lane l owns rows l, l+64, ...  The chain's haplotype
// list and the panel words of the current grid stay in registers across the block's reads (reads are sorted
// by grid).  Products run over the read's bases in order, so each entry is bit-identical to the
// reference's; then divide by the column max and floor (:235-262).
// The walk over a read's bases is serial (a product in the reference's order) and every operand of a step used to be a load
// that depended on the one before it -- offset, then position and quality, then the two table values of the quality: four
// round trips per base with nothing else in flight.  Now the block's read offsets sit in lanes (one load), and the bases come
// in chunks of 64 -- position, quality and the quality's two table values, one base per lane, gathered side by side -- so a
// step takes its operands from registers (v_readlane with a wave-uniform lane).  Same values, same order of multiplications.
// ---------------------------------------------------------------------------------------------
constexpr int kReadsPerWave = 32;
constexpr int kMaxPatternBits = 5;
constexpr int padb_of(int ne) { return ne <= 4 ? 4 : ne <= 8 ? 8 : ne <= 12 ? 12 : 16; }

template <int NEALL, int NW>
__global__ __launch_bounds__(64) void k_ematread(GibbsParams p) {
    constexpr int NE = NEALL / NW, PADB = padb_of(NE), NT = 64 * NW;
    const int c = blockIdx.y, lane = threadIdx.x;
    const int R = p.read_off[c + 1] - p.read_off[c];
    const int r0 = blockIdx.x * kReadsPerWave;
    if (r0 >= R) return;
    const int32_t *rp = p.read_ptr + p.read_off[c] + c;
    const int32_t *u = p.u + p.base_off[c], *bq = p.bq + p.base_off[c];
    const int32_t *which = p.which + (size_t)c * p.Ks;
    int kk[NEALL];
#pragma unroll
    for (int i = 0; i < NEALL; i++) {
        const int k = lane + 64 * i;
        kk[i] = (k < p.Ks) ? which[k] : -1;
    }
    int g_prev = -1;
    uint32_t w[NEALL];
    const double e1 = 1 - p.ref_error, e0 = p.ref_error;
    const int nblk = min(kReadsPerWave, R - r0);
    static_assert(kReadsPerWave < 64, "lane l holds the offset of read r0 + l, lane nblk the block's end");
    const int my_s = (lane <= nblk) ? rp[r0 + lane] : 0;
    const int my_dense = (lane < nblk) ? p.dense_of[p.read_off[c] + r0 + lane] : -1;
    const int b_end = rl_i32(my_s, nblk);   // one past the block's last base
    int chunk0 = rl_i32(my_s, 0), c_u = 0, c_b = 0;
    double c_R = 0, c_A = 0;
    auto load_chunk = [&](int base) {
        chunk0 = base;
        const int idx = base + lane;
        const bool ok = idx < b_end;
        c_u = ok ? u[idx] : 0;
        c_b = ok ? bq[idx] : 0;
        const int ab = c_b < 0 ? -c_b : c_b;
        c_R = p.pR_tab[(c_b > 0 ? 256 : 0) + ab];
        c_A = p.pA_tab[(c_b > 0 ? 256 : 0) + ab];
    };
    load_chunk(chunk0);
    for (int r = r0; r < r0 + nblk; r++) {
        double v[NEALL];
        uint32_t pat[NEALL];
#pragma unroll
        for (int i = 0; i < NEALL; i++) { v[i] = 1.0; pat[i] = 0; }
        const int s = rl_i32(my_s, r - r0);
        int J = rl_i32(my_s, r - r0 + 1) - s - 1;
        if (J >= p.Jmax) J = p.Jmax;
        const int dense = rl_i32(my_dense, r - r0);
        int n_inf = 0;
        double tv = 1.0;   // lane l: the product for allele pattern l (bit j = allele at the read's j-th informative SNP)
        for (int j = 0; j <= J; j++) {
            if (s + j >= chunk0 + 64) load_chunk(s + j);
            const int li = s + j - chunk0;
            const int b = rl_i32(c_b, li);
            int snp = rl_i32(c_u, li);
            if (p.rc_common) {   // all-SNP read: a common SNP maps to its panel column, a rare one has no column
                const int all_snp = snp;
                snp = p.rc_common[all_snp];
                if (snp < 0) {
                    // rare SNP (gibbs-small.cpp:382-401): everyone as ref, then the carriers re-done -- in that order
                    if (b == 0) continue;
                    const double pR = rl_f64(c_R, li), pA = rl_f64(c_A, li);
                    const double xe1 = e0 * pA + e1 * pR;
                    const bool any = (p.rc_any[(size_t)c * p.rc_words + (all_snp >> 5)] >> (all_snp & 31)) & 1u;
                    if (!any) {   // no selected haplotype carries it: a common factor, dropped under rescaling
                        if (!p.rescale) {
#pragma unroll
                            for (int i = 0; i < NEALL; i++) v[i] *= xe1;
                            tv *= xe1;
                        }
                        continue;
                    }
                    const double ratio = (e1 * pA + e0 * pR) / xe1;
#pragma unroll
                    for (int i = 0; i < NEALL; i++) {
                        const uint32_t bit = (kk[i] >= 0 && rare_has_alt(p, kk[i], all_snp)) ? 1u : 0u;
                        v[i] *= xe1;
                        if (bit) v[i] *= ratio;
                        pat[i] |= bit << min(n_inf, 7);
                    }
                    tv *= xe1;
                    if ((lane >> min(n_inf, 31)) & 1) tv *= ratio;
                    n_inf++;
                    continue;
                }
            }
            const int g = snp >> 5;
            if (g != g_prev) {
#pragma unroll
                for (int i = 0; i < NEALL; i++) {
                    w[i] = 0;
                    if (kk[i] >= 0) w[i] = panel_word(p, g, kk[i], p.hm[(size_t)g * p.Kp + kk[i]]);
                }
                g_prev = g;
            }
            if (b == 0) continue;  // no base quality seen yet: factor 1 (host folded the carry-over rule)
            const double pR = rl_f64(c_R, li), pA = rl_f64(c_A, li);
            // the factor of a row is one of two values (its allele at the SNP): formed once, the expression the reference
            // evaluates per row (gibbs-small.cpp:219-226) with e = 1 - ref_error resp. ref_error
            const double f1 = (e1 * pA + (1 - e1) * pR), f0 = (e0 * pA + (1 - e0) * pR);
#pragma unroll
            for (int i = 0; i < NEALL; i++) {
                const uint32_t bit = (w[i] >> (snp & 31)) & 1u;
                v[i] *= bit ? f1 : f0;
                pat[i] |= bit << min(n_inf, 7);
            }
            // the same factor, in the same order, for this lane's pattern
            tv *= ((lane >> min(n_inf, 31)) & 1) ? f1 : f0;
            n_inf++;
        }
        bool degenerate = false;
        double d1 = 1.0;
        if (p.rescale) {
            double x = 0;
#pragma unroll
            for (int i = 0; i < NEALL; i++) if (kk[i] >= 0 && v[i] > x) x = v[i];
            x = wmax(x);
            d1 = 1 / x;
            degenerate = isinf(x) || x == 0 || isinf(d1);
            if (degenerate) {
#pragma unroll
                for (int i = 0; i < NEALL; i++) v[i] = 1;
                tv = 1;
            } else {
#pragma unroll
                for (int i = 0; i < NEALL; i++) {
                    v[i] *= d1;
                    if (v[i] < p.inv_maxdiff) v[i] = p.inv_maxdiff;
                }
                tv *= d1;
                if (tv < p.inv_maxdiff) tv = p.inv_maxdiff;
            }
        }
        // category 1 (gibbs-nipt.cpp:350-372): no entry below 1 - 1e-12
        const double thresh = 1 - 1e-12;
        bool below = false;
#pragma unroll
        for (int i = 0; i < NEALL; i++) if (kk[i] >= 0 && v[i] < thresh) below = true;
        const bool any_below = __any(below);
        if (lane == 0) {
            p.is_cat1[p.read_off[c] + r] = (any_below || p.disable_read_category_usage) ? 0 : 1;
            p.er_nent[p.read_off[c] + r] = (uint8_t)(n_inf <= kMaxPatternBits ? (1 << n_inf) : 64);
        }
        if (dense >= 0) {
            double *out = p.eMatRead + p.eread_off[c] + (size_t)dense * p.Ksp;
#pragma unroll
            for (int i = 0; i < NEALL; i++) out[lane + 64 * i] = (kk[i] >= 0) ? v[i] : 1.0;
        } else {
            // (a degenerate read has every entry 1: pattern 63 for every row)
            p.er_tab[(size_t)(p.read_off[c] + r) * 64 + lane] = (lane < (1 << n_inf) && lane != 63) ? tv : 1.0;
            uint8_t *ix = p.er_idx + p.eridx_off[c] + (size_t)r * NT * PADB;
#pragma unroll
            for (int wv = 0; wv < NW; wv++) {
                uint32_t pk[PADB / 4];
#pragma unroll
                for (int q = 0; q < PADB / 4; q++) pk[q] = 0;
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    const int j = wv + NW * i;   // thread lane + 64 wv of the chain owns rows lane + 64 (wv + NW i)
                    const uint32_t code = (kk[j] >= 0 && !degenerate) ? pat[j] : 63u;
                    pk[i >> 2] |= code << ((i & 3) * 8);
                }
                uint32_t *dst = reinterpret_cast<uint32_t *>(ix + (size_t)(lane + 64 * wv) * PADB);
#pragma unroll
                for (int q = 0; q < PADB / 4; q++) dst[q] = pk[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_gibbs: NW wavefronts (one workgroup) per chain; initialisation, all sweeps, shard passes.
// Thread t of the NT = 64 * NW threads owns small-panel rows t, t + NT, ... (NE of them).
// ---------------------------------------------------------------------------------------------
template <int NE>
struct Col {
    double v[NE];
};

// Raw buffer access (a wave-uniform base in scalar registers + a byte count): lanes whose offset lies beyond the count read 0
// and write nothing WITHOUT a memory request -- bounds that cost no branch and no traffic.  Used for the tail of the last
// small-panel row (Ks = 600: lanes 24..63 of row 9, 320 of a column's 5 120 bytes) and for the unused part of a compact
// read's 64-entry table.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double buf_ld_f64(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void buf_st_f64(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, v), r,
                                          (int)byte_off, 0, 0);
}

template <int NE>
__device__ __forceinline__ void load_col(Col<NE> &c, const double *src, int t, int NT) {
#pragma unroll
    for (int i = 0; i < NE; i++) c.v[i] = src[t + NT * i];
}
template <int NE>
__device__ __forceinline__ void store_col(const Col<NE> &c, double *dst, int t, int NT) {
#pragma unroll
    for (int i = 0; i < NE; i++) dst[t + NT * i] = c.v[i];
}

// 1 / e to within 1 ulp: v_rcp_f64 and two Newton steps (5 instructions; an IEEE division expands to ~15).  The
// sampler takes a read's emission out of a label with it -- x * (1 / e) where the reference writes x / e: the two can
// differ in the last bit, 1e-16 relative, against sampling thresholds compared with a 53-bit uniform.
__device__ __forceinline__ double fast_rcp(double e) {
    double r = __builtin_amdgcn_rcp(e);
    r = __builtin_fma(__builtin_fma(-e, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-e, r, 1.0), r, r);
    return r;
}

template <int NE, int NW>
struct Chain {
    static constexpr int NT = 64 * NW;
    const GibbsParams &p;
    int c, t, lane, wave, R, G, Ks, Ksp;
    double *alpha[3], *beta[3], *eg[3], *cv[3];   // state matrices of the chain's nH labels
    const double *eMatRead;   // dense columns of the reads that have one
    const uint8_t *eridx;     // compact read emissions (GibbsParams)
    const double *ertab;
    const int32_t *dense_of;
    const int32_t *wif;
    const uint8_t *ghr, *cat1, *nent;
    int32_t *H, *Hc;
    double prior;  // 1 / Ks
    bool valid[NE];
    double *red;   // LDS [2][NW][4]
    int par;

    __device__ Chain(const GibbsParams &p_, int c_, int t_, double *red_)
        : p(p_), c(c_), t(t_), lane(t_ & 63), wave(t_ >> 6), red(red_), par(0) {
        G = p.G; Ks = p.Ks; Ksp = p.Ksp;
        R = p.read_off[c + 1] - p.read_off[c];
        const size_t mat = (size_t)G * Ksp;
        for (int h = 0; h < 3; h++) {
            const size_t at = ((size_t)c * p.nH + min(h, p.nH - 1)) * mat;
            alpha[h] = p.alpha + at;
            beta[h] = p.beta + at;
            eg[h] = p.eg + at;
        }
        for (int h = 0; h < 3; h++) cv[h] = p.cvec + ((size_t)c * 3 + h) * G;
        eMatRead = p.eMatRead + p.eread_off[c];
        eridx = p.er_idx + p.eridx_off[c];
        ertab = p.er_tab + (size_t)p.read_off[c] * 64;
        dense_of = p.dense_of + p.read_off[c];
        wif = p.wif + p.read_off[c];
        cat1 = p.is_cat1 + p.read_off[c];
        nent = p.er_nent + p.read_off[c];
        ghr = p.grid_has_read + (size_t)c * G;
        H = p.H + p.read_off[c];
        Hc = p.H_class + p.read_off[c];
        prior = 1.0 / Ks;
        // Ksp is Ks rounded up to 64 rows (gibbs.hip), so only a thread's LAST row can lie beyond Ks: rows t + NT i with
        // i < NE - 1 end at Ksp - NT - 1 < Ksp - 64 < Ks.  Written so that the compiler sees it: the per-element selects
        // on valid[i] of the grid steps (2 v_cndmask per fp64 element) fold away for all rows but the last.
#pragma unroll
        for (int i = 0; i < NE; i++) valid[i] = (i < NE - 1) ? true : (t + NT * i) < Ks;
    }
    // A read's emission column, compact form: this thread's pattern bytes and the lane's table entry (one load each,
    // issued a read ahead), expanded by a cross-lane gather (ds_bpermute: no memory traffic).
    static constexpr int PADB = padb_of(NE);
    struct ErPre {
        uint32_t w[PADB / 4];
        double tv;
    };
    // `nent`: the table entries the read refers to (wave-uniform; 64 = all).  Entry 63 -- the value of padding rows and of
    // degenerate reads -- is 1 by construction and is not fetched.
    __device__ __forceinline__ void ld_pre(ErPre &x, int r, int nent = 64) const {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(eridx + ((size_t)r * NT + t) * PADB);
        if constexpr (PADB == 16) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src);
            x.w[0] = q.x; x.w[1] = q.y; x.w[2] = q.z; x.w[3] = q.w;
        } else if constexpr (PADB == 12) {
            struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };
            const U3 q = *reinterpret_cast<const U3 *>(src);
            x.w[0] = q.a; x.w[1] = q.b; x.w[2] = q.c;
        } else if constexpr (PADB == 8) {
            const uint2 q = *reinterpret_cast<const uint2 *>(src);
            x.w[0] = q.x; x.w[1] = q.y;
        } else {
            x.w[0] = *src;
        }
        const double tv = buf_ld_f64(buf_rsrc(ertab + (size_t)r * 64, (uint32_t)nent * 8), (uint32_t)lane * 8);
        x.tv = lane == 63 ? 1.0 : tv;
    }
    __device__ __forceinline__ void expand(Col<NE> &er, const ErPre &x) const {
        const int lo = __double2loint(x.tv), hi = __double2hiint(x.tv);
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const int src = (int)((x.w[i >> 2] >> ((i & 3) * 8)) & 0xffu) << 2;
            er.v[i] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, hi), __builtin_amdgcn_ds_bpermute(src, lo));
        }
    }
    // (Measured alternatives, both dropped -- r02, 512 chains x 20 000 reads, us per read visit: this form 0.93; table entry
    // and reciprocal both precomputed by k_ematread and gathered with the same 4 ds_bpermute per row 0.95 (the v_rcp chain is
    // not on the critical path, the 8 extra bytes per lane are); table in LDS, one ds_write_b128 per lane and one
    // ds_read_b128 per row instead of the bpermutes 1.29.)
    // compact form, emission and its reciprocal together: the reciprocal is taken once per table entry (this lane's) and
    // gathered like the emission itself -- 1 v_rcp + 2 Newton steps per read instead of per row, for 2 more ds_bpermute per
    // row; the values are those of fast_rcp applied row by row
    __device__ __forceinline__ void expand_with_rcp(Col<NE> &er, Col<NE> &ri, const ErPre &x) const {
        const double tvi = fast_rcp(x.tv);
        const int lo = __double2loint(x.tv), hi = __double2hiint(x.tv);
        const int ilo = __double2loint(tvi), ihi = __double2hiint(tvi);
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const int src = (int)((x.w[i >> 2] >> ((i & 3) * 8)) & 0xffu) << 2;
            er.v[i] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, hi), __builtin_amdgcn_ds_bpermute(src, lo));
            ri.v[i] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, ihi), __builtin_amdgcn_ds_bpermute(src, ilo));
        }
    }
    // emission column of read r with its (wave-uniform) dense row dn
    __device__ __forceinline__ void read_emission(Col<NE> &er, const ErPre &x, int dn) const {
        if (dn >= 0) ld(er, eMatRead + (size_t)dn * Ksp);
        else expand(er, x);
    }

    __device__ __forceinline__ double tm0(int g) const { return p.sigma[g]; }
    // transMatRate_t row 1 as the caller passed it (the reference never recomputes 1 - sigma)
    __device__ __forceinline__ double tm1(int g) const { return p.sigma[p.G - 1 + g]; }

    // workgroup-wide sums of N values per thread; every thread gets the totals.  Waves reduce by DPP, then
    // exchange through a parity-alternating LDS buffer with a bare s_barrier (lgkmcnt only: the column
    // prefetches in flight on vmcnt must not be drained here, which __syncthreads() would do).
    template <int N>
    __device__ __forceinline__ void bsum(double (&x)[N]) {
        if constexpr (N == 1) x[0] = wsum(x[0]);
        if constexpr (N == 2) wsum2(x[0], x[1]);
        if constexpr (N == 3) wsum3(x[0], x[1], x[2]);
        if constexpr (N == 4) wsum4(x[0], x[1], x[2], x[3]);
        static_assert(N >= 1 && N <= 4, "bsum handles 1..4 values");
        if (NW == 1) return;
        double *buf = red + (size_t)par * NW * 4;
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < N; q++) buf[wave * 4 + q] = x[q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < N; q++) {
            double s = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) s += buf[w * 4 + q];
            x[q] = s;
        }
        par ^= 1;
    }
    __device__ __forceinline__ double bsum1(double v) {
        double x[1] = {v};
        bsum<1>(x);
        return x[0];
    }
    __device__ __forceinline__ double sum_col(const Col<NE> &c) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < NE; i++) s += c.v[i];
        return bsum1(s);
    }
    __device__ __forceinline__ void sum_col2(const Col<NE> &c0, const Col<NE> &c1, double &s0, double &s1) {
        double x[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < NE; i++) { x[0] += c0.v[i]; x[1] += c1.v[i]; }
        bsum<2>(x);
        s0 = x[0]; s1 = x[1];
    }
    __device__ __forceinline__ void ld(Col<NE> &c, const double *src) const { load_col(c, src, t, NT); }
    __device__ __forceinline__ void st(const Col<NE> &c, double *dst) const { store_col(c, dst, t, NT); }
    // the same for STATE columns (alpha / beta / eMatGrid) in the sweeps: only the Ks real rows move; the padding rows of a
    // column keep what the initialisation stored (0 for alpha and beta, 1 for eMatGrid) and read as 0 here -- every use of
    // a padding row is either masked by `valid` or a product with an alpha / beta that is 0
    __device__ __forceinline__ void ldm(Col<NE> &c, const double *src) const {
        const __amdgpu_buffer_rsrc_t r = buf_rsrc(src, (uint32_t)Ks * 8);
#pragma unroll
        for (int i = 0; i < NE; i++) c.v[i] = buf_ld_f64(r, (uint32_t)(t + NT * i) * 8);
    }
    __device__ __forceinline__ void stm(const Col<NE> &c, double *dst) const {
        const __amdgpu_buffer_rsrc_t r = buf_rsrc(dst, (uint32_t)Ks * 8);
#pragma unroll
        for (int i = 0; i < NE; i++) buf_st_f64(r, (uint32_t)(t + NT * i) * 8, c.v[i]);
    }
};

// ---- lane-held scalar streams --------------------------------------------------------------
// The per-grid / per-read scalars of a chain (c, sigma, grid_has_read; wif, category, label) are uniform,
// and a uniform value loaded through the vector memory path has to be waited for before it can steer
// control flow -- which would expose one memory round trip per grid and per read and also drain the
// column prefetches (in-order vmcnt).  Instead lane j of every wave holds element base + j of each
// stream: ONE vector load per 64 elements, then v_readlane (no memory) per element, and one store per 64
// elements (by wave 0) for the streams the sampler updates.
template <class CH>
struct GridStreams {   // lane j <-> grid base + j
    double t0, t1;     // transition INTO the grid (forward) or OUT of it (backward)
    double c0, c1;
    int has;           // grid_has_read of the grid (forward) or of grid + 1 (backward)
    int base;
    __device__ void load_fwd(const CH &ch, int b) {
        base = b;
        const int g = b + ch.lane;
        const bool ok = g < ch.G;
        t0 = (ok && g > 0) ? ch.tm0(g - 1) : 1.0;
        t1 = (ok && g > 0) ? ch.tm1(g - 1) : 0.0;
        c0 = ok ? ch.cv[0][g] : 1.0;
        c1 = ok ? ch.cv[1][g] : 1.0;
        has = ok ? ch.ghr[g] : 0;
    }
    __device__ void load_bwd(const CH &ch, int b) {
        base = b;
        const int g = b + ch.lane;
        const bool ok = g < ch.G - 1;
        t0 = ok ? ch.tm0(g) : 1.0;
        t1 = ok ? ch.tm1(g) : 0.0;
        c0 = (g < ch.G) ? ch.cv[0][g] : 1.0;
        c1 = (g < ch.G) ? ch.cv[1][g] : 1.0;
        has = ok ? ch.ghr[g + 1] : 0;
    }
    __device__ void store_c(const CH &ch) const {
        const int g = base + ch.lane;
        if (ch.wave == 0 && g < ch.G) { ch.cv[0][g] = c0; ch.cv[1][g] = c1; }
    }
    __device__ __forceinline__ void set_c(int lane, int j, double a, double b) {
        if (lane == j) { c0 = a; c1 = b; }
    }
};

// the same for the three-label (NIPT) kernels
template <class CH>
struct GridStreams3 {
    double t0, t1, c0, c1, c2;
    int has, base;
    __device__ void load_fwd(const CH &ch, int b) {
        base = b;
        const int g = b + ch.lane;
        const bool ok = g < ch.G;
        t0 = (ok && g > 0) ? ch.tm0(g - 1) : 1.0;
        t1 = (ok && g > 0) ? ch.tm1(g - 1) : 0.0;
        c0 = ok ? ch.cv[0][g] : 1.0;
        c1 = ok ? ch.cv[1][g] : 1.0;
        c2 = ok ? ch.cv[2][g] : 1.0;
        has = ok ? ch.ghr[g] : 0;
    }
    __device__ void load_bwd(const CH &ch, int b) {
        base = b;
        const int g = b + ch.lane;
        const bool ok = g < ch.G - 1;
        t0 = ok ? ch.tm0(g) : 1.0;
        t1 = ok ? ch.tm1(g) : 0.0;
        c0 = (g < ch.G) ? ch.cv[0][g] : 1.0;
        c1 = (g < ch.G) ? ch.cv[1][g] : 1.0;
        c2 = (g < ch.G) ? ch.cv[2][g] : 1.0;
        has = ok ? ch.ghr[g + 1] : 0;
    }
    __device__ void store_c(const CH &ch) const {
        const int g = base + ch.lane;
        if (ch.wave == 0 && g < ch.G) { ch.cv[0][g] = c0; ch.cv[1][g] = c1; ch.cv[2][g] = c2; }
    }
    __device__ __forceinline__ void set_c(int lane, int j, double a, double b, double c) {
        if (lane == j) { c0 = a; c1 = b; c2 = c; }
    }
    __device__ __forceinline__ double c_of(int h, int j) const { return rl_f64(h == 0 ? c0 : (h == 1 ? c1 : c2), j); }
};

template <class CH>
struct ReadStreams {   // lane j <-> read base + j
    int wif, cat1, H, Hc, base, dn, nent;
    double u;
    __device__ void load(const CH &ch, int b, const double *runif, int it) {
        base = b;
        const int r = b + ch.lane;
        const bool ok = r < ch.R;
        wif = ok ? ch.wif[r] : -1;
        nent = ok ? ch.nent[r] : 64;
        dn = ok ? ch.dense_of[r] : -1;
        cat1 = ok ? ch.cat1[r] : 1;
        H = ok ? ch.H[r] : 1;
        Hc = ok ? ch.Hc[r] : 0;
        u = (ok && runif) ? runif[(size_t)ch.R * it + r] : 0.0;
    }
    __device__ void store(const CH &ch) const {
        const int r = base + ch.lane;
        if (ch.wave == 0 && r < ch.R) { ch.H[r] = H; ch.Hc[r] = Hc; }
    }
};

// all waves of the chain must see the global-memory stores of the other waves (columns are private to their
// owning thread, but cv / H / Hc written by wave 0 are re-read by every wave at the next block refill)
template <int NW>
__device__ __forceinline__ void chain_sync() {
    if (NW > 1) __syncthreads();
}


}  // namespace
