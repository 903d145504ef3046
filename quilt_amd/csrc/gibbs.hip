// gibbs.hip -- small-panel Gibbs read-label sampler on gfx950 (MI355X).
//
// What it computes: `rcpp_forwardBackwardGibbsNIPT` (QUILT/src/gibbs-nipt.cpp:2395-3307) for the
// production argument values (diploid, ff = 0): packed-panel read emissions
// (gibbs-small.cpp:116-265), read categories (gibbs-nipt.cpp:338-382), initialisation
// (:1629-1750), the read-by-read Gibbs sweeps (:1756-1956, :733-1295), the shard resampler
// (gibbs-nipt-block.cpp:1975-2355) and hapProbs / genProbs from the packed panel
// (gibbs-small.cpp:472-635).
//
// How (MI355X-first):
//   * a sweep is a chain of ~G + R dependent Ks-wide steps (every accepted move changes what the next
//     read sees), so the unit of parallelism is the chain: ONE workgroup of NW wavefronts (1 when the
//     device is full: one chain per SIMD, 1024 per device; 2 when it is not) owns one (sample, Gibbs
//     chain) and keeps the current grid's alpha / beta / eMatGrid columns of both labels in registers
//     (Ks = 600 -> 10 fp64 per lane per column at NW = 1).  All Ks-wide sums are DPP butterflies, plus one
//     LDS exchange with a bare s_barrier when NW > 1.
//   * columns live in HBM as [grid][Ks padded to 64] fp64, lane-strided, so every column access is
//     coalesced; read emissions are kept in a compact form (per-read table + pattern byte per row, expanded
//     with ds_bpermute; see GibbsParams) and fetched one read ahead.
//   * fp64 throughout and no FMA contraction (-ffp-contract=off).  Deviations from the reference's
//     per-element arithmetic, each <= 1 ulp: the order of the Ks-wide sums; x * (1 / e) with a refined
//     v_rcp_f64 where the reference divides by a read's emission (and by P + Q).  alpha * beta of a grid is held
//     and updated by the moves as the reference's ab_m is (round 4; it used to be re-formed per read).  Sampling thresholds are compared with 53-bit uniforms, so the sampled labels equal the
//     CPU path's under the same uniforms in every test.
//   * the uniforms the reference draws from R's RNG are inputs.
//
// Block Gibbs: for diploid samples `Rcpp_block_gibbs_resampler` (gibbs-nipt-block.cpp:1636-1967) is the
// identity: c3 is all zero (gibbs-nipt.cpp:2678), so logC_after(2) is -inf and then NaN (:1819-1821,
// :1896-1898), every choice_log_probs entry is NaN (:661-675), ir_chosen stays 0 (:741-752), the
// "No change warranted" branch is taken (:830) and the final backward (:1947-1954) reproduces the beta
// the sweep already holds.  The shard resampler is the active step and is implemented here.
// NIPT (ff > 0): the three-label sampler and its block Gibbs are in gibbs3.hip / gibbs_blocks.hpp; this file drives
// them (segments of sweeps with a block pass between them, gibbs_chunk) and holds the panel-facing kernels both modes
// share, including the rare + common forms (k_ematread's rare branch in gibbs_dev.hpp, k_happrobs_rc).
#include "panel.hpp"

#include <chrono>
#include <functional>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>

#include "gibbs_dev.hpp"
#include "gibbs_blocks.hpp"

namespace qa {
int gibbs3_waves(int Ksp, int C, int share);
void launch_gibbs3(const void *gibbs_params, hipStream_t st);
void launch_block_rate3(const void *gibbs_params, hipStream_t st);
void launch_block3(const void *gibbs_params, hipStream_t st);
void launch_resample3(const void *gibbs_params, hipStream_t st);
}

namespace {

// Rcpp_run_forward_haploid (copied-from-stitch.cpp:340-387) for both labels, prior = alphaMat = 1/Ks
template <int NE, int NW>
__device__ void forward_full_both(Chain<NE, NW> &ch) {
    const int G = ch.G, Ksp = ch.Ksp;
    Col<NE> a[2], e[2];
    GridStreams<Chain<NE, NW>> gs;
    for (int g = 0; g < G; g++) {
        if ((g & 63) == 0) {
            if (g) gs.store_c(ch);
            gs.load_fwd(ch, g);
        }
        const int j = g & 63;
        ch.ld(e[0], ch.eg[0] + (size_t)g * Ksp);
        ch.ld(e[1], ch.eg[1] + (size_t)g * Ksp);
        const double s0 = rl_f64(gs.t0, j), s1 = rl_f64(gs.t1, j);
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < NE; i++) {
                if (g == 0) a[h].v[i] = ch.valid[i] ? ch.prior * e[h].v[i] : 0.0;
                else a[h].v[i] = ch.valid[i] ? e[h].v[i] * (s0 * a[h].v[i] + s1 * ch.prior) : 0.0;
            }
        }
        double sm[2];
        ch.sum_col2(a[0], a[1], sm[0], sm[1]);
        const double cc[2] = {1 / sm[0], 1 / sm[1]};
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < NE; i++) a[h].v[i] = a[h].v[i] * cc[h];
            ch.st(a[h], ch.alpha[h] + (size_t)g * Ksp);
        }
        gs.set_c(ch.lane, j, cc[0], cc[1]);
    }
    gs.store_c(ch);
    chain_sync<NW>();
}

// Rcpp_run_backward_haploid (copied-from-stitch.cpp:392-409; FASTER = false) or
// Rcpp_run_backward_haploid_QUILT_faster (:417-440; FASTER = true), both labels; beta(G-1) = c(G-1)
template <int NE, int NW, bool FASTER>
__device__ void backward_both(Chain<NE, NW> &ch) {
    const int G = ch.G, Ksp = ch.Ksp;
    const double one_over_K = 1 / (double)ch.Ks;
    Col<NE> b[2], e[2];
    GridStreams<Chain<NE, NW>> gs;
    gs.load_bwd(ch, (G - 1) & ~63);
    {
        const int j = (G - 1) & 63;
        const double cl[2] = {rl_f64(gs.c0, j), rl_f64(gs.c1, j)};
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < NE; i++) b[h].v[i] = ch.valid[i] ? cl[h] : 0.0;
            ch.stm(b[h], ch.beta[h] + (size_t)(G - 1) * Ksp);
        }
    }
    if (G >= 2) {
        ch.ldm(e[0], ch.eg[0] + (size_t)(G - 1) * Ksp);
        ch.ldm(e[1], ch.eg[1] + (size_t)(G - 1) * Ksp);
    }
    for (int g = G - 2; g >= 0; --g) {
        if ((g & 63) == 63) gs.load_bwd(ch, g & ~63);
        const int j = g & 63;
        Col<NE> en[2];   // next iteration's emission columns (grid g) while this one computes
        ch.ldm(en[0], ch.eg[0] + (size_t)g * Ksp);
        ch.ldm(en[1], ch.eg[1] + (size_t)g * Ksp);
        const double s0 = rl_f64(gs.t0, j), s1 = rl_f64(gs.t1, j);
        const double cg[2] = {rl_f64(gs.c0, j), rl_f64(gs.c1, j)};
        const bool has = !FASTER || rl_i32(gs.has, j) != 0;
        double x[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < NE; i++) {
                if (has) b[h].v[i] = e[h].v[i] * b[h].v[i];
                if (FASTER) x[h] += b[h].v[i];
                else x[h] += ch.valid[i] ? ch.prior * b[h].v[i] : 0.0;
            }
        }
        ch.template bsum<2>(x);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double xx = FASTER ? s1 * x[h] * one_over_K : s1 * x[h];
#pragma unroll
            for (int i = 0; i < NE; i++) b[h].v[i] = ch.valid[i] ? cg[h] * (xx + s0 * b[h].v[i]) : 0.0;
            ch.stm(b[h], ch.beta[h] + (size_t)g * Ksp);
        }
        e[0] = en[0];
        e[1] = en[1];
    }
}

// LEAN: the build for TWO chains per SIMD (256 registers per wave).  A chain is a serial string of dependent instructions --
// one wave issues an instruction every ~11 cycles -- and a second wave on the SIMD fills most of the gaps (measured at
// Ks = 256, whose columns fit 256 registers anyway: 2 048 chains take 1.13 x the time of 1 024).  At 10 rows per lane the
// sweep therefore gives up what it held for its own latency hiding: the next grid's columns are not fetched a grid ahead
// (the other wave runs while they arrive).
#ifndef QA_LEAN_WAVES
#define QA_LEAN_WAVES 2
#endif
template <int NE, int NW, bool LEAN>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(LEAN ? QA_LEAN_WAVES : 1))) void k_gibbs(GibbsParams p) {
    __shared__ double s_red[2 * NW * 4];
    // LEAN: the current grid's eMatGrid columns wait in LDS while the grid's reads are sampled (10 KB per chain, 80 KB per
    // compute unit at two chains per SIMD): a move updates them there, and their 4 * NE registers are free in the read loop
    __shared__ double s_e[LEAN ? 2 * NE * 64 * NW : 1];
    const int c = blockIdx.x, t = threadIdx.x;
    using CH = Chain<NE, NW>;
    CH ch(p, c, t, s_red);
    constexpr int NT = CH::NT;
    const int lane = ch.lane;
    const int G = ch.G, Ksp = ch.Ksp, R = ch.R, Ks = ch.Ks;
    const double prior = ch.prior;
    bool (&valid)[NE] = ch.valid;
    const double *runif = p.seed_reads ? nullptr : p.runif_reads + (size_t)p.read_off[c] * p.n_its;
    const uint64_t seed_reads = p.seed_reads ? p.seed_reads[c] : 0, seed_shard = p.seed_shard ? p.seed_shard[c] : 0;
    const int first_read = p.first_read[c];
    // first_read < 0: this chain starts from its starting labels even in a launch that initialises iteratively
    const bool init_iteratively = p.init_iteratively && first_read >= 0;

    // ---- c = 0 (arma::zeros, gibbs-nipt.cpp:2676-2678), H_class = 0
    for (int h = 0; h < 3; h++)
        for (int g = t; g < G; g += NT) ch.cv[h][g] = 0.0;
    for (int r = t; r < R; r += NT) ch.Hc[r] = 0;

    // ---- rcpp_gibbs_nipt_initialize (:1629-1750)
    {
        Col<NE> one;
#pragma unroll
        for (int i = 0; i < NE; i++) one.v[i] = 1.0;
        for (int h = 0; h < 2; h++)
            for (int g = 0; g < G; g++) ch.st(one, ch.eg[h] + (size_t)g * Ksp);
    }
    chain_sync<NW>();
    if (!init_iteratively) {
        // rcpp_make_eMatGrid_t (copied-from-stitch.cpp:262-281): reads are sorted by grid, so the
        // products of one grid are formed in registers in read order
        ReadStreams<CH> rs;
        rs.base = -1;
        int r = 0;
        while (r < R) {
            if ((r & 63) == 0 && rs.base != r) rs.load(ch, r, nullptr, 0);
            const int g = rl_i32(rs.wif, r & 63);
            Col<NE> e[2];
#pragma unroll
            for (int i = 0; i < NE; i++) e[0].v[i] = e[1].v[i] = 1.0;
            while (r < R) {
                if ((r & 63) == 0 && rs.base != r) rs.load(ch, r, nullptr, 0);
                if (rl_i32(rs.wif, r & 63) != g) break;
                Col<NE> er;
                {
                    typename CH::ErPre x;
                    ch.ld_pre(x, r);
                    ch.read_emission(er, x, rl_i32(rs.dn, r & 63));
                }
                const int h = rl_i32(rs.H, r & 63) - 1;
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    if (h == 0) e[0].v[i] *= er.v[i];
                    else e[1].v[i] *= er.v[i];
                }
                r++;
            }
            ch.st(e[0], ch.eg[0] + (size_t)g * Ksp);
            ch.st(e[1], ch.eg[1] + (size_t)g * Ksp);
        }
        forward_full_both(ch);
        backward_both<NE, NW, false>(ch);   // rcpp_initialize_gibbs_forward_backward (:453-487)
    } else {
        // alpha = beta = 1, c = 1, then only column 0 of alpha is initialised (:1725-1740)
        Col<NE> one;
#pragma unroll
        for (int i = 0; i < NE; i++) one.v[i] = valid[i] ? 1.0 : 0.0;
        for (int h = 0; h < 2; h++) {
            for (int g = 0; g < G; g++) {
                ch.st(one, ch.alpha[h] + (size_t)g * Ksp);
                ch.st(one, ch.beta[h] + (size_t)g * Ksp);
            }
            for (int g = t; g < G; g += NT) ch.cv[h][g] = 1.0;
        }
        chain_sync<NW>();
        for (int h = 0; h < 2; h++) {
            Col<NE> a;
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = valid[i] ? prior * 1.0 : 0.0;
            const double cc = 1 / ch.sum_col(a);
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = a.v[i] * cc;
            ch.st(a, ch.alpha[h]);
            if (t == 0) ch.cv[h][0] = cc;
        }
    }
    chain_sync<NW>();

    int shard_it = 0;
    int status = 0;
    for (int it = 0; it < p.n_its && status == 0; it++) {
        // ================= rcpp_gibbs_nipt_iterate (:1756-1956) =================
        // H_class (record_read_set, :1142-1165) is overwritten for every sampled read in every sweep, so only
        // the last sweep's value is observable: it is computed there only.
        const bool last_sweep = it == p.n_its - 1;
        const bool steady = !init_iteratively || it >= 2;   // every sampled read is in "normal progress" (:817-834)
        Col<NE> a[2];   // alpha of the current grid, both labels (also the reference's alphaHat_m)
        int iRead = 0;  // next unprocessed read
        // software pipeline over reads: the emission column of read iRead is always in flight one read
        // ahead of its use; all per-read scalars come from the lane-held streams
        typename CH::ErPre pre_er;
        if (R > 0) ch.ld_pre(pre_er, 0);
        ReadStreams<CH> rs;
        rs.base = -1;
        GridStreams<CH> gs;
        Col<NE> e[2], bt[2];
        ch.ldm(e[0], ch.eg[0]);
        ch.ldm(e[1], ch.eg[1]);
        ch.ldm(bt[0], ch.beta[0]);
        ch.ldm(bt[1], ch.beta[1]);
        for (int g = 0; g < G; g++) {
            if ((g & 63) == 0) {
                if (g) gs.store_c(ch);
                gs.load_fwd(ch, g);
            }
            const int jg = g & 63;
            const bool has = rl_i32(gs.has, jg) != 0;
            // next grid's eMatGrid and beta columns: issued now, used after this grid's reads (LEAN: fetched at the end of
            // the iteration, straight into the registers of this grid's)
            Col<NE> en[LEAN ? 1 : 2], bn[LEAN ? 1 : 2];
            if constexpr (!LEAN) {
                const size_t gn = (size_t)min(g + 1, G - 1) * Ksp;   // clamped: loads stay unconditional
                ch.ldm(en[0], ch.eg[0] + gn);
                ch.ldm(en[1], ch.eg[1] + gn);
                ch.ldm(bn[0], ch.beta[0] + gn);
                ch.ldm(bn[1], ch.beta[1] + gn);
            }
            double cg[2];
            if (g > 0) {
                // rcpp_alpha_forward_one_QUILT_faster (:671-707), normalize = true
                const double x = rl_f64(gs.t0, jg), t1 = rl_f64(gs.t1, jg);
                const double one_over_K = 1 / (double)Ks;
                const double c2v[2] = {rl_f64(gs.c0, jg), rl_f64(gs.c1, jg)};
                double sp[2];
                ch.sum_col2(a[0], a[1], sp[0], sp[1]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double alphaConst = t1 * sp[h];
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        const double inner = (x * a[h].v[i] + alphaConst * one_over_K);
                        a[h].v[i] = valid[i] ? (has ? e[h].v[i] * inner : inner) : 0.0;
                    }
                }
                double sn[2];
                ch.sum_col2(a[0], a[1], sn[0], sn[1]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double c2 = c2v[h];
                    double aa = 1 / (c2 * sn[h]);
                    cg[h] = c2 * aa;
                    aa *= c2;
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] *= aa;
                }
            } else {
                // rcpp_reinitialize_in_iterations (:712-727)
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] = valid[i] ? prior * e[h].v[i] : 0.0;
                }
                double sn[2];
                ch.sum_col2(a[0], a[1], sn[0], sn[1]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cg[h] = 1 / sn[h];
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] *= cg[h];
                }
            }
            // ---- sample_reads_in_grid (:733-1295), diploid: labels 0 / 1, the third label has prior 0, pC(2) = 1
            bool changed = false;
            // ab_m = alphaHat_m % betaHat_m and pC = its column sums (:853-858): the reference forms them at the grid's first
            // sampled read; nothing changes alpha or beta before that, so they are formed here, for every grid with reads --
            // beta's registers are then free for the whole read loop -- and kept up to date by the moves (:1088-1093)
            Col<NE> ab[2];
            double pC[2] = {1, 1};
            if (has) {
                double s[2] = {0, 0};
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    ab[0].v[i] = a[0].v[i] * bt[0].v[i];
                    ab[1].v[i] = a[1].v[i] * bt[1].v[i];
                    s[0] += ab[0].v[i];
                    s[1] += ab[1].v[i];
                }
                ch.template bsum<2>(s);
                pC[0] = s[0]; pC[1] = s[1];
                if constexpr (LEAN) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
#pragma unroll
                        for (int i = 0; i < NE; i++) s_e[(h * NE + i) * NT + t] = e[h].v[i];
                    }
                }
            }
            int h_rC = 0, h_rA1 = 1;
            bool normal = false, ginit = false, pass = false;
            while (iRead < R) {
                if ((iRead & 63) == 0 && rs.base != iRead) {
                    if (rs.base >= 0) rs.store(ch);
                    rs.load(ch, iRead, runif, it);
                    // counter-based uniforms: 64 at a time, one per lane, instead of one on the scalar unit per read
                    if (!runif) rs.u = stream_uniform(seed_reads, (uint64_t)R * it + iRead + lane);
                }
                const int jr = iRead & 63;
                if (rl_i32(rs.wif, jr) != g) break;
                const int r = iRead;
                const typename CH::ErPre cur_er = pre_er;
                iRead++;
                // unconditional (clamped) so that no control-flow join forces the in-order vmcnt to drain
                // (the next read's table size comes from the lane-held stream; across a stream boundary: the whole table)
                // (measured and dropped in round 4: TWO reads ahead in the lean build -- 2 048 chains 1 115 -> 1 138 ms: the packs are
                // not what the waves wait for)
                ch.ld_pre(pre_er, min(iRead, R - 1), (iRead & 63) ? rl_i32(rs.nent, iRead & 63) : 64);
                if (rl_i32(rs.cat1, jr) != 0) continue;  // reads that cannot discriminate are skipped (:815)
                Col<NE> er, ri;   // the read's emission column and 1 / it (used by normal reads)
                const int dn_r = rl_i32(rs.dn, jr);
                if (dn_r >= 0) {
                    ch.ld(er, ch.eMatRead + (size_t)dn_r * Ksp);
#pragma unroll
                    for (int i = 0; i < NE; i++) ri.v[i] = fast_rcp(er.v[i]);
                } else {
                    ch.expand_with_rcp(er, ri, cur_er);
                }
                if (steady) normal = true;   // (no iterative initialisation, or past its two sweeps: hoisted out of the loop)
                else if (r < first_read && it == 0) pass = true;
                else if (first_read <= r && it == 0) { pass = false; ginit = true; }
                else if (r < first_read && it == 1) { pass = false; ginit = true; }
                else { ginit = false; normal = true; }
                double pA1[2] = {pC[0], pC[1]};
                if (normal) {
                    h_rC = rl_i32(rs.H, jr) - 1;
                    h_rA1 = 1 - h_rC;
                    // dense form for every category (the reference's sparse category-2/3 updates are
                    // algebraically the same sums: test-unit-gibbs-diploid.R:114-124)
                    double s[2] = {0, 0};
                    // wave-uniform branch: two straight-line versions instead of per-element selects (the empty asm keeps
                    // the compiler from converting the branch back into 4 * NE v_cndmask)
                    // (two partial sums per value: the dependent chain of NE fp64 adds is on the read's critical path)
                    double s2[2] = {0, 0};
                    if (h_rC == 0) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < NE; i++) {
                            double &u0 = (i & 1) ? s2[0] : s[0], &u1 = (i & 1) ? s2[1] : s[1];
                            u0 += ab[0].v[i] * ri.v[i];
                            u1 += ab[1].v[i] * er.v[i];
                        }
                        // (the sums pass through an asm of this branch's own: otherwise the two versions are sunk into one
                        // block behind 2 * NE register copies)
                        asm volatile("; current label 0" : "+v"(s[0]), "+v"(s[1]), "+v"(s2[0]), "+v"(s2[1]));
                    } else {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < NE; i++) {
                            double &u0 = (i & 1) ? s2[0] : s[0], &u1 = (i & 1) ? s2[1] : s[1];
                            u0 += ab[1].v[i] * ri.v[i];
                            u1 += ab[0].v[i] * er.v[i];
                        }
                        asm volatile("; current label 1" : "+v"(s[0]), "+v"(s[1]), "+v"(s2[0]), "+v"(s2[1]));
                    }
                    s[0] += s2[0]; s[1] += s2[1];
                    ch.template bsum<2>(s);
                    pA1[h_rC] = s[0];     // the current label loses the read
                    pA1[h_rA1] = s[1];    // the other label gains it
                } else if (ginit) {
                    h_rC = 0; h_rA1 = 1;
                    double s[2] = {0, 0};
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        s[0] += ab[0].v[i] * er.v[i];
                        s[1] += ab[1].v[i] * er.v[i];
                    }
                    ch.template bsum<2>(s);
                    pC[0] = s[0];
                    pA1[1] = s[1];
                }
                // (:998-1046) prior_probs = (0.5, 0.5, 0) and pC(2) = pA(2) = 1: the factors 1 and 0.5 are exact,
                // so the normalised probabilities are P / (P + Q) and Q / (P + Q) bit for bit; label 3 has weight 0
                const double P = pC[0] * pC[1], Q = pA1[0] * pA1[1];
                const double denom = P + Q;
                const double rden = fast_rcp(denom);
                const double norm_pC = P * rden, norm_pA1 = Q * rden;
                const double chance = rl_f64(rs.u, jr);
                const double p0 = (h_rC == 0) ? norm_pC : norm_pA1, p1 = (h_rC == 0) ? norm_pA1 : norm_pC;
                const double cs0 = p0, cs1 = p1 + p0;
                const int h_rN = (chance < cs0) ? 0 : ((chance < cs1) ? 1 : 0);
                if (((h_rN != h_rC) || ginit) && !pass) {
                    changed = true;
                    if (lane == jr) rs.H = h_rN + 1;
                    // eMatGrid's column: in registers, or (LEAN) where it waits in LDS
                    auto mul_e = [&](int h, const Col<NE> &f) {
                        if constexpr (LEAN) {
#pragma unroll
                            for (int i = 0; i < NE; i++) s_e[(h * NE + i) * NT + t] *= f.v[i];
                        } else {
#pragma unroll
                            for (int i = 0; i < NE; i++) e[h].v[i] *= f.v[i];
                        }
                    };
                    if (normal) {   // alphaHat_m, ab_m, eMatGrid of the label that loses the read (:1086-1102)
                        if (h_rC == 0) {
#pragma unroll
                            for (int i = 0; i < NE; i++) { a[0].v[i] *= ri.v[i]; ab[0].v[i] *= ri.v[i]; }
                            mul_e(0, ri);
                        } else {
#pragma unroll
                            for (int i = 0; i < NE; i++) { a[1].v[i] *= ri.v[i]; ab[1].v[i] *= ri.v[i]; }
                            mul_e(1, ri);
                        }
                    }
                    if (h_rN == 0) {   // ... and of the one that gains it (:1090-1108)
#pragma unroll
                        for (int i = 0; i < NE; i++) { a[0].v[i] *= er.v[i]; ab[0].v[i] *= er.v[i]; }
                        mul_e(0, er);
                    } else {
#pragma unroll
                        for (int i = 0; i < NE; i++) { a[1].v[i] *= er.v[i]; ab[1].v[i] *= er.v[i]; }
                        mul_e(1, er);
                    }
                    if (normal || h_rN == 1) { pC[0] = pA1[0]; pC[1] = pA1[1]; }
                }
                if (last_sweep) {
                    // record_read_set (:1142-1165) with the ff = 0 prototypes (:2707-2729)
                    const double x[3] = {p0, p1, 0.0};
                    const double rlc[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0.5 / (0.5 + 0.5), 0.5 / (0.5 + 0.5), 0},
                                              {0.5 / (0.5 + 0.0), 0, 0.0 / (0.5 + 0.0)},
                                              {0, 0.5 / (0.5 + 0.0), 0.0 / (0.5 + 0.0)}, {0.5, 0.5, 0.0}};
                    double local_min = 2;
                    int which = 8;
                    for (int i = 0; i < 7; i++) {
                        const double y = fabs(rlc[i][0] - x[0]) + fabs(rlc[i][1] - x[1]) + fabs(rlc[i][2] - x[2]);
                        if (y < local_min) { local_min = y; which = i; }
                    }
                    if (lane == jr) rs.Hc = (local_min < p.class_sum_cutoff) ? which + 1 : 0;
                }
            }
#ifndef QA_LEAN_LATE_E   // (round 4: 2 048 chains 1 134 -> 1 112 ms; -DQA_LEAN_LATE_E restores the late form for A/B runs)
            // LEAN: the next grid's eMatGrid columns are requested NOW -- e's registers are free (the grid's own columns wait in
            // LDS) -- ahead of the changed columns' stores and of everything else the end of a grid does; beta's follow at the end.
            // The columns a move changed go from LDS to memory through beta's registers (dead since alpha * beta was formed).
            if constexpr (LEAN) {
                const size_t gn0 = (size_t)min(g + 1, G - 1) * Ksp;
                ch.ldm(e[0], ch.eg[0] + gn0);
                ch.ldm(e[1], ch.eg[1] + gn0);
            }
#endif
            if (changed) {
                // re-inject the moved columns and renormalise (:1262-1292)
                double sm[2];
                ch.sum_col2(a[0], a[1], sm[0], sm[1]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double alphaConst = 1 / sm[h];
                    cg[h] *= alphaConst;
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] = a[h].v[i] * alphaConst;
#ifndef QA_LEAN_LATE_E
                    if constexpr (LEAN) {
#pragma unroll
                        for (int i = 0; i < NE; i++) bt[h].v[i] = s_e[(h * NE + i) * NT + t];
                        ch.stm(bt[h], ch.eg[h] + (size_t)g * Ksp);
                    } else {
                        ch.stm(e[h], ch.eg[h] + (size_t)g * Ksp);
                    }
#else
                    if constexpr (LEAN) {
#pragma unroll
                        for (int i = 0; i < NE; i++) e[h].v[i] = s_e[(h * NE + i) * NT + t];
                    }
                    ch.stm(e[h], ch.eg[h] + (size_t)g * Ksp);
#endif
                }
            }
            // alphaHat_t is not read again inside the call (the shard pass runs its own forward): only the state left
            // by the last sweep is observable (hapProbs, state_out), so only that sweep writes it
            if (last_sweep) {
#pragma unroll
                for (int h = 0; h < 2; h++) ch.stm(a[h], ch.alpha[h] + (size_t)g * Ksp);
            }
            gs.set_c(lane, jg, cg[0], cg[1]);
            if constexpr (LEAN) {
#ifndef QA_DBG_NO_LATE_LOADS   // (developer timing build: what the exposed loads of the next grid's columns cost; results are wrong)
                const size_t gn = (size_t)min(g + 1, G - 1) * Ksp;
#ifdef QA_LEAN_LATE_E
                ch.ldm(e[0], ch.eg[0] + gn);
                ch.ldm(e[1], ch.eg[1] + gn);
#endif
                ch.ldm(bt[0], ch.beta[0] + gn);
                ch.ldm(bt[1], ch.beta[1] + gn);
#endif
            } else {
                e[0] = en[0]; e[1] = en[1];
                bt[0] = bn[0]; bt[1] = bn[1];
            }
        }
        gs.store_c(ch);
        if (rs.base >= 0) rs.store(ch);
        chain_sync<NW>();
#ifndef QA_DBG_SKIP_BWD   // (developer timing builds: scripts/perf_gibbs.py with QUILT_AMD_LIB)
        backward_both<NE, NW, true>(ch);
#endif
        // ---- underflow check (:2959-2969)
        {
            double s[2] = {0, 0};
            for (int g = t; g < G; g += NT) { s[0] += ch.cv[0][g]; s[1] += ch.cv[1][g]; }
            ch.template bsum<2>(s);
            if (!isfinite(s[0]) || !isfinite(s[1])) status = 1;
        }
        if (p.per_it) {
            // add_to_per_it_likelihoods (:1583-1621): -sum(log c_h) and the number of reads per label after this sweep
            double s[4] = {0, 0, 0, 0};
            for (int g = t; g < G; g += NT) { s[0] -= log(ch.cv[0][g]); s[1] -= log(ch.cv[1][g]); }
            for (int r = t; r < R; r += NT) { const int h = ch.H[r]; s[2] += h == 1 ? 1.0 : 0.0; s[3] += h == 2 ? 1.0 : 0.0; }
            ch.template bsum<4>(s);
            if (t == 0) {
                double *o = p.per_it + ((size_t)c * p.n_its + it) * 8;
                o[0] = s[0]; o[1] = s[1]; o[2] = 0; o[3] = s[2]; o[4] = s[3]; o[5] = 0; o[6] = 0; o[7] = 0;
            }
        }
        if (status) break;
        bool to_block = false;
        for (int i = 0; i < p.n_block; i++) if (p.block_its[i] == it) to_block = true;
#ifdef QA_DBG_SKIP_SHARD
        to_block = false;
#endif
        if (to_block && p.do_shard) {
            // ============ Rcpp_shard_block_gibbs_resampler (gibbs-nipt-block.cpp:1975-2355), ff = 0,
            // shard_check_every_pair: one left-to-right pass deciding at every grid whether everything
            // to the right swaps haplotypes ============
            const double *ru = p.runif_shard + ((size_t)c * p.n_block + shard_it) * (G - 1);
            shard_it++;
            double mloc1, mloc2, mlc1 = 0, mlc2 = 0;
            {
                double s[2] = {0, 0};
                for (int g = t; g < G; g += NT) { s[0] -= log(ch.cv[0][g]); s[1] -= log(ch.cv[1][g]); }
                ch.template bsum<2>(s);
                mloc1 = s[0]; mloc2 = s[1];
            }
            bool flip = false;
            int ir = 0;
            Col<NE> s_a[2];
            GridStreams<CH> ss;
            ReadStreams<CH> rr;
            rr.base = -1;
            bool rr_dirty = false;
            for (int g = 0; g < G; g++) {
                if ((g & 63) == 0) {
                    if (g) ss.store_c(ch);
                    ss.load_fwd(ch, g);
                }
                const int jg = g & 63;
                const double oc1 = rl_f64(ss.c0, jg), oc2 = rl_f64(ss.c1, jg);
                // everything to the right of an accepted switch swaps haplotypes: the grid's eMatGrid columns are fetched
                // crosswise then (a wave-uniform choice of address, not 4 * NE selects) and written back in their new places
                Col<NE> e2[2];
                ch.ldm(e2[0], ch.eg[flip ? 1 : 0] + (size_t)g * Ksp);
                ch.ldm(e2[1], ch.eg[flip ? 0 : 1] + (size_t)g * Ksp);
                // (LEAN: beta's columns are fetched where they are used, after the grid's eMatGrid columns are dead: fewer
                // spilled registers in this loop)
                Col<NE> b1, b2;
                if (!LEAN && g < G - 1) {
                    ch.ldm(b1, ch.beta[0] + (size_t)g * Ksp);
                    ch.ldm(b2, ch.beta[1] + (size_t)g * Ksp);
                }
                double cn[2];
                if (g == 0) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
#pragma unroll
                        for (int i = 0; i < NE; i++) s_a[h].v[i] = valid[i] ? prior * e2[h].v[i] : 0.0;
                    }
                    double sm[2];
                    ch.sum_col2(s_a[0], s_a[1], sm[0], sm[1]);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        cn[h] = 1 / sm[h];
#pragma unroll
                        for (int i = 0; i < NE; i++) s_a[h].v[i] *= cn[h];
                    }
                } else {
                    if (flip) {
                        ch.stm(e2[0], ch.eg[0] + (size_t)g * Ksp);
                        ch.stm(e2[1], ch.eg[1] + (size_t)g * Ksp);
                    }
                    // rcpp_alpha_forward_one (gibbs-nipt.cpp:627-657), alphaMat = 1/Ks, normalize
                    const double x = rl_f64(ss.t0, jg), t1 = rl_f64(ss.t1, jg);
                    double sp[2];
                    ch.sum_col2(s_a[0], s_a[1], sp[0], sp[1]);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const double alphaConst = t1 * sp[h];
                        const double c2 = (h == 0) ? oc1 : oc2;
#pragma unroll
                        for (int i = 0; i < NE; i++)
                            s_a[h].v[i] = valid[i] ? c2 * e2[h].v[i] * (x * s_a[h].v[i] + alphaConst * prior) : 0.0;
                    }
                    double sm[2];
                    ch.sum_col2(s_a[0], s_a[1], sm[0], sm[1]);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const double c2 = (h == 0) ? oc1 : oc2;
                        const double aa = 1 / sm[h];
                        cn[h] = c2 * aa;
#pragma unroll
                        for (int i = 0; i < NE; i++) s_a[h].v[i] *= aa;
                    }
                }
                if (last_sweep) {
                    ch.stm(s_a[0], ch.alpha[0] + (size_t)g * Ksp);
                    ch.stm(s_a[1], ch.alpha[1] + (size_t)g * Ksp);
                }
                ss.set_c(lane, jg, cn[0], cn[1]);
                mlc1 -= log(cn[0]);
                mlc2 -= log(cn[1]);
                while (ir < R) {
                    if ((ir & 63) == 0 && rr.base != ir) {
                        if (rr.base >= 0 && rr_dirty) rr.store(ch);
                        rr.load(ch, ir, nullptr, 0);
                        rr_dirty = false;
                    }
                    if (rl_i32(rr.wif, ir & 63) != g) break;
                    if (flip) {
                        if (lane == (ir & 63)) rr.H = 3 - rr.H;
                        rr_dirty = true;
                    }
                    ir++;
                }
                if (g < G - 1) {
                    if constexpr (LEAN) {
                        ch.ldm(b1, ch.beta[0] + (size_t)g * Ksp);
                        ch.ldm(b2, ch.beta[1] + (size_t)g * Ksp);
                    }
                    double s[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        s[0] += s_a[0].v[i] * b1.v[i];
                        s[1] += s_a[1].v[i] * b2.v[i];
                        s[2] += s_a[1].v[i] * b1.v[i];
                        s[3] += s_a[0].v[i] * b2.v[i];
                    }
                    ch.template bsum<4>(s);
                    const double pA1 = mlc1 + mloc1 + log(s[0]);
                    const double pA2 = mlc2 + mloc2 + log(s[1]);
                    const double pB1 = mlc2 + mloc1 + log(s[2]);
                    const double pB2 = mlc1 + mloc2 + log(s[3]);
                    const double diff = pB1 + pB2 - pA1 - pA2;
                    double probs1 = 1;
                    const double probs2 = exp(diff);
                    const double ps = probs1 + probs2;
                    probs1 /= ps;
                    const double ug = seed_shard ? stream_uniform(seed_shard, (uint64_t)(shard_it - 1) * (G - 1) + g) : ru[g];
                    flip = ug > probs1;
                }
                mloc1 += log(oc1);
                mloc2 += log(oc2);
            }
            ss.store_c(ch);
            if (rr.base >= 0 && rr_dirty) rr.store(ch);
            chain_sync<NW>();
            backward_both<NE, NW, false>(ch);
        }
    }
    if (t == 0) p.status[c] = status;
}

// ---------------------------------------------------------------------------------------------
// k_happrobs: gamma = alpha * beta / c on the fly, scattered to the 32 SNPs of the grid by each
// haplotype's word (gibbs-small.cpp:540-634).  One 256-thread block per (grid, chain): thread =
// (bit b, slice of 8 over k).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_happrobs(GibbsParams p) {
    const int g = blockIdx.x, c = blockIdx.y;
    const int b = threadIdx.x & 31, part = threadIdx.x >> 5;
    __shared__ double s_g[3][8][32], s_t[3][8];
    // gamma columns and panel words sized by the call (nH x Ksp doubles + Ksp words: 18 KB at Ks = 600, two labels, against the
    // 34 KB of fixed [3][1024] arrays): twice the workgroups per compute unit for a kernel that waits on gathers
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int Ks = p.Ks, Ksp = p.Ksp, G = p.G;
    double *const s_gk0 = reinterpret_cast<double *>(s_dyn);
    uint32_t *const s_w = reinterpret_cast<uint32_t *>(s_gk0 + (size_t)p.nH * Ksp);
    auto gk = [&](int h, int k) -> double & { return s_gk0[(size_t)h * Ksp + k]; };
    const int32_t *which = p.which + (size_t)c * Ks;
    const size_t mat = (size_t)G * Ksp;
    // gamma once per (label, haplotype), read with consecutive lanes on consecutive haplotypes (it used to be re-formed by
    // each of the 32 bit lanes from broadcast loads: 70 ms per launch, load-issue bound); the sums below keep their order
    for (int k = threadIdx.x; k < Ks; k += 256) {
        const int kk = which[k];
        s_w[k] = panel_word(p, g, kk, p.hm[(size_t)g * p.Kp + kk]);
        for (int h = 0; h < p.nH; h++) {
            const size_t o = ((size_t)c * p.nH + h) * mat + (size_t)g * Ksp + k;
            const double x = 1 / p.cvec[((size_t)c * 3 + h) * G + g];
            gk(h, k) = (p.alpha[o] * p.beta[o]) * x;
        }
    }
    __syncthreads();
    const int s = 32 * g, nLocal = min(32, p.T - s);
    double acc[3] = {0, 0, 0}, tot[3] = {0, 0, 0};
    if (p.nH == 2) {
        for (int k = part; k < Ks; k += 8) {
            const bool on = (s_w[k] >> b) & 1u;
            const double g0 = gk(0, k), g1 = gk(1, k);
            tot[0] += g0; tot[1] += g1;
            if (on) { acc[0] += g0; acc[1] += g1; }
        }
    } else {   // three labels, written out like the two: a loop over a run-time label count indexes acc / tot dynamically (scratch)
        for (int k = part; k < Ks; k += 8) {
            const bool on = (s_w[k] >> b) & 1u;
            const double g0 = gk(0, k), g1 = gk(1, k), g2 = gk(2, k);
            tot[0] += g0; tot[1] += g1; tot[2] += g2;
            if (on) { acc[0] += g0; acc[1] += g1; acc[2] += g2; }
        }
    }
#pragma unroll
    for (int h = 0; h < 3; h++) {
        if (h < p.nH) {
            s_g[h][part][b] = acc[h];
            if (b == 0) s_t[h][part] = tot[h];
        }
    }
    __syncthreads();
    if (part == 0 && b < nLocal) {
        double g1[3] = {0, 0, 0.0 * (1 - p.ref_error) + 0.0 * p.ref_error};
#pragma unroll
        for (int h = 0; h < 3; h++) {
            if (h < p.nH) {
                double on = 0, all = 0;
                for (int q = 0; q < 8; q++) { on += s_g[h][q][b]; all += s_t[h][q]; }
                const double off = all - on;  // sum over haplotypes whose bit is 0
                g1[h] = on * (1 - p.ref_error) + off * p.ref_error;
            }
        }
        const double g0 = g1[0], gB = g1[1], g2 = g1[2];
        double *hp = p.hapProbs + ((size_t)c * p.T + s + b) * 3;
        double *gm = p.genProbsM + ((size_t)c * p.T + s + b) * 3;
        double *gf = p.genProbsF + ((size_t)c * p.T + s + b) * 3;
        hp[0] = g0; hp[1] = gB; hp[2] = g2;
        gm[0] = (1 - g0) * (1 - gB);
        gm[1] = (g0 * (1 - gB) + (1 - g0) * gB);
        gm[2] = g0 * gB;
        gf[0] = (1 - g0) * (1 - g2);
        gf[1] = (g0 * (1 - g2) + (1 - g0) * g2);
        gf[2] = g0 * g2;
    }
}


// ---------------------------------------------------------------------------------------------
// k_happrobs_rc: hapProbs / genProbs over ALL SNPs for the rare + common call
// (rcpp_calculate_genProbs_and_hapProbs_final_rare_common, gibbs-small.cpp:711-867).  One 256-thread block per
// (all-SNP grid, chain): gamma of the grid's column goes to LDS once; thread = (SNP b of the grid, slice of 8 over k).
// A common SNP reads its allele from the panel word of its common grid (the block's 32 SNPs span at most two of them);
// a rare SNP from the haplotype's rare list, or is ref_error outright when no selected haplotype carries it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_happrobs_rc(GibbsParams p) {
    const int g = blockIdx.x, c = blockIdx.y;
    const int b = threadIdx.x & 31, part = threadIdx.x >> 5;
    __shared__ double s_on[3][8][32], s_all[3][8];
    __shared__ int s_cg;
    // gamma columns, the two common grids' panel words and the chain's haplotype list sized by the call (see k_happrobs)
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int Ks = p.Ks, Ksp = p.Ksp, G = p.G;
    double *const s_gam0 = reinterpret_cast<double *>(s_dyn);
    uint32_t *const s_w0 = reinterpret_cast<uint32_t *>(s_gam0 + (size_t)p.nH * Ksp);
    int32_t *const s_which = reinterpret_cast<int32_t *>(s_w0 + 2 * (size_t)Ksp);
    auto gam = [&](int h, int k) -> double & { return s_gam0[(size_t)h * Ksp + k]; };
    auto sw = [&](int q, int k) -> uint32_t & { return s_w0[(size_t)q * Ksp + k]; };
    const int32_t *which = p.which + (size_t)c * Ks;
    const int s = 32 * g, nLocal = min(32, p.T - s);
    // the common grid of the block's first common SNP: the 32 SNPs looked at together by the first wave (it used to be one
    // thread walking them with a dependent load each)
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        const int cs = i < nLocal ? p.rc_common[s + i] : -1;
        const unsigned long long m = __ballot(cs >= 0);
        const int first = m ? __ffsll((long long)m) - 1 : -1;
        const int cs_first = __shfl(cs, first < 0 ? 0 : first, 64);
        if (i == 0) s_cg = first < 0 ? -1 : (cs_first >> 5);
    }
    __syncthreads();
    const int cg = s_cg;
    const size_t mat = (size_t)G * Ksp;
    for (int k = threadIdx.x; k < Ks; k += 256) {
        const int kk = which[k];
        s_which[k] = kk;
        for (int q = 0; q < 2; q++) {
            const int gq = cg + q;
            sw(q, k) = (cg >= 0 && gq < p.rc_Gc) ? panel_word(p, gq, kk, p.hm[(size_t)gq * p.Kp + kk]) : 0u;
        }
        for (int h = 0; h < p.nH; h++) {
            const double *a = p.alpha + ((size_t)c * p.nH + h) * mat + (size_t)g * Ksp;
            const double *be = p.beta + ((size_t)c * p.nH + h) * mat + (size_t)g * Ksp;
            const double x = 1 / p.cvec[((size_t)c * 3 + h) * G + g];
            gam(h, k) = (a[k] * be[k]) * x;
        }
    }
    __syncthreads();
    const int snp = s + b;
    const int cs = b < nLocal ? p.rc_common[snp] : -1;
    const bool rare = b < nLocal && cs < 0;
    const bool any = rare && ((p.rc_any[(size_t)c * p.rc_words + (snp >> 5)] >> (snp & 31)) & 1u);
    const int nH = p.nH;
    double on[3] = {0, 0, 0};
    if (b < nLocal && rare && any && p.rc_pairs) {
        // the rows that carry this rare SNP, from the chain's (SNP, row) pairs of this grid: ascending in the row, so the partial
        // sum of this thread's rows (row = part mod 8) forms in the order the walk over all rows gave it.  (That walk searched the
        // rare list of each of the 600 selected haplotypes -- three dependent loads a search -- for every rare SNP somebody
        // carries: 110 us for such a thread, 39 % of the blocks held one, 200 ms per all-SNP launch.)
        const int32_t *po = p.rc_pair_off + (size_t)c * (G + 1);
        const uint16_t *pp = p.rc_pairs + p.rc_pair_base[c];
        for (int i = po[g]; i < po[g + 1]; i++) {
            const uint32_t e = pp[i];
            const int k = (int)(e & 1023u);
            if ((int)(e >> 10) == b && (k & 7) == part) {
#pragma unroll
                for (int h = 0; h < 3; h++) if (h < nH) on[h] += gam(h, k);   // (compile-time label index: on[] stays in registers)
            }
        }
    } else if (b < nLocal && (!rare || any)) {
        const int q = rare ? 0 : (cs >> 5) - cg, bit = cs & 31;
        for (int k = part; k < Ks; k += 8) {
            const bool alt = rare ? rare_has_alt(p, s_which[k], snp) : ((sw(q, k) >> bit) & 1u);
#pragma unroll
            for (int h = 0; h < 3; h++) {
                if (h < nH && alt) on[h] += gam(h, k);
            }
        }
    }
    for (int h = 0; h < 3; h++) s_on[h][part][b] = on[h];
    if (b == 0) {   // the column total does not depend on the SNP
        double t[3] = {0, 0, 0};
        for (int k = part; k < Ks; k += 8) {
#pragma unroll
            for (int h = 0; h < 3; h++) if (h < nH) t[h] += gam(h, k);
        }
        for (int h = 0; h < 3; h++) s_all[h][part] = t[h];
    }
    __syncthreads();
    if (part == 0 && b < nLocal) {
        double g1[3] = {0, 0, 0};
#pragma unroll
        for (int h = 0; h < 3; h++) {
            if (h < nH) {
                double o = 0, t = 0;
                for (int q = 0; q < 8; q++) { o += s_on[h][q][b]; t += s_all[h][q]; }
                if (!rare) g1[h] = o * (1 - p.ref_error) + (t - o) * p.ref_error;
                else if (!any) g1[h] = p.ref_error;
                else g1[h] = t * p.ref_error + o * (1 - 2 * p.ref_error);
            }
        }
        const double g0 = g1[0], gB = g1[1], g2 = g1[2];
        double *hp = p.hapProbs + ((size_t)c * p.T + snp) * 3;
        double *gm = p.genProbsM + ((size_t)c * p.T + snp) * 3;
        double *gf = p.genProbsF + ((size_t)c * p.T + snp) * 3;
        hp[0] = g0; hp[1] = gB; hp[2] = g2;
        gm[0] = (1 - g0) * (1 - gB);
        gm[1] = (g0 * (1 - gB) + (1 - g0) * gB);
        gm[2] = g0 * gB;
        gf[0] = (1 - g0) * (1 - g2);
        gf[1] = (g0 * (1 - g2) + (1 - g0) * g2);
        gf[2] = g0 * g2;
    }
}


// ---------------------------------------------------------------------------------------------
// k_ematread_dense: `rcpp_make_eMatRead_t` (copied-from-stitch.cpp:115-229) for dense per-SNP
// haplotype dosages (the 2-3 "haplotypes" of calculate_eMatRead_t_vs_haplotypes, functions.R:2975-3020).
// One thread per (read, chain): K is 2 or 3, the products run over the read's bases in order.
// ---------------------------------------------------------------------------------------------
struct DenseParams {
    // rare + common form (qa_rcpp_make_eMatRead_t_rare_common): eHaps is [C][K][Tc] over the COMMON SNPs, a rare SNP's entry is
    // 0.5 (get_initial_read_labels, rare_common.R:61-107), the reads are per SAMPLE with a chain -> sample map
    const int32_t *common_index = nullptr;   // [T] all-SNP index -> common-SNP index or -1
    const int32_t *chain_sample = nullptr;   // [C] or null (reads per chain)
    const int32_t *out_off = nullptr;        // [C] first output row of the chain (with chain_sample)
    int Tc = 0;
    int C, K, T, Jmax, rescale, hap_major;
    double inv_maxdiff;
    const double *eHaps;      // [C][T][K], or [C][K][T] with hap_major
    const int32_t *read_off, *read_ptr, *base_off, *u, *bq;
    const double *pR_tab, *pA_tab;
    double *out;              // [sum R][K]
};

__global__ __launch_bounds__(64) void k_ematread_dense(DenseParams p) {
    const int c = blockIdx.y;
    const int sr = p.chain_sample ? p.chain_sample[c] : c;   // whose reads
    const int R = p.read_off[sr + 1] - p.read_off[sr];
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const int32_t *rp = p.read_ptr + p.read_off[sr] + sr;
    const int32_t *u = p.u + p.base_off[sr], *bq = p.bq + p.base_off[sr];
    const double *eh = p.eHaps + (size_t)c * (p.common_index ? p.Tc : p.T) * p.K;
    double v[3] = {1, 1, 1};
    const int s = rp[r];
    int J = rp[r + 1] - s - 1;
    if (J >= p.Jmax) J = p.Jmax;
    for (int j = 0; j <= J; j++) {
        const int b = bq[s + j];
        if (b == 0) continue;
        const int ab = b < 0 ? -b : b;
        const double pR = p.pR_tab[(b > 0 ? 256 : 0) + ab], pA = p.pA_tab[(b > 0 ? 256 : 0) + ab];
        if (p.common_index) {
            const int cs = p.common_index[u[s + j]];
            for (int k = 0; k < p.K; k++) { const double ek = cs >= 0 ? eh[(size_t)k * p.Tc + cs] : 0.5; v[k] *= (ek * pA + (1 - ek) * pR); }
        } else if (p.hap_major) {
            const double *e = eh + u[s + j];
            for (int k = 0; k < p.K; k++) { const double ek = e[(size_t)k * p.T]; v[k] *= (ek * pA + (1 - ek) * pR); }
        } else {
            const double *e = eh + (size_t)u[s + j] * p.K;
            for (int k = 0; k < p.K; k++) v[k] *= (e[k] * pA + (1 - e[k]) * pR);
        }
    }
    if (p.rescale) {
        double x = 0;
        for (int k = 0; k < p.K; k++) if (v[k] > x) x = v[k];
        const double d1 = 1 / x;
        if (isinf(x) || x == 0 || isinf(d1)) {
            for (int k = 0; k < p.K; k++) v[k] = 1;
        } else {
            for (int k = 0; k < p.K; k++) {
                v[k] *= d1;
                if (v[k] < p.inv_maxdiff) v[k] = p.inv_maxdiff;
            }
        }
    }
    double *o = p.out + (size_t)((p.out_off ? p.out_off[c] : p.read_off[c]) + r) * p.K;
    for (int k = 0; k < p.K; k++) o[k] = v[k];
}

// rcpp_int_contract(round(hapProbs)) per chain and label: hap [C][T][3] -> words [C][3][G]
__global__ __launch_bounds__(256) void k_pack_hap_words(const double *hap, int T, int G, int32_t *words) {
    const int g = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, c = blockIdx.z;
    if (g >= G) return;
    uint32_t w = 0;
    for (int b = 0; b < 32; b++) {
        const int t = 32 * g + b;
        if (t < T && hap[((size_t)c * T + t) * 3 + h] > 0.5) w |= 1u << b;
    }
    words[((size_t)c * 3 + h) * G + g] = (int32_t)w;
}

// hap [C][T][3] -> out [C][nL][T] (haploid dosages label by label: qa_gibbs_opts_t.hap_major_out)
__global__ __launch_bounds__(256) void k_hap_major(const double *hap, int T, int nL, double *out) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (t >= T) return;
    for (int h = 0; h < nL; h++) out[((size_t)c * nL + h) * T + t] = hap[((size_t)c * T + t) * 3 + h];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace qa {

struct GibbsScratch {
    DBuf<int32_t> which, read_off, read_ptr, base_off, u, bq, wif, block_its, first_read, H, H_class, status;
    DBuf<uint8_t> ghr, is_cat1, er_nent;
    DBuf<int32_t> dense_of;
    DBuf<size_t> eridx_off;
    ABuf<uint8_t> er_idx;
    ABuf<double> er_tab;
    DBuf<double> tabs, runif_reads, runif_shard, tm;
    ABuf<double> eMatRead, alpha, beta, eg, cvec, hap, gm, gf;   // carved from the panel's arena per call
    DBuf<size_t> eread_off;
    DBuf<uint64_t> seeds;
    DBuf<uint32_t> rc_any;
    DBuf<uint16_t> rc_pairs;
    DBuf<int32_t> rc_pair_off;
    DBuf<size_t> rc_pair_base;
    DBuf<double> blk_rate2, ff_chain, per_it;
    DBuf<int32_t> blk_where, blk_tab, blk_n;
};

}  // namespace qa

struct GibbsHolder {
    qa::GibbsScratch s;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace {

thread_local std::unique_ptr<GibbsHolder> g_gibbs;

// the reference carries pR / pA over bases with bq == 0, even across reads (gibbs-small.cpp:139-181,
// copied-from-stitch.cpp:139-175): fold that rule into an "effective" base quality (input marshalling)
// The bq == 0 carry-over ("fold_zero_base_qualities"): a base without a quality takes the last quality seen in its chain's reads
// (0 until one is seen); |bq| <= 255.  Applied per chain by the callers' host threads (gibbs_chunk, make_eMatRead_t_impl).

// eps tables with the host libm (what the reference's pow() is), so the device needs no pow:
// [0..255] pR for bq < 0, [256..511] pR for bq > 0, [512..767] pA for bq < 0, [768..1023] pA for bq > 0
std::vector<double> base_quality_tables() {
    std::vector<double> tabs(4 * 256);
    for (int q = 0; q < 256; q++) {
        const double en = std::pow(10, (double)(-q) / 10), ep = std::pow(10, -(double)q / 10);
        tabs[q] = 1 - en;
        tabs[256 + q] = ep / 3;
        tabs[512 + q] = en / 3;
        tabs[768 + q] = 1 - ep;
    }
    return tabs;
}

template <int NEALL, int NW>
void launch_ematread(const GibbsParams &prm, int maxR, hipStream_t st) {
    hipLaunchKernelGGL((k_ematread<NEALL, NW>), dim3((maxR + kReadsPerWave - 1) / kReadsPerWave, prm.C), dim3(64), 0, st, prm);
    QA_HIP(hipGetLastError());
}

template <int NE, int NW, bool LEAN = false>
void launch_gibbs_kernel(const GibbsParams &prm, hipStream_t st) {
    hipLaunchKernelGGL((k_gibbs<NE, NW, LEAN>), dim3(prm.C), dim3(64 * NW), 0, st, prm);
    QA_HIP(hipGetLastError());
}

// Geometry of one chain: NW waves x NE rows per thread with 64 * NW * NE == Ksp.
// The single-wave kernel (10 columns per lane) holds 512 registers, i.e. one chain per SIMD, 1024 per device, and a
// chain's time is its serial per-read latency: measured at Ksubset = 600 with 128 chains NW = 2 is fastest (0.64 s vs
// 0.77 NW = 1, 0.70 NW = 5, 0.86 NW = 10: more waves = costlier exchange per read), with 896 chains NW = 1 (0.94 s vs
// 1.36 NW = 2, which no longer fits in one wave of workgroups).  Geometries with several waves exist for Ksp = 640.
int choose_gibbs_waves(int Ksp, int C, int share) {
    if (Ksp != 640) return 1;
    int nw = 1;
    // share: host threads sharing the device; 0: device phases (qa_panel_set_exclusive) -- launches that fit run together, one
    // SIMD slot per wave, and a phase lasts as long as its slowest launch: one wave per chain always
    if (share > 0 && (long)C * 2 <= 1024 / share) nw = 2;
    // device phases, a handful of chains (one to eight samples: the quick-start's shape): nothing shares the phase that two waves
    // per chain could hold up, and a chain's serial time is all there is -- one sample through the whole pipeline 3.53 -> 3.15 s
    // (scripts/perf_latency.py; labels identical in every geometry).  Larger launches keep one wave: a set's 128 / 256 phasing
    // chains run beside another set's main chains only while their waves fit the 1 024 SIMD slots together.
    // Round 6: up to 256 chains (a single job of 32 samples -- BASELINE configs[1] as stated -- is one launch set of 224 + 32
    // chains): two waves per chain still leave half of the 1 024 SIMD slots to whatever shares the phase, and such a job is all
    // serial time (tests/test_configs_gpu.py records it).
    if (share == 0 && C <= 256) nw = 2;
    if (const char *forced = getenv("QA_GIBBS_NW")) {   // test hook: exercise every geometry
        const int f = atoi(forced);
        if (f == 1 || f == 2 || f == 5 || f == 10) nw = f;
    }
    return nw;
}

// More chains than SIMDs in one launch: the 256-register build, two chains per SIMD, all of them resident at once (2 048 chains:
// 1.24 s against 2 x 0.72 s for two launches of the 512-register build; alone on a SIMD the lean build is the slower one,
// 0.91 s).  QA_GIBBS_LEAN = 1 / 0: test hook forcing / forbidding it.
bool use_lean_build(int C) {
    if (const char *f = getenv("QA_GIBBS_LEAN")) return atoi(f) != 0;
    return C > 1024;
}

void launch_gibbs(const GibbsParams &prm, int maxR, hipStream_t st, hipEvent_t *ev, bool want_probs, int nw,
                  const std::function<void()> &nipt_sweeps) {
    const int NE1 = prm.Ksp / 64;   // rows per lane with one wave
    QA_HIP(hipEventRecord(ev[0], st));
    switch (NE1) {
        case 1: launch_ematread<1, 1>(prm, maxR, st); break;
        case 2: launch_ematread<2, 1>(prm, maxR, st); break;
        case 3: launch_ematread<3, 1>(prm, maxR, st); break;
        case 4: launch_ematread<4, 1>(prm, maxR, st); break;
        case 5: launch_ematread<5, 1>(prm, maxR, st); break;
        case 6: launch_ematread<6, 1>(prm, maxR, st); break;
        case 7: launch_ematread<7, 1>(prm, maxR, st); break;
        case 8: launch_ematread<8, 1>(prm, maxR, st); break;
        case 9: launch_ematread<9, 1>(prm, maxR, st); break;
        case 10:
            if (nw == 10) launch_ematread<10, 10>(prm, maxR, st);
            else if (nw == 5) launch_ematread<10, 5>(prm, maxR, st);
            else if (nw == 2) launch_ematread<10, 2>(prm, maxR, st);
            else launch_ematread<10, 1>(prm, maxR, st);
            break;
        case 12: launch_ematread<12, 1>(prm, maxR, st); break;
        case 16: launch_ematread<16, 1>(prm, maxR, st); break;
        default: throw std::runtime_error("Ksubset geometry not built (Ksubset / 64 rounded up must be 1..10, 12 or 16)");
    }
    QA_HIP(hipEventRecord(ev[1], st));
    if (prm.nH == 3) {
        nipt_sweeps();   // three-label sampler (NIPT) with its block passes: gibbs3.hip, driven from gibbs_chunk
    } else if (NE1 == 10) {
        if (nw == 10) launch_gibbs_kernel<1, 10>(prm, st);
        else if (nw == 5) launch_gibbs_kernel<2, 5>(prm, st);
        else if (nw == 2) launch_gibbs_kernel<5, 2>(prm, st);
        else if (use_lean_build(prm.C)) launch_gibbs_kernel<10, 1, true>(prm, st);
        else launch_gibbs_kernel<10, 1>(prm, st);
    } else {
        switch (NE1) {
            case 1: launch_gibbs_kernel<1, 1>(prm, st); break;
            case 2: launch_gibbs_kernel<2, 1>(prm, st); break;
            case 3: launch_gibbs_kernel<3, 1>(prm, st); break;
            case 4: launch_gibbs_kernel<4, 1>(prm, st); break;
            case 5: launch_gibbs_kernel<5, 1>(prm, st); break;
            case 6: launch_gibbs_kernel<6, 1>(prm, st); break;
            case 7: launch_gibbs_kernel<7, 1>(prm, st); break;
            case 8: launch_gibbs_kernel<8, 1>(prm, st); break;
            case 9: launch_gibbs_kernel<9, 1>(prm, st); break;
            case 12: launch_gibbs_kernel<12, 1>(prm, st); break;
            default: launch_gibbs_kernel<16, 1>(prm, st); break;
        }
    }
    QA_HIP(hipEventRecord(ev[2], st));
    if (want_probs) {   // return_hapProbs / return_genProbs (functions.R:2566-2599): skipped when nobody asks
        // dynamic LDS: nH gamma columns of Ksp doubles + the panel words (+ the haplotype list for the rare + common form)
        const size_t lds_hp = (size_t)prm.nH * prm.Ksp * 8 + (size_t)prm.Ksp * 4;
        const size_t lds_rc = (size_t)prm.nH * prm.Ksp * 8 + (size_t)2 * prm.Ksp * 4 + (size_t)prm.Ksp * 4;
        if (prm.rc_common) hipLaunchKernelGGL(k_happrobs_rc, dim3(prm.G, prm.C), dim3(256), lds_rc, st, prm);
        else hipLaunchKernelGGL(k_happrobs, dim3(prm.G, prm.C), dim3(256), lds_hp, st, prm);
        QA_HIP(hipGetLastError());
    }
    QA_HIP(hipEventRecord(ev[3], st));
}

}  // namespace

static int gibbs_chunk(qa_panel_t *pn, size_t arena_need, const qa_rare_common *rc, const qa_gibbs_opts_t *o, const double *ff_chain, const int64_t *rep_off, const int32_t *rep_id, int per_it_off, int32_t n_chain, const int32_t *which_haps_to_use_1based,
                   const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                   const int32_t *wif, const double *runif_reads, const int32_t *first_read,
                   const double *runif_shard, int32_t *H, int32_t *H_class, double *hapProbs_t,
                   double *genProbsM_t, double *genProbsF_t, int32_t *underflow_problem, double *state_out,
                   const uint64_t *seed_reads, const uint64_t *seed_shard) {
    {
        if (!g_gibbs) g_gibbs.reset(new GibbsHolder());
        auto &S = g_gibbs->s;
        const bool tmg = getenv("QA_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double T0 = now();
        // the Gibbs launches of a handle that shares the device go to its own CU partition (qa_panel_set_cu_partition): the
        // chains hold whole register files for the launch's lifetime, and spread over every CU they would leave no CU free
        // for the other handle's full-panel workgroups
        hipStream_t st = pn->gibbs_stream ? pn->gibbs_stream : pn->stream;
        const int C = n_chain, G = rc ? rc->G_all : pn->G, T = rc ? rc->T_all : pn->T, Ks = o->Ks;
        const int Ksp = (Ks + 63) / 64 * 64, NE = Ksp / 64;
        const int rc_words = rc ? (T + 31) / 32 : 0;
        std::vector<uint32_t> rc_any((size_t)C * rc_words, 0u);
        const int n_its = o->n_gibbs_burn_in_its + o->n_gibbs_sample_its;
        const int totR = read_off[C];
        // bases: per chain the CSR block read_ptr[read_off[c] + c .. read_off[c+1] + c] is local (starts at 0)
        std::vector<int32_t> base_off(C + 1, 0), which0((size_t)C * Ks), bq_eff;
        std::vector<size_t> eoff(C), ixoff(C);
        std::vector<uint8_t> ghr((size_t)C * G, 0);
        std::vector<int32_t> dense_of(std::max(totR, 1), -1);
        const int nw = o->ff != 0.0 ? qa::gibbs3_waves(Ksp, C, pn->sharers()) : choose_gibbs_waves(Ksp, C, pn->exclusive ? 0 : pn->share);
        const int er_nt = 64 * nw, er_padb = padb_of(NE / nw);
        int maxR = 0;
        size_t etot = 0, ixtot = 0;
        for (int c = 0; c < C; c++) {   // offsets first (cheap), then the per-chain passes over the reads in parallel
            const int R = read_off[c + 1] - read_off[c];
            maxR = std::max(maxR, R);
            const int32_t *rp = read_ptr + read_off[c] + c;
            base_off[c + 1] = base_off[c] + rp[R];
            ixoff[c] = ixtot;
            ixtot += (size_t)R * er_nt * er_padb;
        }
        const int totB = base_off[C];
        // opts->reads_same_as: chains that share their sample's reads.  rep[c] = the first chain OF THIS LAUNCH with chain c's
        // reads (c itself without the option); only the representatives' bases are looked at and uploaded.
        std::vector<int32_t> rep(C);
        std::vector<int64_t> hbase(C);   // where a chain's bases are looked at on the host, relative to this launch's u / bq
        bool aliased = false;
        {
            // keyed on the representative's INDEX in the call (reads_same_as[c]), not on where its bases lie: a chain without
            // bases has the offset of the chain behind it and would be taken for sharing that chain's reads
            std::map<int32_t, int32_t> first_of;
            for (int c = 0; c < C; c++) {
                rep[c] = c;
                hbase[c] = rep_off ? rep_off[c] : (int64_t)base_off[c];
                if (!rep_off) continue;
                auto it = first_of.find(rep_id[c]);
                if (it == first_of.end()) { first_of.emplace(rep_id[c], c); aliased |= rep_off[c] != (int64_t)base_off[c]; continue; }
                const int r0 = it->second;
                const int R = read_off[c + 1] - read_off[c];
                // the same reads means the same read boundaries, not just as many reads and bases
                if (read_off[r0 + 1] - read_off[r0] != R || base_off[r0 + 1] - base_off[r0] != base_off[c + 1] - base_off[c] ||
                    std::memcmp(read_ptr + read_off[r0] + r0, read_ptr + read_off[c] + c, sizeof(int32_t) * ((size_t)R + 1)) != 0)
                    throw std::runtime_error("reads_same_as names a chain with other reads");
                rep[c] = r0;
                aliased = true;
            }
        }
        bq_eff.resize((size_t)std::max(totB, 1));
        std::vector<int32_t> n_dense(C, 0);
        {
            // validation, grid_has_read, 0-based haplotypes, the bq == 0 carry-over (fold_zero_base_qualities) and the
            // choice of the reads that keep a dense column (more informative bases -- k_ematread skips bq == 0 -- than the
            // pattern width): independent per chain, spread over host threads
            const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), C));
            std::vector<std::string> errs(n_thr);
            int pass = 0;   // 0: the representatives (they fold their base qualities in place); 1: the chains that share a representative's
            auto work = [&](int tid) {
                try {
                    for (int c = tid; c < C; c += n_thr) {
                        if ((rep[c] == c) != (pass == 0)) continue;
                        const int R = read_off[c + 1] - read_off[c];
                        const int32_t *rp = read_ptr + read_off[c] + c;
                        const int64_t hb = hbase[c];          // where this chain's bases are looked at (u, bq)
                        const size_t fb = (size_t)base_off[rep[c]];   // the representative's block of bq_eff
                        if (pass == 0) std::memcpy(&bq_eff[fb], bq + hb, sizeof(int32_t) * (size_t)(base_off[c + 1] - base_off[c]));
                        for (int r = 0; r < R; r++) {
                            const int g = wif[read_off[c] + r];
                            if (g < 0 || g >= G) throw std::runtime_error("read grid index out of range");
                            if (r > 0 && g < wif[read_off[c] + r - 1]) throw std::runtime_error("reads must be sorted by grid");
                            ghr[(size_t)c * G + g] = 1;
                        }
                        for (int k = 0; k < Ks; k++) {
                            const int v = which_haps_to_use_1based[(size_t)c * Ks + k] - 1;
                            if (v < 0 || v >= pn->K) throw std::runtime_error("which_haps_to_use out of range");
                            which0[(size_t)c * Ks + k] = v;
                            if (rc)   // rare SNPs some selected haplotype carries the alt of (rare_per_snp_info, rare_common.R:313-322)
                                for (int64_t i = rc->h_rare_ptr[v]; i < rc->h_rare_ptr[v + 1]; i++) {
                                    const int t = rc->h_rare_snp[i];
                                    rc_any[(size_t)c * rc_words + (t >> 5)] |= 1u << (t & 31);
                                }
                        }
                        const int32_t *cu = u + hb;
                        int last = 0, nd = 0;
                        for (int r = 0; r < R; r++) {
                            int J = rp[r + 1] - rp[r] - 1;
                            if (J >= o->Jmax) J = o->Jmax;
                            int n_inf = 0;
                            for (int j = 0; j <= J; j++) {
                                int32_t b = bq_eff[fb + rp[r] + j];
                                if (b == 0) b = last; else last = b;
                                if (pass == 0) bq_eff[fb + rp[r] + j] = b;   // (a sharing chain reads the folded values: folding them again changes nothing)
                                if (b > 255 || b < -255) throw std::runtime_error("|base quality| > 255");
                                const int t = cu[rp[r] + j];
                                if (t < 0 || t >= T) throw std::runtime_error("read SNP index out of range");
                                // a rare SNP nobody selected carries is a common factor: no pattern bit (k_ematread)
                                const bool informative = !rc || rc->h_common_index[t] >= 0 ||
                                                         ((rc_any[(size_t)c * rc_words + (t >> 5)] >> (t & 31)) & 1u);
                                n_inf += (b != 0) && informative;
                            }
                            if (n_inf > kMaxPatternBits) dense_of[read_off[c] + r] = nd++;
                        }
                        n_dense[c] = nd;
                    }
                } catch (const std::exception &e) {
                    errs[tid] = e.what();
                }
            };
            for (pass = 0; pass < (aliased ? 2 : 1); pass++) {
                std::vector<std::thread> th;
                for (int i = 1; i < n_thr; i++) th.emplace_back(work, i);
                work(0);
                for (auto &t : th) t.join();
                for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
            }
        }
        for (int c = 0; c < C; c++) {
            eoff[c] = etot;
            etot += (size_t)n_dense[c] * Ksp;
        }
        const double T1 = now();
        const std::vector<double> tabs = base_quality_tables();
        std::vector<double> tm((size_t)2 * std::max(G - 1, 1));
        for (int g = 0; g < G - 1; g++) {
            tm[g] = rc ? rc->h_sigma[g] : pn->h_sigma[g];
            tm[(size_t)G - 1 + g] = rc ? rc->h_tm1[g] : pn->h_tm1[g];
        }

        S.which.ensure(which0.size()); S.which.upload(which0.data(), which0.size(), st);
        S.read_off.ensure(C + 1); S.read_off.upload(read_off, C + 1, st);
        S.read_ptr.ensure(totR + C); S.read_ptr.upload(read_ptr, totR + C, st);
        if (!aliased) {
            S.base_off.ensure(C + 1); S.base_off.upload(base_off.data(), C + 1, st);
            S.u.ensure(std::max(totB, 1)); S.u.upload(u, totB, st);
            S.bq.ensure(std::max(totB, 1)); S.bq.upload(bq_eff.data(), totB, st);
        } else {
            // the representatives' bases back to back; every chain's device offset is its representative's (the kernels take
            // base_off[c] as the start of the chain's bases and nothing else)
            std::vector<int32_t> dev_off(C + 1, 0), cu_h, cb_h;
            int at = 0;
            for (int c = 0; c < C; c++)
                if (rep[c] == c) { dev_off[c] = at; at += base_off[c + 1] - base_off[c]; }
            for (int c = 0; c < C; c++) dev_off[c] = dev_off[rep[c]];
            dev_off[C] = at;
            cu_h.resize((size_t)std::max(at, 1)); cb_h.resize((size_t)std::max(at, 1));
            for (int c = 0; c < C; c++)
                if (rep[c] == c) {
                    const size_t nb_c = (size_t)(base_off[c + 1] - base_off[c]);
                    std::memcpy(&cu_h[(size_t)dev_off[c]], u + hbase[c], sizeof(int32_t) * nb_c);
                    std::memcpy(&cb_h[(size_t)dev_off[c]], &bq_eff[(size_t)base_off[c]], sizeof(int32_t) * nb_c);
                }
            S.base_off.ensure(C + 1); S.base_off.upload(dev_off.data(), C + 1, st);
            S.u.ensure(std::max(at, 1)); S.u.upload(cu_h.data(), at, st);
            S.bq.ensure(std::max(at, 1)); S.bq.upload(cb_h.data(), at, st);
        }
        S.wif.ensure(std::max(totR, 1)); S.wif.upload(wif, totR, st);
        S.ghr.ensure(ghr.size()); S.ghr.upload(ghr.data(), ghr.size(), st);
        S.tabs.ensure(tabs.size()); S.tabs.upload(tabs.data(), tabs.size(), st);
        S.tm.ensure(tm.size()); S.tm.upload(tm.data(), tm.size(), st);
        S.block_its.ensure(std::max(o->n_block_gibbs_iterations, 1));
        S.block_its.upload(o->block_gibbs_iterations, o->n_block_gibbs_iterations, st);
        S.first_read.ensure(C); S.first_read.upload(first_read, C, st);
        if (!seed_reads) {
            S.runif_reads.ensure(std::max<size_t>((size_t)totR * n_its, 1));
            S.runif_reads.upload(runif_reads, (size_t)totR * n_its, st);
        } else {
            S.runif_reads.ensure(1);
            S.seeds.ensure((size_t)2 * C);
            S.seeds.upload(seed_reads, C, st);
            qa::staged_upload(S.seeds.p + C, seed_shard ? seed_shard : seed_reads, sizeof(uint64_t) * C, st);
        }
        // diploid: the shard passes' uniforms; NIPT: the block passes' (include/quilt_amd.h)
        const bool nipt = o->ff != 0.0;
        const size_t nshard_used = nipt ? (size_t)totR * o->n_block_gibbs_iterations * 2
                                        : (size_t)C * o->n_block_gibbs_iterations * (G - 1);
        S.runif_shard.ensure(std::max<size_t>(nshard_used, 1));
        // (read only when a pass will draw from it: a caller that switches the passes off may hand over anything -- an array
        // shorter than the passes would need was read past its end here until round 6: found by AddressSanitizer on the host code)
        if (runif_shard && !seed_shard && o->n_block_gibbs_iterations > 0 && o->perform_block_gibbs && (nipt || o->do_shard_block_gibbs))
            S.runif_shard.upload(runif_shard, nshard_used, st);
        S.eread_off.ensure(C); S.eread_off.upload(eoff.data(), C, st);
        S.eridx_off.ensure(C); S.eridx_off.upload(ixoff.data(), C, st);
        S.dense_of.ensure(dense_of.size()); S.dense_of.upload(dense_of.data(), dense_of.size(), st);
        S.is_cat1.ensure(std::max(totR, 1));
        S.er_nent.ensure(std::max(totR, 1));
        const int nH = o->ff != 0.0 ? 3 : 2;
        const size_t mat = (size_t)C * nH * G * Ksp;
        S.H.ensure(std::max(totR, 1)); S.H.upload(H, totR, st);
        S.H_class.ensure(std::max(totR, 1));
        S.status.ensure(C);
        if (o->per_it_out) S.per_it.ensure((size_t)C * n_its * 8);
        if (ff_chain) { S.ff_chain.ensure(C); S.ff_chain.upload(ff_chain, C, st); }
        if (rc) { S.rc_any.ensure(std::max<size_t>(rc_any.size(), 1)); S.rc_any.upload(rc_any.data(), rc_any.size(), st); }
        // (SNP, row) pairs per chain for k_happrobs_rc (only when the probabilities are asked for)
        const bool want_pairs = rc && (hapProbs_t || genProbsM_t || genProbsF_t || o->hap_words_out || o->hap_major_out) && Ks <= 1024;
        if (want_pairs) {
            std::vector<size_t> pbase((size_t)C + 1, 0);
            std::vector<std::vector<uint32_t>> keys((size_t)C);   // SNP << 10 | row, sorted
            {
                const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), C));
                auto work = [&](int tid) {
                    for (int c = tid; c < C; c += n_thr) {
                        auto &kv = keys[(size_t)c];
                        for (int k = 0; k < Ks; k++) {
                            const int v = which0[(size_t)c * Ks + k];
                            for (int64_t i = rc->h_rare_ptr[v]; i < rc->h_rare_ptr[v + 1]; i++) kv.push_back(((uint32_t)rc->h_rare_snp[i] << 10) | (uint32_t)k);
                        }
                        std::sort(kv.begin(), kv.end());
                    }
                };
                std::vector<std::thread> th;
                for (int i = 1; i < n_thr; i++) th.emplace_back(work, i);
                work(0);
                for (auto &t : th) t.join();
            }
            for (int c = 0; c < C; c++) pbase[(size_t)c + 1] = pbase[(size_t)c] + keys[(size_t)c].size();
            std::vector<uint16_t> pairs(std::max<size_t>(pbase[(size_t)C], 1));
            std::vector<int32_t> poff((size_t)C * (G + 1), 0);
            for (int c = 0; c < C; c++) {
                const auto &kv = keys[(size_t)c];
                int32_t *po = poff.data() + (size_t)c * (G + 1);
                size_t i = 0;
                for (int g = 0; g < G; g++) {
                    po[g] = (int32_t)i;
                    while (i < kv.size() && (int)((kv[i] >> 10) >> 5) == g) {
                        pairs[pbase[(size_t)c] + i] = (uint16_t)((((kv[i] >> 10) & 31u) << 10) | (kv[i] & 1023u));
                        i++;
                    }
                }
                po[G] = (int32_t)i;
            }
            S.rc_pairs.ensure(pairs.size()); S.rc_pairs.upload(pairs.data(), pairs.size(), st);
            S.rc_pair_off.ensure(poff.size()); S.rc_pair_off.upload(poff.data(), poff.size(), st);
            S.rc_pair_base.ensure((size_t)C); S.rc_pair_base.upload(pbase.data(), (size_t)C, st);
        }
        // ---- everything above is host work and uploads into this thread's own buffers; from here on the launch set has the
        // device (exclusive phases: queue behind the other handles' launch sets) and the arena
        const double hold_t0 = now();
        qa::GateHold hold;
        hold.acquire(pn->gate(), &pn->arena, C * nw, arena_need);
        const double T1g = now();
        qa::Arena &arena = hold.arena();
        S.er_idx.arena = S.er_tab.arena = &arena;
        S.eMatRead.arena = S.alpha.arena = S.beta.arena = S.eg.arena = S.cvec.arena = S.hap.arena = S.gm.arena = S.gf.arena = &arena;
        arena.require(arena_need);
        arena.reset();
        S.eMatRead.ensure(std::max<size_t>(etot, 1));
        S.er_idx.ensure(std::max<size_t>(ixtot, 1));
        S.er_tab.ensure(std::max<size_t>((size_t)totR * 64, 1));
        S.alpha.ensure(mat); S.beta.ensure(mat); S.eg.ensure(mat);
        S.cvec.ensure((size_t)C * 3 * G);
        if (hapProbs_t || genProbsM_t || genProbsF_t || o->hap_words_out || o->hap_major_out) {
            S.hap.ensure((size_t)C * T * 3); S.gm.ensure((size_t)C * T * 3); S.gf.ensure((size_t)C * T * 3);
        }

        GibbsParams prm{};
        prm.hm = pn->hm.p; prm.B = pn->B.p; prm.sp_off = pn->sp_off.p; prm.sp_k = pn->sp_k.p;
        prm.sp_word = pn->sp_word.p; prm.sigma = S.tm.p; prm.Kp = pn->Kp; prm.G = G; prm.T = T;
        prm.nMaxDH = pn->nMaxDH; prm.ref_error = pn->ref_error;
        prm.C = C; prm.Ks = Ks; prm.Ksp = Ksp; prm.NE = NE; prm.which = S.which.p;
        prm.read_off = S.read_off.p; prm.read_ptr = S.read_ptr.p; prm.base_off = S.base_off.p;
        prm.u = S.u.p; prm.bq = S.bq.p; prm.wif = S.wif.p; prm.grid_has_read = S.ghr.p;
        prm.pR_tab = S.tabs.p; prm.pA_tab = S.tabs.p + 512;
        prm.Jmax = o->Jmax; prm.inv_maxdiff = 1 / o->maxDifferenceBetweenReads; prm.rescale = o->rescale_eMatRead_t;
        prm.n_its = n_its; prm.n_burn_in = o->n_gibbs_burn_in_its; prm.block_its = S.block_its.p;
        prm.n_block = o->perform_block_gibbs ? o->n_block_gibbs_iterations : 0;
        prm.do_shard = o->do_shard_block_gibbs; prm.init_iteratively = o->gibbs_initialize_iteratively;
        prm.disable_read_category_usage = o->disable_read_category_usage;
        prm.class_sum_cutoff = o->class_sum_cutoff;
        prm.nH = nH;
        prm.it_begin = 0; prm.it_end = n_its;
        prm.ff = o->ff;
        if (o->per_it_out) {
            QA_HIP(hipMemsetAsync(S.per_it.p, 0, sizeof(double) * C * n_its * 8, st));
            prm.per_it = S.per_it.p;
        }
        if (ff_chain) prm.ff_chain = S.ff_chain.p;
        prm.runif_reads = S.runif_reads.p; prm.first_read = S.first_read.p; prm.runif_shard = S.runif_shard.p;
        prm.seed_reads = seed_reads ? S.seeds.p : nullptr;
        prm.seed_shard = (seed_reads && seed_shard) ? S.seeds.p + C : nullptr;
        prm.eMatRead = S.eMatRead.p; prm.eread_off = S.eread_off.p; prm.is_cat1 = S.is_cat1.p; prm.er_nent = S.er_nent.p;
        prm.er_idx = S.er_idx.p; prm.eridx_off = S.eridx_off.p; prm.er_tab = S.er_tab.p; prm.dense_of = S.dense_of.p;
        prm.er_nt = er_nt; prm.er_padb = er_padb;
        prm.alpha = S.alpha.p; prm.beta = S.beta.p; prm.eg = S.eg.p; prm.cvec = S.cvec.p;
        prm.H = S.H.p; prm.H_class = S.H_class.p; prm.status = S.status.p;
        prm.hapProbs = S.hap.p; prm.genProbsM = S.gm.p; prm.genProbsF = S.gf.p;
        if (rc) {
            prm.rc_common = rc->common_index.p; prm.rc_rare_ptr = rc->rare_ptr.p; prm.rc_rare_snp = rc->rare_snp.p;
            prm.rc_any = S.rc_any.p; prm.rc_words = rc_words; prm.rc_Gc = pn->G;
            if (want_pairs) { prm.rc_pairs = S.rc_pairs.p; prm.rc_pair_off = S.rc_pair_off.p; prm.rc_pair_base = S.rc_pair_base.p; }
        }

        for (auto &e : g_gibbs->ev) if (!e) QA_HIP(hipEventCreate(&e));
        const bool want_probs = hapProbs_t || genProbsM_t || genProbsF_t || o->hap_words_out || o->hap_major_out;
        QA_HIP(hipStreamSynchronize(st));
        const double T2 = now();
        // NIPT: the sweeps are cut at the block-Gibbs iterations; between two segments the switch rate per grid boundary
        // comes back to the host, which defines the blocks (scalar logic per chain, gibbs_blocks.hpp) for the block kernel
        auto nipt_sweeps = [&]() {
            GibbsParams q = prm;
            q.ff = o->ff;
            // The three-label sampler's 256-register build (two chains per SIMD, gibbs3.hip) is built and tested but NOT chosen by
            // default: alone on the device 1 792 chains take 2.33-2.58 s in it against 2 x 1.24 s for two launches of 896 at one
            // chain per SIMD -- a sweep's grid steps are HBM-bound either way (5.6 TB/s) and the read visits, which two waves per
            // SIMD should overlap, pay for eMatGrid's trips through LDS and ~600 bytes of scratch per lane.
            // QA_GIBBS3_LEAN=1 selects it (tests, measurements).
            const char *lean3_env = getenv("QA_GIBBS3_LEAN");   // (read per call: the tests switch it inside one process)
            const bool lean3_on = lean3_env && atoi(lean3_env) != 0;
            q.lean3 = (Ksp == 640 && nw == 1 && lean3_on) ? 1 : 0;
            q.blk_n_pass = std::max(o->n_block_gibbs_iterations, 1);
            std::vector<int> passes;
            if (o->perform_block_gibbs)
                for (int i = 0; i < o->n_block_gibbs_iterations; i++) {
                    const int b = o->block_gibbs_iterations[i];
                    if (b >= 0 && b < n_its && std::find(passes.begin(), passes.end(), b) == passes.end()) passes.push_back(b);
                }
            std::sort(passes.begin(), passes.end());
            if (!passes.empty()) {
                S.blk_rate2.ensure((size_t)C * G); S.blk_where.ensure((size_t)C * G);
                S.blk_tab.ensure((size_t)C * 4 * G); S.blk_n.ensure(C);
                q.blk_rate2 = S.blk_rate2.p; q.blk_where = S.blk_where.p; q.blk_tab = S.blk_tab.p; q.blk_n = S.blk_n.p;
            }
            std::vector<double> rate2;
            std::vector<int32_t> h_where, h_tab, h_n;
            int it0 = 0;
            q.rebuild = 0;
            for (size_t j = 0; j < passes.size(); j++) {
                q.it_begin = it0; q.it_end = passes[j] + 1;
                qa::launch_gibbs3(&q, st);
                q.rebuild = 1;   // (what follows a block pass starts by re-forming the state from its labels)
                qa::launch_block_rate3(&q, st);
                rate2.resize((size_t)C * G);
                S.blk_rate2.download(rate2.data(), rate2.size(), st);
                std::vector<int32_t> seg_status(C);
                S.status.download(seg_status.data(), C, st);
                QA_HIP(hipStreamSynchronize(st));
                const double tg0 = now();
                h_where.assign((size_t)C * G, -1); h_tab.assign((size_t)C * 4 * G, 0); h_n.assign(C, 0);
                const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), C));
                auto work = [&](int tid) {
                    for (int c = tid; c < C; c += n_thr) {
                        const int R = read_off[c + 1] - read_off[c];
                        if (seg_status[c] != 0 || R < 1) continue;   // underflowed chain: stopped, the caller retries it
                        const std::vector<int32_t> blocked = qa::define_blocked_grids(
                            rate2.data() + (size_t)c * G, o->L_grid, G, o->shuffle_bin_radius, o->block_gibbs_quantile_prob);
                        const qa::BlockTable T = qa::make_gibbs_considers(blocked, wif + read_off[c], R);
                        h_n[c] = T.n_blocks;
                        std::copy(T.grid_where.begin(), T.grid_where.end(), h_where.begin() + (size_t)c * G);
                        int32_t *tb = h_tab.data() + (size_t)c * 4 * G;
                        for (int b = 0; b < T.n_blocks; b++) {
                            tb[b] = T.grid_start[b]; tb[G + b] = T.grid_end[b];
                            tb[2 * G + b] = T.reads_start[b]; tb[3 * G + b] = T.reads_end[b];
                        }
                    }
                };
                std::vector<std::thread> th;
                for (int i = 1; i < n_thr; i++) th.emplace_back(work, i);
                work(0);
                for (auto &t : th) t.join();
                const double tg1 = now();
                S.blk_where.upload(h_where.data(), h_where.size(), st);
                S.blk_tab.upload(h_tab.data(), h_tab.size(), st);
                S.blk_n.upload(h_n.data(), h_n.size(), st);
                q.blk_pass = (int)j;
                // opts->draw_uniforms: the pass's uniforms come from the caller WHEN THE REFERENCE DRAWS THEM -- runif_block now
                // (gibbs-nipt.cpp:3016), the per-read re-draws after the relabelling, one for every read whose class leaves a choice, in
                // read order (rcpp_sample_H_using_H_class, gibbs-nipt-block.cpp:213-246): an R caller's stream then stays in step
                // with the CPU package's.  Chain by chain, from this (the calling) thread.
                const bool draw_cb = o->draw_uniforms != nullptr;
                q.defer_resample = draw_cb ? 1 : 0;
                std::vector<double> ub;
                if (draw_cb) {
                    for (int c = 0; c < C; c++) {
                        const int R = read_off[c + 1] - read_off[c];
                        if (R < 1) continue;
                        ub.assign((size_t)R, 0.0);
                        o->draw_uniforms(o->draw_uniforms_ctx, per_it_off + c, (int)j, 0, R, ub.data());
                        qa::staged_upload(S.runif_shard.p + ((size_t)read_off[c] * q.blk_n_pass + (size_t)j * R) * 2, ub.data(), sizeof(double) * (size_t)R, st);
                    }
                }
                qa::launch_block3(&q, st);
                if (draw_cb) {
                    std::vector<int32_t> hc((size_t)std::max(totR, 1));
                    S.H_class.download(hc.data(), totR, st);
                    QA_HIP(hipStreamSynchronize(st));
                    std::vector<double> draws;
                    for (int c = 0; c < C; c++) {
                        const int R = read_off[c + 1] - read_off[c];
                        if (R < 1 || seg_status[c] != 0) continue;
                        const int32_t *h = hc.data() + read_off[c];
                        int n = 0;
                        for (int r = 0; r < R; r++) n += !(h[r] >= 1 && h[r] <= 3);
                        draws.assign((size_t)std::max(n, 1), 0.0);
                        if (n > 0) o->draw_uniforms(o->draw_uniforms_ctx, per_it_off + c, (int)j, 1, n, draws.data());
                        ub.assign((size_t)R, 0.0);
                        for (int r = 0, k = 0; r < R; r++)
                            if (!(h[r] >= 1 && h[r] <= 3)) ub[(size_t)r] = draws[(size_t)k++];
                        qa::staged_upload(S.runif_shard.p + ((size_t)read_off[c] * q.blk_n_pass + (size_t)j * R) * 2 + R, ub.data(), sizeof(double) * (size_t)R, st);
                    }
                    qa::launch_resample3(&q, st);
                }
                if (tmg) fprintf(stderr, "[qa_gibbs C=%d] block pass %d: device idle for the host's block tables %.1f ms (tables %.1f on %d threads, uploads enqueued %.1f)\n",
                                 C, (int)j, (now() - tg0) * 1e3, (tg1 - tg0) * 1e3, n_thr, (now() - tg1) * 1e3);
                it0 = passes[j] + 1;
            }
            if (passes.empty() || it0 < n_its || q.rebuild) {   // (also with no sweeps left: the rebuild after the last pass)
                q.it_begin = it0; q.it_end = n_its;
                qa::launch_gibbs3(&q, st);
            }
        };
        launch_gibbs(prm, maxR, st, g_gibbs->ev, want_probs, nw, nipt_sweeps);
        S.H.download(H, totR, st);
        if (H_class) S.H_class.download(H_class, totR, st);
        if (o->per_it_out) S.per_it.download(o->per_it_out + (size_t)per_it_off * n_its * 8, (size_t)C * n_its * 8, st);
        std::vector<int32_t> status(C);
        S.status.download(status.data(), C, st);
        QA_HIP(hipStreamSynchronize(st));
        const double T3 = now();
        hold.mark();
        if (hapProbs_t) S.hap.download(hapProbs_t, (size_t)C * T * 3, st);
        if (o->hap_words_out) {   // (the genotype-probability buffers are free by now: the words are staged in S.gm)
            int32_t *d_words = reinterpret_cast<int32_t *>(S.gm.p);
            hipLaunchKernelGGL(k_pack_hap_words, dim3((G + 255) / 256, 3, C), dim3(256), 0, st, S.hap.p, T, G, d_words);
            QA_HIP(hipGetLastError());
            if (genProbsM_t) throw std::runtime_error("hap_words_out cannot be combined with genProbs outputs");
            qa::staged_download(o->hap_words_out + (size_t)per_it_off * 3 * G, d_words, sizeof(int32_t) * (size_t)C * 3 * G, st);
        }
        if (o->hap_major_out) {   // (staged in the free S.gf: label-major rows, then one transfer)
            const int nL = o->hap_major_labels;
            if (genProbsF_t || nL < 1 || nL > 3) throw std::runtime_error("hap_major_out: 1..3 labels, not combined with genProbs outputs");
            hipLaunchKernelGGL(k_hap_major, dim3((T + 255) / 256, C), dim3(256), 0, st, S.hap.p, T, nL, S.gf.p);
            QA_HIP(hipGetLastError());
            qa::staged_download(o->hap_major_out + (size_t)per_it_off * nL * T, S.gf.p, sizeof(double) * (size_t)C * nL * T, st);
        }
        if (genProbsM_t) S.gm.download(genProbsM_t, (size_t)C * T * 3, st);
        if (genProbsF_t) S.gf.download(genProbsF_t, (size_t)C * T * 3, st);
        QA_HIP(hipStreamSynchronize(st));
        if (!(state_out && C == 1)) hold.release();
        {
            float ms[3];
            for (int i = 0; i < 3; i++) QA_HIP(hipEventElapsedTime(&ms[i], g_gibbs->ev[i], g_gibbs->ev[i + 1]));
            // algorithmic bytes: per sweep and label 6 column streams of Ks x G fp64, plus every
            // read's emission column once per sweep; initialisation and each shard pass ~ one sweep without reads
            const double col = 2.0 * Ks * (double)G * 48.0;
            const double sweeps = (double)n_its * (C * col + (double)totR * Ks * 8.0) +
                                  (1.0 + 2.0 * prm.n_block) * C * col;
            const double t_e = qa::profile_clock_ms(g_gibbs->ev[0]);
            // work units: read visits + grid steps over all chains; serial: those of the longest chain (the launch's
            // critical path: sweeps, initial forward / backward, shard or block passes)
            const double passes = (double)n_its + 1.0 + 2.0 * prm.n_block;
            const double units = (double)n_its * totR + passes * (double)C * G;
            const double serial = (double)n_its * maxR + passes * (double)G;
            qa::profile_add(qa::PK_EMATREAD, ms[0], (double)totR * Ks * 8.0, t_e);
            qa::profile_add(nH == 3 ? qa::PK_GIBBS3 : ((Ksp / 64 == 10 && nw == 1 && use_lean_build(C)) ? qa::PK_GIBBS_LEAN : qa::PK_GIBBS), ms[1],
                            sweeps * (nH / 2.0), t_e + ms[0], units, serial, (double)C);
            if (want_probs) qa::profile_add(qa::PK_HAPPROBS, ms[2], C * (double)nH * Ks * (double)G * 16.0, t_e + ms[0] + ms[1]);
            if (tmg)
                fprintf(stderr, "[qa_gibbs C=%d] host prep %.3f s, tables+uploads %.3f s (of which queued for the device %.3f s), kernels %.3f s (events %.3f), downloads %.3f s\n", C,
                        T1 - T0, T2 - T1, T1g - hold_t0, T3 - T2, (ms[0] + ms[1] + ms[2]) / 1e3, now() - T3);
        }
        int ret = QA_OK;
        for (int c = 0; c < C; c++) {
            if (underflow_problem) underflow_problem[c] = status[c];
            if (status[c]) ret = QA_UNDERFLOW;
        }
        if (state_out && C == 1) {
            // debugging / test aid: alpha, beta, eMatGrid of both labels ([6][G][Ks]) then c ([3][G])
            std::vector<double> tmp((size_t)G * Ksp);
            const qa::ABuf<double> *src[3] = {&S.alpha, &S.beta, &S.eg};
            size_t o2 = 0;
            for (int m = 0; m < 3; m++)
                for (int h = 0; h < 2; h++) {
                    QA_HIP(hipMemcpy(tmp.data(), src[m]->p + (size_t)h * G * Ksp, sizeof(double) * tmp.size(),
                                     hipMemcpyDeviceToHost));
                    for (int g = 0; g < G; g++)
                        for (int k = 0; k < Ks; k++) state_out[o2 + (size_t)g * Ks + k] = tmp[(size_t)g * Ksp + k];
                    o2 += (size_t)G * Ks;
                }
            QA_HIP(hipMemcpy(state_out + o2, S.cvec.p, sizeof(double) * 3 * G, hipMemcpyDeviceToHost));
        }
        return ret;
    }
}



extern "C" {

static int gibbs_batch_impl(qa_panel_t *pn, const qa_rare_common *rc, const qa_gibbs_opts_t *o, int32_t n_chain,
                   const int32_t *which_haps_to_use_1based,
                   const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                   const int32_t *wif, const double *runif_reads, const int32_t *first_read,
                   const double *runif_shard, int32_t *H, int32_t *H_class, double *hapProbs_t,
                   double *genProbsM_t, double *genProbsF_t, int32_t *underflow_problem, double *state_out,
                   const uint64_t *seed_reads, const uint64_t *seed_shard) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!pn || !o || n_chain <= 0 || !which_haps_to_use_1based || !read_off || !read_ptr || !u || !bq || !wif ||
        (!runif_reads && !seed_reads) || !first_read || !H) {
        qa::set_error("qa_gibbs_batch: null argument");
        return QA_ERR_INVALID;
    }
    if ((o->ff != 0.0) != (o->sample_is_diploid == 0) || o->ff < 0 || o->ff >= 1) {
        qa::set_error("qa_gibbs_batch: ff > 0 goes with sample_is_diploid = 0 (NIPT), ff = 0 with sample_is_diploid = 1");
        return QA_ERR_INVALID;
    }
    if (o->ff != 0.0 && o->perform_block_gibbs && o->n_block_gibbs_iterations > 0 &&
        (!o->L_grid || o->shuffle_bin_radius <= 0 || !(o->block_gibbs_quantile_prob > 0 && o->block_gibbs_quantile_prob < 1) ||
         (!runif_shard && !seed_shard && !seed_reads && !o->draw_uniforms))) {
        qa::set_error("qa_gibbs_batch: the NIPT block Gibbs needs opts->L_grid, shuffle_bin_radius, block_gibbs_quantile_prob "
                      "and the block passes' uniforms (runif_shard, seeds, or opts->draw_uniforms)");
        return QA_ERR_INVALID;
    }
    if (o->draw_uniforms && (o->ff == 0.0 || seed_reads || seed_shard)) {
        qa::set_error("qa_gibbs_batch: opts->draw_uniforms serves the NIPT block passes (ff > 0) of a call with explicit uniforms "
                      "(runif_reads given, no seeds)");
        return QA_ERR_INVALID;
    }
    // the uniforms of the shard passes (diploid) / block passes (NIPT): explicit, or the per-chain seed of the counter-based
    // stream -- never uninitialised device memory
    if (seed_shard && !seed_reads) {
        qa::set_error("qa_gibbs_batch: seed_shard goes with seed_reads (the two counter-based streams of a chain)");
        return QA_ERR_INVALID;
    }
    {
        const bool passes = o->perform_block_gibbs && o->n_block_gibbs_iterations > 0 && (o->ff != 0.0 || o->do_shard_block_gibbs);
        if (passes && !runif_shard && !(seed_reads && seed_shard) && !(o->ff != 0.0 && o->draw_uniforms)) {
            qa::set_error("qa_gibbs_batch: block / shard passes requested without their uniforms (runif_shard, or seed_reads and "
                          "seed_shard)");
            return QA_ERR_INVALID;
        }
    }
    if (o->ff_chain) {
        bool ok = o->ff != 0.0;
        for (int c = 0; c < n_chain && ok; c++) ok = o->ff_chain[c] > 0 && o->ff_chain[c] < 1;
        if (!ok) {
            qa::set_error("qa_gibbs_batch: ff_chain goes with ff > 0 (NIPT) and holds fetal fractions in (0, 1)");
            return QA_ERR_INVALID;
        }
    }
    if (o->reads_same_as)
        for (int c = 0; c < n_chain; c++) {
            const int r0 = o->reads_same_as[c];
            if (r0 < 0 || r0 > c || o->reads_same_as[r0] != r0 || read_off[r0 + 1] - read_off[r0] != read_off[c + 1] - read_off[c]) {
                qa::set_error("qa_gibbs_batch: reads_same_as[%d] must name an earlier chain (or the chain itself) that names itself and has as many reads", c);
                return QA_ERR_INVALID;
            }
        }
    if (o->Ks <= 0 || o->Ks > 1024) {
        qa::set_error("qa_gibbs_batch: Ksubset = %d outside 1..1024", o->Ks);
        return QA_ERR_UNSUPPORTED;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(pn->device));
        const int G = rc ? rc->G_all : pn->G, T = rc ? rc->T_all : pn->T, Ks = o->Ks, Ksp = (Ks + 63) / 64 * 64;
        const int n_its = o->n_gibbs_burn_in_its + o->n_gibbs_sample_its;
        const int nb = o->n_block_gibbs_iterations;
        const bool want_probs = hapProbs_t || genProbsM_t || genProbsF_t || o->hap_words_out || o->hap_major_out;
        // chains are processed in chunks that fit the device arena (read emissions + 6 Ks x G state matrices each)
        std::vector<size_t> base_of(n_chain + 1, 0);
        for (int c = 0; c < n_chain; c++) {
            const int R = read_off[c + 1] - read_off[c];
            base_of[c + 1] = base_of[c] + (size_t)(read_ptr + read_off[c] + c)[R];
        }
        const size_t budget = pn->plan_budget();
        // device bytes per chain: read emissions -- pattern bytes + table (512 B) per read, a dense Ks-column for the reads
        // with more bases than the pattern width (an upper bound of those that end up dense) -- and the state matrices.
        // Pattern bytes per read: 64 lanes x padb_of(rows per lane) with one wave, 128 x 8 with two (the geometries the
        // launcher picks: <= 1 024 B); the many-wave geometries of the QA_GIBBS_NW test hook take up to 2 560 B
        const size_t pat_bytes = getenv("QA_GIBBS_NW") ? 2560 : 1024;
        std::vector<size_t> adds(n_chain);
        // Rare + common form: a read is dense only when it covers more than kMaxPatternBits INFORMATIVE SNPs -- common SNPs and
        // rare SNPs some selected haplotype carries (gibbs_chunk below, k_ematread) -- which most all-SNP reads do not, although
        // they cover three times the bases.  Counting bases here (the bound the common-SNP form uses) priced a chain at ~290 MB
        // instead of ~230 MB and cut every all-SNP launch of 896 chains into two of 448, each at a chain's full serial time
        // (round 5: 20 of the QUILT2-default bench's 44 launches).  So count as gibbs_chunk does, chains over host threads.
        std::vector<size_t> n_long_of(n_chain, 0);
        {
            const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), n_chain));
            const int rc_words = rc ? (T + 31) / 32 : 0;
            std::vector<std::string> errs(n_thr);
            auto work = [&](int tid) {
                try {
                    std::vector<uint32_t> any(rc_words);
                    for (int c = tid; c < n_chain; c += n_thr) {
                        const size_t R = read_off[c + 1] - read_off[c];
                        const int32_t *rp = read_ptr + read_off[c] + c;
                        size_t n_long = 0;
                        if (!rc) {
                            for (size_t r = 0; r < R; r++) n_long += std::min(rp[r + 1] - rp[r], o->Jmax + 1) > kMaxPatternBits;
                        } else {
                            std::fill(any.begin(), any.end(), 0u);
                            for (int k = 0; k < Ks; k++) {
                                const int v = which_haps_to_use_1based[(size_t)c * Ks + k] - 1;
                                if (v < 0 || v >= pn->K) throw std::runtime_error("which_haps_to_use out of range");
                                for (int64_t i = rc->h_rare_ptr[v]; i < rc->h_rare_ptr[v + 1]; i++) {
                                    const int t = rc->h_rare_snp[i];
                                    any[t >> 5] |= 1u << (t & 31);
                                }
                            }
                            const int32_t *cu = u + base_of[o->reads_same_as ? o->reads_same_as[c] : c];
                            for (size_t r = 0; r < R; r++) {
                                const int nb_r = std::min(rp[r + 1] - rp[r], o->Jmax + 1);
                                int n_inf = 0;
                                for (int j = 0; j < nb_r; j++) {   // (an upper bound of gibbs_chunk's count: it also drops bases of quality 0)
                                    const int t = cu[rp[r] + j];
                                    if (t < 0 || t >= T) throw std::runtime_error("read SNP index out of range");
                                    n_inf += rc->h_common_index[t] >= 0 || ((any[t >> 5] >> (t & 31)) & 1u);
                                }
                                n_long += n_inf > kMaxPatternBits;
                            }
                        }
                        n_long_of[c] = n_long;
                    }
                } catch (const std::exception &e) {
                    errs[tid] = e.what();
                }
            };
            std::vector<std::thread> th;
            for (int i = 1; i < n_thr; i++) th.emplace_back(work, i);
            work(0);
            for (auto &t : th) t.join();
            for (const auto &e : errs)
                if (!e.empty()) throw std::runtime_error(e);
        }
        for (int c = 0; c < n_chain; c++) {
            const size_t R = read_off[c + 1] - read_off[c];
            adds[c] = R * (pat_bytes + 512) + n_long_of[c] * Ksp * 8 + (size_t)(o->ff != 0.0 ? 9 : 6) * G * Ksp * 8 + (size_t)3 * G * 8 +
                      (want_probs ? (size_t)9 * T * 8 : 0) + 8192;
        }
        // A launch costs a chain's serial latency whatever it carries, so when the chains do not fit one launch they are cut
        // into EQUAL launches (1 000 + 24 would cost two full launches for 1 024 chains; 512 + 512 costs the same two but
        // leaves both at one chain per SIMD)
        int n_launch = 1;
        {
            size_t need = (size_t)1 << 20;
            for (int c = 0, first = 1; c < n_chain; c++, first = 0) {
                if (!first && need + adds[c] > budget) { n_launch++; need = (size_t)1 << 20; }
                need += adds[c];
            }
        }
        const int target = (n_chain + n_launch - 1) / n_launch;
        int c0 = 0, ret = QA_OK;
        while (c0 < n_chain) {
            size_t need = (size_t)1 << 20;
            int c1 = c0;
            while (c1 < n_chain && c1 - c0 < target) {
                if (c1 > c0 && need + adds[c1] > budget) break;
                need += adds[c1];
                c1++;
            }
            std::vector<int32_t> ro(c1 - c0 + 1);
            for (int i = 0; i <= c1 - c0; i++) ro[i] = read_off[c0 + i] - read_off[c0];
            std::vector<int64_t> rep_off;   // opts->reads_same_as: a chain's bases, relative to this launch's u / bq (may lie before them)
            if (o->reads_same_as) {
                rep_off.resize((size_t)(c1 - c0));
                for (int c = c0; c < c1; c++) rep_off[(size_t)(c - c0)] = (int64_t)base_of[(size_t)o->reads_same_as[c]] - (int64_t)base_of[(size_t)c0];
            }
            const int st = gibbs_chunk(
                pn, need, rc, o, o->ff_chain ? o->ff_chain + c0 : nullptr, o->reads_same_as ? rep_off.data() : nullptr, o->reads_same_as ? o->reads_same_as + c0 : nullptr, c0, c1 - c0, which_haps_to_use_1based + (size_t)c0 * Ks, ro.data(), read_ptr + read_off[c0] + c0,
                u + base_of[c0], bq + base_of[c0], wif + read_off[c0],
                runif_reads ? runif_reads + (size_t)read_off[c0] * n_its : nullptr, first_read + c0,
                runif_shard ? runif_shard + (o->ff != 0.0 ? (size_t)read_off[c0] * nb * 2 : (size_t)c0 * nb * (G - 1)) : nullptr,
                H + read_off[c0],
                H_class ? H_class + read_off[c0] : nullptr, hapProbs_t ? hapProbs_t + (size_t)c0 * T * 3 : nullptr,
                genProbsM_t ? genProbsM_t + (size_t)c0 * T * 3 : nullptr, genProbsF_t ? genProbsF_t + (size_t)c0 * T * 3 : nullptr,
                underflow_problem ? underflow_problem + c0 : nullptr, state_out, seed_reads ? seed_reads + c0 : nullptr,
                seed_shard ? seed_shard + c0 : nullptr);
            if (st < 0) return st;
            if (st == QA_UNDERFLOW) ret = QA_UNDERFLOW;
            c0 = c1;
        }
        return ret;
    });
}

int qa_gibbs_batch(qa_panel_t *pn, const qa_gibbs_opts_t *o, int32_t n_chain, const int32_t *which_haps_to_use_1based,
                   const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                   const int32_t *wif, const double *runif_reads, const int32_t *first_read,
                   const double *runif_shard, int32_t *H, int32_t *H_class, double *hapProbs_t,
                   double *genProbsM_t, double *genProbsF_t, int32_t *underflow_problem, double *state_out,
                   const uint64_t *seed_reads, const uint64_t *seed_shard) {
    return gibbs_batch_impl(pn, nullptr, o, n_chain, which_haps_to_use_1based, read_off, read_ptr, u, bq, wif, runif_reads,
                            first_read, runif_shard, H, H_class, hapProbs_t, genProbsM_t, genProbsF_t, underflow_problem,
                            state_out, seed_reads, seed_shard);
}

int qa_gibbs_batch_rare_common(qa_panel_t *pn, const qa_rare_common_t *rc, const qa_gibbs_opts_t *o, int32_t n_chain,
                               const int32_t *which_haps_to_use_1based, const int32_t *read_off, const int32_t *read_ptr,
                               const int32_t *u, const int32_t *bq, const int32_t *wif, const double *runif_reads,
                               const int32_t *first_read, const double *runif_shard, int32_t *H, int32_t *H_class,
                               double *hapProbs_t, double *genProbsM_t, double *genProbsF_t,
                               int32_t *underflow_problem, double *state_out, const uint64_t *seed_reads,
                               const uint64_t *seed_shard) {
    if (!rc || !pn) {
        qa::set_error("qa_gibbs_batch_rare_common: null argument");
        return QA_ERR_INVALID;
    }
    if (rc->K != pn->K || rc->device != pn->device) {
        qa::set_error("qa_gibbs_batch_rare_common: the rare/common handle belongs to another panel or device");
        return QA_ERR_INVALID;
    }
    return gibbs_batch_impl(pn, rc, o, n_chain, which_haps_to_use_1based, read_off, read_ptr, u, bq, wif, runif_reads,
                            first_read, runif_shard, H, H_class, hapProbs_t, genProbsM_t, genProbsF_t, underflow_problem,
                            state_out, seed_reads, seed_shard);
}

int qa_rare_common_create(qa_panel_t *pn, int32_t nSNPs_all, const uint8_t *snp_is_common, const int64_t *rare_ptr,
                          const int32_t *rare_snp_1based, const double *transMatRate_t_all, qa_rare_common_t **out) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!pn || !snp_is_common || !rare_ptr || !transMatRate_t_all || !out || nSNPs_all < pn->T) {
        qa::set_error("qa_rare_common_create: null argument or fewer SNPs than the panel has");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(pn->device));
        std::unique_ptr<qa_rare_common> rc(new qa_rare_common());
        rc->device = pn->device; rc->K = pn->K; rc->T_all = nSNPs_all; rc->G_all = (nSNPs_all + 31) / 32;
        rc->h_common_index.assign(nSNPs_all, -1);
        int n_common = 0;
        for (int t = 0; t < nSNPs_all; t++) if (snp_is_common[t]) rc->h_common_index[t] = n_common++;
        if (n_common != pn->T) throw std::runtime_error("sum(snp_is_common) differs from the panel's number of SNPs");
        const int64_t n_rare = rare_ptr[pn->K];
        if (rare_ptr[0] != 0 || n_rare < 0 || (n_rare > 0 && !rare_snp_1based)) throw std::runtime_error("bad rare_per_hap_info CSR");
        rc->h_rare_ptr.assign(rare_ptr, rare_ptr + pn->K + 1);
        rc->h_rare_snp.resize(n_rare);
        for (int k = 0; k < pn->K; k++) {
            if (rare_ptr[k + 1] < rare_ptr[k]) throw std::runtime_error("bad rare_per_hap_info CSR");
            for (int64_t i = rare_ptr[k]; i < rare_ptr[k + 1]; i++) {
                const int t = rare_snp_1based[i] - 1;
                if (t < 0 || t >= nSNPs_all || snp_is_common[t]) throw std::runtime_error("rare_per_hap_info names a SNP that is not rare");
                if (i > rare_ptr[k] && t <= rc->h_rare_snp[i - 1]) throw std::runtime_error("rare_per_hap_info must ascend within a haplotype");
                rc->h_rare_snp[i] = t;
            }
        }
        const int G = rc->G_all;
        rc->h_sigma.resize(std::max(G - 1, 0)); rc->h_tm1.resize(std::max(G - 1, 0));
        for (int g = 0; g < G - 1; g++) { rc->h_sigma[g] = transMatRate_t_all[2 * (size_t)g]; rc->h_tm1[g] = transMatRate_t_all[2 * (size_t)g + 1]; }
        rc->common_index.alloc(nSNPs_all); rc->common_index.upload(rc->h_common_index.data(), nSNPs_all, pn->stream);
        rc->rare_ptr.alloc(pn->K + 1); rc->rare_ptr.upload(rc->h_rare_ptr.data(), pn->K + 1, pn->stream);
        rc->rare_snp.alloc(std::max<int64_t>(n_rare, 1)); rc->rare_snp.upload(rc->h_rare_snp.data(), n_rare, pn->stream);
        QA_HIP(hipStreamSynchronize(pn->stream));
        *out = rc.release();
        return (int)QA_OK;
    });
}

void qa_rare_common_destroy(qa_rare_common_t *rc) { delete rc; }

int qa_nipt_block_table(const double *rate2, const int32_t *L_grid, int32_t nGrids, int32_t shuffle_bin_radius,
                        double block_gibbs_quantile_prob, const int32_t *wif0, int32_t nReads, int32_t *blocked_grid,
                        int32_t *grid_start, int32_t *grid_end, int32_t *reads_start, int32_t *reads_end,
                        int32_t *grid_where, int32_t *n_blocks) {
    if (!rate2 || !L_grid || nGrids < 2 || !wif0 || nReads < 1 || !blocked_grid || !grid_start || !grid_end || !reads_start ||
        !reads_end || !grid_where || !n_blocks) {
        qa::set_error("qa_nipt_block_table: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        const std::vector<int32_t> blocked = qa::define_blocked_grids(rate2, L_grid, nGrids, shuffle_bin_radius, block_gibbs_quantile_prob);
        const qa::BlockTable T = qa::make_gibbs_considers(blocked, wif0, nReads);
        std::copy(blocked.begin(), blocked.end(), blocked_grid);
        std::copy(T.grid_where.begin(), T.grid_where.end(), grid_where);
        for (int b = 0; b < T.n_blocks; b++) {
            grid_start[b] = T.grid_start[b]; grid_end[b] = T.grid_end[b];
            reads_start[b] = T.reads_start[b]; reads_end[b] = T.reads_end[b];
        }
        *n_blocks = T.n_blocks;
        return (int)QA_OK;
    });
}

static int make_eMatRead_t_impl(qa_panel_t *pn, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                                const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                                double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                                double *eMatRead_t, int hap_major) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!pn || nSNPs <= 0 || n_chain <= 0 || K < 1 || K > 3 || !eHaps || !read_off || !read_ptr || !u || !bq || !eMatRead_t) {
        qa::set_error("qa_rcpp_make_eMatRead_t: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(pn->device));
        hipStream_t st = pn->stream;
        const int C = n_chain, T = nSNPs;
        std::vector<int32_t> base_off(C + 1, 0);
        int maxR = 0;
        for (int c = 0; c < C; c++) {
            const int R = read_off[c + 1] - read_off[c];
            maxR = std::max(maxR, R);
            base_off[c + 1] = base_off[c] + (read_ptr + read_off[c] + c)[R];
        }
        const int totR = read_off[C], totB = base_off[C];
        // validation, the copy of the qualities and the bq == 0 carry-over are per chain: on the call's host threads (the driver's
        // read-confidence call carries every chain of a launch set: 90 M bases, 0.25 s on one thread -- and at the end of a
        // stream nothing runs beside it)
        std::vector<int32_t> bq_eff((size_t)std::max(totB, 1));
        {
            const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), C));
            std::vector<std::string> errs(n_thr);
            auto work = [&](int tid) {
                try {
                    for (int c = tid; c < C; c += n_thr) {
                        const int R = read_off[c + 1] - read_off[c];
                        const int32_t *rp = read_ptr + read_off[c] + c;
                        const size_t b0 = (size_t)base_off[c];
                        for (int i = 0; i < rp[R]; i++) {
                            if (u[b0 + i] < 0 || u[b0 + i] >= T) throw std::runtime_error("read SNP index out of range");
                            bq_eff[b0 + i] = bq[b0 + i];
                        }
                        int last = 0;   // fold_zero_base_qualities for this chain
                        for (int r = 0; r < R; r++) {
                            int J = rp[r + 1] - rp[r] - 1;
                            if (J >= Jmax) J = Jmax;
                            for (int j = 0; j <= J; j++) {
                                int32_t &q = bq_eff[b0 + rp[r] + j];
                                if (q == 0) q = last; else last = q;
                                if (q > 255 || q < -255) throw std::runtime_error("|base quality| > 255");
                            }
                        }
                    }
                } catch (const std::exception &e) {
                    errs[tid] = e.what();
                }
            };
            std::vector<std::thread> th;
            for (int i = 1; i < n_thr; i++) th.emplace_back(work, i);
            work(0);
            for (auto &t : th) t.join();
            for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
        }
        const std::vector<double> tabs = base_quality_tables();
        // carved from the handle's arena (no launch set of this handle is in flight during this call): a call-local
        // hipMalloc / hipFree pair would synchronise the device with the other host threads' launches
        qa::ABuf<double> d_e, d_tabs, d_out;
        qa::ABuf<int32_t> d_ro, d_rp, d_bo, d_u, d_bq;
        {
            auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
            const size_t n_e = (size_t)C * T * K, n_out = std::max<size_t>((size_t)totR * K, 1), n_b = std::max(totB, 1);
            const size_t need = pad(n_e * 8) + pad(tabs.size() * 8) + pad(n_out * 8) + 2 * pad((size_t)(C + 1) * 4) +
                                pad((size_t)(totR + C) * 4) + 2 * pad(n_b * 4) + 4096;
            pn->arena.require(need);
            pn->arena.reset();
            for (auto *b : {&d_e, &d_tabs, &d_out}) b->arena = &pn->arena;
            for (auto *b : {&d_ro, &d_rp, &d_bo, &d_u, &d_bq}) b->arena = &pn->arena;
            d_e.ensure(n_e); d_tabs.ensure(tabs.size()); d_out.ensure(n_out);
            d_ro.ensure(C + 1); d_rp.ensure(totR + C); d_bo.ensure(C + 1); d_u.ensure(n_b); d_bq.ensure(n_b);
        }
        d_e.upload(eHaps, (size_t)C * T * K, st); d_tabs.upload(tabs.data(), tabs.size(), st);
        d_ro.upload(read_off, C + 1, st); d_rp.upload(read_ptr, totR + C, st); d_bo.upload(base_off.data(), C + 1, st);
        d_u.upload(u, totB, st); d_bq.upload(bq_eff.data(), totB, st);
        DenseParams prm{};
        prm.C = C; prm.K = K; prm.T = T; prm.Jmax = Jmax; prm.rescale = rescale_eMatRead_t; prm.hap_major = hap_major;
        prm.inv_maxdiff = 1 / maxDifferenceBetweenReads; prm.eHaps = d_e.p; prm.read_off = d_ro.p;
        prm.read_ptr = d_rp.p; prm.base_off = d_bo.p; prm.u = d_u.p; prm.bq = d_bq.p;
        prm.pR_tab = d_tabs.p; prm.pA_tab = d_tabs.p + 512; prm.out = d_out.p;
        hipLaunchKernelGGL(k_ematread_dense, dim3((maxR + 63) / 64, C), dim3(64), 0, st, prm);
        QA_HIP(hipGetLastError());
        d_out.download(eMatRead_t, (size_t)totR * K, st);
        QA_HIP(hipStreamSynchronize(st));
        return QA_OK;
    });
}

// get_initial_read_labels' read likelihoods (rare_common.R:61-107) for many chains at once WITHOUT spreading the haplotypes over
// all SNPs on the host: hap_common [n_chain][K][T_common] (hap-major, the last seek iteration's haploid dosages as the driver
// holds them), 0.5 at the rare SNPs supplied by the kernel through the all-SNP -> common index of the qa_rare_common_t; the
// all-SNP reads once per SAMPLE (n_sample of them, flattened as everywhere) with chain_sample[c] naming a chain's sample.
// Output rows chain after chain, [reads of the chain's sample][K].  Same numbers as qa_rcpp_make_eMatRead_t_nsnps on the expanded
// haplotypes (same products in the same order); what it saves per launch set of 896 chains at 192 000 SNPs: 2.75 GB of
// host-side expansion and staged upload, and six of seven copies of the reads (2.8 GB).
int qa_rcpp_make_eMatRead_t_rare_common(qa_panel_t *pn, const qa_rare_common_t *rc, int32_t n_chain, int32_t n_sample,
                                        const int32_t *chain_sample, int32_t K, const double *hap_common, const int32_t *read_off,
                                        const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                                        double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                                        double *eMatRead_t) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!pn || !rc || n_chain <= 0 || n_sample <= 0 || !chain_sample || K < 1 || K > 3 || !hap_common || !read_off || !read_ptr || !u ||
        !bq || !eMatRead_t || rc->K != pn->K) {
        qa::set_error("qa_rcpp_make_eMatRead_t_rare_common: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(pn->device));
        hipStream_t st = pn->stream;
        const int C = n_chain, NS = n_sample, T = rc->T_all, Tc = pn->T;
        std::vector<int32_t> base_off(NS + 1, 0), out_off(C + 1, 0);
        int maxR = 0;
        for (int i = 0; i < NS; i++) {
            const int R = read_off[i + 1] - read_off[i];
            maxR = std::max(maxR, R);
            base_off[i + 1] = base_off[i] + (read_ptr + read_off[i] + i)[R];
        }
        for (int c = 0; c < C; c++) {
            if (chain_sample[c] < 0 || chain_sample[c] >= NS) throw std::runtime_error("chain_sample out of range");
            out_off[c + 1] = out_off[c] + (read_off[chain_sample[c] + 1] - read_off[chain_sample[c]]);
        }
        const int totR = read_off[NS], totB = base_off[NS], totOut = out_off[C];
        std::vector<int32_t> bq_eff((size_t)std::max(totB, 1));
        {
            const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), NS));
            std::vector<std::string> errs(n_thr);
            auto work = [&](int tid) {
                try {
                    for (int i = tid; i < NS; i += n_thr) {
                        const int R = read_off[i + 1] - read_off[i];
                        const int32_t *rp = read_ptr + read_off[i] + i;
                        const size_t b0 = (size_t)base_off[i];
                        for (int q = 0; q < rp[R]; q++) {
                            if (u[b0 + q] < 0 || u[b0 + q] >= T) throw std::runtime_error("read SNP index out of range");
                            bq_eff[b0 + q] = bq[b0 + q];
                        }
                        int last = 0;   // fold_zero_base_qualities, per sample (a chain's reads are its sample's)
                        for (int r = 0; r < R; r++) {
                            int J = rp[r + 1] - rp[r] - 1;
                            if (J >= Jmax) J = Jmax;
                            for (int j = 0; j <= J; j++) {
                                int32_t &q = bq_eff[b0 + rp[r] + j];
                                if (q == 0) q = last; else last = q;
                                if (q > 255 || q < -255) throw std::runtime_error("|base quality| > 255");
                            }
                        }
                    }
                } catch (const std::exception &e) {
                    errs[tid] = e.what();
                }
            };
            std::vector<std::thread> th;
            for (int i = 1; i < n_thr; i++) th.emplace_back(work, i);
            work(0);
            for (auto &t : th) t.join();
            for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
        }
        const std::vector<double> tabs = base_quality_tables();
        qa::ABuf<double> d_e, d_tabs, d_out;
        qa::ABuf<int32_t> d_ro, d_rp, d_bo, d_u, d_bq, d_cs, d_oo;
        {
            auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
            const size_t n_e = (size_t)C * Tc * K, n_out = std::max<size_t>((size_t)totOut * K, 1), n_b = std::max(totB, 1);
            const size_t need = pad(n_e * 8) + pad(tabs.size() * 8) + pad(n_out * 8) + 2 * pad((size_t)(NS + 1) * 4) +
                                pad((size_t)(totR + NS) * 4) + 2 * pad(n_b * 4) + 2 * pad((size_t)(C + 1) * 4) + 4096;
            pn->arena.require(need);
            pn->arena.reset();
            for (auto *b : {&d_e, &d_tabs, &d_out}) b->arena = &pn->arena;
            for (auto *b : {&d_ro, &d_rp, &d_bo, &d_u, &d_bq, &d_cs, &d_oo}) b->arena = &pn->arena;
            d_e.ensure(n_e); d_tabs.ensure(tabs.size()); d_out.ensure(n_out);
            d_ro.ensure(NS + 1); d_rp.ensure(totR + NS); d_bo.ensure(NS + 1); d_u.ensure(n_b); d_bq.ensure(n_b);
            d_cs.ensure(C); d_oo.ensure(C + 1);
        }
        d_e.upload(hap_common, (size_t)C * Tc * K, st); d_tabs.upload(tabs.data(), tabs.size(), st);
        d_ro.upload(read_off, NS + 1, st); d_rp.upload(read_ptr, totR + NS, st); d_bo.upload(base_off.data(), NS + 1, st);
        d_u.upload(u, totB, st); d_bq.upload(bq_eff.data(), totB, st);
        d_cs.upload(chain_sample, C, st); d_oo.upload(out_off.data(), C + 1, st);
        DenseParams prm{};
        prm.C = C; prm.K = K; prm.T = T; prm.Tc = Tc; prm.Jmax = Jmax; prm.rescale = rescale_eMatRead_t; prm.hap_major = 1;
        prm.common_index = rc->common_index.p; prm.chain_sample = d_cs.p; prm.out_off = d_oo.p;
        prm.inv_maxdiff = 1 / maxDifferenceBetweenReads; prm.eHaps = d_e.p; prm.read_off = d_ro.p;
        prm.read_ptr = d_rp.p; prm.base_off = d_bo.p; prm.u = d_u.p; prm.bq = d_bq.p;
        prm.pR_tab = d_tabs.p; prm.pA_tab = d_tabs.p + 512; prm.out = d_out.p;
        hipLaunchKernelGGL(k_ematread_dense, dim3((maxR + 63) / 64, C), dim3(64), 0, st, prm);
        QA_HIP(hipGetLastError());
        d_out.download(eMatRead_t, (size_t)totOut * K, st);
        QA_HIP(hipStreamSynchronize(st));
        return QA_OK;
    });
}

int qa_rcpp_make_eMatRead_t_nsnps(qa_panel_t *pn, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                                  const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                                  double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                                  double *eMatRead_t) {
    return make_eMatRead_t_impl(pn, nSNPs, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxDifferenceBetweenReads, Jmax,
                                rescale_eMatRead_t, eMatRead_t, 0);
}

int qa_rcpp_make_eMatRead_t_hap_major(qa_panel_t *pn, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps,
                                      const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                                      double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                                      double *eMatRead_t) {
    return make_eMatRead_t_impl(pn, nSNPs, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxDifferenceBetweenReads, Jmax,
                                rescale_eMatRead_t, eMatRead_t, 1);
}

int qa_rcpp_make_eMatRead_t(qa_panel_t *pn, int32_t n_chain, int32_t K, const double *eHaps, const int32_t *read_off,
                            const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                            double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                            double *eMatRead_t) {
    if (!pn) {
        qa::set_error("qa_rcpp_make_eMatRead_t: bad argument");
        return QA_ERR_INVALID;
    }
    return qa_rcpp_make_eMatRead_t_nsnps(pn, pn->T, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxDifferenceBetweenReads,
                                         Jmax, rescale_eMatRead_t, eMatRead_t);
}

}  // extern "C"
