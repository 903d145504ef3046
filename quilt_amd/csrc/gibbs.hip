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
//     read sees), so the unit of parallelism is the chain: ONE 64-lane wavefront owns one (sample,
//     Gibbs chain) and keeps the current grid's alpha / beta / eMatGrid columns of both labels in
//     VGPRs (Ks = 600 -> 10 fp64 per lane per column).  All Ks-wide sums are DPP/shuffle butterflies
//     -- no LDS, no barrier anywhere on the serial path.  Hundreds of chains fill the chip.
//   * columns live in HBM as [grid][Ks padded to 64] fp64, lane-strided, so every column load/store is
//     10 coalesced 512-byte wave accesses; the next read's emission column is prefetched while the
//     current one is resolved.
//   * fp64 throughout and no FMA contraction (-ffp-contract=off): per-element arithmetic is bit-for-
//     bit the reference's; only the order of the Ks-wide sums differs (1e-16 relative), which is what
//     lets the sampled read labels match the CPU path under the same uniforms.
//   * the uniforms the reference draws from R's RNG are inputs.
//
// Block Gibbs: for diploid samples `Rcpp_block_gibbs_resampler` (gibbs-nipt-block.cpp:1636-1967) is the
// identity: c3 is all zero (gibbs-nipt.cpp:2678), so logC_after(2) is -inf and then NaN (:1819-1821,
// :1896-1898), every choice_log_probs entry is NaN (:661-675), ir_chosen stays 0 (:741-752), the
// "No change warranted" branch is taken (:830) and the final backward (:1947-1954) reproduces the beta
// the sweep already holds.  The shard resampler is the active step and is implemented here.
// NIPT (ff > 0) block Gibbs is not implemented yet (QA_ERR_UNSUPPORTED).
#include "panel.hpp"

#include <algorithm>
#include <cmath>
#include <memory>

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
    int disable_read_category_usage;
    double class_sum_cutoff;
    const double *runif_reads; // [C][R_c * n_its] at offset read_off[c] * n_its
    const int32_t *first_read; // [C]
    const double *runif_shard; // [C][n_block][G-1]
    // state (per chain)
    double *eMatRead;        // at eread_off[c] doubles: [R_c][Ksp]
    const size_t *eread_off; // [C]
    uint8_t *is_cat1;        // [sum R]
    double *alpha, *beta, *eg;  // [C][2][G][Ksp]
    double *cvec;            // [C][3][G]
    int32_t *H;              // [sum R] labels 1-based (in/out)
    int32_t *H_class;        // [sum R]
    int32_t *status;         // [C] 0 ok, 1 underflow
    // outputs
    double *hapProbs, *genProbsM, *genProbsF;  // [C][T][3]
};

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
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

// ---------------------------------------------------------------------------------------------
// k_ematread: P(read r | small-panel haplotype k) (gibbs-small.cpp:148-263).  One wave per
// (read, chain); lane l owns rows l, l+64, ...  Products run over the read's bases in order, so each
// entry is bit-identical to the reference's; then divide by the column max and floor (:235-262).
// ---------------------------------------------------------------------------------------------
template <int NE>
__global__ __launch_bounds__(64) void k_ematread(GibbsParams p) {
    const int c = blockIdx.y, lane = threadIdx.x;
    const int R = p.read_off[c + 1] - p.read_off[c];
    const int r = blockIdx.x;
    if (r >= R) return;
    const int32_t *rp = p.read_ptr + p.read_off[c] + c;
    const int32_t *u = p.u + p.base_off[c], *bq = p.bq + p.base_off[c];
    const int32_t *which = p.which + (size_t)c * p.Ks;
    int kk[NE];
    double v[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const int k = lane + 64 * i;
        kk[i] = (k < p.Ks) ? which[k] : -1;
        v[i] = 1.0;
    }
    const int s = rp[r];
    int J = rp[r + 1] - s - 1;
    if (J >= p.Jmax) J = p.Jmax;
    int g_prev = -1;
    uint32_t w[NE];
    for (int j = 0; j <= J; j++) {
        const int b = bq[s + j];
        const int snp = u[s + j];
        const int g = snp >> 5;
        if (g != g_prev) {
#pragma unroll
            for (int i = 0; i < NE; i++) {
                w[i] = 0;
                if (kk[i] >= 0) w[i] = panel_word(p, g, kk[i], p.hm[(size_t)g * p.Kp + kk[i]]);
            }
            g_prev = g;
        }
        if (b == 0) continue;  // no base quality seen yet: factor 1 (host folded the carry-over rule)
        const int ab = b < 0 ? -b : b;
        const double pR = p.pR_tab[(b > 0 ? 256 : 0) + ab], pA = p.pA_tab[(b > 0 ? 256 : 0) + ab];
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const double e = ((w[i] >> (snp & 31)) & 1u) ? 1 - p.ref_error : p.ref_error;
            v[i] *= (e * pA + (1 - e) * pR);
        }
    }
    if (p.rescale) {
        double x = 0;
#pragma unroll
        for (int i = 0; i < NE; i++) if (kk[i] >= 0 && v[i] > x) x = v[i];
        x = wmax(x);
        const double d1 = 1 / x;
        if (isinf(x) || x == 0 || isinf(d1)) {
#pragma unroll
            for (int i = 0; i < NE; i++) v[i] = 1;
        } else {
#pragma unroll
            for (int i = 0; i < NE; i++) {
                v[i] *= d1;
                if (v[i] < p.inv_maxdiff) v[i] = p.inv_maxdiff;
            }
        }
    }
    // category 1 (gibbs-nipt.cpp:350-372): no entry below 1 - 1e-12
    const double thresh = 1 - 1e-12;
    bool below = false;
#pragma unroll
    for (int i = 0; i < NE; i++) if (kk[i] >= 0 && v[i] < thresh) below = true;
    const bool any_below = __any(below);
    double *out = p.eMatRead + p.eread_off[c] + (size_t)r * p.Ksp;
#pragma unroll
    for (int i = 0; i < NE; i++) out[lane + 64 * i] = (kk[i] >= 0) ? v[i] : 1.0;
    if (lane == 0) p.is_cat1[p.read_off[c] + r] = (any_below || p.disable_read_category_usage) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// k_gibbs: one wave per chain; initialisation, all sweeps, shard passes.
// ---------------------------------------------------------------------------------------------
template <int NE>
struct Col {
    double v[NE];
};

template <int NE>
__device__ __forceinline__ void load_col(Col<NE> &c, const double *src, int lane) {
#pragma unroll
    for (int i = 0; i < NE; i++) c.v[i] = src[lane + 64 * i];
}
template <int NE>
__device__ __forceinline__ void store_col(const Col<NE> &c, double *dst, int lane) {
#pragma unroll
    for (int i = 0; i < NE; i++) dst[lane + 64 * i] = c.v[i];
}
template <int NE>
__device__ __forceinline__ double sum_col(const Col<NE> &c) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < NE; i++) s += c.v[i];
    return wsum(s);
}

template <int NE>
struct Chain {
    const GibbsParams &p;
    int c, lane, R, G, Ks, Ksp;
    double *alpha[2], *beta[2], *eg[2], *cv[3];
    const double *eMatRead;
    const int32_t *wif;
    const uint8_t *ghr, *cat1;
    int32_t *H, *Hc;
    double prior;  // 1 / Ks
    bool valid[NE];

    __device__ Chain(const GibbsParams &p_, int c_, int lane_) : p(p_), c(c_), lane(lane_) {
        G = p.G; Ks = p.Ks; Ksp = p.Ksp;
        R = p.read_off[c + 1] - p.read_off[c];
        const size_t mat = (size_t)G * Ksp;
        for (int h = 0; h < 2; h++) {
            alpha[h] = p.alpha + ((size_t)c * 2 + h) * mat;
            beta[h] = p.beta + ((size_t)c * 2 + h) * mat;
            eg[h] = p.eg + ((size_t)c * 2 + h) * mat;
        }
        for (int h = 0; h < 3; h++) cv[h] = p.cvec + ((size_t)c * 3 + h) * G;
        eMatRead = p.eMatRead + p.eread_off[c];
        wif = p.wif + p.read_off[c];
        cat1 = p.is_cat1 + p.read_off[c];
        ghr = p.grid_has_read + (size_t)c * G;
        H = p.H + p.read_off[c];
        Hc = p.H_class + p.read_off[c];
        prior = 1.0 / Ks;
#pragma unroll
        for (int i = 0; i < NE; i++) valid[i] = (lane + 64 * i) < Ks;
    }

    // Rcpp_run_forward_haploid (copied-from-stitch.cpp:340-387), prior = alphaMat = 1/Ks
    __device__ void forward_full(int h) {
        Col<NE> a, e;
        load_col(e, eg[h], lane);
#pragma unroll
        for (int i = 0; i < NE; i++) a.v[i] = valid[i] ? prior * e.v[i] : 0.0;
        double cc = 1 / sum_col(a);
#pragma unroll
        for (int i = 0; i < NE; i++) a.v[i] = a.v[i] * cc;
        if (lane == 0) cv[h][0] = cc;
        store_col(a, alpha[h], lane);
        for (int g = 1; g < G; g++) {
            const double sig = tm0(g - 1);
            load_col(e, eg[h] + (size_t)g * Ksp, lane);
            const double t1 = tm1(g - 1);
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = valid[i] ? e.v[i] * (sig * a.v[i] + t1 * prior) : 0.0;
            cc = 1 / sum_col(a);
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] *= cc;
            if (lane == 0) cv[h][g] = cc;
            store_col(a, alpha[h] + (size_t)g * Ksp, lane);
        }
    }
    __device__ __forceinline__ double tm0(int g) const { return p.sigma[g]; }
    // transMatRate_t row 1 is stored by the reference as (1 - sigma) computed in R; the host passes it
    __device__ __forceinline__ double tm1(int g) const { return p.sigma[p.G - 1 + g]; }

    // Rcpp_run_backward_haploid (copied-from-stitch.cpp:392-409); beta(G-1) must be set
    __device__ void backward_generic(int h) {
        Col<NE> b, e;
        load_col(b, beta[h] + (size_t)(G - 1) * Ksp, lane);
        for (int g = G - 2; g >= 0; --g) {
            load_col(e, eg[h] + (size_t)(g + 1) * Ksp, lane);
            double s = 0;
#pragma unroll
            for (int i = 0; i < NE; i++) {
                b.v[i] = e.v[i] * b.v[i];
                s += valid[i] ? prior * b.v[i] : 0.0;
            }
            s = wsum(s);
            const double x = tm1(g) * s, cg = cv[h][g], s0 = tm0(g);
#pragma unroll
            for (int i = 0; i < NE; i++) b.v[i] = valid[i] ? cg * (x + s0 * b.v[i]) : 0.0;
            store_col(b, beta[h] + (size_t)g * Ksp, lane);
        }
    }
    // Rcpp_run_backward_haploid_QUILT_faster (copied-from-stitch.cpp:417-440)
    __device__ void backward_faster(int h) {
        Col<NE> b, e;
        const double one_over_K = 1 / (double)Ks;
        const double cl = cv[h][G - 1];
#pragma unroll
        for (int i = 0; i < NE; i++) b.v[i] = valid[i] ? cl : 0.0;
        store_col(b, beta[h] + (size_t)(G - 1) * Ksp, lane);
        for (int g = G - 2; g >= 0; --g) {
            if (ghr[g + 1]) {
                load_col(e, eg[h] + (size_t)(g + 1) * Ksp, lane);
#pragma unroll
                for (int i = 0; i < NE; i++) b.v[i] = e.v[i] * b.v[i];
            }
            const double x = tm1(g) * sum_col(b) * one_over_K, cg = cv[h][g], s0 = tm0(g);
#pragma unroll
            for (int i = 0; i < NE; i++) b.v[i] = valid[i] ? cg * (x + s0 * b.v[i]) : 0.0;
            store_col(b, beta[h] + (size_t)g * Ksp, lane);
        }
    }
};

template <int NE>
__global__ __launch_bounds__(64) void k_gibbs(GibbsParams p) {
    const int c = blockIdx.x, lane = threadIdx.x;
    Chain<NE> ch(p, c, lane);
    const int G = ch.G, Ksp = ch.Ksp, R = ch.R, Ks = ch.Ks;
    const double prior = ch.prior;
    bool (&valid)[NE] = ch.valid;
    const double *runif = p.runif_reads + (size_t)p.read_off[c] * p.n_its;
    const int first_read = p.first_read[c];

    // ---- c = 0 (arma::zeros, gibbs-nipt.cpp:2676-2678), H_class = 0
    for (int h = 0; h < 3; h++)
        for (int g = lane; g < G; g += 64) ch.cv[h][g] = 0.0;
    for (int r = lane; r < R; r += 64) ch.Hc[r] = 0;

    // ---- rcpp_gibbs_nipt_initialize (:1629-1750)
    {
        Col<NE> one;
#pragma unroll
        for (int i = 0; i < NE; i++) one.v[i] = 1.0;
        for (int h = 0; h < 2; h++)
            for (int g = 0; g < G; g++) store_col(one, ch.eg[h] + (size_t)g * Ksp, lane);
    }
    if (!p.init_iteratively) {
        // rcpp_make_eMatGrid_t (copied-from-stitch.cpp:262-281): reads are sorted by grid, so the
        // products of one grid are formed in registers in read order
        int r = 0;
        while (r < R) {
            const int g = ch.wif[r];
            Col<NE> e[2];
#pragma unroll
            for (int i = 0; i < NE; i++) e[0].v[i] = e[1].v[i] = 1.0;
            while (r < R && ch.wif[r] == g) {
                Col<NE> er;
                load_col(er, ch.eMatRead + (size_t)r * Ksp, lane);
                const int h = ch.H[r] - 1;
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    if (h == 0) e[0].v[i] *= er.v[i];
                    else e[1].v[i] *= er.v[i];
                }
                r++;
            }
            store_col(e[0], ch.eg[0] + (size_t)g * Ksp, lane);
            store_col(e[1], ch.eg[1] + (size_t)g * Ksp, lane);
        }
        for (int h = 0; h < 2; h++) {
            ch.forward_full(h);
            Col<NE> b;
            const double cl = ch.cv[h][G - 1];
#pragma unroll
            for (int i = 0; i < NE; i++) b.v[i] = valid[i] ? cl : 0.0;
            store_col(b, ch.beta[h] + (size_t)(G - 1) * Ksp, lane);
            ch.backward_generic(h);
        }
    } else {
        // alpha = beta = 1, c = 1, then only column 0 of alpha is initialised (:1725-1740)
        Col<NE> one;
#pragma unroll
        for (int i = 0; i < NE; i++) one.v[i] = valid[i] ? 1.0 : 0.0;
        for (int h = 0; h < 2; h++) {
            for (int g = 0; g < G; g++) {
                store_col(one, ch.alpha[h] + (size_t)g * Ksp, lane);
                store_col(one, ch.beta[h] + (size_t)g * Ksp, lane);
            }
            for (int g = lane; g < G; g += 64) ch.cv[h][g] = 1.0;
            Col<NE> a;
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = valid[i] ? prior * 1.0 : 0.0;
            const double cc = 1 / sum_col(a);
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = a.v[i] * cc;
            store_col(a, ch.alpha[h], lane);
            if (lane == 0) ch.cv[h][0] = cc;
        }
    }

    const double rlc3_0 = 0.5 / (0.5 + 0.5), rlc3_1 = 0.5 / (0.5 + 0.5);  // ff = 0 prototypes (:2707-2729)
    int shard_it = 0;
    int status = 0;
    for (int it = 0; it < p.n_its && status == 0; it++) {
        // ================= rcpp_gibbs_nipt_iterate (:1756-1956) =================
        Col<NE> a[2];   // alpha of the current grid, both labels
        int iRead = 0;  // next unprocessed read
        for (int g = 0; g < G; g++) {
            Col<NE> e[2];
            const bool has = ch.ghr[g] != 0;
            load_col(e[0], ch.eg[0] + (size_t)g * Ksp, lane);
            load_col(e[1], ch.eg[1] + (size_t)g * Ksp, lane);
            double cg[2];
            if (g > 0) {
                // rcpp_alpha_forward_one_QUILT_faster (:671-707), normalize = true
                const double x = ch.tm0(g - 1), t1 = ch.tm1(g - 1);
                const double one_over_K = 1 / (double)Ks;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double alphaConst = t1 * sum_col(a[h]);
                    const double c2 = ch.cv[h][g];
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        const double inner = (x * a[h].v[i] + alphaConst * one_over_K);
                        a[h].v[i] = valid[i] ? (has ? e[h].v[i] * inner : inner) : 0.0;
                    }
                    double aa = 1 / (c2 * sum_col(a[h]));
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
                    cg[h] = 1 / sum_col(a[h]);
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] *= cg[h];
                }
            }
            // ---- sample_reads_in_grid (:733-1295)
            bool grid_started = false, changed = false;
            Col<NE> am[2], ab[2];
            double pC[3] = {1, 1, 1}, pA1[3] = {1, 1, 1}, pA2[3] = {1, 1, 1};
            int h_rC = 0, h_rA1 = 1, h_rA2 = 2;
            bool normal = false, ginit = false, pass = false;
            while (iRead < R && ch.wif[iRead] == g) {
                const int r = iRead;
                iRead++;
                if (ch.cat1[r]) continue;  // diploid: reads that cannot discriminate are skipped (:815)
                if (!p.init_iteratively) normal = true;
                else if (r < first_read && it == 0) pass = true;
                else if (first_read <= r && it == 0) { pass = false; ginit = true; }
                else if (r < first_read && it == 1) { pass = false; ginit = true; }
                else { ginit = false; normal = true; }
                if (!grid_started) {
                    Col<NE> b;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        load_col(b, ch.beta[h] + (size_t)g * Ksp, lane);
#pragma unroll
                        for (int i = 0; i < NE; i++) {
                            am[h].v[i] = a[h].v[i];
                            ab[h].v[i] = a[h].v[i] * b.v[i];
                        }
                        pC[h] = sum_col(ab[h]);
                    }
                    pC[2] = 1;
                    grid_started = true;
                }
                Col<NE> er;
                load_col(er, ch.eMatRead + (size_t)r * Ksp, lane);
                if (normal) {
                    h_rC = ch.H[r] - 1;
                    h_rA1 = (h_rC == 0) ? 1 : 0;
                    h_rA2 = (h_rC == 2) ? 1 : 2;
                    for (int h = 0; h < 3; h++) pA1[h] = pA2[h] = pC[h];
                    // dense form for every category (the reference's sparse category-2/3 updates are
                    // algebraically the same sums: test-unit-gibbs-diploid.R:114-124)
                    double s1 = 0, s2 = 0;
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        const double xc = (h_rC == 0) ? ab[0].v[i] : ab[1].v[i];
                        const double xa = (h_rC == 0) ? ab[1].v[i] : ab[0].v[i];
                        s1 += xc / er.v[i];
                        s2 += xa * er.v[i];
                    }
                    pA1[h_rC] = wsum(s1);
                    pA1[h_rA1] = wsum(s2);
                    pA2[h_rA1] = pC[h_rA1];
                    pA2[h_rC] = pA1[h_rC];
                } else if (ginit) {
                    h_rC = 0; h_rA1 = 1; h_rA2 = 2;
                    for (int h = 0; h < 3; h++) pA1[h] = pA2[h] = pC[h];
                    double s1 = 0, s2 = 0;
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        s1 += ab[0].v[i] * er.v[i];
                        s2 += ab[1].v[i] * er.v[i];
                    }
                    pC[0] = wsum(s1);
                    pA1[1] = wsum(s2);
                } else {
                    for (int h = 0; h < 3; h++) pA1[h] = pA2[h] = pC[h];
                }
                const double prior_probs[3] = {0.5, (1 - 0.0) * 0.5, 0.0 * 0.5};
                const double prod_pC = (pC[0] * pC[1] * pC[2]) * prior_probs[h_rC];
                const double prod_pA1 = (pA1[0] * pA1[1] * pA1[2]) * prior_probs[h_rA1];
                const double prod_pA2 = (pA2[0] * pA2[1] * pA2[2]) * prior_probs[h_rA2];
                const double denom = prod_pC + prod_pA1 + prod_pA2;
                const double norm_pC = prod_pC / denom, norm_pA1 = prod_pA1 / denom, norm_pA2 = prod_pA2 / denom;
                const double chance = runif[(size_t)R * it + r];
                double cs[3] = {0, 0, 0};
                cs[h_rC] = norm_pC;
                cs[h_rA1] = norm_pA1;
                cs[h_rA2] = norm_pA2;
                cs[1] += cs[0];
                cs[2] += cs[1];
                int h_rN = 0;
                for (int i = 2; i >= 0; i--) if (chance < cs[i]) h_rN = i;
                if (((h_rN != h_rC) || ginit) && !pass && h_rN < 2) {
                    changed = true;
                    if (lane == 0) ch.H[r] = h_rN + 1;
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        if (normal) {
                            if (h_rC == 0) { am[0].v[i] /= er.v[i]; ab[0].v[i] /= er.v[i]; e[0].v[i] /= er.v[i]; }
                            else { am[1].v[i] /= er.v[i]; ab[1].v[i] /= er.v[i]; e[1].v[i] /= er.v[i]; }
                        }
                        if (h_rN == 0) { am[0].v[i] *= er.v[i]; ab[0].v[i] *= er.v[i]; e[0].v[i] *= er.v[i]; }
                        else { am[1].v[i] *= er.v[i]; ab[1].v[i] *= er.v[i]; e[1].v[i] *= er.v[i]; }
                    }
                    if (normal) {
                        for (int i = 0; i < 3; i++) pC[i] = (h_rN == h_rA1) ? pA1[i] : pA2[i];
                    } else if (ginit) {
                        if (h_rN == 1) for (int i = 0; i < 3; i++) pC[i] = pA1[i];
                    }
                }
                // record_read_set (:1142-1165)
                {
                    double x[3];
                    x[h_rC] = norm_pC;
                    x[h_rA1] = norm_pA1;
                    x[h_rA2] = norm_pA2;
                    const double rlc[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {rlc3_0, rlc3_1, 0},
                                              {0.5 / (0.5 + 0.0), 0, 0.0 / (0.5 + 0.0)},
                                              {0, 0.5 / (0.5 + 0.0), 0.0 / (0.5 + 0.0)}, {0.5, 0.5, 0.0}};
                    double local_min = 2;
                    int which = 8;
                    for (int i = 0; i < 7; i++) {
                        const double y = fabs(rlc[i][0] - x[0]) + fabs(rlc[i][1] - x[1]) + fabs(rlc[i][2] - x[2]);
                        if (y < local_min) { local_min = y; which = i; }
                    }
                    if (lane == 0) ch.Hc[r] = (local_min < p.class_sum_cutoff) ? which + 1 : 0;
                }
            }
            if (changed) {
                // re-inject the moved columns and renormalise (:1262-1292)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const double alphaConst = 1 / sum_col(am[h]);
                    cg[h] *= alphaConst;
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] = am[h].v[i] * alphaConst;
                    store_col(e[h], ch.eg[h] + (size_t)g * Ksp, lane);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                store_col(a[h], ch.alpha[h] + (size_t)g * Ksp, lane);
                if (lane == 0) ch.cv[h][g] = cg[h];
            }
        }
            ch.backward_faster(0);
        ch.backward_faster(1);
        // ---- underflow check (:2959-2969)
        for (int h = 0; h < 2; h++) {
            double s = 0;
            for (int g = lane; g < G; g += 64) s += ch.cv[h][g];
            s = wsum(s);
            if (!isfinite(s)) status = 1;
        }
        if (status) break;
        bool to_block = false;
        for (int i = 0; i < p.n_block; i++) if (p.block_its[i] == it) to_block = true;
        if (to_block && p.do_shard) {
            // ============ Rcpp_shard_block_gibbs_resampler (gibbs-nipt-block.cpp:1975-2355), ff = 0,
            // shard_check_every_pair: one left-to-right pass deciding at every grid whether everything
            // to the right swaps haplotypes ============
            const double *ru = p.runif_shard + ((size_t)c * p.n_block + shard_it) * (G - 1);
            shard_it++;
            double mloc1 = 0, mloc2 = 0, mlc1 = 0, mlc2 = 0;
            {
                double s1 = 0, s2 = 0;
                // sequential order matters little here; keep the reference's running form per lane 0
                for (int g = 0; g < G; g++) { s1 -= log(ch.cv[0][g]); s2 -= log(ch.cv[1][g]); }
                mloc1 = s1; mloc2 = s2;
            }
            bool flip = false;
            int ir = 0;
            Col<NE> s_a[2];
            for (int g = 0; g < G; g++) {
                const double oc1 = ch.cv[0][g], oc2 = ch.cv[1][g];
                Col<NE> e[2];
                load_col(e[0], ch.eg[0] + (size_t)g * Ksp, lane);
                load_col(e[1], ch.eg[1] + (size_t)g * Ksp, lane);
                double cn[2];
                if (g == 0) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
#pragma unroll
                        for (int i = 0; i < NE; i++) s_a[h].v[i] = valid[i] ? prior * e[h].v[i] : 0.0;
                        cn[h] = 1 / sum_col(s_a[h]);
#pragma unroll
                        for (int i = 0; i < NE; i++) s_a[h].v[i] *= cn[h];
                    }
                } else {
                    if (flip) {
                        Col<NE> t = e[0]; e[0] = e[1]; e[1] = t;
                        store_col(e[0], ch.eg[0] + (size_t)g * Ksp, lane);
                        store_col(e[1], ch.eg[1] + (size_t)g * Ksp, lane);
                    }
                    // rcpp_alpha_forward_one (gibbs-nipt.cpp:627-657), alphaMat = 1/Ks, normalize
                    const double x = ch.tm0(g - 1), t1 = ch.tm1(g - 1);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const double alphaConst = t1 * sum_col(s_a[h]);
                        const double c2 = (h == 0) ? oc1 : oc2;
#pragma unroll
                        for (int i = 0; i < NE; i++)
                            s_a[h].v[i] = valid[i] ? c2 * e[h].v[i] * (x * s_a[h].v[i] + alphaConst * prior) : 0.0;
                        const double aa = 1 / sum_col(s_a[h]);
                        cn[h] = c2 * aa;
#pragma unroll
                        for (int i = 0; i < NE; i++) s_a[h].v[i] *= aa;
                    }
                }
                store_col(s_a[0], ch.alpha[0] + (size_t)g * Ksp, lane);
                store_col(s_a[1], ch.alpha[1] + (size_t)g * Ksp, lane);
                if (lane == 0) { ch.cv[0][g] = cn[0]; ch.cv[1][g] = cn[1]; }
                mlc1 -= log(cn[0]);
                mlc2 -= log(cn[1]);
                while (ir < R && ch.wif[ir] == g) {
                    if (flip && lane == 0) ch.H[ir] = 3 - ch.H[ir];
                    ir++;
                }
                if (g < G - 1) {
                    Col<NE> b1, b2;
                    load_col(b1, ch.beta[0] + (size_t)g * Ksp, lane);
                    load_col(b2, ch.beta[1] + (size_t)g * Ksp, lane);
                    double s11 = 0, s22 = 0, s21 = 0, s12 = 0;
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        s11 += s_a[0].v[i] * b1.v[i];
                        s22 += s_a[1].v[i] * b2.v[i];
                        s21 += s_a[1].v[i] * b1.v[i];
                        s12 += s_a[0].v[i] * b2.v[i];
                    }
                    s11 = wsum(s11); s22 = wsum(s22); s21 = wsum(s21); s12 = wsum(s12);
                    const double pA1 = mlc1 + mloc1 + log(s11);
                    const double pA2 = mlc2 + mloc2 + log(s22);
                    const double pB1 = mlc2 + mloc1 + log(s21);
                    const double pB2 = mlc1 + mloc2 + log(s12);
                    const double diff = pB1 + pB2 - pA1 - pA2;
                    double probs1 = 1;
                    const double probs2 = exp(diff);
                    const double ps = probs1 + probs2;
                    probs1 /= ps;
                    flip = ru[g] > probs1;
                }
                mloc1 += log(oc1);
                mloc2 += log(oc2);
            }
                    for (int h = 0; h < 2; h++) {
                Col<NE> b;
                const double cl = ch.cv[h][G - 1];
#pragma unroll
                for (int i = 0; i < NE; i++) b.v[i] = valid[i] ? cl : 0.0;
                store_col(b, ch.beta[h] + (size_t)(G - 1) * Ksp, lane);
                ch.backward_generic(h);
            }
        }
    }
    if (lane == 0) p.status[c] = status;
}

// ---------------------------------------------------------------------------------------------
// k_happrobs: gamma = alpha * beta / c on the fly, scattered to the 32 SNPs of the grid by each
// haplotype's word (gibbs-small.cpp:540-634).  One 256-thread block per (grid, chain): thread =
// (bit b, slice of 8 over k).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_happrobs(GibbsParams p) {
    const int g = blockIdx.x, c = blockIdx.y;
    const int b = threadIdx.x & 31, part = threadIdx.x >> 5;
    __shared__ double s_g[2][8][32], s_t[2][8];
    __shared__ uint32_t s_w[1024];
    const int Ks = p.Ks, Ksp = p.Ksp, G = p.G;
    const int32_t *which = p.which + (size_t)c * Ks;
    for (int k = threadIdx.x; k < Ks; k += 256) {
        const int kk = which[k];
        s_w[k] = panel_word(p, g, kk, p.hm[(size_t)g * p.Kp + kk]);
    }
    __syncthreads();
    const int s = 32 * g, nLocal = min(32, p.T - s);
    const size_t mat = (size_t)G * Ksp;
    double acc[2] = {0, 0}, tot[2] = {0, 0};
    for (int h = 0; h < 2; h++) {
        const double *a = p.alpha + ((size_t)c * 2 + h) * mat + (size_t)g * Ksp;
        const double *be = p.beta + ((size_t)c * 2 + h) * mat + (size_t)g * Ksp;
        const double x = 1 / p.cvec[((size_t)c * 3 + h) * G + g];
        for (int k = part; k < Ks; k += 8) {
            const double gk = (a[k] * be[k]) * x;
            tot[h] += gk;
            if ((s_w[k] >> b) & 1u) acc[h] += gk;
        }
        s_g[h][part][b] = acc[h];
        if (b == 0) s_t[h][part] = tot[h];
    }
    __syncthreads();
    if (part == 0 && b < nLocal) {
        double g1[2];
        for (int h = 0; h < 2; h++) {
            double on = 0, all = 0;
            for (int q = 0; q < 8; q++) { on += s_g[h][q][b]; all += s_t[h][q]; }
            const double off = all - on;  // sum over haplotypes whose bit is 0
            g1[h] = on * (1 - p.ref_error) + off * p.ref_error;
        }
        const double g0 = g1[0], gB = g1[1], g2 = 0.0 * (1 - p.ref_error) + 0.0 * p.ref_error;
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
// k_ematread_dense: `rcpp_make_eMatRead_t` (copied-from-stitch.cpp:115-229) for dense per-SNP
// haplotype dosages (the 2-3 "haplotypes" of calculate_eMatRead_t_vs_haplotypes, functions.R:2975-3020).
// One thread per (read, chain): K is 2 or 3, the products run over the read's bases in order.
// ---------------------------------------------------------------------------------------------
struct DenseParams {
    int C, K, T, Jmax, rescale;
    double inv_maxdiff;
    const double *eHaps;      // [C][T][K]
    const int32_t *read_off, *read_ptr, *base_off, *u, *bq;
    const double *pR_tab, *pA_tab;
    double *out;              // [sum R][K]
};

__global__ __launch_bounds__(64) void k_ematread_dense(DenseParams p) {
    const int c = blockIdx.y;
    const int R = p.read_off[c + 1] - p.read_off[c];
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const int32_t *rp = p.read_ptr + p.read_off[c] + c;
    const int32_t *u = p.u + p.base_off[c], *bq = p.bq + p.base_off[c];
    const double *eh = p.eHaps + (size_t)c * p.T * p.K;
    double v[3] = {1, 1, 1};
    const int s = rp[r];
    int J = rp[r + 1] - s - 1;
    if (J >= p.Jmax) J = p.Jmax;
    for (int j = 0; j <= J; j++) {
        const int b = bq[s + j];
        if (b == 0) continue;
        const int ab = b < 0 ? -b : b;
        const double pR = p.pR_tab[(b > 0 ? 256 : 0) + ab], pA = p.pA_tab[(b > 0 ? 256 : 0) + ab];
        const double *e = eh + (size_t)u[s + j] * p.K;
        for (int k = 0; k < p.K; k++) v[k] *= (e[k] * pA + (1 - e[k]) * pR);
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
    double *o = p.out + (size_t)(p.read_off[c] + r) * p.K;
    for (int k = 0; k < p.K; k++) o[k] = v[k];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace qa {

struct GibbsScratch {
    DBuf<int32_t> which, read_off, read_ptr, base_off, u, bq, wif, block_its, first_read, H, H_class, status;
    DBuf<uint8_t> ghr, is_cat1;
    DBuf<double> tabs, runif_reads, runif_shard, eMatRead, alpha, beta, eg, cvec, hap, gm, gf, tm;
    DBuf<size_t> eread_off;
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
void fold_zero_base_qualities(std::vector<int32_t> &bq_eff, int C, const int32_t *read_off, const int32_t *read_ptr,
                              const std::vector<int32_t> &base_off, int Jmax) {
    for (int c = 0; c < C; c++) {
        const int R = read_off[c + 1] - read_off[c];
        const int32_t *rp = read_ptr + read_off[c] + c;
        int last = 0;
        for (int r = 0; r < R; r++) {
            int J = rp[r + 1] - rp[r] - 1;
            if (J >= Jmax) J = Jmax;
            for (int j = 0; j <= J; j++) {
                int32_t &b = bq_eff[(size_t)base_off[c] + rp[r] + j];
                if (b == 0) b = last; else last = b;
                if (b > 255 || b < -255) throw std::runtime_error("|base quality| > 255");
            }
        }
    }
}

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

template <int NE>
void launch_gibbs(const GibbsParams &prm, int maxR, hipStream_t st, hipEvent_t *ev) {
    QA_HIP(hipEventRecord(ev[0], st));
    hipLaunchKernelGGL(k_ematread<NE>, dim3(maxR, prm.C), dim3(64), 0, st, prm);
    QA_HIP(hipGetLastError());
    QA_HIP(hipEventRecord(ev[1], st));
    hipLaunchKernelGGL(k_gibbs<NE>, dim3(prm.C), dim3(64), 0, st, prm);
    QA_HIP(hipGetLastError());
    QA_HIP(hipEventRecord(ev[2], st));
    hipLaunchKernelGGL(k_happrobs, dim3(prm.G, prm.C), dim3(256), 0, st, prm);
    QA_HIP(hipGetLastError());
    QA_HIP(hipEventRecord(ev[3], st));
}

}  // namespace

extern "C" {

int qa_gibbs_batch(qa_panel_t *pn, const qa_gibbs_opts_t *o, int32_t n_chain, const int32_t *which_haps_to_use_1based,
                   const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                   const int32_t *wif, const double *runif_reads, const int32_t *first_read,
                   const double *runif_shard, int32_t *H, int32_t *H_class, double *hapProbs_t,
                   double *genProbsM_t, double *genProbsF_t, int32_t *underflow_problem, double *state_out) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!pn || !o || n_chain <= 0 || !which_haps_to_use_1based || !read_off || !read_ptr || !u || !bq || !wif ||
        !runif_reads || !first_read || !H) {
        qa::set_error("qa_gibbs_batch: null argument");
        return QA_ERR_INVALID;
    }
    if (o->ff != 0.0 || !o->sample_is_diploid) {
        qa::set_error("qa_gibbs_batch: only the diploid sampler (ff = 0, sample_is_diploid) is implemented on the device");
        return QA_ERR_UNSUPPORTED;
    }
    if (o->Ks <= 0 || o->Ks > 1024) {
        qa::set_error("qa_gibbs_batch: Ksubset = %d outside 1..1024", o->Ks);
        return QA_ERR_UNSUPPORTED;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(pn->device));
        if (!g_gibbs) g_gibbs.reset(new GibbsHolder());
        auto &S = g_gibbs->s;
        hipStream_t st = pn->stream;
        const int C = n_chain, G = pn->G, T = pn->T, Ks = o->Ks;
        const int Ksp = (Ks + 63) / 64 * 64, NE = Ksp / 64;
        const int n_its = o->n_gibbs_burn_in_its + o->n_gibbs_sample_its;
        const int totR = read_off[C];
        // bases: per chain the CSR block read_ptr[read_off[c] + c .. read_off[c+1] + c] is local (starts at 0)
        std::vector<int32_t> base_off(C + 1, 0), which0((size_t)C * Ks), bq_eff;
        std::vector<size_t> eoff(C);
        std::vector<uint8_t> ghr((size_t)C * G, 0);
        int maxR = 0;
        size_t etot = 0;
        for (int c = 0; c < C; c++) {
            const int R = read_off[c + 1] - read_off[c];
            maxR = std::max(maxR, R);
            const int32_t *rp = read_ptr + read_off[c] + c;
            base_off[c + 1] = base_off[c] + rp[R];
            eoff[c] = etot;
            etot += (size_t)R * Ksp;
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
            }
        }
        const int totB = base_off[C];
        bq_eff.assign(bq, bq + totB);
        fold_zero_base_qualities(bq_eff, C, read_off, read_ptr, base_off, o->Jmax);
        const std::vector<double> tabs = base_quality_tables();
        std::vector<double> tm((size_t)2 * std::max(G - 1, 1));
        for (int g = 0; g < G - 1; g++) { tm[g] = pn->h_sigma[g]; tm[(size_t)G - 1 + g] = pn->h_tm1[g]; }

        S.which.ensure(which0.size()); S.which.upload(which0.data(), which0.size(), st);
        S.read_off.ensure(C + 1); S.read_off.upload(read_off, C + 1, st);
        S.read_ptr.ensure(totR + C); S.read_ptr.upload(read_ptr, totR + C, st);
        S.base_off.ensure(C + 1); S.base_off.upload(base_off.data(), C + 1, st);
        S.u.ensure(std::max(totB, 1)); S.u.upload(u, totB, st);
        S.bq.ensure(std::max(totB, 1)); S.bq.upload(bq_eff.data(), totB, st);
        S.wif.ensure(std::max(totR, 1)); S.wif.upload(wif, totR, st);
        S.ghr.ensure(ghr.size()); S.ghr.upload(ghr.data(), ghr.size(), st);
        S.tabs.ensure(tabs.size()); S.tabs.upload(tabs.data(), tabs.size(), st);
        S.tm.ensure(tm.size()); S.tm.upload(tm.data(), tm.size(), st);
        S.block_its.ensure(std::max(o->n_block_gibbs_iterations, 1));
        S.block_its.upload(o->block_gibbs_iterations, o->n_block_gibbs_iterations, st);
        S.first_read.ensure(C); S.first_read.upload(first_read, C, st);
        S.runif_reads.ensure(std::max<size_t>((size_t)totR * n_its, 1));
        S.runif_reads.upload(runif_reads, (size_t)totR * n_its, st);
        const size_t nshard = (size_t)C * std::max(o->n_block_gibbs_iterations, 1) * std::max(G - 1, 1);
        S.runif_shard.ensure(nshard);
        if (runif_shard && o->n_block_gibbs_iterations > 0)
            S.runif_shard.upload(runif_shard, (size_t)C * o->n_block_gibbs_iterations * (G - 1), st);
        S.eread_off.ensure(C); S.eread_off.upload(eoff.data(), C, st);
        S.eMatRead.ensure(std::max<size_t>(etot, 1));
        S.is_cat1.ensure(std::max(totR, 1));
        const size_t mat = (size_t)C * 2 * G * Ksp;
        S.alpha.ensure(mat); S.beta.ensure(mat); S.eg.ensure(mat);
        S.cvec.ensure((size_t)C * 3 * G);
        S.H.ensure(std::max(totR, 1)); S.H.upload(H, totR, st);
        S.H_class.ensure(std::max(totR, 1));
        S.status.ensure(C);
        S.hap.ensure((size_t)C * T * 3); S.gm.ensure((size_t)C * T * 3); S.gf.ensure((size_t)C * T * 3);

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
        prm.runif_reads = S.runif_reads.p; prm.first_read = S.first_read.p; prm.runif_shard = S.runif_shard.p;
        prm.eMatRead = S.eMatRead.p; prm.eread_off = S.eread_off.p; prm.is_cat1 = S.is_cat1.p;
        prm.alpha = S.alpha.p; prm.beta = S.beta.p; prm.eg = S.eg.p; prm.cvec = S.cvec.p;
        prm.H = S.H.p; prm.H_class = S.H_class.p; prm.status = S.status.p;
        prm.hapProbs = S.hap.p; prm.genProbsM = S.gm.p; prm.genProbsF = S.gf.p;

        for (auto &e : g_gibbs->ev) if (!e) QA_HIP(hipEventCreate(&e));
        switch (NE) {
            case 1: launch_gibbs<1>(prm, maxR, st, g_gibbs->ev); break;
            case 2: launch_gibbs<2>(prm, maxR, st, g_gibbs->ev); break;
            case 3: launch_gibbs<3>(prm, maxR, st, g_gibbs->ev); break;
            case 4: launch_gibbs<4>(prm, maxR, st, g_gibbs->ev); break;
            case 5: launch_gibbs<5>(prm, maxR, st, g_gibbs->ev); break;
            case 6: launch_gibbs<6>(prm, maxR, st, g_gibbs->ev); break;
            case 7: launch_gibbs<7>(prm, maxR, st, g_gibbs->ev); break;
            case 8: launch_gibbs<8>(prm, maxR, st, g_gibbs->ev); break;
            case 9: launch_gibbs<9>(prm, maxR, st, g_gibbs->ev); break;
            case 10: launch_gibbs<10>(prm, maxR, st, g_gibbs->ev); break;
            case 12: launch_gibbs<12>(prm, maxR, st, g_gibbs->ev); break;
            case 16: launch_gibbs<16>(prm, maxR, st, g_gibbs->ev); break;
            default: throw std::runtime_error("Ksubset geometry not built (NE must be 1..10, 12 or 16)");
        }
        S.H.download(H, totR, st);
        if (H_class) S.H_class.download(H_class, totR, st);
        std::vector<int32_t> status(C);
        S.status.download(status.data(), C, st);
        if (hapProbs_t) S.hap.download(hapProbs_t, (size_t)C * T * 3, st);
        if (genProbsM_t) S.gm.download(genProbsM_t, (size_t)C * T * 3, st);
        if (genProbsF_t) S.gf.download(genProbsF_t, (size_t)C * T * 3, st);
        QA_HIP(hipStreamSynchronize(st));
        {
            float ms[3];
            for (int i = 0; i < 3; i++) QA_HIP(hipEventElapsedTime(&ms[i], g_gibbs->ev[i], g_gibbs->ev[i + 1]));
            // algorithmic bytes: per sweep and label 6 column streams of Ks x G fp64, plus every
            // read's emission column once per sweep; initialisation and each shard pass ~ one sweep without reads
            const double col = 2.0 * Ks * (double)G * 48.0;
            const double sweeps = (double)n_its * (C * col + (double)totR * Ks * 8.0) +
                                  (1.0 + 2.0 * prm.n_block) * C * col;
            qa::profile_add(qa::PK_EMATREAD, ms[0], (double)totR * Ks * 8.0);
            qa::profile_add(qa::PK_GIBBS, ms[1], sweeps);
            qa::profile_add(qa::PK_HAPPROBS, ms[2], C * 2.0 * Ks * (double)G * 16.0);
        }
        int rc = QA_OK;
        for (int c = 0; c < C; c++) {
            if (underflow_problem) underflow_problem[c] = status[c];
            if (status[c]) rc = QA_UNDERFLOW;
        }
        if (state_out && C == 1) {
            // debugging / test aid: alpha, beta, eMatGrid of both labels ([6][G][Ks]) then c ([3][G])
            std::vector<double> tmp((size_t)G * Ksp);
            const qa::DBuf<double> *src[3] = {&S.alpha, &S.beta, &S.eg};
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
        return rc;
    });
}


int qa_rcpp_make_eMatRead_t(qa_panel_t *pn, int32_t n_chain, int32_t K, const double *eHaps, const int32_t *read_off,
                            const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                            double maxDifferenceBetweenReads, int32_t Jmax, int32_t rescale_eMatRead_t,
                            double *eMatRead_t) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!pn || n_chain <= 0 || K < 1 || K > 3 || !eHaps || !read_off || !read_ptr || !u || !bq || !eMatRead_t) {
        qa::set_error("qa_rcpp_make_eMatRead_t: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(pn->device));
        hipStream_t st = pn->stream;
        const int C = n_chain, T = pn->T;
        std::vector<int32_t> base_off(C + 1, 0);
        int maxR = 0;
        for (int c = 0; c < C; c++) {
            const int R = read_off[c + 1] - read_off[c];
            maxR = std::max(maxR, R);
            base_off[c + 1] = base_off[c] + (read_ptr + read_off[c] + c)[R];
        }
        const int totR = read_off[C], totB = base_off[C];
        std::vector<int32_t> bq_eff(bq, bq + totB);
        fold_zero_base_qualities(bq_eff, C, read_off, read_ptr, base_off, Jmax);
        const std::vector<double> tabs = base_quality_tables();
        qa::DBuf<double> d_e((size_t)C * T * K), d_tabs(tabs.size()), d_out(std::max<size_t>((size_t)totR * K, 1));
        qa::DBuf<int32_t> d_ro(C + 1), d_rp(totR + C), d_bo(C + 1), d_u(std::max(totB, 1)), d_bq(std::max(totB, 1));
        d_e.upload(eHaps, (size_t)C * T * K, st); d_tabs.upload(tabs.data(), tabs.size(), st);
        d_ro.upload(read_off, C + 1, st); d_rp.upload(read_ptr, totR + C, st); d_bo.upload(base_off.data(), C + 1, st);
        d_u.upload(u, totB, st); d_bq.upload(bq_eff.data(), totB, st);
        DenseParams prm{};
        prm.C = C; prm.K = K; prm.T = T; prm.Jmax = Jmax; prm.rescale = rescale_eMatRead_t;
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

}  // extern "C"
