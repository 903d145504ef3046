// fullpass.hip -- full-panel haploid Li-Stephens forward/backward on gfx950 (MI355X).
//
// What it computes: QUILT/src/reference-single.cpp `Rcpp_haploid_dosage_versus_refs`
// (:2189-2413) = `Rcpp_build_eMatDH` (:272-329) + forward v3 (:878-1131) + backward v3
// (:1781-2179) + top-K picker (:129-194), for use_eMatDH = TRUE.
//
// How (MI355X-first, not the reference's loop nest):
//   * the recursion is sequential in the grid index g and needs one all-K sum per grid, so ONE
//     workgroup owns one (sample, chain, label) pass and keeps all K alpha (resp. beta) values in
//     VGPRs (fp32, 16 * NCH per lane); the per-grid sum is a wave-shuffle + LDS reduce with a single
//     s_barrier.  Hundreds of independent passes fill the 256 CUs -- no inter-workgroup traffic.
//   * per grid each lane streams 16 haplotype codes with one aligned 16-byte load of the uint8
//     `hapMatcherR` row (coalesced 1 KiB per wave instruction), gathers the emission of its code
//     from a <=256-entry fp32 table staged in LDS, and (dosage passes) checkpoints alpha to HBM in
//     a lane-interleaved order that makes every store a fully coalesced 1 KiB dwordx4 store.
//   * the backward pass re-reads that checkpoint, forms gamma = alpha * beta in registers, and
//     histograms gamma * sigma by haplotype code with fixed-point integer LDS atomics (32 bank-private
//     copies; ds_add_f32 is ~30x slower on gfx950); the histogram of grid g+1 is folded after the
//     block-sum barrier of grid g; the 32 x nMaxDH dosage mat-vec is off the serial path (k_dosage).
//   * the best-haplotype lists come from a second family of kernels with fp64 state (k_fwd64 / k_bwd64 /
//     k_topk<double>: register + LDS resident state, haplotype codes DMA'd global -> LDS): which of several
//     nearly tied haplotypes the reference (double arithmetic) reports decides the next small panel, and
//     fp32 state ranks them differently.  See the comment above those kernels.
//   * alpha/beta are renormalised every grid (the reference's always_normalize = TRUE semantics,
//     equivalent for dosage / gamma / sum(log c): test-unit-reference-single.R:588-642), which is
//     what makes fp32 state safe; emissions are built in fp64 and rounded once.
//
// Algorithmic HBM bytes: dosage pass 10 * K * G (1 B code + 4 B alpha store
// forward; 1 B code + 4 B alpha load backward), ranking pass (1 + 0.1 * 8) * 2 * K * G.
#include "fullpass_dev.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <utility>

namespace {

// ---------------------------------------------------------------------------------------------
// k_emat: emission tables.  One block per (grid, pass); thread d computes the fp64 emission of
// distinct word d-1 (reference-single.cpp:294-327), the block finds min / max, applies
// normalize_emissions (:985-990) and writes the table.  Row 0 is written as 0: haplotypes with code 0
// ("specials") get their own emission from `esp` (:1002-1042).
// The per-SNP factors of the words' products are formed once per block (they do not depend on the
// word) and the block's min / max are wave reductions plus one 4-entry exchange (min and max are
// exact: any order gives the same value) -- the first form of this kernel evaluated the factors per
// word and SNP and ran three 8-step LDS trees with a barrier per step.
// ---------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void k_emat(PassParams prm) {
    const int g = blockIdx.x, p = blockIdx.y, t = threadIdx.x;
    __shared__ double2 s_e[32];
    __shared__ double s_red[3][4];
    __shared__ int s_var;
    const int s = 32 * g;
    const int nLocal = min(32, prm.T - s);
    const double2 *gl = reinterpret_cast<const double2 *>(prm.gl) + (size_t)p * prm.T + s;
    if (t == 0) s_var = 0;
    __syncthreads();
    if (t < nLocal) {
        const double2 v = gl[t];   // x = P(reads | ref), y = P(reads | alt)
        const double eps = prm.ref_error, ome = 1.0 - eps;
        s_e[t] = make_double2(v.x * ome + v.y * eps, v.x * eps + v.y * ome);
        if (v.x != 1.0 || v.y != 1.0) s_var = 1;  // benign race: all writers store 1
    }
    __syncthreads();
    const bool has_variant = s_var != 0 || g == 1 || g == 0;
    TS *out = static_cast<TS *>(prm.emat) + ((size_t)p * prm.G + g) * kMaxRow;
    const int so = prm.sp_off[g], sn = prm.sp_off[g + 1] - so;
    // special emissions: the grid's list as it is, or (lazy: the fp64 ranking kernels) followed by 16 zero entries that
    // keep the zero-coded padding beyond K at 0 (fullpass64.hip)
    TS *esp_out = static_cast<TS *>(prm.esp) + (size_t)p * prm.esp_stride + so + (prm.lazy && sn > 0 ? 16 * prm.sp_gidx[g] : 0);
    if (prm.lazy && sn > 0 && t < 16) esp_out[sn + t] = TS(0);
    if (!has_variant) {
        // reference shortcut (:1078-1088): emission is 1 for every haplotype
        out[t] = t >= prm.nrow ? TS(0) : (t == 0) ? (sn > 0 ? TS(1) : TS(0)) : TS(1);
        for (int i = t; i < sn; i += 256) esp_out[i] = TS(1);
        if (prm.lazy && t == 0) prm.emin[(size_t)p * prm.G + g] = -1.0;   // "no variant": the forward takes the shortcut
        return;  // (never taken for g == 0)
    }
    const bool row = t >= 1 && t < prm.nrow;
    double e = 0;
    if (row) e = word_emission((uint32_t)prm.B[(size_t)g * prm.nMaxDH + (t - 1)], s_e, nLocal);
    // min over rows 1..nMaxDH (row 0 of the reference's table starts at 1 and takes the min), then the max with row 0 in it
    const int wv = t >> 6;
    const double wmin = wave_min(row ? e : 1.0);
    if ((t & 63) == 0) s_red[0][wv] = wmin;
    __syncthreads();
    const double row0 = fmin(fmin(s_red[0][0], s_red[0][1]), fmin(s_red[0][2], s_red[0][3]));
    const double wmax = wave_max(row ? e : row0);
    if ((t & 63) == 0) s_red[1][wv] = wmax;
    __syncthreads();
    const double emax = fmax(fmax(s_red[1][0], s_red[1][1]), fmax(s_red[1][2], s_red[1][3]));
    double scale = 1.0, sp_scale = 1.0;
    if (prm.normalize_emissions) {
        if (emax < 1.0) scale = 1.0 / emax;
        sp_scale = 1.0 / emax;  // specials are always divided by emission_max (:1034)
    }
    if (g == 0) {
        if (prm.lazy) {
            scale = sp_scale = 1.0;   // the reference initialises alpha(0) from the raw emissions (:2314-2347)
        } else {
            // state that cannot hold the raw emissions: scale by 1 / max here and fold the factor back into c[0] in k_fwd
            scale = sp_scale = 1.0 / emax;
        }
        if (t == 0) prm.escale0[p] = scale;
    }
    // row 0 (code 0): 0 normally, so the zero padding K..Kq and any stray code drop out; 1 on grids
    // that hold specials, whose own emission `esp` is applied by the kernels' rare path
    out[t] = t >= prm.nrow ? TS(0) : (t == 0) ? (sn > 0 ? TS(1) : TS(0)) : (TS)(e * scale);   // kMaxRow == blockDim
    double sp_min = 2.0;
    for (int i = t; i < sn; i += 256) {
        double es = word_emission(prm.sp_word[so + i], s_e, nLocal) * sp_scale;
        esp_out[i] = (TS)es;
        sp_min = fmin(sp_min, es);
    }
    if (prm.lazy) {
        // min_emission_prob of the grid (:1044-1057): the smallest table row after normalisation (row 0 holds the column
        // minimum, the initial 1 included), then the specials
        const double wsp = wave_min(sp_min);
        if ((t & 63) == 0) s_red[2][wv] = wsp;
        __syncthreads();
        if (t == 0) {
            const double m = fmin(row0 * scale, fmin(fmin(s_red[2][0], s_red[2][1]), fmin(s_red[2][2], s_red[2][3])));
            prm.emin[(size_t)p * prm.G + g] = m;
            if (g == 1) prm.emin_b1[p] = s_var != 0 ? m : -1.0;   // the backward pass does not force grid 1 (:1866-1877)
        }
    }
}

// Multiply the state of the haplotypes with code 0 ("specials": grids with more than nMaxDH distinct
// words) by their own emission (reference-single.cpp:1002-1042 / :1902-1964).  Row 0 of the LDS table
// is 1 on such grids, so x holds the un-emitted value.  A chunk's specials are consecutive entries of
// the grid's ascending list: one lower_bound per chunk that holds a zero code, then in order.
template <typename TS, int NCH>
__device__ __forceinline__ void apply_special_emissions(TS (&x)[NCH][16], const uint4 (&dh)[NCH], const PassParams &prm,
                                                        const TS *esp, int g, int NT, int t) {
    const int lo = prm.sp_off[g], hi = prm.sp_off[g + 1];
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        if (!any_zero_code(dh[j])) continue;
        const int k0 = (j * NT + t) * 16;
        if (k0 >= prm.K) continue;
        int at = special_lower_bound(prm.sp_k, lo, hi, k0);
        const uint32_t w[4] = {dh[j].x, dh[j].y, dh[j].z, dh[j].w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t code = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
            if (code == 0 && k0 + i < prm.K) {
                x[j][i] *= esp[at];
                at++;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_fwd: forward recursion.  alpha_k <- (alpha_k + psi/sigma) * e_k, renormalised every grid
// (reference-single.cpp:935-1107 with always_normalize).  One block per pass.
// ---------------------------------------------------------------------------------------------
template <typename TS, int NCH, int MAXT>
__global__ __launch_bounds__(MAXT) void k_fwd(PassParams prm) {
    using V = typename Vec<TS>::V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TS *etab = reinterpret_cast<TS *>(smem);                                     // [2][256]
    double *red = reinterpret_cast<double *>(smem + 2 * kMaxRow * sizeof(TS));   // [2][16]
    const int p = blockIdx.x, t = threadIdx.x, NT = blockDim.x, nwaves = NT >> 6;
    const int K = prm.K, G = prm.G;
    const int flags = prm.flags[p];
    const bool store_all = (flags & 15) != 0;
    const TS *emat = static_cast<const TS *>(prm.emat) + (size_t)p * G * kMaxRow;
    const TS *esp = static_cast<const TS *>(prm.esp) + (size_t)p * prm.esp_stride;
    V *aout = reinterpret_cast<V *>(static_cast<TS *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride);
    const int32_t *slot = prm.alpha_slot + (size_t)p * G;
    const size_t col_vecs = (size_t)prm.Kq / Vec<TS>::EPV;

    TS a[NCH][16];
    uint4 dh[NCH];
    const TS invK = TS(1) / (TS)K;
    // Chunks past K carry 0 and must stay 0: their codes are 0 (never loaded), so they see a finite table entry, and the
    // additive term is switched off per chunk (one select per chunk instead of a compare per element).  `tail_star` < 16
    // marks the one lane whose chunk (row jstar) straddles K.
    uint32_t valid = 0;
    const int jstar = (K / 16) / NT;
    const int tail_star = ((K / 16) % NT == t && (K & 15)) ? (K & 15) : 16;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int k0 = (j * NT + t) * 16;
        valid |= (k0 < K ? 1u : 0u) << j;
#pragma unroll
        for (int i = 0; i < 16; i++) a[j][i] = (k0 + i < K) ? invK : TS(0);
        dh[j] = make_uint4(0, 0, 0, 0);
        if (k0 < K) dh[j] = *reinterpret_cast<const uint4 *>(prm.hm + k0);
    }
    for (int i = t; i < kMaxRow; i += NT) etab[i] = (i < prm.nrow) ? emat[i] : TS(0);
    __syncthreads();

    for (int g = 0; g < G; g++) {
        const TS *et = etab + (g & 1) * kMaxRow;
        TS etn[4] = {0, 0, 0, 0};  // next grid's table: rows t, t+NT, .. (NT >= 64, nrow <= 256)
        if (g + 1 < G) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (t + r * NT < prm.nrow) etn[r] = emat[(size_t)(g + 1) * kMaxRow + t + r * NT];
        }
        double sg = 0.0, sig = 1.0;
        if (g > 0) {
            sig = prm.sigma[g - 1];
            sg = (1.0 - sig) / (double)K / sig;  // psi / sigma with A_prev = 1
        }
        const TS s = (TS)sg;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            const uint32_t w[4] = {dh[j].x, dh[j].y, dh[j].z, dh[j].w};
            const TS sj = ((valid >> j) & 1u) ? s : TS(0);
            TS e[16];   // the 16 table look-ups first, back to back, then the arithmetic
#pragma unroll
            for (int i = 0; i < 16; i++) e[i] = et[(w[i >> 2] >> ((i & 3) * 8)) & 0xffu];
#pragma unroll
            for (int i = 0; i < 16; i++) a[j][i] = (a[j][i] + sj) * e[i];
        }
        if (prm.sp_off[g + 1] > prm.sp_off[g]) apply_special_emissions<TS, NCH>(a, dh, prm, esp, g, NT, t);
        TS psum = 0;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            if (j == jstar && tail_star < 16) {   // one lane of the block
#pragma unroll
                for (int i = 0; i < 16; i++) a[j][i] = i < tail_star ? a[j][i] : TS(0);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) psum += a[j][i];
        }
        if (g + 1 < G) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (t + r * NT < kMaxRow) etab[((g + 1) & 1) * kMaxRow + t + r * NT] = etn[r];
        }
        const double A = block_sum((double)psum, red + (g & 1) * 16, t, nwaves);
        // the codes of the next grid: issued as early as the registers allow (temporaries are dead)
        if (g + 1 < G) {
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const int k0 = (j * NT + t) * 16;
                if (k0 < K) dh[j] = *reinterpret_cast<const uint4 *>(prm.hm + (size_t)(g + 1) * prm.Kp + k0);
            }
        }
        const double invA = 1.0 / A;
        const TS inv = (TS)invA;
        if (t == 0) prm.c[(size_t)p * G + g] = (g == 0) ? invA * prm.escale0[p] : invA / sig;
        const int sl = store_all ? g : slot[g];
#pragma unroll
        for (int j = 0; j < NCH; j++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[j][i] *= inv;
            if (sl >= 0 && (j * NT + t) * 16 < K) store_chunk<TS>(aout + (size_t)sl * col_vecs, a[j], j, NT, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_bwd: backward recursion; FULL adds gamma + the code histogram of the dosage passes; beta at the
// thinned grids goes to k_topk.  reference-single.cpp:1854-2177.
// ---------------------------------------------------------------------------------------------
template <typename TS, int NCH, int MAXT, bool FULL>
__global__ __launch_bounds__(MAXT) void k_bwd(PassParams prm) {
    using V = typename Vec<TS>::V;
    constexpr int EPV = Vec<TS>::EPV, NV = 16 / EPV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TS *etab = reinterpret_cast<TS *>(smem);                                     // [2][256]
    double *red = reinterpret_cast<double *>(smem + 2 * kMaxRow * sizeof(TS));   // [2][16]
    using Hist = typename Vec<TS>::Hist;   // uint32 at 2^-31 (fp32 state) / uint64 at 2^-62 (fp64 state)
    Hist *hist = reinterpret_cast<Hist *>(smem + 2 * kMaxRow * sizeof(TS) + 2 * 16 * 8);  // [2][kMaxRow][32] (FULL)
    const int p = blockIdx.x, t = threadIdx.x, NT = blockDim.x, nwaves = NT >> 6;
    const int lane = t & 63;
    const int K = prm.K, G = prm.G;
    const int flags = prm.flags[p];
    const bool want_dosage = FULL && (flags & 1) != 0;
    const bool store_all = (flags & 15) != 0;
    const bool want_gamma = FULL && (flags & 4) != 0, want_beta = FULL && (flags & 8) != 0;
    const TS *emat = static_cast<const TS *>(prm.emat) + (size_t)p * G * kMaxRow;
    const TS *esp = static_cast<const TS *>(prm.esp) + (size_t)p * prm.esp_stride;
    const V *ain = reinterpret_cast<const V *>(static_cast<const TS *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride);
    const int32_t *slot = prm.alpha_slot + (size_t)p * G;
    const size_t col_vecs = (size_t)prm.Kq / EPV;
    const double *cvec = prm.c + (size_t)p * G;

    TS b[NCH][16];
    uint4 dh[NCH];  // codes of grid g+1 during the emission phase, then reloaded with grid g's
    uint32_t valid = 0;   // see k_fwd
    const int jstar = (K / 16) / NT;
    const int tail_star = ((K / 16) % NT == t && (K & 15)) ? (K & 15) : 16;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int k0 = (j * NT + t) * 16;
        valid |= (k0 < K ? 1u : 0u) << j;
#pragma unroll
        for (int i = 0; i < 16; i++) b[j][i] = (k0 + i < K) ? TS(1) : TS(0);
        dh[j] = make_uint4(0, 0, 0, 0);
    }
    // fold one grid's 32 bank-private histogram copies into prm.mg (thread = code; copies visited in a rotated order
    // so that a wave's reads spread over the banks), re-zeroing them
    auto fold_histogram = [&](int g_of) {
        Hist *h = hist + (g_of & 1) * kMaxRow * kHistCopies;
        Hist *mgo = static_cast<Hist *>(prm.mg) + ((size_t)p * G + g_of) * kMaxRow;
        for (int code = t; code < prm.nrow; code += NT) {
            Hist v = 0;
#pragma unroll 8
            for (int c = 0; c < kHistCopies; c++) {
                const int at = code * kHistCopies + ((c + lane) & (kHistCopies - 1));
                v += h[at];
                h[at] = 0;
            }
            mgo[code] = v;
        }
    };
    if (want_dosage)
        for (int i = t; i < 2 * kMaxRow * kHistCopies; i += NT) hist[i] = 0;
    for (int i = t; i < kMaxRow; i += NT)
        etab[((G - 1) & 1) * kMaxRow + i] = (i < prm.nrow) ? emat[(size_t)(G - 1) * kMaxRow + i] : TS(0);
    __syncthreads();

    for (int g = G - 1; g >= 0; --g) {
        double sig = 1.0;  // "not_jump_prob" of the reference: sigma_g, or 1 at the last grid
        if (g < G - 1) {
            TS etn[4] = {0, 0, 0, 0};  // table of grid g (the emission side of iteration g-1)
            if (g > 0) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (t + r * NT < prm.nrow) etn[r] = emat[(size_t)g * kMaxRow + t + r * NT];
            }
            sig = prm.sigma[g];
            const TS *et = etab + ((g + 1) & 1) * kMaxRow;
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const uint32_t w[4] = {dh[j].x, dh[j].y, dh[j].z, dh[j].w};
                TS e[16];   // look-ups first, back to back
#pragma unroll
                for (int i = 0; i < 16; i++) e[i] = et[(w[i >> 2] >> ((i & 3) * 8)) & 0xffu];
#pragma unroll
                for (int i = 0; i < 16; i++) b[j][i] *= e[i];
            }
            if (prm.sp_off[g + 2] > prm.sp_off[g + 1]) apply_special_emissions<TS, NCH>(b, dh, prm, esp, g + 1, NT, t);
            TS psum = 0;
#pragma unroll
            for (int j = 0; j < NCH; j++)
#pragma unroll
                for (int i = 0; i < 16; i++) psum += b[j][i];
            // codes of grid g: histogram side below (dosage passes), emission side of the next iteration.
            // Issued here, after the last use of grid g+1's codes, so one register set serves both.
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const int k0 = (j * NT + t) * 16;
                if (k0 < K) dh[j] = *reinterpret_cast<const uint4 *>(prm.hm + (size_t)g * prm.Kp + k0);
            }
            // buffer (g & 1) was last read in iteration g + 1, before that iteration's barrier
            if (g > 0) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (t + r * NT < kMaxRow) etab[(g & 1) * kMaxRow + t + r * NT] = etn[r];
            }
            const double S = block_sum((double)psum, red + (g & 1) * 16, t, nwaves);
            // every wave is past its atomics of grid g+1 (they precede this barrier), and none reaches the atomics of
            // grid g-1 (same buffer) before the next barrier: fold grid g+1 here, without a barrier of its own
            if constexpr (FULL) {
                if (want_dosage && (store_all || slot[g + 1] >= 0)) fold_histogram(g + 1);
            }
            const TS add = (TS)((1.0 - sig) / (double)K / sig * S);
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const TS aj = ((valid >> j) & 1u) ? add : TS(0);
#pragma unroll
                for (int i = 0; i < 16; i++) b[j][i] += aj;
                if (j == jstar && tail_star < 16) {   // one lane of the block
#pragma unroll
                    for (int i = 0; i < 16; i++) b[j][i] = i < tail_star ? b[j][i] : TS(0);
                }
            }
        }

        if (g == G - 1) {
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const int k0 = (j * NT + t) * 16;
                if (k0 < K) dh[j] = *reinterpret_cast<const uint4 *>(prm.hm + (size_t)g * prm.Kp + k0);
            }
        }
        const int tcol = prm.thin_col[g];
        if constexpr (FULL) {
            const int sl = store_all ? g : slot[g];
            const bool need_gamma = (want_dosage || want_gamma) && sl >= 0;
            if (need_gamma) {
                const V *src = ain + (size_t)sl * col_vecs;
                const bool has_sp = prm.sp_off[g + 1] > prm.sp_off[g];
                Hist *h = hist + (g & 1) * kMaxRow * kHistCopies;
                const TS fs = (TS)sig;
                // alpha is streamed through a 2-chunk register pipeline: chunk j+1 is in flight while chunk
                // j is consumed (holding all 16 * NCH values would double the register footprint)
                V av[2][NV];
                if (t * 16 < K) {
#pragma unroll
                    for (int q = 0; q < NV; q++) av[0][q] = src[alpha_vec_index<NV>(0, q, NT, t)];
                }
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    const int k0 = (j * NT + t) * 16;
                    if (j + 1 < NCH && ((j + 1) * NT + t) * 16 < K) {
#pragma unroll
                        for (int q = 0; q < NV; q++) av[(j + 1) & 1][q] = src[alpha_vec_index<NV>(j + 1, q, NT, t)];
                    }
                    if (k0 < K) {
                        const uint32_t w[4] = {dh[j].x, dh[j].y, dh[j].z, dh[j].w};
                        int sp_at = 0;
                        if (want_dosage && has_sp && any_zero_code(dh[j]))
                            sp_at = special_lower_bound(prm.sp_k, prm.sp_off[g], prm.sp_off[g + 1], k0);
#pragma unroll
                        for (int q = 0; q < NV; q++) {
                            TS gq[EPV];
#pragma unroll
                            for (int r = 0; r < EPV; r++) {
                                const int i = EPV * q + r;
                                const TS gk = vget(av[j & 1][q], r) * b[j][i];  // 0 beyond K: b is 0 there
                                gq[r] = gk * fs;
                                if (want_dosage) {
                                    const uint32_t code = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
                                    if constexpr (sizeof(TS) == 8)
                                        atomicAdd(&h[code * kHistCopies + (lane & 31)], (Hist)((double)gq[r] * kHistScale64 + 0.5));
                                    else
                                        atomicAdd(&h[code * kHistCopies + (lane & 31)], (Hist)((float)gq[r] * kHistScale + 0.5f));
                                    if (has_sp && code == 0 && k0 + i < K) {
                                        // gamma of a special haplotype goes to its own list (:2096-2128)
                                        static_cast<TS *>(prm.gsp)[(size_t)p * prm.n_special + sp_at] = gk;
                                        sp_at++;
                                    }
                                }
                            }
                            if (want_gamma) {
                                V *dst = reinterpret_cast<V *>(prm.gamma_out) + ((size_t)p * G + g) * col_vecs;
                                dst[alpha_vec_index<NV>(j, q, NT, t)] = vmake(gq);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (tcol >= 0 && prm.beta_thin) {
            // thinned grid: hand the (unscaled) beta column to k_topk, which forms gamma = alpha * beta
            // and picks the top matches off the serial path (:2020-2031)
            V *dst = reinterpret_cast<V *>(prm.beta_thin) + ((size_t)p * prm.n_thin + tcol) * col_vecs;
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                if ((j * NT + t) * 16 >= K) continue;
                store_chunk<TS>(dst, b[j], j, NT, t);
            }
        }
        // beta *= c_g * sigma_g   (:2165-2166)
        const TS x = (TS)(cvec[g] * sig);
#pragma unroll
        for (int j = 0; j < NCH; j++) {
#pragma unroll
            for (int i = 0; i < 16; i++) b[j][i] *= x;
        }
        if (want_beta) {
            V *dst = reinterpret_cast<V *>(prm.beta_out) + ((size_t)p * G + g) * col_vecs;
#pragma unroll
            for (int j = 0; j < NCH; j++) store_chunk<TS>(dst, b[j], j, NT, t);
        }
    }
    if constexpr (FULL) {
        if (want_dosage && (store_all || slot[0] >= 0)) {
            __syncthreads();
            fold_histogram(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_topk: best haplotypes at one thinned grid of one pass (reference-single.cpp:129-194, :2020-2031).
// threshold = K_top-th largest gamma counted with multiplicity; every k with gamma >= threshold is
// reported with gamma * not_jump_prob.  One block per (thinned column, pass); alpha and beta columns
// are in the same lane-interleaved order, so gamma is an elementwise product of two coalesced streams.
// ---------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void k_topk(PassParams prm, int NT) {
    using V = typename Vec<TS>::V;
    constexpr int EPV = Vec<TS>::EPV, NV = 16 / EPV;
    // one block per (thinned column, pass), or per entry of the to-do list (the grids the fused picker of the fp64
    // ranking kernels handed over)
    const int tcol = prm.topk_todo ? prm.topk_todo[2 * blockIdx.x + 1] : blockIdx.x;
    const int p = prm.topk_todo ? prm.topk_todo[2 * blockIdx.x] : blockIdx.y;
    const int t = threadIdx.x, lane = t & 63;
    __shared__ TS s_top[4][kMaxTop];
    __shared__ int s_cnt;
    __shared__ int s_g;
    if (t == 0) {
        s_cnt = 0;
        int g = 0;
        for (int i = 0; i < prm.G; i++) if (prm.thin_col[i] == tcol) g = i;
        s_g = g;
    }
    __syncthreads();
    const int g = s_g;
    const int flags = prm.flags[p];
    const int sl = (flags & 15) ? g : prm.alpha_slot[(size_t)p * prm.G + g];
    const size_t col_vecs = (size_t)prm.Kq / EPV;
    const V *av = reinterpret_cast<const V *>(static_cast<const TS *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride) + (size_t)sl * col_vecs;
    const V *bv = reinterpret_cast<const V *>(prm.beta_thin) + ((size_t)p * prm.n_thin + tcol) * col_vecs;
    const int Ktop = prm.K_top;
    const int nvec = prm.Kq / EPV;
    TS ltop[kMaxTop];
#pragma unroll
    for (int q = 0; q < kMaxTop; q++) ltop[q] = 0;
    auto k_of = [&](int v, int r) {
        const int j = v / (NT * NV), rem = v % (NT * NV);
        const int tt = (rem / (64 * NV)) * 64 + (rem & 63), q = (rem >> 6) % NV;
        return (j * NT + tt) * 16 + EPV * q + r;
    };
    auto gamma_of = [&](int v, TS (&gq)[EPV]) {
        const V a4 = av[v], b4 = bv[v];
#pragma unroll
        for (int r = 0; r < EPV; r++) gq[r] = vget(a4, r) * vget(b4, r);
    };
    for (int v = t; v < nvec; v += 256) {
        if (k_of(v, 0) >= prm.K) continue;
        TS gq[EPV];
        gamma_of(v, gq);
#pragma unroll
        for (int r = 0; r < EPV; r++) {
            TS x = (k_of(v, r) < prm.K) ? gq[r] : TS(0);
#pragma unroll
            for (int z = 0; z < kMaxTop; z++) {
                if (z < Ktop) {
                    const TS hi = ltop[z] > x ? ltop[z] : x;
                    x = ltop[z] > x ? x : ltop[z];
                    ltop[z] = hi;
                }
            }
        }
    }
    // wave merge: pop the maximum Ktop times
    for (int r = 0; r < Ktop; r++) {
        const TS m = wave_max(ltop[0]);
        const unsigned long long owners = __ballot(ltop[0] == m);
        const int first = __ffsll((long long)owners) - 1;
        if (lane == first) {
#pragma unroll
            for (int q = 0; q < kMaxTop - 1; q++) ltop[q] = ltop[q + 1];
            ltop[kMaxTop - 1] = 0;
        }
        if (lane == 0) s_top[t >> 6][r] = m;
    }
    __syncthreads();
    TS thr = 0;
    {
        TS head = 0;
        int pos = 0;
        if (lane < 4) head = s_top[lane][0];
        for (int r = 0; r < Ktop; r++) {
            const TS m = wave_max(head);
            thr = m;
            const unsigned long long owners = __ballot(lane < 4 && head == m);
            const int first = __ffsll((long long)owners) - 1;
            if (lane == first) {
                pos++;
                head = (pos < Ktop) ? s_top[lane][pos] : TS(0);
            }
        }
    }
    const TS fs = (g < prm.G - 1) ? (TS)prm.sigma[g] : TS(1);
    int32_t *oi = prm.top_idx + ((size_t)p * prm.n_thin + tcol) * prm.top_cap;
    TS *ov = static_cast<TS *>(prm.top_val) + ((size_t)p * prm.n_thin + tcol) * prm.top_cap;
    for (int v = t; v < nvec; v += 256) {
        if (k_of(v, 0) >= prm.K) continue;
        TS gq[EPV];
        gamma_of(v, gq);
#pragma unroll
        for (int r = 0; r < EPV; r++) {
            const int kk = k_of(v, r);
            if (kk < prm.K && gq[r] >= thr) {
                const int at = atomicAdd(&s_cnt, 1);
                if (at < prm.top_cap) { oi[at] = kk; ov[at] = gq[r] * fs; }
            }
        }
    }
    __syncthreads();
    const int n_all = s_cnt;
    if (n_all > prm.top_cap && prm.truncate_lists) {
        // More matches than the list holds (exact ties at the threshold, e.g. haplotypes identical over the
        // whole region).  Keep what the host logic can ever look at first: every gamma above the
        // threshold, then the tied ones by ascending haplotype -- i.e. the head of the ordered list.
        __shared__ int s_n;
        auto block_count = [&](int kb) {   // #matches with gamma > thr, or gamma == thr and k <= kb
            __syncthreads();
            if (t == 0) s_n = 0;
            __syncthreads();
            int mine = 0;
            for (int v = t; v < nvec; v += 256) {
                if (k_of(v, 0) >= prm.K) continue;
                TS gq[EPV];
                gamma_of(v, gq);
#pragma unroll
                for (int r = 0; r < EPV; r++) {
                    const int kk = k_of(v, r);
                    if (kk < prm.K && (gq[r] > thr || (gq[r] == thr && kk <= kb))) mine++;
                }
            }
            atomicAdd(&s_n, mine);
            __syncthreads();
            return s_n;
        };
        int lo = -1, hi = prm.K - 1;   // smallest kb with count(kb) >= top_cap (count(-1) < top_cap: fewer than K_top above thr)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (block_count(mid) >= prm.top_cap) hi = mid; else lo = mid;
        }
        const int kb = (block_count(lo) >= prm.top_cap) ? lo : hi;
        __syncthreads();
        if (t == 0) s_cnt = 0;
        __syncthreads();
        for (int v = t; v < nvec; v += 256) {
            if (k_of(v, 0) >= prm.K) continue;
            TS gq[EPV];
            gamma_of(v, gq);
#pragma unroll
            for (int r = 0; r < EPV; r++) {
                const int kk = k_of(v, r);
                if (kk < prm.K && (gq[r] > thr || (gq[r] == thr && kk <= kb))) {
                    const int at = atomicAdd(&s_cnt, 1);
                    if (at < prm.top_cap) { oi[at] = kk; ov[at] = gq[r] * fs; }
                }
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        prm.top_cnt[(size_t)p * prm.n_thin + tcol] = n_all;
        const int n = min(min(s_cnt, prm.top_cap), n_all);
        // order as everything_per_hap_rejig_haps does (functions.R:2161-2170): value descending, ties by
        // ascending haplotype (R's stable order() on the k-ascending list).  Lists are ~K_top long.
        if (n <= 64 && (n_all <= prm.top_cap || prm.truncate_lists)) {
            for (int i = 1; i < n; i++) {
                const int ki = oi[i];
                const TS vi = ov[i];
                int j = i - 1;
                while (j >= 0 && (ov[j] < vi || (ov[j] == vi && oi[j] > ki))) { oi[j + 1] = oi[j]; ov[j + 1] = ov[j]; j--; }
                oi[j + 1] = ki;
                ov[j + 1] = vi;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_dosage: dosage[32g + b] = sigma_g * ( sum_d IE[d, 32g+b] * mg[d] + sum_specials gamma * (bit ? 1-eps : eps) )
// (reference-single.cpp:2092-2139).
// One wave per four consecutive grids, a lane per (grid, pair of SNPs): the wave's 128 dosages are 1 KiB of consecutive
// memory, stored with one instruction (16 bytes per lane), a block's four waves write 4 KiB -- the rows usually lie in the
// caller's pinned buffer, across PCIe.  (The first form ran one block of 32 x 8 threads per grid, 256 bytes per block:
// 512 000 blocks per 256 passes, 8 ms per launch for 131 MB.)  The sums keep that form's order: the codes d = 1 + q,
// 9 + q, ... into partial sum q (q = 0..7), the partial sums added in order.
// ---------------------------------------------------------------------------------------------
constexpr int kDosageGridsPerBlock = 16;

template <typename TS>
__global__ __launch_bounds__(256) void k_dosage(PassParams prm) {
    using Hist = typename Vec<TS>::Hist;
    const int p = blockIdx.y;
    if (!(prm.flags[p] & 1)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * kDosageGridsPerBlock + wave * 4 + (lane >> 4);
    if (g >= prm.G) return;                 // (no barrier below)
    const int b0 = 2 * (lane & 15), b1 = b0 + 1;   // this lane's SNPs of the grid
    const int s = 32 * g;
    const int nLocal = min(32, prm.T - s);
    const Hist *mg = static_cast<const Hist *>(prm.mg) + ((size_t)p * prm.G + g) * kMaxRow;
    const double eps = prm.ref_error, ome = 1.0 - eps;
    const double unit = sizeof(TS) == 8 ? prm.hist_unit : 1.0 / (double)kHistScale;
    // per partial sum q: histogram part (already times sigma_g) of both SNPs, specials (raw gamma), the grid's mass
    double acc0[8], acc1[8], sp0[8], sp1[8], mass[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc0[q] = acc1[q] = sp0[q] = sp1[q] = mass[q] = 0;
    const int32_t *Bg = prm.B + (size_t)g * prm.nMaxDH;
    const double *IE0 = prm.IE ? prm.IE + (size_t)(s + min(b0, nLocal - 1)) * prm.nMaxDH : nullptr;
    const double *IE1 = prm.IE ? prm.IE + (size_t)(s + min(b1, nLocal - 1)) * prm.nMaxDH : nullptr;
    mass[0] += (double)mg[0] * unit;        // bin 0 holds the specials' share of the grid's total gamma * sigma
    for (int d0 = 1; d0 < prm.nrow; d0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int d = d0 + q;
            if (d < prm.nrow) {
                const double m = (double)mg[d] * unit;
                mass[(q + 1) & 7] += m;     // (d & 7: d0 = 1 mod 8)
                double ie0, ie1;
                if (prm.IE) {
                    ie0 = IE0[d - 1];
                    ie1 = IE1[d - 1];
                } else {
                    const uint32_t w = (uint32_t)Bg[d - 1];
                    ie0 = ((w >> b0) & 1u) ? ome : eps;
                    ie1 = ((w >> b1) & 1u) ? ome : eps;
                }
                acc0[q] += ie0 * m;
                acc1[q] += ie1 * m;
            }
        }
    }
    const int so = prm.sp_off[g], sn = prm.sp_off[g + 1] - so;
    for (int i0 = 0; i0 < sn; i0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = i0 + q;
            if (i < sn) {
                const double gk = (double)static_cast<const TS *>(prm.gsp)[(size_t)p * prm.n_special + so + i];
                const uint32_t w = prm.sp_word[so + i];
                sp0[q] += gk * (((w >> b0) & 1u) ? ome : eps);
                sp1[q] += gk * (((w >> b1) & 1u) ? ome : eps);
            }
        }
    }
    const double sig = (g < prm.G - 1) ? prm.sigma[g] : 1.0;
    double tot0 = 0, tot1 = 0, ms = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        tot0 += acc0[q] + sp0[q] * sig;
        tot1 += acc1[q] + sp1[q] * sig;
        ms += mass[q];
    }
    // gamma * sigma_g of a grid (what the histogram accumulates) sums to 1 (reference-single.cpp:2048-2091, :2170-2176;
    // the reference's test suite checks colSums(gamma_t) == 1).  Dividing by the mass actually accumulated removes the
    // common-mode drift of the state's scale (with fp32 state ~3e-8 per grid, 6e-5 over 2 000 grids), which is all
    // the dosage of a long region would otherwise inherit from 2 000 renormalisations in single precision.
    if (ms > 0) {
        tot0 /= ms;
        tot1 /= ms;
    }
    double *dst = prm.dosage + (size_t)p * prm.T + s + b0;
    if (b1 < nLocal && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        *reinterpret_cast<double2 *>(dst) = make_double2(tot0, tot1);
    } else {
        if (b0 < nLocal) dst[0] = tot0;
        if (b1 < nLocal) dst[1] = tot1;
    }
}

// ---------------------------------------------------------------------------------------------
// k_make_gl: per-label genotype likelihoods of a pass from the sample's reads and the current read
// labels: make_gl_from_u_bq (QUILT/R/reference-single.R:19-42) + Rcpp_make_gl_bound
// (reference-single.cpp:68-94).  One thread per (SNP, pass); the bases covering a SNP come from a
// per-sample SNP-major index (built once on the host), in read order, so the product order is the
// reference's.
// ---------------------------------------------------------------------------------------------
struct GlParams {
    int P, T;
    const int32_t *pass_sample;  // [P]
    const int32_t *pass_label;   // [P] 1-based
    const int32_t *pass_hoff;    // [P] offset of the pass's chain into H
    const int32_t *snp_ptr;      // [n_sample][T + 1] (local offsets)
    const int32_t *ent_off;      // [n_sample] offset into ent_*
    const int32_t *ent_read;     // read index (within the sample) of each base
    const int32_t *ent_bq;       // its signed base quality
    const int32_t *H;            // labels
    const double *pR_tab, *pA_tab;
    double minGLValue;
    double *gl;                  // [P][T][2]
};

__global__ __launch_bounds__(256) void k_make_gl(GlParams p) {
    const int t = blockIdx.x * 256 + threadIdx.x, pi = blockIdx.y;
    if (t >= p.T) return;
    const int s = p.pass_sample[pi], lab = p.pass_label[pi];
    const int32_t *sp = p.snp_ptr + (size_t)s * (p.T + 1);
    const int32_t *er = p.ent_read + p.ent_off[s], *eb = p.ent_bq + p.ent_off[s];
    const int32_t *H = p.H + p.pass_hoff[pi];
    double a = 1.0, b = 1.0;
    for (int i = sp[t]; i < sp[t + 1]; i++) {
        const int bq = eb[i];
        if (bq == 0 || H[er[i]] != lab) continue;
        const int ab = bq < 0 ? -bq : bq;
        a *= p.pR_tab[(bq > 0 ? 256 : 0) + ab];
        b *= p.pA_tab[(bq > 0 ? 256 : 0) + ab];
    }
    if (p.minGLValue > 0 && (a < p.minGLValue || b < p.minGLValue)) {
        if (a > b) { b = b / a; a = 1; if (b < p.minGLValue) b = p.minGLValue; }
        else       { a = a / b; b = 1; if (a < p.minGLValue) a = p.minGLValue; }
    }
    double2 *out = reinterpret_cast<double2 *>(p.gl) + (size_t)pi * p.T + t;
    *out = make_double2(a, b);
}

// un-permute a lane-interleaved [cols][Kq] matrix into a K x cols double matrix (column-major)
template <typename TS>
__global__ void k_unpermute(const TS *src, double *dst, int K, int Kq, int NT, int cols, size_t dst_ld) {
    constexpr int EPV = Vec<TS>::EPV, NV = 16 / EPV;
    const int col = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K || col >= cols) return;
    const int chunk = k >> 4, e = k & 15;
    const int j = chunk / NT, t = chunk % NT;
    const size_t vec = alpha_vec_index<NV>(j, e / EPV, NT, t);
    dst[(size_t)col * dst_ld + k] = (double)src[(size_t)col * Kq + vec * EPV + (e % EPV)];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct qa_panel::Scratch {
    qa::ABuf<double> gl, c, dosage, escale0, unperm, emin, emin_b1, spill, fw_add, fw_xs;
    qa::ABuf<char> emat, esp, alpha, gamma, beta, beta_thin, top_val, mg, gsp;   // fp32 or fp64 elements (the launch decides)
    qa::ABuf<int32_t> thin_col, flags, alpha_slot, top_cnt, top_idx;
    qa::DBuf<int32_t> todo;   // (grid, pass) pairs handed to k_topk: persistent, grow-only
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    explicit Scratch(qa::Arena *a) {
        gl.arena = c.arena = dosage.arena = escale0.arena = unperm.arena = emin.arena = emin_b1.arena = spill.arena = fw_add.arena = fw_xs.arena = a;
        emat.arena = esp.arena = alpha.arena = mg.arena = gsp.arena = gamma.arena = beta.arena = beta_thin.arena = top_val.arena = a;
        thin_col.arena = flags.arena = alpha_slot.arena = top_cnt.arena = top_idx.arena = a;
    }
    ~Scratch() {
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    }
};

void qa::drop_pass_scratch(qa_panel *p) {
    delete p->scratch;
    p->scratch = nullptr;
}

qa_panel::~qa_panel() {
    delete scratch;
    if (gibbs_stream) (void)hipStreamDestroy(gibbs_stream);
    if (pass_stream) (void)hipStreamDestroy(pass_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

namespace {

thread_local double g_timing[6] = {0, 0, 0, 0, 0, 0};   // emat, forward, backward, dosage, separate top-K, total

// Three families of kernels run a pass:
//   KIND_F32       fp32 state (k_fwd / k_bwd<float>): the dosage passes, any output
//   KIND_F64_RANK  fp64 state, the reference's lazy normalisation, fused top-K (fullpass64.hip): best-haplotype lists only
//   KIND_F64_DOS   fp64 state, the reference's lazy normalisation, alpha stored at every second grid (k_bwd64d re-forms the
//                  others: PassParams::fw_add), gamma histogram for the dosage
//                  (k_fwd64 + k_bwd64d, fullpass64.hip): the DOSAGE passes of qa_panel_set_dosage_precision(64)
//   KIND_F64_FULL  fp64 state through the generic kernels (k_fwd / k_bwd<double>, one wave per SIMD): any output in
//                  double (alphaHat_t / betaHat_t / gamma_t of the single-pass entry point in that mode); not tuned (it spills)
//   KIND_F64_REF   VALIDATION MODE (qa_panel_set_sum_order(panel, 1), fullpass_ref.hip): fp64 state, the reference's lazy
//                  normalisation, every K-wide sum added in the reference's order by one lane; any output; slow on purpose
enum PassKind { KIND_F32 = 0, KIND_F64_RANK = 1, KIND_F64_FULL = 2, KIND_F64_DOS = 3, KIND_F64_REF = 4 };
struct Geometry { int NT, NCH; PassKind kind; bool f64() const { return kind != KIND_F32; } };

// Register-resident geometry: NT threads (multiple of 64) x NCH chunks of 16 haplotypes per lane.  fp32 state:
// NT <= 512 so that each wave may use 256 VGPRs.  The smallest NCH that covers K gives the most waves; tiny panels
// still get >= 2 chunks per lane for ILP.  KIND_F64_RANK: 512 threads, chunk rows of 8192 haplotypes (fullpass64.hip).
// KIND_F64_FULL: 256 threads, NCH = ceil(K / 4096) rounded up to a built variant.
constexpr int kNchList32[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12};
constexpr int kNchList64[] = {1, 2, 4, 6, 8, 10, 12, 13, 14};
Geometry pick_geometry(int K, PassKind kind = KIND_F32) {
    const int chunks = (K + 15) / 16;
    if (kind == KIND_F64_RANK || kind == KIND_F64_DOS) {
        int nch = qa::fb64_chunks(K);
        if (kind == KIND_F64_DOS && nch && qa::fb64_dos_lds_bytes(K) > 160 * 1024) nch = 0;
        return {nch ? 512 : 0, nch, kind};
    }
    if (kind == KIND_F64_FULL || kind == KIND_F64_REF) {   // (the validation kernels write the generic kernels' layout)
        const int need = (K + 4095) / 4096;
        for (int nch : kNchList64) if (nch >= need) return {256, nch, kind};
        return {0, 0, kind};
    }
    auto fit = [&](int nch) -> int {
        int nt = (chunks + nch - 1) / nch;
        nt = std::max((nt + 63) / 64 * 64, 64);
        if (nt > 512) return 0;
        if (nch == 1 && chunks > 128) return 0;
        return nt;
    };
    for (int nch : kNchList32) if (int nt = fit(nch)) return {nt, nch, kind};
    return {0, 0, kind};
}


// device bytes one pass needs in run_passes (mirrors its carves, with alignment slack)
size_t pass_bytes(const qa_panel *pn, const Geometry &geo, int n_thin, bool stores_all, bool gamma, bool beta,
                  bool device_gl) {
    const size_t Kq = (size_t)geo.NT * geo.NCH * 16, G = pn->G, T = pn->T, es = geo.f64() ? 8 : 4;
    const size_t cols = stores_all ? G : (size_t)std::max(n_thin, 1);
    (void)device_gl;
    const size_t nsp = (size_t)pn->n_special + 16 * (size_t)pn->n_sp_grids + 16;
    const size_t col = geo.kind == KIND_F64_DOS ? qa::fb64_alpha_col_elems(pn->K) : Kq;
    size_t b = T * 16 + G * 4 + G * kMaxRow * (es + 8) + 8 + nsp * (es + 8) + cols * col * es + G * 16 + T * 8;
    if (gamma) b += G * Kq * es;
    if (beta) b += G * Kq * es;
    if (n_thin > 0) b += (size_t)n_thin * Kq * es + (size_t)n_thin * (4 + 8 + 64 * 12);   // (lists of up to 64 entries)
    if (geo.kind == KIND_F64_RANK || geo.kind == KIND_F64_DOS) b += (size_t)qa::fb64_spill_rows(pn->K) * 8192 * 8;   // streamed chunk rows
    if (geo.kind == KIND_F64_REF) b += qa::fb_ref_state_doubles((int)Kq) * 8;   // state + gamma column in k order
    return b + 256 * 24;
}

// how many homogeneous passes fit, and make the arena big enough for them
int plan_chunk(qa_panel *pn, size_t per_pass, int remaining) {
    const size_t fixed = (size_t)pn->G * 8 + ((size_t)2 << 20);
    const size_t budget = pn->plan_budget();
    long n = budget > fixed ? (long)((budget - fixed) / per_pass) : 0;
    n = std::max<long>(1, std::min<long>(n, remaining));
    // One pass is one workgroup and a compute unit holds one such workgroup: a launch runs in rounds of n_cu passes.  When
    // the passes have to be split anyway, split at whole rounds (292 passes cost two rounds, 256 + 36, like 512 would).
    static const int n_cu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return std::max(prop.multiProcessorCount, 1);
    }();
    if (n < remaining && n > n_cu) n = n / n_cu * n_cu;
    pn->require_scratch(fixed + (size_t)n * per_pass);
    pn->A().reset();
    return (int)n;
}

template <typename TS, int NCH, int MAXT, bool FULL>
void launch_fb(const PassParams &prm, int NT, hipStream_t s, hipEvent_t e_mid) {
    const size_t lds_f = 2 * kMaxRow * sizeof(TS) + 2 * 16 * 8;
    const size_t lds_b = lds_f + (FULL ? (size_t)2 * kMaxRow * kHistCopies * sizeof(typename Vec<TS>::Hist) : 0);
    hipLaunchKernelGGL((k_fwd<TS, NCH, MAXT>), dim3(prm.P), dim3(NT), lds_f, s, prm);
    QA_HIP(hipGetLastError());
    QA_HIP(hipEventRecord(e_mid, s));
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bwd<TS, NCH, MAXT, FULL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    hipLaunchKernelGGL((k_bwd<TS, NCH, MAXT, FULL>), dim3(prm.P), dim3(NT), lds_b, s, prm);
    QA_HIP(hipGetLastError());
}

void launch_fb_any(const Geometry &geo, const PassParams &prm, hipStream_t st, hipEvent_t e_mid) {
    if (geo.kind == KIND_F64_DOS) {
        qa::launch_fb64_dosage(&prm, st, e_mid);
        return;
    }
    if (geo.kind == KIND_F64_RANK) {
        qa::launch_fb64(&prm, st, e_mid);
        return;
    }
    if (geo.kind == KIND_F64_REF) {
        qa::launch_fb_ref(&prm, geo.NT, st, e_mid);
        return;
    }
    if (geo.kind == KIND_F64_FULL) {
        switch (geo.NCH) {
#ifndef QA_FAST_BUILD
            case 1: launch_fb<double, 1, 256, true>(prm, geo.NT, st, e_mid); break;
            case 4: launch_fb<double, 4, 256, true>(prm, geo.NT, st, e_mid); break;
            case 6: launch_fb<double, 6, 256, true>(prm, geo.NT, st, e_mid); break;
            case 8: launch_fb<double, 8, 256, true>(prm, geo.NT, st, e_mid); break;
            case 10: launch_fb<double, 10, 256, true>(prm, geo.NT, st, e_mid); break;
            case 12: launch_fb<double, 12, 256, true>(prm, geo.NT, st, e_mid); break;
            case 14: launch_fb<double, 14, 256, true>(prm, geo.NT, st, e_mid); break;
#endif
            case 2: launch_fb<double, 2, 256, true>(prm, geo.NT, st, e_mid); break;
            case 13: launch_fb<double, 13, 256, true>(prm, geo.NT, st, e_mid); break;
            default: throw std::runtime_error("geometry not built");
        }
        return;
    }
    switch (geo.NCH) {
#ifndef QA_FAST_BUILD
        case 1: launch_fb<float, 1, 512, true>(prm, geo.NT, st, e_mid); break;
        case 3: launch_fb<float, 3, 512, true>(prm, geo.NT, st, e_mid); break;
        case 4: launch_fb<float, 4, 512, true>(prm, geo.NT, st, e_mid); break;
        case 5: launch_fb<float, 5, 512, true>(prm, geo.NT, st, e_mid); break;
        case 6: launch_fb<float, 6, 512, true>(prm, geo.NT, st, e_mid); break;
        case 8: launch_fb<float, 8, 512, true>(prm, geo.NT, st, e_mid); break;
        case 10: launch_fb<float, 10, 512, true>(prm, geo.NT, st, e_mid); break;
        case 12: launch_fb<float, 12, 512, true>(prm, geo.NT, st, e_mid); break;
#endif
        case 2: launch_fb<float, 2, 512, true>(prm, geo.NT, st, e_mid); break;
        case 7: launch_fb<float, 7, 512, true>(prm, geo.NT, st, e_mid); break;
        default: throw std::runtime_error("geometry not built");
    }
}

// the kernels behind the dosage passes of a handle: fp32 state, or (qa_panel_set_dosage_precision(64)) the fp64 dosage
// kernels -- the generic fp64 kernels when K exceeds those kernels' on-chip capacity
PassKind dosage_kind(const qa_panel *pn) {
    if (pn->sum_order_ref) return KIND_F64_REF;
    // (panels beyond the fp32 kernels' 98 304 haplotypes: the fp64 dosage kernels, whose chunk rows past the seventh stream)
    if (!pn->dosage_fp64 && pick_geometry(pn->K, KIND_F32).NT) return KIND_F32;
    return pick_geometry(pn->K, KIND_F64_DOS).NT ? KIND_F64_DOS : KIND_F64_FULL;
}

// the kernels behind the best-haplotype lists of a handle with fp64 ranking
PassKind rank_kind(const qa_panel *pn) { return pn->sum_order_ref ? KIND_F64_REF : KIND_F64_RANK; }

struct BatchOut {
    double *dosage = nullptr;        // [P][T] (row p, or dosage_rows[p] when given)
    const int32_t *dosage_rows = nullptr;
    double *c = nullptr;             // [P][G]
    double *alphaHat_t = nullptr;    // single-pass API only
    double *betaHat_t = nullptr;
    double *gamma_t = nullptr;
    double *gammaSmall_t = nullptr;
    bool gamma_small_unscaled = false;  // gammaSmall_t without return_gamma_t: no sigma factor (:2170-2176)
    // best_haps_stuff_list of every (pass, thinned column), appended in pass-major order
    std::vector<std::vector<std::pair<int32_t, double>>> *lists = nullptr;
    bool truncate_lists = false;   // batched drivers: lists capped at top_cap entries (head of the ordered list)
    int top_cap = 64;              // initial (or, truncating, final) capacity per list
    // truncating drivers: the device arrays as they are, [P][n_thin][top_cap], instead of `lists`
    std::vector<int32_t> *flat_idx = nullptr;
    std::vector<double> *flat_val = nullptr;
    bool order_by_value = false;
    std::vector<int32_t> *true_counts = nullptr;   // untruncated list lengths  // lists ordered as everything_per_hap_rejig_haps wants (else ascending k)
};

// runs P passes; flags per pass as in PassParams
int run_passes(qa_panel *pn, int P, const double *gl, const int32_t *h_flags, const int32_t *thin_col_h,
               int K_top, int normalize_emissions, const BatchOut &out, PassKind kind = KIND_F32,
               int always_normalize = 0, double norm_threshold = 1e-100) {
    if (K_top > kMaxTop) {
        qa::set_error("K_top_matches = %d > %d not supported", K_top, kMaxTop);
        return QA_ERR_UNSUPPORTED;
    }
    const Geometry geo = pick_geometry(pn->K, kind);
    if (geo.NT == 0) {
        qa::set_error("K = %d exceeds the on-chip capacity of the %s full-pass kernels", pn->K,
                      kind == KIND_F32 ? "fp32" : kind == KIND_F64_RANK ? "fp64 ranking" : kind == KIND_F64_DOS ? "fp64 dosage" :
                      kind == KIND_F64_REF ? "reference-order validation" : "generic fp64");
        return QA_ERR_UNSUPPORTED;
    }
    const bool f64 = geo.f64();
    const size_t es = f64 ? 8 : 4;
    if (kind == KIND_F64_RANK)
        for (int p = 0; p < P; p++)
            if (h_flags[p] & 15) throw std::runtime_error("fp64 ranking passes carry no dosage / gamma / beta outputs");
    if (kind == KIND_F64_DOS) {
        for (int p = 0; p < P; p++)
            if (!(h_flags[p] & 1) || (h_flags[p] & 12)) throw std::runtime_error("fp64 dosage passes yield the dosage (and c) only");
        if (K_top > 0) throw std::runtime_error("fp64 dosage passes carry no best-haplotype lists (the ranking passes do)");
    }
    QA_HIP(hipSetDevice(pn->device));
    if (!pn->scratch) pn->scratch = new qa_panel::Scratch(&pn->A());
    auto &S = *pn->scratch;
    hipStream_t st = pn->pass_stream ? pn->pass_stream : pn->stream;
    for (auto &e : S.ev) if (!e) QA_HIP(hipEventCreate(&e));
    const int G = pn->G, T = pn->T, K = pn->K;
    const int Kq = geo.NT * geo.NCH * 16;
    int n_thin = 0;
    for (int g = 0; g < G; g++) if (thin_col_h[g] >= 0) n_thin = std::max(n_thin, thin_col_h[g] + 1);

    // alpha checkpoint slots: all grids for dosage / gamma / beta passes, thinned grids otherwise
    std::vector<int32_t> slot((size_t)P * G, -1);
    size_t max_cols = 0;
    bool any_gamma = false, any_beta = false;
    for (int p = 0; p < P; p++) {
        const int f = h_flags[p];
        size_t cols;
        if ((f & 15) && kind == KIND_F64_DOS) {   // every second column: k_bwd64d re-forms the odd grids' (PassParams::fw_add)
            for (int g = 0; g < G; g += 2) slot[(size_t)p * G + g] = g / 2;
            cols = (G + 1) / 2;
        } else if (f & 15) {
            for (int g = 0; g < G; g++) slot[(size_t)p * G + g] = g;
            cols = G;
        } else {
            int n = 0;
            for (int g = 0; g < G; g++) if (thin_col_h[g] >= 0) slot[(size_t)p * G + g] = n++;
            cols = std::max(n, 1);
        }
        max_cols = std::max(max_cols, cols);
        any_gamma |= (f & 4) != 0;
        any_beta |= (f & 8) != 0;
    }
    const bool any_top = n_thin > 0 && K_top > 0;
    // the fp64 ranking kernels pick the lists themselves when only the ordered head of each list is wanted (the driver)
    const bool fused = any_top && kind == KIND_F64_RANK && out.truncate_lists && out.top_cap <= 64;
    const size_t alpha_col = kind == KIND_F64_DOS ? qa::fb64_alpha_col_elems(K) : (size_t)Kq;
    const size_t alpha_stride = max_cols * alpha_col;
    const bool lazy = kind == KIND_F64_RANK || kind == KIND_F64_DOS || kind == KIND_F64_REF;
    const size_t esp_stride = (size_t)pn->n_special + (lazy ? 16 * (size_t)pn->n_sp_grids : 0) + 16;

    if (gl) {   // host gl; otherwise the caller has filled S.gl on the device already (k_make_gl)
        S.gl.ensure((size_t)P * T * 2);
        S.gl.upload(gl, (size_t)P * T * 2, st);
    }
    S.thin_col.ensure(G);
    S.thin_col.upload(thin_col_h, G, st);
    S.flags.ensure(P);
    S.flags.upload(h_flags, P, st);
    S.alpha_slot.ensure((size_t)P * G);
    S.alpha_slot.upload(slot.data(), (size_t)P * G, st);
    S.emat.ensure((size_t)P * G * kMaxRow * es);
    S.escale0.ensure(P);
    S.emin.ensure((size_t)P * G);
    S.emin_b1.ensure(P);
    S.esp.ensure((size_t)P * esp_stride * es);
    S.gsp.ensure(std::max<size_t>((size_t)P * pn->n_special, 1) * es);
    S.alpha.ensure((size_t)P * alpha_stride * es + ((size_t)1 << 20));   // (slack: k_bwd64d's idle lanes fetch a fixed line past a short column)
    S.c.ensure((size_t)P * G);
    if (kind == KIND_F64_DOS) { S.fw_add.ensure((size_t)P * G); S.fw_xs.ensure((size_t)P * G); }
    S.mg.ensure((size_t)P * G * kMaxRow * (f64 ? 8 : 4));
    S.dosage.ensure((size_t)P * T);
    if (any_gamma) S.gamma.ensure((size_t)P * G * Kq * es);
    if (any_beta) S.beta.ensure((size_t)P * G * Kq * es);
    if (any_top) S.beta_thin.ensure((size_t)P * n_thin * Kq * es);
    int top_cap = out.top_cap;
    S.top_cnt.ensure(std::max<size_t>((size_t)P * std::max(n_thin, 1), 1));
    const size_t spill_stride = kind == KIND_F64_REF ? qa::fb_ref_state_doubles(Kq)   // the validation kernels' state (when not in LDS)
                                : lazy ? (size_t)qa::fb64_spill_rows(K) * 8192 : 0;   // doubles per pass: chunk rows streamed through HBM
    if (spill_stride) S.spill.ensure((size_t)P * spill_stride);

    PassParams prm{};
    prm.hm = pn->hm.p; prm.B = pn->B.p; prm.sp_off = pn->sp_off.p; prm.sp_k = pn->sp_k.p;
    prm.sp_word = pn->sp_word.p; prm.sp_gidx = pn->sp_gidx.p; prm.sp_chunk_at = pn->sp_chunk_at.p;
    prm.sigma = pn->sigma.p; prm.tm1 = pn->tm1.p; prm.IE = pn->ie_derived ? nullptr : pn->IE.p;
    prm.K = K; prm.Kp = pn->Kp; prm.G = G; prm.T = T; prm.nMaxDH = pn->nMaxDH; prm.nrow = pn->nrow;
    prm.n_special = pn->n_special; prm.ref_error = pn->ref_error;
    prm.P = P; prm.gl = S.gl.p; prm.thin_col = S.thin_col.p; prm.n_thin = n_thin; prm.flags = S.flags.p;
    prm.normalize_emissions = normalize_emissions;
    prm.lazy = lazy ? 1 : 0; prm.always_normalize = always_normalize; prm.norm_threshold = norm_threshold;
    prm.grid0_left_to_right = pn->sum_order_grid0_ltr ? 1 : 0;
    prm.emin = S.emin.p; prm.emin_b1 = S.emin_b1.p; prm.esp_stride = (int)esp_stride;

    prm.spill = spill_stride ? S.spill.p : nullptr; prm.spill_pass_stride = spill_stride;
    prm.fw_add = kind == KIND_F64_DOS ? S.fw_add.p : nullptr; prm.fw_xs = kind == KIND_F64_DOS ? S.fw_xs.p : nullptr;
    prm.emat = S.emat.p; prm.esp = S.esp.p; prm.escale0 = S.escale0.p; prm.alpha = S.alpha.p; prm.alpha_slot = S.alpha_slot.p;
    prm.alpha_pass_stride = alpha_stride; prm.Kq = Kq; prm.alpha_col_elems = alpha_col;
    prm.hist_unit = kind == KIND_F64_DOS ? 1.0 / 2251799813685248.0 /* 2^-51: k_bwd64d */ : 1.0 / kHistScale64; prm.c = S.c.p; prm.mg = S.mg.p; prm.gsp = S.gsp.p;
    prm.gamma_out = any_gamma ? S.gamma.p : nullptr; prm.beta_out = any_beta ? S.beta.p : nullptr;
    // Dosage rows that go to consecutive rows of a qa_host_alloc buffer are written there by k_dosage itself (every element
    // once, 4 KiB of consecutive bytes per workgroup): the transfer rides under the kernel instead of following it.
    bool dosage_direct = false;
    if (out.dosage && P > 0) {
        const size_t r0 = out.dosage_rows ? (size_t)out.dosage_rows[0] : 0;
        dosage_direct = true;
        for (int p = 0; p < P && dosage_direct; p++)
            dosage_direct = (h_flags[p] & 1) && (out.dosage_rows ? (size_t)out.dosage_rows[p] : (size_t)p) == r0 + (size_t)p;
        dosage_direct = dosage_direct && qa::pinned_registry().covers(out.dosage + r0 * T, sizeof(double) * (size_t)P * T);
        if (dosage_direct) prm.dosage = out.dosage + r0 * T;
    }
    if (!dosage_direct) prm.dosage = S.dosage.p;
    prm.K_top = any_top ? K_top : 0;
    prm.beta_thin = any_top ? S.beta_thin.p : nullptr;
    prm.fused_topk = fused ? 1 : 0;
    prm.top_cap = top_cap;
    prm.truncate_lists = out.truncate_lists ? 1 : 0;
    if (any_top) {
        S.top_idx.ensure((size_t)P * n_thin * top_cap);
        S.top_val.ensure((size_t)P * n_thin * top_cap * es);
        prm.top_cnt = S.top_cnt.p; prm.top_idx = S.top_idx.p; prm.top_val = S.top_val.p;
    }

    QA_HIP(hipEventRecord(S.ev[0], st));
    if (f64) hipLaunchKernelGGL(k_emat<double>, dim3(G, P), dim3(256), 0, st, prm);
    else hipLaunchKernelGGL(k_emat<float>, dim3(G, P), dim3(256), 0, st, prm);
    QA_HIP(hipGetLastError());
    QA_HIP(hipEventRecord(S.ev[1], st));
    launch_fb_any(geo, prm, st, S.ev[2]);
    QA_HIP(hipEventRecord(S.ev[3], st));
    const dim3 dgrid((G + kDosageGridsPerBlock - 1) / kDosageGridsPerBlock, P);
    if (kind == KIND_F64_REF) { /* the validation backward kernel wrote the dosage itself, sums in the reference's order */ }
    else if (f64) hipLaunchKernelGGL(k_dosage<double>, dgrid, dim3(256), 0, st, prm);
    else hipLaunchKernelGGL(k_dosage<float>, dgrid, dim3(256), 0, st, prm);
    QA_HIP(hipGetLastError());
    QA_HIP(hipEventRecord(S.ev[4], st));
    std::vector<int32_t> cnt;
    int n_handed_over = 0;
    if (fused) {
        // the backward kernel wrote the lists; the grids whose candidates overflowed its LDS list (top_cnt = -1: ties)
        // left their beta column for k_topk
        cnt.resize((size_t)P * n_thin);
        S.top_cnt.download(cnt.data(), cnt.size(), st);
        std::vector<int32_t> todo;
        for (size_t i = 0; i < cnt.size(); i++)
            if (cnt[i] < 0) { todo.push_back((int32_t)(i / n_thin)); todo.push_back((int32_t)(i % n_thin)); }
        n_handed_over = (int)todo.size() / 2;
        if (n_handed_over) {
            S.todo.ensure(todo.size());
            S.todo.upload(todo.data(), todo.size(), st);
            prm.topk_todo = S.todo.p;
            hipLaunchKernelGGL(k_topk<double>, dim3(n_handed_over), dim3(256), 0, st, prm, geo.NT);
            QA_HIP(hipGetLastError());
            S.top_cnt.download(cnt.data(), cnt.size(), st);
            QA_HIP(hipStreamSynchronize(st));
            prm.topk_todo = nullptr;
        }
    }
    for (int attempt = 0; any_top && !fused && attempt < 2; attempt++) {
        prm.top_cap = top_cap;
        S.top_idx.ensure((size_t)P * n_thin * top_cap);
        S.top_val.ensure((size_t)P * n_thin * top_cap * es);
        prm.top_cnt = S.top_cnt.p; prm.top_idx = S.top_idx.p; prm.top_val = S.top_val.p;
        if (f64) hipLaunchKernelGGL(k_topk<double>, dim3(n_thin, P), dim3(256), 0, st, prm, geo.NT);
        else hipLaunchKernelGGL(k_topk<float>, dim3(n_thin, P), dim3(256), 0, st, prm, geo.NT);
        QA_HIP(hipGetLastError());
        cnt.resize((size_t)P * n_thin);
        S.top_cnt.download(cnt.data(), cnt.size(), st);
        QA_HIP(hipStreamSynchronize(st));
        int mx = 0;
        for (int32_t v : cnt) mx = std::max(mx, v);
        if (mx <= top_cap || out.truncate_lists) break;
        top_cap = mx;  // pathological ties (e.g. a label without reads): redo with room for all
    }
    QA_HIP(hipEventRecord(S.ev[5], st));
    QA_HIP(hipStreamSynchronize(st));
    float ms;
    for (int i = 0; i < 5; i++) {
        QA_HIP(hipEventElapsedTime(&ms, S.ev[i], S.ev[i + 1]));
        g_timing[i] = ms;
    }
    QA_HIP(hipEventElapsedTime(&ms, S.ev[0], S.ev[5]));
    g_timing[5] = ms;
    {
        // algorithmic HBM bytes of this launch set: per cell 1 B code + one state element
        // (4 B, or 8 B with fp64 state) on every stored column, forward (alpha) and backward (alpha re-read) alike;
        // the separate top-K picker re-reads alpha and beta at the thinned grids twice (outside the SURVEY contract:
        // it is what the fused picker of the fp64 ranking kernels removes); k_dosage reads the code histograms
        double cells_all = 0, cells_thin = 0, n_dos = 0;
        for (int p = 0; p < P; p++) {
            if (h_flags[p] & 15) cells_all += (double)K * G; else cells_thin += (double)K * G;
            if (h_flags[p] & 1) n_dos += 1;
        }
        const double frac = G > 0 ? (double)n_thin / G : 0;
        const double per_dir = cells_all * (1.0 + es) + cells_thin * (1.0 + es * frac);
        const double topk_bytes = (any_top && !fused) ? (double)P * n_thin * K * es * 4.0
                                                      : (double)n_handed_over * K * es * 4.0;
        const double t_e = qa::profile_clock_ms(S.ev[0]);
        double at = t_e;
        qa::profile_add(qa::PK_EMAT, g_timing[0], (double)P * T * 16.0 + (double)P * G * kMaxRow * es, at); at += g_timing[0];
        const int pk_f = kind == KIND_F32 ? qa::PK_FWD : kind == KIND_F64_RANK ? qa::PK_FWD64 : kind == KIND_F64_DOS ? qa::PK_FWD64D : qa::PK_FWD64G;
        const int pk_b = kind == KIND_F32 ? qa::PK_BWD : kind == KIND_F64_RANK ? qa::PK_BWD64 : kind == KIND_F64_DOS ? qa::PK_BWD64D : qa::PK_BWD64G;
        // (the fp64 dosage forward stores every second column: half of the state bytes; its backward reads each stored column twice)
        const double per_dir_fwd = kind == KIND_F64_DOS ? cells_all * (1.0 + es * 0.5) : per_dir;
        qa::profile_add(pk_f, g_timing[1], per_dir_fwd, at); at += g_timing[1];
        qa::profile_add(pk_b, g_timing[2], per_dir, at); at += g_timing[2];
        if (n_dos > 0) qa::profile_add(qa::PK_DOSAGE, g_timing[3], n_dos * G * kMaxRow * (f64 ? 8.0 : 4.0) + n_dos * T * 8.0, at);
        at += g_timing[3];
        if (topk_bytes > 0) qa::profile_add(qa::PK_TOPK, g_timing[4], topk_bytes, at);
    }

    // ---- copy results back
    if (out.c) S.c.download(out.c, (size_t)P * G, st);
    if (out.dosage && !dosage_direct) {
        // runs of consecutive dosage passes come back in one staged transfer each, then scatter to their rows
        // (a run whose destination rows are consecutive too -- the batch calls' layout -- lands in the caller's buffer
        // directly: one transfer, staged piecewise or, for a qa_host_alloc buffer, written by the copy kernel itself)
        const int max_rows = std::max<int>(1, (int)(qa::kStagePiece / (sizeof(double) * T)));
        auto row_of = [&](int p) { return (size_t)(out.dosage_rows ? out.dosage_rows[p] : p); };
        std::vector<double> tmp;
        for (int p = 0; p < P;) {
            if (!(h_flags[p] & 1)) { p++; continue; }
            int n = 1;
            while (p + n < P && (h_flags[p + n] & 1) && row_of(p + n) == row_of(p) + (size_t)n) n++;
            if (n > 1 || max_rows == 1) {
                qa::staged_download(out.dosage + row_of(p) * T, S.dosage.p + (size_t)p * T, sizeof(double) * T * n, st);
                p += n;
                continue;
            }
            while (p + n < P && n < max_rows && (h_flags[p + n] & 1)) n++;
            tmp.resize((size_t)n * T);
            qa::staged_download(tmp.data(), S.dosage.p + (size_t)p * T, sizeof(double) * T * n, st);
            for (int i = 0; i < n; i++)
                memcpy(out.dosage + row_of(p + i) * T, tmp.data() + (size_t)i * T, sizeof(double) * T);
            p += n;
        }
    }
    auto unpermute_to_host = [&](const char *base, size_t elem_off, int cols, double *dst) {
        S.unperm.ensure((size_t)K * cols);
        if (f64)
            hipLaunchKernelGGL(k_unpermute<double>, dim3((K + 255) / 256, cols), dim3(256), 0, st,
                               reinterpret_cast<const double *>(base) + elem_off, S.unperm.p, K, Kq, geo.NT, cols, (size_t)K);
        else
            hipLaunchKernelGGL(k_unpermute<float>, dim3((K + 255) / 256, cols), dim3(256), 0, st,
                               reinterpret_cast<const float *>(base) + elem_off, S.unperm.p, K, Kq, geo.NT, cols, (size_t)K);
        QA_HIP(hipGetLastError());
        S.unperm.download(dst, (size_t)K * cols, st);
        QA_HIP(hipStreamSynchronize(st));
    };
    if (P == 1) {
        if (out.alphaHat_t) {
            if (h_flags[0] & 15) {
                unpermute_to_host(S.alpha.p, 0, G, out.alphaHat_t);
            } else {
                // only column 0 and the thinned columns exist (reference-single.cpp:2264-2268)
                std::vector<double> col(K);
                for (int g = 0; g < G; g++) {
                    const int sl = slot[g];
                    if (sl < 0) continue;
                    unpermute_to_host(S.alpha.p, (size_t)sl * Kq, 1, col.data());
                    memcpy(out.alphaHat_t + (size_t)g * K, col.data(), sizeof(double) * K);
                }
            }
        }
        if (out.gamma_t && any_gamma) unpermute_to_host(S.gamma.p, 0, G, out.gamma_t);
        if (out.betaHat_t && any_beta) unpermute_to_host(S.beta.p, 0, G, out.betaHat_t);
        if (out.gammaSmall_t && any_gamma) {
            std::vector<double> col(K);
            for (int g = 0; g < G; g++) {
                if (thin_col_h[g] < 0) continue;
                unpermute_to_host(S.gamma.p, (size_t)g * Kq, 1, col.data());
                if (out.gamma_small_unscaled && g < G - 1)
                    for (int k = 0; k < K; k++) col[k] /= pn->h_sigma[g];
                memcpy(out.gammaSmall_t + (size_t)thin_col_h[g] * K, col.data(), sizeof(double) * K);
            }
        }
    }
    int status = QA_OK;
    if (any_top && out.true_counts) *out.true_counts = cnt;
    if (any_top && (out.lists || out.flat_idx)) {
        const size_t n = (size_t)P * n_thin;
        std::vector<int32_t> idx_local;
        std::vector<double> val_local;
        std::vector<int32_t> &idx = out.flat_idx ? *out.flat_idx : idx_local;
        std::vector<double> &val = out.flat_val ? *out.flat_val : val_local;
        idx.resize(n * top_cap);
        val.resize(n * top_cap);
        S.top_idx.download(idx.data(), idx.size(), st);
        if (f64) {
            qa::staged_download(val.data(), S.top_val.p, val.size() * 8, st);
        } else {
            std::vector<float> v32(val.size());
            qa::staged_download(v32.data(), S.top_val.p, v32.size() * 4, st);
            for (size_t i = 0; i < val.size(); i++) val[i] = v32[i];
        }
        for (size_t i = 0; out.lists && i < n; i++) {
            std::vector<std::pair<int32_t, double>> tmp;
            const int nq = std::min<int>(cnt[i], top_cap);
            tmp.reserve(nq);
            for (int q = 0; q < nq; q++) tmp.emplace_back(idx[i * top_cap + q], val[i * top_cap + q]);
            if (!out.order_by_value) {
                std::sort(tmp.begin(), tmp.end());  // ascending k, the reference's emission order
            } else if (nq > 64) {                    // (k_topk orders lists of up to 64 entries itself)
                std::sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) {
                    return a.second > b.second || (a.second == b.second && a.first < b.first);
                });
            }
            out.lists->push_back(std::move(tmp));
        }
    }
    QA_HIP(hipStreamSynchronize(st));
    return status;
}

}  // namespace

extern "C" {

int qa_last_fullpass_timing_ms(double out[5]) {
    if (!out) return QA_ERR_INVALID;
    for (int i = 0; i < 3; i++) out[i] = g_timing[i];
    out[3] = g_timing[3] + g_timing[4];
    out[4] = g_timing[5];
    return QA_OK;
}

// pack best-haps lists into the caller's CSR arrays
static int pack_lists(const std::vector<std::vector<std::pair<int32_t, double>>> &lists, int32_t *best_ptr,
                      int32_t *best_idx, double *best_val, int64_t best_cap) {
    if (!best_ptr) return QA_OK;
    int64_t total = 0;
    best_ptr[0] = 0;
    for (size_t i = 0; i < lists.size(); i++) {
        total += (int64_t)lists[i].size();
        best_ptr[i + 1] = (int32_t)total;
    }
    if (total > best_cap || !best_idx || !best_val) {
        qa::set_error("best-haps capacity %lld < needed %lld", (long long)best_cap, (long long)total);
        return QA_ERR_CAPACITY;
    }
    for (size_t i = 0; i < lists.size(); i++)
        for (size_t q = 0; q < lists[i].size(); q++) {
            best_idx[best_ptr[i] + q] = lists[i][q].first;
            best_val[best_ptr[i] + q] = (double)lists[i][q].second;
        }
    return QA_OK;
}

int qa_Rcpp_haploid_dosage_versus_refs(
    qa_panel_t *panel, const double *gl, const int32_t *gammaSmall_cols_to_get,
    const qa_fullpass_opts_t *o, double *alphaHat_t, double *betaHat_t, double *c, double *gamma_t,
    double *gammaSmall_t, double *dosage, int32_t *best_ptr, int32_t *best_idx, double *best_val,
    int64_t best_cap) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!panel || !gl || !o || !gammaSmall_cols_to_get) {
        qa::set_error("qa_Rcpp_haploid_dosage_versus_refs: null argument");
        return QA_ERR_INVALID;
    }
    // A C caller that zero-initialises the options would ask the lazily normalised fp64 passes never to renormalise between
    // grid 0 and the last grid: alpha underflows over a long region, 1 / run_total = inf, NaN dosages.  A threshold that is not
    // positive (or NaN) is therefore read as "not set" = the reference's default (reference-single.cpp:2216, 1e-100).
    const double norm_threshold = o->min_emission_prob_normalization_threshold > 0 ? o->min_emission_prob_normalization_threshold : 1e-100;
    return qa::guarded([&] {
        qa::GateHold hold;
        hold.acquire(panel->gate(), &panel->arena);
        const int G = panel->G;
        std::vector<int32_t> thin(G, -1);
        const bool use_thin = o->get_best_haps_from_thinned_sites || o->return_gammaSmall_t;
        if (use_thin) for (int g = 0; g < G; g++) thin[g] = gammaSmall_cols_to_get[g];
        int32_t f = 0;
        if (o->return_dosage) f |= 1;
        // reference-single.cpp:2264-2268: alpha is kept everywhere unless only thinned outputs are wanted
        const bool only_thin = use_thin && !o->return_gamma_t && !o->return_dosage && !o->return_betaHat_t;
        if (!only_thin) f |= 2;
        if (o->return_gamma_t || o->return_gammaSmall_t) f |= 4;
        if (o->return_betaHat_t) f |= 8;
        BatchOut out;
        out.dosage = o->return_dosage ? dosage : nullptr;
        out.c = c;
        out.alphaHat_t = alphaHat_t;
        out.betaHat_t = o->return_betaHat_t ? betaHat_t : nullptr;
        out.gamma_t = o->return_gamma_t ? gamma_t : nullptr;
        out.gammaSmall_t = o->return_gammaSmall_t ? gammaSmall_t : nullptr;
        out.gamma_small_unscaled = !o->return_gamma_t;
        std::vector<std::vector<std::pair<int32_t, double>>> lists;
        QA_HIP(hipSetDevice(panel->device));
        int nt = 0;
        for (int g = 0; g < G; g++) nt = std::max(nt, thin[g] + 1);
        const bool want_lists = o->get_best_haps_from_thinned_sites != 0;
        const int K_top = want_lists ? o->K_top_matches : 0;
        auto plan = [&](PassKind kind, int32_t flags) {
            const Geometry geo1 = pick_geometry(panel->K, kind);
            if (geo1.NT == 0) throw std::runtime_error("K exceeds the on-chip capacity of the full-pass kernels");
            const size_t need = pass_bytes(panel, geo1, nt, (flags & 15) != 0, (flags & 4) != 0, (flags & 8) != 0, false) +
                                (size_t)panel->K * G * 8 /* un-permute staging */;
            plan_chunk(panel, need, 1);
        };
        // fp64 dosage: the tuned kernels yield dosage and c; a call that also wants alpha / beta / gamma matrices takes the
        // generic fp64 kernels
        const bool matrices = alphaHat_t || o->return_betaHat_t || o->return_gamma_t || o->return_gammaSmall_t;
        const bool f32_fits = pick_geometry(panel->K, KIND_F32).NT != 0;
        const PassKind main_kind = panel->sum_order_ref ? KIND_F64_REF   // validation mode: one pass of the reference-order kernels yields everything
                                   : (!panel->dosage_fp64 && f32_fits) ? KIND_F32 : (matrices || !o->return_dosage) ? KIND_F64_FULL : dosage_kind(panel);
        if (pick_geometry(panel->K, main_kind).NT == 0 && !(want_lists && panel->rank_fp64 && only_thin)) {
            // K x nGrids outputs (alphaHat_t / betaHat_t / gamma_t / gammaSmall_t) come from kernels that keep the whole state
            // on chip; the dosage and the best-haplotype lists (what the driver path asks for) have no such limit
            qa::set_error("K = %d: alphaHat_t / betaHat_t / gamma_t / gammaSmall_t outputs are limited to K <= 57 344 haplotypes (state on "
                          "chip); dosage, c and best_haps_stuff_list are available for any K", panel->K);
            return QA_ERR_UNSUPPORTED;
        }
        int st;
        if (main_kind == KIND_F64_DOS) {
            std::vector<int32_t> no_thin(G, -1);
            const int32_t f1 = 1;
            plan(KIND_F64_DOS, f1);
            st = run_passes(panel, 1, gl, &f1, no_thin.data(), 0, o->normalize_emissions, out, KIND_F64_DOS,
                            o->always_normalize, norm_threshold);
            if (st == QA_OK && want_lists) {
                BatchOut out2;
                out2.lists = &lists;
                const int32_t f0 = 0;
                const PassKind rk = panel->rank_fp64 ? rank_kind(panel) : KIND_F32;
                plan(rk, 0);
                st = run_passes(panel, 1, gl, &f0, thin.data(), K_top, o->normalize_emissions, out2, rk,
                                o->always_normalize, norm_threshold);
            }
        } else if (want_lists && panel->rank_fp64 && only_thin) {
            // only the lists (and alpha at the thinned grids, c): the fp64 ranking pass, which follows the reference's
            // normalisation schedule (always_normalize / min_emission_prob_normalization_threshold honoured)
            out.lists = &lists;
            plan(rank_kind(panel), 0);
            st = run_passes(panel, 1, gl, &f, thin.data(), K_top, o->normalize_emissions, out, rank_kind(panel),
                            o->always_normalize, norm_threshold);
        } else if (want_lists && panel->rank_fp64 && main_kind == KIND_F32) {
            // the best-haplotype lists come from a pass with fp64 state, so that their membership and order are
            // the reference's; every other output comes from the fp32 pass
            plan(KIND_F32, f);
            st = run_passes(panel, 1, gl, &f, thin.data(), 0, o->normalize_emissions, out, KIND_F32);
            if (st != QA_OK) return st;
            BatchOut out2;
            out2.lists = &lists;
            const int32_t f0 = 0;
            plan(rank_kind(panel), 0);
            st = run_passes(panel, 1, gl, &f0, thin.data(), K_top, o->normalize_emissions, out2, rank_kind(panel),
                            o->always_normalize, norm_threshold);
        } else {
            // one pass yields everything: fp32 state with fp32 ranking, or fp64 state (qa_panel_set_dosage_precision(64))
            if (want_lists) out.lists = &lists;
            plan(main_kind, f);
            st = run_passes(panel, 1, gl, &f, thin.data(), K_top, o->normalize_emissions, out, main_kind,
                            o->always_normalize, norm_threshold);
        }
        if (st != QA_OK || !want_lists) return st;
        return pack_lists(lists, best_ptr, best_idx, best_val, best_cap);
    });
}

int qa_fullpass_batch(qa_panel_t *panel, int32_t n_pass, const double *gl, const int32_t *want_dosage,
                      const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches, double *dosage,
                      int32_t *best_ptr, int32_t *best_idx, double *best_val, int64_t best_cap) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!panel || !gl || n_pass <= 0 || !want_dosage || !gammaSmall_cols_to_get) {
        qa::set_error("qa_fullpass_batch: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        qa::GateHold hold;
        hold.acquire(panel->gate(), &panel->arena);
        std::vector<int32_t> f(n_pass);
        for (int i = 0; i < n_pass; i++) f[i] = want_dosage[i] ? 1 : 0;
        // chunk the passes so that the alpha checkpoints fit in HBM (K = 50 000, G = 2 000: 0.4 GB per dosage
        // pass, 0.08 GB per thin pass); chunks are homogeneous so that every pass of a chunk has the same footprint
        QA_HIP(hipSetDevice(panel->device));
        PassKind main_kind = dosage_kind(panel);
        if (main_kind == KIND_F64_DOS && !panel->rank_fp64 && K_top_matches > 0) main_kind = KIND_F64_FULL;   // (lists from the dosage pass itself)
        // fp64 ranking passes beside the dosage passes (fp32 state, or the fp64 dosage kernels); with the generic fp64 kernels,
        // or fp32 ranking, one pass yields both
        const bool exact = panel->rank_fp64 && K_top_matches > 0 && main_kind != KIND_F64_FULL;
        const Geometry geo = pick_geometry(panel->K, main_kind), geo64 = pick_geometry(panel->K, rank_kind(panel));
        if (geo.NT == 0 || (exact && geo64.NT == 0))
            throw std::runtime_error("K exceeds the on-chip capacity of the full-pass kernels");
        const int G = panel->G, T = panel->T;
        int n_thin = 0;
        for (int g = 0; g < G; g++) n_thin = std::max(n_thin, gammaSmall_cols_to_get[g] + 1);
        std::vector<std::vector<std::pair<int32_t, double>>> lists;
        std::vector<int32_t> zeros(n_pass, 0);
        int done = 0;
        int status = QA_OK;
        while (done < n_pass && status == QA_OK) {
            const bool dos = f[done] != 0;
            int run = 0;
            while (done + run < n_pass && (f[done + run] != 0) == dos) run++;
            if (!exact) {
                // (thin passes of the fp64-dosage mode still take the faster ranking kernels)
                const bool rank = !dos && panel->rank_fp64 && K_top_matches > 0 && geo64.NT != 0;
                const int n = plan_chunk(panel, pass_bytes(panel, rank ? geo64 : geo, n_thin, dos, false, false, false), run);
                BatchOut out;
                out.dosage = dosage ? dosage + (size_t)done * T : nullptr;
                out.lists = &lists;
                status = run_passes(panel, n, gl + (size_t)done * T * 2, f.data() + done, gammaSmall_cols_to_get,
                                    K_top_matches, 1, out, rank ? rank_kind(panel) : main_kind);
                done += n;
                continue;
            }
            // fp64-state ranking passes for the lists; dosage passes separately
            int n = plan_chunk(panel, pass_bytes(panel, geo64, n_thin, false, false, false, false), run);
            if (dos) n = std::min(n, plan_chunk(panel, pass_bytes(panel, geo, 0, true, false, false, false), run));
            if (dos) {
                BatchOut out;
                out.dosage = dosage ? dosage + (size_t)done * T : nullptr;
                std::vector<int32_t> no_thin(G, -1);
                status = run_passes(panel, n, gl + (size_t)done * T * 2, f.data() + done, no_thin.data(), 0, 1, out, main_kind);
                if (status != QA_OK) break;
                plan_chunk(panel, pass_bytes(panel, geo64, n_thin, false, false, false, false), n);
            }
            BatchOut out2;
            out2.lists = &lists;
            status = run_passes(panel, n, gl + (size_t)done * T * 2, zeros.data(), gammaSmall_cols_to_get, K_top_matches, 1,
                                out2, rank_kind(panel));
            done += n;
        }
        if (status != QA_OK) return status;
        return pack_lists(lists, best_ptr, best_idx, best_val, best_cap);
    });
}


// selection arguments of qa_fullpass_reads_select_batch (nullptr: plain qa_fullpass_reads_batch)
struct SelectArgs {
    int32_t Ksubset, Knew;
    const int32_t *which;
    const uint64_t *seed;
    int32_t *which_next, *status;
};

static int fullpass_reads_impl(qa_panel_t *panel, int32_t n_chain, int32_t n_label, int32_t n_sample,
                               const int32_t *chain_sample, const int32_t *read_off, const int32_t *read_ptr,
                               const int32_t *u, const int32_t *bq, const int32_t *H, const int32_t *want_dosage,
                               const int32_t *want_top, const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches,
                               double minGLValue, double *dosage, int32_t top_width, int32_t *top_idx, float *top_val,
                               int32_t *top_cnt, const SelectArgs *sel) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!panel || n_chain <= 0 || n_label < 1 || n_label > 3 || n_sample <= 0 || !chain_sample || !read_off || !read_ptr ||
        !u || !bq || !H || !want_dosage || !gammaSmall_cols_to_get || top_width < K_top_matches || top_width > 64) {
        qa::set_error("qa_fullpass_reads_batch: bad argument");
        return QA_ERR_INVALID;
    }
    if (sel && (sel->Ksubset < 1 || sel->Ksubset > panel->K || sel->Knew < 0 || sel->Knew > sel->Ksubset || !sel->which ||
                !sel->seed || !sel->which_next || !sel->status || K_top_matches < 1)) {
        qa::set_error("qa_fullpass_reads_select_batch: bad selection argument");
        return QA_ERR_INVALID;
    }
    if (sel)
        for (size_t i = 0; i < (size_t)n_chain * sel->Ksubset; i++)
            if (sel->which[i] < 1 || sel->which[i] > panel->K) {
                qa::set_error("qa_fullpass_reads_select_batch: which_haps_to_use out of range");
                return QA_ERR_INVALID;
            }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(panel->device));
        if (!panel->scratch) panel->scratch = new qa_panel::Scratch(&panel->A());
        auto &S = *panel->scratch;
        hipStream_t st = panel->pass_stream ? panel->pass_stream : panel->stream;
        const int G = panel->G, T = panel->T;
        int n_thin = 0;
        for (int g = 0; g < G; g++) n_thin = std::max(n_thin, gammaSmall_cols_to_get[g] + 1);
        const bool tmg = getenv("QA_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double T0 = now();
        double t_kern = 0, t_run = 0;
        // ---- per-sample SNP-major index of the bases (input marshalling, O(bases))
        std::vector<int32_t> base_off(n_sample + 1, 0), snp_ptr((size_t)n_sample * (T + 1), 0), ent_off(n_sample, 0);
        for (int s = 0; s < n_sample; s++) {
            const int R = read_off[s + 1] - read_off[s];
            base_off[s + 1] = base_off[s] + (read_ptr + read_off[s] + s)[R];
        }
        const int totB = base_off[n_sample];
        std::vector<int32_t> ent_read(std::max(totB, 1)), ent_bq(std::max(totB, 1));
        {
            // a counting sort per sample, samples independent: spread over host threads (0.3 s on one thread for 256 samples of
            // 50 000 bases -- in front of every full-panel launch set, and uncovered whenever one host thread has the device to itself)
            const int n_thr = std::max(1, std::min<int>(qa::host_threads_cap(), n_sample));
            std::vector<std::string> errs(n_thr);
            auto work = [&](int tid) {
                try {
                    std::vector<int32_t> fill;
                    for (int s = tid; s < n_sample; s += n_thr) {
                        const int R = read_off[s + 1] - read_off[s];
                        const int32_t *rp = read_ptr + read_off[s] + s;
                        const int32_t *su = u + base_off[s], *sb = bq + base_off[s];
                        int32_t *sp = snp_ptr.data() + (size_t)s * (T + 1);
                        ent_off[s] = base_off[s];
                        for (int i = 0; i < rp[R]; i++) {
                            if (su[i] < 0 || su[i] >= T) throw std::runtime_error("SNP index out of range");
                            if (sb[i] > 255 || sb[i] < -255) throw std::runtime_error("|base quality| > 255");
                            sp[su[i] + 1]++;
                        }
                        for (int t = 0; t < T; t++) sp[t + 1] += sp[t];
                        fill.assign(sp, sp + T);
                        for (int r = 0; r < R; r++)
                            for (int i = rp[r]; i < rp[r + 1]; i++) {
                                const int at = fill[su[i]]++;
                                ent_read[(size_t)base_off[s] + at] = r;
                                ent_bq[(size_t)base_off[s] + at] = sb[i];
                            }
                    }
                } catch (const std::exception &e) {
                    errs[tid] = e.what();
                }
            };
            if (n_thr == 1) {
                work(0);
            } else {
                std::vector<std::thread> th;
                for (int t = 0; t < n_thr; t++) th.emplace_back(work, t);
                for (auto &t : th) t.join();
            }
            for (const auto &e : errs)
                if (!e.empty()) throw std::runtime_error(e);
        }
        // chain -> offset of its labels in H (chains are laid out back to back, each with its sample's R)
        std::vector<int32_t> hoff(n_chain + 1, 0);
        for (int c = 0; c < n_chain; c++) {
            const int s = chain_sample[c];
            if (s < 0 || s >= n_sample) throw std::runtime_error("chain_sample out of range");
            hoff[c + 1] = hoff[c] + (read_off[s + 1] - read_off[s]);
        }
        const int P = n_chain * n_label;
        // Work groups of passes (pass = chain x label).  With fp64 ranking (the default) the dosage comes from a pass
        // with fp32 state and the best-haplotype lists from a pass with fp64 state; a chain that wants both runs both.
        struct Group { std::vector<int32_t> ids; int32_t flag; int K_top; PassKind kind; };
        std::vector<Group> groups;
        {
            PassKind main_kind = dosage_kind(panel);
            if (main_kind == KIND_F64_DOS && !panel->rank_fp64) main_kind = KIND_F64_FULL;   // (lists from the dosage pass itself)
            const bool exact = panel->rank_fp64 && main_kind != KIND_F64_FULL;   // else one pass yields dosage and lists
            Group gd{{}, 1, 0, main_kind}, gdt{{}, 1, K_top_matches, main_kind},
                gt{{}, 0, K_top_matches, panel->rank_fp64 ? rank_kind(panel) : KIND_F32};
            for (int c = 0; c < n_chain; c++) {
                const bool dos = want_dosage[c] != 0, top = K_top_matches > 0 && (!want_top || want_top[c] != 0);
                for (int l = 0; l < n_label; l++) {
                    const int id = c * n_label + l;
                    if (exact) {
                        if (dos) gd.ids.push_back(id);
                        if (top) gt.ids.push_back(id);
                    } else {
                        if (dos && top) gdt.ids.push_back(id);
                        else if (dos) gd.ids.push_back(id);
                        else if (top) gt.ids.push_back(id);
                    }
                }
            }
            for (Group *g : {&gd, &gdt, &gt}) if (!g->ids.empty()) groups.push_back(std::move(*g));
        }
        // eps tables from the host libm (convertScaledBQtoProbs, as copied-from-stitch.cpp:166-175)
        std::vector<double> tabs(4 * 256);
        for (int q = 0; q < 256; q++) {
            const double e = std::pow(10, -(double)q / 10);
            tabs[q] = 1 - e; tabs[256 + q] = e / 3; tabs[512 + q] = e / 3; tabs[768 + q] = 1 - e;
        }
        // per-call device buffers, carved from the handle's grow-only side arena (no hipMalloc / hipFree per call)
        const size_t n_out_all = (size_t)P * n_thin;
        {
            auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
            size_t need = 0;
            for (size_t b : {(size_t)P * 4, (size_t)P * 4, (size_t)P * 4, snp_ptr.size() * 4, (size_t)n_sample * 4, ent_read.size() * 4,
                             ent_bq.size() * 4, (size_t)std::max(hoff[n_chain], 1) * 4, tabs.size() * 8})
                need += pad(b);
            if (sel)
                for (size_t b : {std::max<size_t>(n_out_all * top_width, 1) * 4, std::max<size_t>(n_out_all, 1) * 4, (size_t)P * 4,
                                 (size_t)n_chain * sel->Ksubset * 4, (size_t)n_chain * sel->Ksubset * 4, (size_t)n_chain * 4,
                                 (size_t)n_chain * 4, (size_t)n_chain * 8})
                    need += pad(b);
            if (need > panel->aux.cap) panel->aux.require(need + need / 4);
            panel->aux.reset();
        }
        auto carve_i32 = [&](size_t n) { qa::ABuf<int32_t> b; b.arena = &panel->aux; b.ensure(std::max<size_t>(n, 1)); return b; };
        qa::ABuf<int32_t> d_ps = carve_i32(P), d_pl = carve_i32(P), d_ph = carve_i32(P), d_sp = carve_i32(snp_ptr.size()),
                          d_eo = carve_i32(n_sample), d_er = carve_i32(ent_read.size()), d_eb = carve_i32(ent_bq.size()),
                          d_H = carve_i32(std::max(hoff[n_chain], 1));
        qa::ABuf<double> d_tabs;
        d_tabs.arena = &panel->aux;
        d_tabs.ensure(tabs.size());
        d_sp.upload(snp_ptr.data(), snp_ptr.size(), st); d_eo.upload(ent_off.data(), n_sample, st);
        d_er.upload(ent_read.data(), ent_read.size(), st); d_eb.upload(ent_bq.data(), ent_bq.size(), st);
        d_H.upload(H, hoff[n_chain], st); d_tabs.upload(tabs.data(), tabs.size(), st);

        const size_t n_out = (size_t)P * n_thin;
        if (top_cnt) std::fill(top_cnt, top_cnt + n_out, 0);
        if (top_idx) std::fill(top_idx, top_idx + n_out * top_width, -1);
        if (top_val) std::fill(top_val, top_val + n_out * top_width, 0.f);

        int status = QA_OK;
        std::vector<int32_t> no_thin(G, -1);
        // call-wide list table for the device-side selection: [chain * n_label + label][thinned grid][top_width]
        qa::ABuf<int32_t> d_top_all, d_cnt_all, d_rows;
        if (sel) {
            d_top_all = carve_i32(n_out * top_width);
            d_cnt_all = carve_i32(n_out);
            d_rows = carve_i32(P);
        }
        // (no memset: the scatter kernel writes every entry of every row of a chain that wants lists, -1 past the list's
        // end, and the selection reads no other rows; the runtime's fill is a blit with 512-thread workgroups, which waits
        // for a compute unit free of the other host thread's Gibbs waves -- half a second per occurrence in the r02 trace)
        const bool lists_to_host = top_idx || top_val;
        // ---- everything above is host work and uploads into this handle's own buffers; the launch sets below have the
        // device (exclusive phases: queue behind the other handles' launch sets) and the arena
        const double T1q = now();
        qa::GateHold hold;
        hold.acquire(panel->gate(), &panel->arena);
        const double T1 = now();
        for (const Group &grp : groups) {
            const Geometry geo = pick_geometry(panel->K, grp.kind);
            if (geo.NT == 0) throw std::runtime_error("K exceeds the on-chip capacity of the full-pass kernels");
            const int n_grp = (int)grp.ids.size();
            std::vector<int32_t> ps(n_grp), pl(n_grp), ph(n_grp), flags(n_grp, grp.flag);
            for (int i = 0; i < n_grp; i++) {
                const int c = grp.ids[i] / n_label, l = grp.ids[i] % n_label;
                ps[i] = chain_sample[c]; pl[i] = l + 1; ph[i] = hoff[c];
            }
            QA_HIP(hipStreamSynchronize(st));   // the previous group's launches read d_ps / d_pl / d_ph
            d_ps.upload(ps.data(), n_grp, st); d_pl.upload(pl.data(), n_grp, st); d_ph.upload(ph.data(), n_grp, st);
            const int nt_grp = grp.K_top > 0 ? n_thin : 0;
            int done = 0;
            while (done < n_grp && status == QA_OK) {
                const int n = plan_chunk(panel, pass_bytes(panel, geo, nt_grp, grp.flag != 0, false, false, true), n_grp - done);
                S.gl.ensure((size_t)n * T * 2);
                GlParams gp{};
                gp.P = n; gp.T = T; gp.pass_sample = d_ps.p + done; gp.pass_label = d_pl.p + done; gp.pass_hoff = d_ph.p + done;
                gp.snp_ptr = d_sp.p; gp.ent_off = d_eo.p; gp.ent_read = d_er.p; gp.ent_bq = d_eb.p; gp.H = d_H.p;
                gp.pR_tab = d_tabs.p; gp.pA_tab = d_tabs.p + 512; gp.minGLValue = minGLValue; gp.gl = S.gl.p;
                hipLaunchKernelGGL(k_make_gl, dim3((T + 255) / 256, n), dim3(256), 0, st, gp);
                QA_HIP(hipGetLastError());
                BatchOut out;
                out.dosage = grp.flag ? dosage : nullptr;
                out.dosage_rows = grp.ids.data() + done;
                std::vector<int32_t> fidx;
                std::vector<double> fval;
                if (lists_to_host) {   // with the selection on the device the lists need not cross PCIe
                    out.flat_idx = &fidx;
                    out.flat_val = &fval;
                }
                out.top_cap = top_width;      // k_topk keeps the ordered head of each list: all the driver reads
                out.order_by_value = true;
                out.truncate_lists = true;
                std::vector<int32_t> true_cnt;
                out.true_counts = &true_cnt;
                const double tr = now();
                status = run_passes(panel, n, nullptr, flags.data() + done, grp.K_top > 0 ? gammaSmall_cols_to_get : no_thin.data(),
                                    grp.K_top, 1, out, grp.kind);
                t_run += now() - tr;
                t_kern += g_timing[5] / 1e3;
                if (status != QA_OK) break;
                // compact, already ordered lists: the first top_width entries of every (pass, thinned grid)
                if (grp.K_top > 0 && sel) {   // this launch set's lists to their rows of the call-wide device table
                    d_rows.upload(grp.ids.data() + done, n, st);
                    qa::launch_scatter_lists(S.top_idx.p, S.top_cnt.p, d_rows.p, n, n_thin, top_width, d_top_all.p, d_cnt_all.p, st);
                    QA_HIP(hipStreamSynchronize(st));   // d_rows is re-used by the next launch set
                }
                if (grp.K_top > 0) {
                    for (int i = 0; i < n * n_thin; i++) {
                        const size_t o = (size_t)grp.ids[done + i / n_thin] * n_thin + (i % n_thin);
                        const int len = lists_to_host ? std::min<int>(true_cnt[i], top_width) : 0;
                        if (top_cnt) top_cnt[o] = true_cnt[i];
                        for (int q = 0; q < len; q++) {
                            if (top_idx) top_idx[o * top_width + q] = fidx[(size_t)i * top_width + q];
                            if (top_val) top_val[o * top_width + q] = (float)fval[(size_t)i * top_width + q];
                        }
                    }
                }
                done += n;
            }
            if (status != QA_OK) break;
        }
        if (sel && status == QA_OK) {
            // everything_select_good_haps for every chain that asked for lists (select.hip), on the lists still on the device
            qa::ABuf<int32_t> d_which = carve_i32((size_t)n_chain * sel->Ksubset), d_next = carve_i32((size_t)n_chain * sel->Ksubset),
                              d_stat = carve_i32(n_chain), d_want = carve_i32(n_chain);
            qa::ABuf<uint64_t> d_seed;
            d_seed.arena = &panel->aux;
            d_seed.ensure(n_chain);
            std::vector<int32_t> want(n_chain);
            for (int c = 0; c < n_chain; c++) want[c] = K_top_matches > 0 && (!want_top || want_top[c] != 0);
            d_which.upload(sel->which, (size_t)n_chain * sel->Ksubset, st);
            d_seed.upload(sel->seed, n_chain, st);
            d_want.upload(want.data(), n_chain, st);
            qa::SelectParams sp{};
            sp.n_label = n_label; sp.n_thin = n_thin; sp.top_width = top_width; sp.K_top_matches = K_top_matches; sp.K = panel->K;
            sp.Ksubset = sel->Ksubset; sp.Knew = sel->Knew; sp.top = d_top_all.p; sp.which = d_which.p; sp.seed = d_seed.p;
            sp.want = d_want.p; sp.out = d_next.p; sp.status = d_stat.p;
            hipEvent_t e0, e1;
            QA_HIP(hipEventCreate(&e0)); QA_HIP(hipEventCreate(&e1));
            QA_HIP(hipEventRecord(e0, st));
            const bool launched = qa::launch_select(sp, n_chain, st);
            QA_HIP(hipEventRecord(e1, st));
            if (launched) {
                d_next.download(sel->which_next, (size_t)n_chain * sel->Ksubset, st);
                d_stat.download(sel->status, n_chain, st);
            } else {   // tables beyond the LDS: every chain is left to the caller's host path
                for (int c = 0; c < n_chain; c++) sel->status[c] = want[c] ? 1 : -1;
            }
            QA_HIP(hipStreamSynchronize(st));
            float ms = 0;
            QA_HIP(hipEventElapsedTime(&ms, e0, e1));
            qa::profile_add(qa::PK_SELECT, ms, (double)n_out * top_width * 4.0 + (double)n_chain * sel->Ksubset * 8.0,
                            qa::profile_clock_ms(e0), n_chain);
            QA_HIP(hipEventDestroy(e0)); QA_HIP(hipEventDestroy(e1));
            // a truncated list only matters to the exhausted branch, which the device leaves to the host (status 1)
        }
        hold.release();
        if (tmg)
            fprintf(stderr, "[qa_fullpass_reads P=%d] index+uploads %.3f s, queued for the device %.3f s, run_passes %.3f s (device %.3f s), scatter etc %.3f s\n", P,
                    T1q - T0, T1 - T1q, t_run, t_kern, now() - T1 - t_run);
        return status;
    });
}

int qa_fullpass_reads_batch(qa_panel_t *panel, int32_t n_chain, int32_t n_label, int32_t n_sample,
                            const int32_t *chain_sample, const int32_t *read_off, const int32_t *read_ptr,
                            const int32_t *u, const int32_t *bq, const int32_t *H, const int32_t *want_dosage,
                            const int32_t *want_top, const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches,
                            double minGLValue,
                            double *dosage, int32_t top_width, int32_t *top_idx, float *top_val, int32_t *top_cnt) {
    return fullpass_reads_impl(panel, n_chain, n_label, n_sample, chain_sample, read_off, read_ptr, u, bq, H, want_dosage,
                               want_top, gammaSmall_cols_to_get, K_top_matches, minGLValue, dosage, top_width, top_idx,
                               top_val, top_cnt, nullptr);
}

int qa_fullpass_reads_select_batch(qa_panel_t *panel, int32_t n_chain, int32_t n_label, int32_t n_sample,
                                   const int32_t *chain_sample, const int32_t *read_off, const int32_t *read_ptr,
                                   const int32_t *u, const int32_t *bq, const int32_t *H, const int32_t *want_dosage,
                                   const int32_t *want_top, const int32_t *gammaSmall_cols_to_get, int32_t K_top_matches,
                                   double minGLValue, double *dosage, int32_t top_width, int32_t *top_idx, float *top_val,
                                   int32_t *top_cnt, int32_t Ksubset, int32_t Knew, const int32_t *which_haps_to_use,
                                   const uint64_t *seed_select, int32_t *which_next, int32_t *select_status) {
    const SelectArgs sel{Ksubset, Knew, which_haps_to_use, seed_select, which_next, select_status};
    return fullpass_reads_impl(panel, n_chain, n_label, n_sample, chain_sample, read_off, read_ptr, u, bq, H, want_dosage,
                               want_top, gammaSmall_cols_to_get, K_top_matches, minGLValue, dosage, top_width, top_idx,
                               top_val, top_cnt, &sel);
}

}  // extern "C"
