// fullpass_dev.hpp -- launch parameters and device helpers shared by the full-panel pass kernels: fullpass.hip (emission
// tables, the fp32-state dosage passes, the generic fp64 passes, dosage mat-vec, top-K picker, host side) and
// fullpass64.hip (the fp64-state ranking passes).  Everything lives in an unnamed namespace: each translation unit gets its
// own copy.
#pragma once

#include "panel.hpp"

#include <type_traits>
#include <utility>

namespace {

constexpr int kMaxRow = 256;      // nMaxDH + 1 <= 256
constexpr int kHistCopies = 32;   // bank-private copies of the gamma histogram
// The histogram accumulates gamma * sigma_g (in [0, 1], summing to 1 over a grid) as fixed point with LDS integer
// atomics: on gfx950 ds_add_u32 runs ~30x faster than ds_add_f32 (scripts/micro/lds_atomic_rate.hip: 0.10 vs 3.0 clk per
// lane-op).  fp32-state passes: 32 bits, round-to-nearest at 2^-31 keeps the per-grid error near 3e-8 (random walk over K
// adds).  fp64-state passes: 64 bits at 2^-62 (2e-19 per add: below fp64 rounding of the sum).
constexpr float kHistScale = 2147483648.f;              // 2^31
constexpr double kHistScale64 = 4611686018427387904.0;  // 2^62
constexpr int kMaxTop = 8;        // K_top_matches supported in registers
constexpr int kCandCap = 1024;    // candidates of the fused top-K picker kept in LDS (fp64 ranking passes)

struct PassParams {
    // panel
    const uint8_t *hm;       // [G][Kp]
    const int32_t *B;        // [G][nMaxDH]
    const int32_t *sp_off;   // [G+1]
    const int32_t *sp_k;
    const uint32_t *sp_word;
    const int32_t *sp_gidx;      // [G] index of the grid among those holding specials, or -1
    const int32_t *sp_chunk_at;  // [n_sp_grids][Kp / 16] first entry (absolute) of the grid's special list at or after each 16-haplotype chunk
    const double *sigma;     // [G-1]  transMatRate_t row 0
    const double *tm1;       // [G-1]  transMatRate_t row 1 (1 - sigma as the caller passed it)
    const double *IE;        // [T][nMaxDH] or null
    int K, Kp, G, T, nMaxDH, nrow, n_special;
    double ref_error;
    // per launch
    int P;                   // passes
    const double *gl;        // [P][T][2]
    const int32_t *thin_col; // [G]  -1 or thinned column
    int n_thin;
    const int32_t *flags;    // [P] bit0: dosage pass; bit1: store all alpha; bit2: store gamma; bit3: store beta
    int normalize_emissions;
    // fp64 ranking passes (fullpass64.hip): the reference's own normalisation schedule (reference-single.cpp:1096-1107)
    int lazy;                // k_emat: raw grid-0 emissions and emin (the lazily normalised forward needs both)
    int always_normalize;
    int grid0_left_to_right; // validation kernels only (fullpass_ref.hip): 0 = grid 0's sum(alphaHat_t_col) as Armadillo's sum() adds it
                             // (two accumulators over the even / odd k), 1 = left to right (qa_panel_set_sum_order(panel, 2))
    double norm_threshold;   // min_emission_prob_normalization_threshold
    double *emin;            // [P][G] min emission of the grid after normalisation (:1044-1057), -1: grid without variant
    double *emin_b1;         // [P] emin of grid 1 as the BACKWARD pass reads it: -1 when grid 1 holds no variant.  The forward pass
                             // forces grid 1 to count as a grid with a variant (:964-966, so emin[1] >= 0 always); the backward
                             // pass tests grid g + 1 itself and does not (:1866-1877)
    // scratch / outputs
    void *emat;             // [P][G][kMaxRow]
    void *esp;              // [P][esp_stride]
    int esp_stride;          // n_special, or (lazy) n_special + 16 * (grids with specials): see k_emat
    double *escale0;         // [P] factor applied to the grid-0 emissions (folded back into c[0])
    void *alpha;            // [P][n_alpha_cols][Kq]   (lane-interleaved order)
    const int32_t *alpha_slot; // [P][G] -> column slot in alpha, or -1
    size_t alpha_pass_stride; // elements
    int Kq;                  // NT * NCH * 16 (padded K of the launch geometry)
    size_t alpha_col_elems;  // elements between two stored alpha columns of a pass (Kq, or the tighter pitch of the fp64 dosage passes)
    double hist_unit;        // value of one unit of the fp64 histogram (mg as uint64): 2^-62 (generic kernels) or 2^-51 (k_bwd64d)
    double *c;               // [P][G]
    void *mg;                // [P][G][kMaxRow]  histogram of gamma * sigma_g by code, fixed point (dosage passes): uint32 at
                             // 2^-31 (fp32 state) or uint64 at 2^-62 (fp64 state)
    void *gsp;               // [P][n_special]   gamma of special haplotypes (float / double)
    void *gamma_out;        // [P][G][Kq] or null
    void *beta_out;         // [P][G][Kq] or null
    void *beta_thin;        // [P][n_thin][Kq] unscaled beta at the thinned grids, or null
    double *dosage;          // [P][T]
    int K_top;
    int top_cap;             // capacity per (pass, thinned column)
    int truncate_lists;      // keep only the head (first top_cap in rejig order) of over-long lists
    int fused_topk;          // fp64 ranking passes: the backward kernel picks the lists itself (top_cnt = -1: left to k_topk)
    int32_t *top_cnt;        // [P][n_thin]
    int32_t *top_idx;        // [P][n_thin][top_cap]
    void *top_val;          // [P][n_thin][top_cap]
    const int32_t *topk_todo;  // k_topk: null (every (thinned column, pass)), or [n][2] = (pass, column) pairs to do
    // fp64 kernels of fullpass64.hip with K beyond their on-chip capacity (7 chunk rows of 8 192 haplotypes): the state of the
    // chunk rows past the seventh lives here, [P][rows][8][512] double2 (a row in the layout of an LDS row), read and written by
    // the thread that owns the elements once per grid
    // fp64 dosage passes (k_fwd64 + k_bwd64d): alpha is handed over at every SECOND grid only (alpha_slot[g] = g / 2 for even g,
    // -1 for odd g) and k_bwd64d re-forms an odd grid's column from the even grid's below it, repeating the forward step's
    // operations with the grid's two scalars kept here: the addend and the rescaling factor (1 where the grid was not renormalised)
    double *fw_add, *fw_xs;  // [P][G] or null (every column handed over)
    double *spill;
    size_t spill_pass_stride;  // doubles
};

// Emission of one distinct word (reference-single.cpp:294-327): the product over the grid's SNPs of P(reads | allele), with
// e[b] = (P(reads | the haplotype carries ref at b), P(reads | alt)) = (gl.x * (1 - eps) + gl.y * eps, gl.x * eps + gl.y * (1 - eps))
// formed ONCE per SNP by k_emat (the same expression every word would evaluate for itself: results unchanged).
__device__ __forceinline__ double word_emission(uint32_t w, const double2 *e, int nLocal) {
    double prob = 1.0;
    for (int b = 0; b < nLocal; b++) {
        const double2 v = e[b];
        prob *= ((w >> b) & 1u) ? v.y : v.x;
    }
    return prob;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// first position in grid g's ascending special list whose haplotype index is >= k (rare path)
__device__ __noinline__ int special_lower_bound(const int32_t *sp_k, int lo, int hi, int k) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sp_k[mid] < k) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ bool has_zero_byte(uint32_t v) { return ((v - 0x01010101u) & ~v & 0x80808080u) != 0; }
__device__ __forceinline__ bool any_zero_code(const uint4 &d) {
    return has_zero_byte(d.x) || has_zero_byte(d.y) || has_zero_byte(d.z) || has_zero_byte(d.w);
}

// The recursions run with fp32 state (dosage passes: 16-byte vectors of 4) or fp64 state (16-byte vectors of 2).
// Checkpoints are written in 16-byte vectors either way.
template <typename TS> struct Vec;
template <> struct Vec<float> { using V = float4; static constexpr int EPV = 4; using Hist = uint32_t; };
template <> struct Vec<double> { using V = double2; static constexpr int EPV = 2; using Hist = unsigned long long; };
__device__ __forceinline__ float vget(const float4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; }
__device__ __forceinline__ double vget(const double2 &v, int r) { return r == 0 ? v.x : v.y; }
__device__ __forceinline__ float4 vmake(const float *x) { return make_float4(x[0], x[1], x[2], x[3]); }
__device__ __forceinline__ double2 vmake(const double *x) { return make_double2(x[0], x[1]); }

// vector index of (chunk j, vector q) of this thread in the lane-interleaved checkpoint layout: each
// (wave, j, q) owns 64 consecutive 16-byte vectors (1 KiB) => every dwordx4 store / load is coalesced.
template <int NV>
__device__ __forceinline__ size_t alpha_vec_index(int j, int q, int NT, int t) {
    return (size_t)j * NT * NV + (size_t)((t >> 6) * NV + q) * 64 + (t & 63);
}

template <typename TS>
__device__ __forceinline__ void store_chunk(typename Vec<TS>::V *dst, const TS (&x)[16], int j, int NT, int t) {
    constexpr int EPV = Vec<TS>::EPV, NV = 16 / EPV;
#pragma unroll
    for (int q = 0; q < NV; q++) dst[alpha_vec_index<NV>(j, q, NT, t)] = vmake(&x[EPV * q]);
}

// block-wide sum of one double per thread: wave shuffle, then LDS across waves.  `buf` is one of two
// alternating 16-entry buffers so that one barrier per call suffices.
__device__ __forceinline__ double block_sum(double v, double *buf, int t, int nwaves) {
    v = wave_sum(v);
    if ((t & 63) == 0) buf[t >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < nwaves; w++) s += buf[w];
    return s;
}

// compile-time loop: the chunk loops are too large for `#pragma unroll` to be honoured, and register-resident
// state needs constant indices
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

}  // namespace

namespace qa {
// fullpass64.hip: the fp64-state ranking passes.  fb64_chunks: chunk rows (of 512 x 16 haplotypes) the geometry needs for K
// haplotypes, 0 when K exceeds its on-chip capacity.
int fb64_chunks(int K);
// chunk rows beyond the on-chip seven, whose state is streamed through HBM (PassParams::spill); 0 up to K = 57 344
int fb64_spill_rows(int K);
size_t fb64_lds_bytes(int K);
void launch_fb64(const void *pass_params, hipStream_t st, hipEvent_t e_mid);
// the fp64-state DOSAGE passes: k_fwd64 storing every second column + k_bwd64d (re-forms the others; gamma histogram for k_dosage)
size_t fb64_alpha_col_elems(int K);
size_t fb64_dos_lds_bytes(int K);
void launch_fb64_dosage(const void *pass_params, hipStream_t st, hipEvent_t e_mid);

// fullpass_ref.hip: the VALIDATION kernels (qa_panel_set_sum_order): every K-wide sum in the reference's order
size_t fb_ref_state_doubles(int Kq);
void launch_fb_ref(const void *pass_params, int NT, hipStream_t st, hipEvent_t e_mid);

// select.hip: everything_select_good_haps on the device (one wave per chain)
struct SelectParams {
    int n_label, n_thin, top_width, K_top_matches, K, Ksubset, Knew;
    const int32_t *top;      // [chain][label][thinned grid][top_width]: 0-based haplotypes, best first, -1 padded
    const int32_t *which;    // [chain][Ksubset]: the chain's current small panel, 1-based
    const uint64_t *seed;    // [chain]: key of the chain's selection stream
    const int32_t *want;     // [chain] or nullptr: chains without a selection get status -1
    int32_t *out;            // [chain][Ksubset]: previously selected (Ksubset - Knew), then the Knew new ones, 1-based
    int32_t *status;         // [chain]: 0 selected; 1 ranks exhausted before Knew were found (host path); -1 not wanted
};
size_t select_lds_bytes(const SelectParams &p);
bool launch_select(const SelectParams &p, int n_chain, hipStream_t st);
void launch_scatter_lists(const int32_t *src_idx, const int32_t *src_cnt, const int32_t *rows, int n, int n_thin, int top_cap,
                          int32_t *dst_idx, int32_t *dst_cnt, hipStream_t st);
}  // namespace qa
