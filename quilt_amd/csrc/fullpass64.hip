// fullpass64.hip -- the fp64-state "ranking" passes of the full-panel forward/backward on gfx950 (MI355X): best-haplotype
// lists at the thinned grids (get_best_haps_from_thinned_sites), QUILT/src/reference-single.cpp:878-1131 (forward v3),
// :1781-2179 (backward v3), :129-194 (top-K picker).
//
// Why fp64 state: which of several nearly tied haplotypes the reference (double arithmetic) reports decides the next small
// panel; fp32 state ranks them differently.  These kernels therefore follow the reference operation by operation -- its
// LAZY normalisation schedule included (alpha is rescaled only when the running product of per-grid minimum emissions
// drops below min_emission_prob_normalization_threshold, :1096-1107; c_g = 1 / sigma otherwise), the analytic column sum
// on grids without variants (:1078-1088), the raw grid-0 emissions (:2314-2347) -- so that per element the arithmetic is
// the reference's; only the order of the K-wide sums differs.
//
// How (MI355X-first):
//   * one workgroup of 512 threads (8 waves, TWO per SIMD, <= 256 VGPRs each) owns a pass and keeps all K state values on
//     chip: the first NR chunk rows (a row = 512 lanes x 16 haplotypes) in registers, the last NL rows in LDS
//     ([row][vector][lane] double2: conflict-free b128).  K = 50 000: 4 rows in registers (128 VGPRs), 2 full rows + one
//     64-lane row in LDS (136 KB).  With two waves per SIMD one wave's LDS gathers (16 table look-ups per chunk) overlap
//     the other's fp64 arithmetic; the previous one-wave-per-SIMD kernels spent 40 % of their cycles parked on those
//     gathers.
//   * lazy normalisation removes the per-element rescaling: forward (x + a) * e and one add for the column sum (3 fp64
//     operations per cell), backward ((x + v) [* s]) * e with the factor s = c_g * sigma_g skipped when it is exactly 1
//     (it is whenever the grid was not renormalised and fl(fl(1 / sigma) * sigma) == 1).
//   * a grid's haplotype codes (1 B per cell, the only full-rate HBM stream) go straight to registers, one 16-byte load per
//     chunk, issued a whole grid ahead; the 2 KiB emission table of the next grid is staged through registers into the
//     other half of an LDS double buffer and published by the per-grid barrier of the block-wide sum.
//   * the top-K picker is fused into the backward kernel at the thinned grids (beta is on chip there, only alpha is
//     streamed): a lower bound of the K_top-th largest gamma from the per-lane maxima, the few candidates above it
//     collected in LDS, exact selection and ordering (value descending, ties by ascending haplotype: functions.R:2161-2170)
//     by one wave.  A grid whose candidates overflow the LDS list (ties: a label without reads, duplicated haplotypes)
//     hands its beta column to k_topk instead (top_cnt = -1).
//
// Algorithmic HBM bytes: per cell 1 B code forward + 1 B backward, plus 8 B of alpha written and read at
// the thinned grids.
#include "fullpass_dev.hpp"

namespace {

constexpr int kNT = 512;          // threads per pass
constexpr int kRowHaps = kNT * 16;
constexpr int kNStream = 8;       // per-grid scalar streams staged in LDS, 64 grids at a time

// NS: chunk rows beyond the on-chip seven, streamed through HBM (PassParams::spill) -- K > 57 344; all rows are then full
// (the codes' row pitch Kp covers them: the padding has code 0, emission 0, state 0)
struct Geo64 { int NCH, NR, NL, n_last, NS; };

// LDS bytes besides the state rows: emission tables [2][256], block sums [2][16], scalar streams, picker scratch
constexpr size_t kLdsFixed = 2 * kMaxRow * 8 + 2 * 16 * 8 + kNStream * 64 * 8 + 8 * kMaxTop * 8 + 64 + (size_t)kCandCap * 12;
constexpr size_t kLdsMax = 160 * 1024;

inline size_t lds_bytes(const Geo64 &g) { return kLdsFixed + (g.NL > 0 ? ((size_t)(g.NL - 1) * kNT + g.n_last) * 128 : 0); }
inline Geo64 geo64(int K) {
    Geo64 g{};
    g.NCH = (K + kRowHaps - 1) / kRowHaps;
    const int rem = K - (g.NCH - 1) * kRowHaps;
    g.n_last = ((rem + 15) / 16 + 63) / 64 * 64;
    g.NR = std::min(g.NCH, 4);
    g.NL = g.NCH - g.NR;
    if (lds_bytes(g) > kLdsMax) { g.NR = 5; g.NL = g.NCH - 5; }

    // (Round 5 tried eight rows on chip for K up to 65 536 -- SIX rows in registers, two in LDS -- instead of streaming the eighth:
    // the <6, 2> kernels spill 112-132 registers (against 16) and the HRC-size bench line fell from 30.1 to 28.3 samples/s; not kept.)
    if (g.NCH > 7) {   // 5 rows in registers, 2 in LDS (what 7 full rows take: 152 / 155 KB of LDS), the rest streamed
        g.NS = g.NCH - 7;
        g.NCH = 7; g.NR = 5; g.NL = 2; g.n_last = kNT;
        return g;
    }
    if (lds_bytes(g) > kLdsMax) g.NCH = 0;
    return g;
}

struct Lds {
    double *etab;        // [2][256]
    double *red;         // [2][16]
    double *sc;          // [kNStream][64] scalar streams of the current block of 64 grids
    double *wtop;        // [8][kMaxTop]
    int *misc;           // [0] candidate count; doubles at misc + 4: threshold
    double *cand_v;      // [kCandCap]
    int *cand_k;         // [kCandCap]
    double2 *state;      // LDS rows
    __device__ __forceinline__ explicit Lds(char *smem) {
        etab = reinterpret_cast<double *>(smem);
        red = etab + 2 * kMaxRow;
        sc = red + 2 * 16;
        wtop = sc + kNStream * 64;
        misc = reinterpret_cast<int *>(wtop + 8 * kMaxTop);
        cand_v = reinterpret_cast<double *>(misc + 16);
        cand_k = reinterpret_cast<int *>(cand_v + kCandCap);
        state = reinterpret_cast<double2 *>(cand_k + kCandCap);
    }
};

// state vectors of LDS row r for thread t: st[q * stride], q = 0..7 (rows before the last hold 512 lanes, the last n_last)
template <int NL, typename LT>
__device__ __forceinline__ double2 *lds_row(const LT &L, int r, int t) { return L.state + (size_t)r * 8 * kNT + t; }
template <int NL>
__device__ __forceinline__ int lds_stride(int r, int n_last) { return r == NL - 1 ? n_last : kNT; }

// state vectors of streamed row r (0-based among the streamed rows) of pass p for thread t: st[q * kNT], q = 0..7
__device__ __forceinline__ double2 *spill_row(const PassParams &prm, int p, int r, int t) {
    return reinterpret_cast<double2 *>(prm.spill + (size_t)p * prm.spill_pass_stride) + (size_t)r * 8 * kNT + t;
}

__device__ __forceinline__ bool has_zero_byte2(uint32_t a, uint32_t b) { return has_zero_byte(a) || has_zero_byte(b); }

// Haplotypes with code 0 ("specials": words beyond the grid's nMaxDH most frequent) take their own emission; table row 0
// is 1 on grids that hold any (reference-single.cpp:1002-1042 / :1902-1964).  A chunk's specials are consecutive entries
// of the grid's ascending list, starting at sp_chunk_at[grid][chunk] (precomputed at panel upload: no search here).
// (Zero codes of the padding beyond K count as specials too: they come after the real ones and read the 16 zero entries
// k_emat appends to every grid's list, which keeps their state at 0.)  Returns the number of zero codes seen.
__device__ __forceinline__ int special_half(double (&x)[8], uint32_t w0, uint32_t w1, const double *esp_at) {
    const uint32_t w[2] = {w0, w1};
    int at = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t code = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
        if (code == 0) {
            x[i] *= esp_at[at];
            at++;
        }
    }
    return at;
}

// Eight haplotypes (half a chunk): forward x <- (x + v) * table[code]; backward x <- ((x + v) * s) * table[code] (the
// "+ v" and "* s" are the previous grid's, applied when the chunk is touched anyway).  Returns their sum.  The padding
// beyond K has code 0, whose emission is 0 on every grid (table row 0, or the zero entries appended to the grid's
// special list): its state stays 0 without any masking.
template <bool BWD>
__device__ __forceinline__ double half_step(double (&x)[8], uint32_t w0, uint32_t w1, const double *et, double v, double s) {
    const uint32_t w[2] = {w0, w1};
    constexpr int GB = BWD ? 8 : 4;   // look-ups in batches, back to back, then the arithmetic (forward: smaller batches,
                                      // it is the kernel shorter of registers)
#pragma unroll
    for (int h = 0; h < 8 / GB; h++) {
        double e[GB];
#pragma unroll
        for (int i = 0; i < GB; i++) e[i] = et[(w[(GB * h + i) >> 2] >> (((GB * h + i) & 3) * 8)) & 0xffu];
#pragma unroll
        for (int i = 0; i < GB; i++) x[GB * h + i] = BWD ? ((x[GB * h + i] + v) * s) * e[i] : (x[GB * h + i] + v) * e[i];
    }
    return 0.0;
}
__device__ __forceinline__ double sum8(const double (&x)[8]) {
    return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
}

// a register-resident chunk (16 haplotypes): both halves, specials, sum
template <bool BWD>
__device__ __forceinline__ double reg_chunk(double (&x)[16], const uint4 &d, const double *et, double v, double s, const int32_t *sp_at,
                                            const double *esp, int k0) {
    double (&lo)[8] = *reinterpret_cast<double (*)[8]>(&x[0]);
    double (&hi)[8] = *reinterpret_cast<double (*)[8]>(&x[8]);
    half_step<BWD>(lo, d.x, d.y, et, v, s);
    half_step<BWD>(hi, d.z, d.w, et, v, s);
    if (sp_at && any_zero_code(d)) {
        const double *e0 = esp + sp_at[k0 >> 4];
        const int n0 = special_half(lo, d.x, d.y, e0);
        special_half(hi, d.z, d.w, e0 + n0);
    }
    return sum8(lo) + sum8(hi);
}

// an LDS-resident chunk, half at a time (8 doubles of temporaries instead of 16)
template <bool BWD>
__device__ __forceinline__ double lds_chunk(double2 *st, int stride, const uint4 &d, const double *et, double v, double s,
                                            const int32_t *sp_at, const double *esp, int k0) {
    double tot = 0;
    int n0 = 0;
    const bool sp = sp_at && any_zero_code(d);
    const double *e0 = sp ? esp + sp_at[k0 >> 4] : esp;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double x[8];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double2 u = st[(4 * h + q) * stride];
            x[2 * q] = u.x;
            x[2 * q + 1] = u.y;
        }
        const uint32_t w0 = h ? d.z : d.x, w1 = h ? d.w : d.y;
        half_step<BWD>(x, w0, w1, et, v, s);
        if (sp) n0 += special_half(x, w0, w1, e0 + n0);
#pragma unroll
        for (int q = 0; q < 4; q++) st[(4 * h + q) * stride] = make_double2(x[2 * q], x[2 * q + 1]);
        tot += sum8(x);
    }
    return tot;
}

// a wave-uniform double that lives in vector registers -> scalar registers
__device__ __forceinline__ double uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// One wave-instruction: 64 lanes x 16 B, global (wave-uniform base + per-lane byte offset) -> LDS (wave base in M0 + lane
// * 16): no staging registers.  Inline asm on purpose: with the builtin the compiler treats every later LDS read as possibly
// aliasing the DMA in flight and puts `s_waitcnt vmcnt(0)` in front of each.  The DMA is invisible to the compiler's own
// vmcnt bookkeeping, which only makes its waits stricter (in-order counter); the one wait that matters -- table landed
// before the block's barrier -- is placed by hand (block_sum64).
__device__ __forceinline__ void dma16(const void *gbase_uniform, uint32_t lane_off, uint32_t lds_wave_base) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_wave_base);
    const uint64_t src = (uint64_t)(uintptr_t)gbase_uniform;
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(src >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)src);   // (the builtin returns int)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(dst), "s"(base)
                 : "memory");
}
// wave 0: DMA grid g's 2 KiB emission table (256 doubles = 2 x 64 lanes x 16 B) into half `buf` of the LDS double buffer
__device__ __forceinline__ void dma_table(const double *emat, int g, uint32_t lds_etab, int buf, int wave, int lane) {
    if (wave == 0) {
        dma16(emat + (size_t)g * kMaxRow, 16 * lane, lds_etab + buf * kMaxRow * 8);
        dma16(emat + (size_t)g * kMaxRow + 128, 16 * lane, lds_etab + buf * kMaxRow * 8 + 1024);
    }
}

// block-wide sum with a bare barrier (no fence: the only LDS traffic to publish is `buf` and the staged table, both
// waited for explicitly), so that the code loads in flight for the next grid are not drained at every grid
// 64-lane sum by DPP row shifts / row broadcasts (no LDS pipe, no per-lane shuffle addresses to keep in registers); the
// total lands in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_get<0x111, 0xf>(v);  // row_shr:1
    v += dpp_get<0x112, 0xf>(v);  // row_shr:2
    v += dpp_get<0x114, 0xf>(v);  // row_shr:4
    v += dpp_get<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row total
    v += dpp_get<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

template <int TABLE_WAIT>
__device__ __forceinline__ double block_sum64(double v, double *buf, int wave, int lane, int nwaves) {
    v = wave_sum_dpp(v);
    if (lane == 63) buf[wave] = v;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TABLE_WAIT) : "memory");   // wave 0's table DMA precedes TABLE_WAIT code loads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    double s = 0;
    for (int w = 0; w < nwaves; w++) s += buf[w];
    return uniform(s);
}

// Per-grid scalars (sigma, 1 - sigma, min emission, c, flags ...) are wave-uniform.  Fetched through the vector memory path
// at the point of use they would have to be waited for with vmcnt(0), which also drains the code loads issued a grid
// ahead (in-order counter) and exposes one memory round trip per grid -- and so would any register spilled to scratch.
// They are therefore staged in LDS 64 grids at a time (wave 0 loads, two barriers per 64 grids) and read back as
// broadcasts; uniform() moves a broadcast value to scalar registers.
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
enum { SC_SIG = 0, SC_TM1, SC_EMIN, SC_SPG, SC_SLOT, SC_C, SC_TCOL };

// The thread index recomputed from the hardware lane counter (2 VALU instructions).  Per-lane address offsets derived from
// it inside the grid loop are loop-variant for the compiler, which would otherwise hoist a dozen of them out of the loop,
// run out of registers and reload them from scratch every grid -- and a scratch reload is a vector-memory operation whose
// wait drains the code loads in flight.
__device__ __forceinline__ int fresh_tid(int wave) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return wave * 64 + l;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int NR, int NL, bool SP = false>
__global__ __launch_bounds__(kNT) void k_fwd64(PassParams prm, int n_last) {
    constexpr int NCH = NR + NL, NT = kNT, nwaves = NT >> 6;
    const int NS = SP ? prm.Kq / kRowHaps - NCH : 0;   // chunk rows streamed through HBM
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L(smem);
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t lds_etab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int K = prm.K, G = prm.G;
    const double *emat = static_cast<const double *>(prm.emat) + (size_t)p * G * kMaxRow;
    const double *esp = static_cast<const double *>(prm.esp) + (size_t)p * prm.esp_stride;
    const double *emin = prm.emin + (size_t)p * G;
    double2 *aout = reinterpret_cast<double2 *>(static_cast<double *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride);
    const int32_t *slot = prm.alpha_slot + (size_t)p * G;
    const size_t col_vecs = (size_t)prm.alpha_col_elems / 2;
    const double double_K = uniform((double)K), one_over_K = uniform(1 / (double)K);

    double a[NR][16];
    uint4 dh[NCH];
    // every load of the main loop is unconditional (row pitch Kp covers whole chunk rows; grid indices are clamped), so that
    // the compiler can count the loads in flight and wait for exactly the one it needs
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int k0 = (j * NT + t) * 16;
        if (j < NR) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[j][i] = 0.0;
        } else if (j - NR < NL - 1 || t < n_last) {
            double2 *st = lds_row<NL>(L, j - NR, t);
            const int stride = lds_stride<NL>(j - NR, n_last);
#pragma unroll
            for (int q = 0; q < 8; q++) st[q * stride] = make_double2(0.0, 0.0);
        }
        dh[j] = *reinterpret_cast<const uint4 *>(prm.hm + k0);
    }
    if constexpr (SP) {
        for (int js = 0; js < NS; js++) {
            double2 *st = spill_row(prm, blockIdx.x, js, t);
#pragma unroll
            for (int q = 0; q < 8; q++) st[q * NT] = make_double2(0.0, 0.0);
        }
    }
    if (t < kMaxRow) L.etab[t] = emat[t];
    __syncthreads();

    // the reference's scalars (reference-single.cpp:935-946, :1092-1107)
    double prev_sum = 1, running_min = 1;
    for (int g = 0; g < G; g++) {
        const int buf = g & 1, jl = g & 63;
        if (jl == 0) {   // next block of 64 grids: flush c, refill the scalar streams
            __syncthreads();
            if (t < 64) {
                if (g > 0) prm.c[(size_t)p * G + g - 64 + lane] = L.sc[SC_C * 64 + lane];
                const int gi = clampi(g + lane, 0, G - 1), gm = clampi(g + lane - 1, 0, G > 1 ? G - 2 : 0);
                L.sc[SC_SIG * 64 + lane] = G > 1 ? prm.sigma[gm] : 1.0;
                L.sc[SC_TM1 * 64 + lane] = G > 1 ? prm.tm1[gm] : 0.0;
                L.sc[SC_EMIN * 64 + lane] = emin[gi];
                L.sc[SC_SPG * 64 + lane] = (double)prm.sp_gidx[gi];
                L.sc[SC_SLOT * 64 + lane] = (double)slot[gi];
            }
            __syncthreads();
        }
        // next grid's table -> the other half of the LDS double buffer (last read in iteration g - 1)
        const int tt = fresh_tid(wave);   // == t, see fresh_tid
        dma_table(emat, clampi(g + 1, 0, G - 1), lds_etab, buf ^ 1, wave, tt & 63);
        const double *et = L.etab + buf * kMaxRow;
        const int sp_g = (int)uniform(L.sc[SC_SPG * 64 + jl]);   // index of the grid among those that hold specials, or -1
        const int32_t *sp_at = sp_g >= 0 ? prm.sp_chunk_at + (size_t)sp_g * (prm.Kq >> 4) : nullptr;
        const double em = uniform(L.sc[SC_EMIN * 64 + jl]);
        const bool has_variant = em >= 0;
        double sig = 1.0, addend = one_over_K;   // grid 0: alpha = e / K (:2314-2347) = (0 + 1/K) * e
        if (g > 0) {
            sig = uniform(L.sc[SC_SIG * 64 + jl]);
            const double jump_prob = uniform(L.sc[SC_TM1 * 64 + jl]) / double_K;
            const double jump_prob_plus = prm.always_normalize ? jump_prob : jump_prob * prev_sum;
            addend = uniform(jump_prob_plus / sig);
        }
        const uint8_t *hm_next = prm.hm + (size_t)clampi(g + 1, 0, G - 1) * prm.Kp;
        double psum = 0;
        static_for<NCH>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const uint4 d = dh[j];
            const int k0 = (j * NT + tt) * 16;
            if constexpr (j < NR) {
                psum += reg_chunk<false>(a[j], d, et, addend, 1.0, sp_at, esp, k0);
            } else {
                if (j - NR < NL - 1 || wave * 64 < n_last)
                    psum += lds_chunk<false>(lds_row<NL>(L, j - NR, tt), lds_stride<NL>(j - NR, n_last), d, et, addend, 1.0, sp_at, esp, k0);
            }
            // the codes of the next grid: a whole grid ahead of their use
            dh[j] = reinterpret_cast<const uint4 *>(hm_next + j * kRowHaps)[tt];
        });
        if constexpr (SP) {   // the streamed rows: codes of this grid fetched now, state read and written in place
            const uint8_t *hm_cur = prm.hm + (size_t)g * prm.Kp;
            for (int js = 0; js < NS; js++) {
                const int jg = NCH + js, k0 = (jg * NT + tt) * 16;
                const uint4 d = reinterpret_cast<const uint4 *>(hm_cur + (size_t)jg * kRowHaps)[tt];
                psum += lds_chunk<false>(spill_row(prm, p, js, tt), NT, d, et, addend, 1.0, sp_at, esp, k0);
            }
        }
        double run_total = block_sum64<NCH>(psum, L.red + buf * 16, wave, lane, nwaves);
        double cg = 1.0;
        if (g > 0) {
            if (has_variant) running_min = uniform(running_min * em);
            else run_total = uniform(prev_sum / sig);   // (:1078-1088): every emission is 1, the sum is known
            cg = cg / sig;
        }
        double xs_of_grid = 1.0;   // (for k_bwd64d, which re-forms the columns of the odd grids: PassParams::fw_add)
        if (g == 0 || prm.always_normalize || running_min < prm.norm_threshold || g == G - 1) {
            const double xs = 1 / run_total;
            xs_of_grid = xs;
#pragma unroll
            for (int j = 0; j < NR; j++) {
#pragma unroll
                for (int i = 0; i < 16; i++) a[j][i] *= xs;
            }
#pragma unroll
            for (int r = 0; r < NL; r++) {
                if (r < NL - 1 || wave * 64 < n_last) {
                    double2 *st = lds_row<NL>(L, r, tt);
                    const int stride = lds_stride<NL>(r, n_last);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        double2 v = st[q * stride];
                        v.x *= xs; v.y *= xs;
                        st[q * stride] = v;
                    }
                }
            }
            if constexpr (SP) {
                for (int js = 0; js < NS; js++) {
                    double2 *st = spill_row(prm, p, js, tt);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        double2 v = st[q * NT];
                        v.x *= xs; v.y *= xs;
                        st[q * NT] = v;
                    }
                }
            }
            cg = (g == 0) ? xs : cg / run_total;
            run_total = 1;
            running_min = 1;
        }
        prev_sum = run_total;
        if (tt == 0) {
            L.sc[SC_C * 64 + jl] = cg;
            if (prm.fw_add) {
                prm.fw_add[(size_t)p * G + g] = addend;
                prm.fw_xs[(size_t)p * G + g] = xs_of_grid;
            }
        }
        const int sl = (int)uniform(L.sc[SC_SLOT * 64 + jl]);
        if (sl >= 0) {   // a thinned grid: the column goes out as the reference stores it (:1109-1113)
            double2 *dst = aout + (size_t)sl * col_vecs;
#pragma unroll
            for (int j = 0; j < NR; j++)
                if ((j * NT + tt) * 16 < K) store_chunk<double>(dst, a[j], j, NT, tt);
#pragma unroll
            for (int r = 0; r < NL; r++) {
                if ((r < NL - 1 || wave * 64 < n_last) && ((NR + r) * NT + tt) * 16 < K) {
                    const double2 *st = lds_row<NL>(L, r, tt);
                    const int stride = lds_stride<NL>(r, n_last);
#pragma unroll
                    for (int q = 0; q < 8; q++) dst[alpha_vec_index<8>(NR + r, q, NT, tt)] = st[q * stride];
                }
            }
            if constexpr (SP) {
                for (int js = 0; js < NS; js++) {
                    if (((NCH + js) * NT + tt) * 16 >= K) continue;
                    const double2 *st = spill_row(prm, p, js, tt);
#pragma unroll
                    for (int q = 0; q < 8; q++) dst[alpha_vec_index<8>(NCH + js, q, NT, tt)] = st[q * NT];
                }
            }
        }
    }
    __syncthreads();
    if (t < 64) {   // the last block of c
        const int gb = (G - 1) & ~63;
        if (gb + lane < G) prm.c[(size_t)p * G + gb + lane] = L.sc[SC_C * 64 + lane];
    }
}

// ---------------------------------------------------------------------------------------------
// backward, with the top-K picker at the thinned grids
// ---------------------------------------------------------------------------------------------
// the K_top largest of one value per lane, descending, into out[0..Ktop) (lane 0 writes); values are >= 0
__device__ __forceinline__ void wave_top(double v, int Ktop, double *out, int lane) {
    for (int r = 0; r < Ktop; r++) {
        const double m = wave_max(v);
        const unsigned long long owners = __ballot(v == m);
        const int first = __ffsll((long long)owners) - 1;
        if (lane == first) v = -1.0;
        if (lane == 0) out[r] = m;
    }
}

template <int NR, int NL, bool SP = false>
__global__ __launch_bounds__(kNT) void k_bwd64(PassParams prm, int n_last) {
    constexpr int NCH = NR + NL, NT = kNT, nwaves = NT >> 6;
    const int NS = SP ? prm.Kq / kRowHaps - NCH : 0;   // chunk rows streamed through HBM
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L(smem);
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t lds_etab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int K = prm.K, G = prm.G;
    const double *emat = static_cast<const double *>(prm.emat) + (size_t)p * G * kMaxRow;
    const double *esp = static_cast<const double *>(prm.esp) + (size_t)p * prm.esp_stride;
    const double *emin = prm.emin + (size_t)p * G;
    const size_t col_vecs = (size_t)prm.alpha_col_elems / 2;   // (beta_thin, if asked for, keeps the Kq pitch k_topk reads)
    const size_t col_vecs_q = (size_t)prm.Kq / 2;
    const double *cvec = prm.c + (size_t)p * G;
    const int32_t *slot = prm.alpha_slot + (size_t)p * G;
    const double2 *ain = reinterpret_cast<const double2 *>(static_cast<const double *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride);
    const double double_K = uniform((double)K);

    double b[NR][16];
    uint4 dh[NCH];   // codes of grid g + 1 during iteration g
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int k0 = (j * NT + t) * 16;
        if (j < NR) {
#pragma unroll
            for (int i = 0; i < 16; i++) b[j][i] = (k0 + i < K) ? 1.0 : 0.0;   // beta(G-1) = 1 / not_jump_prob = 1 (:1866-1870)
        } else if (j - NR < NL - 1 || t < n_last) {
            double2 *st = lds_row<NL>(L, j - NR, t);
            const int stride = lds_stride<NL>(j - NR, n_last);
#pragma unroll
            for (int q = 0; q < 8; q++) st[q * stride] = make_double2(k0 + 2 * q < K ? 1.0 : 0.0, k0 + 2 * q + 1 < K ? 1.0 : 0.0);
        }
        dh[j] = *reinterpret_cast<const uint4 *>(prm.hm + (size_t)(G - 1) * prm.Kp + k0);   // (unconditional: see k_fwd64)
    }
    if constexpr (SP) {
        for (int js = 0; js < NS; js++) {
            double2 *st = spill_row(prm, p, js, t);
            const int k0 = ((NCH + js) * NT + t) * 16;
#pragma unroll
            for (int q = 0; q < 8; q++) st[q * NT] = make_double2(k0 + 2 * q < K ? 1.0 : 0.0, k0 + 2 * q + 1 < K ? 1.0 : 0.0);
        }
    }
    if (t < kMaxRow) L.etab[((G - 1) & 1) * kMaxRow + t] = emat[(size_t)(G - 1) * kMaxRow + t];
    if (t == 0) L.misc[0] = 0;
    __syncthreads();

    // "+ val" and "* x" of the previous iteration, owed by the state (applied when a chunk is next touched)
    double val_prev = 0.0, x_prev = 1.0;
    double B_prev = 1, B_prev_star = uniform(double_K * cvec[G - 1] * 1.0);   // (:1871)
    for (int g = G - 1; g >= 0; --g) {
        const int jl = g & 63;
        if (jl == 63 || g == G - 1) {   // next block of 64 grids (descending): refill the scalar streams; entry l = grid base + l
            __syncthreads();
            if (t < 64) {
                const int gb = g & ~63;
                const int gi = clampi(gb + lane, 0, G - 1), g1 = clampi(gb + lane + 1, 0, G - 1), gs = clampi(gb + lane, 0, G > 1 ? G - 2 : 0);
                L.sc[SC_SIG * 64 + lane] = G > 1 ? prm.sigma[gs] : 1.0;
                L.sc[SC_TM1 * 64 + lane] = G > 1 ? prm.tm1[gs] : 0.0;
                L.sc[SC_EMIN * 64 + lane] = g1 == 1 ? prm.emin_b1[p] : emin[g1];   // of grid + 1 (grid 1: as the backward pass sees it)
                L.sc[SC_SPG * 64 + lane] = (double)prm.sp_gidx[g1];        // of grid + 1
                L.sc[SC_C * 64 + lane] = cvec[gi];
                L.sc[SC_SLOT * 64 + lane] = (double)slot[gi];
                L.sc[SC_TCOL * 64 + lane] = (double)prm.thin_col[gi];
            }
            __syncthreads();
        }
        const double c_g = uniform(L.sc[SC_C * 64 + jl]);
        double not_jump_prob = 1.0, val = 0.0;
        if (g < G - 1) {
            const int buf = (g + 1) & 1;   // grid g+1's table: the emission side
            const int tt = fresh_tid(wave);   // == t, see fresh_tid
            dma_table(emat, g, lds_etab, buf ^ 1, wave, tt & 63);   // grid g's table, for iteration g - 1 (that half was last read in g + 1)
            const double jump_prob = uniform(L.sc[SC_TM1 * 64 + jl]) / double_K;
            not_jump_prob = uniform(L.sc[SC_SIG * 64 + jl]);
            const double *et = L.etab + buf * kMaxRow;
            const int sp_g = (int)uniform(L.sc[SC_SPG * 64 + jl]);
            const int32_t *sp_at = sp_g >= 0 ? prm.sp_chunk_at + (size_t)sp_g * (prm.Kq >> 4) : nullptr;
            const bool has_variant = uniform(L.sc[SC_EMIN * 64 + jl]) >= 0;
            const uint8_t *hm_next = prm.hm + (size_t)g * prm.Kp;   // codes of grid g, for iteration g - 1
            double psum = 0;
            static_for<NCH>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const uint4 d = dh[j];
                const int k0 = (j * NT + tt) * 16;
                if constexpr (j < NR) {
                    psum += reg_chunk<true>(b[j], d, et, val_prev, x_prev, sp_at, esp, k0);
                } else {
                    if (j - NR < NL - 1 || wave * 64 < n_last)
                        psum += lds_chunk<true>(lds_row<NL>(L, j - NR, tt), lds_stride<NL>(j - NR, n_last), d, et, val_prev, x_prev, sp_at, esp, k0);
                }
                dh[j] = reinterpret_cast<const uint4 *>(hm_next + j * kRowHaps)[tt];
            });
            if constexpr (SP) {   // the streamed rows: codes of grid g + 1 fetched now
                const uint8_t *hm_e = prm.hm + (size_t)(g + 1) * prm.Kp;
                for (int js = 0; js < NS; js++) {
                    const int jg = NCH + js, k0 = (jg * NT + tt) * 16;
                    const uint4 d = reinterpret_cast<const uint4 *>(hm_e + (size_t)jg * kRowHaps)[tt];
                    psum += lds_chunk<true>(spill_row(prm, p, js, tt), NT, d, et, val_prev, x_prev, sp_at, esp, k0);
                }
            }
            const double sum_e_times_b = block_sum64<NCH>(psum, L.red + (g & 1) * 16, wave, lane, nwaves);
            // (:1945-1982)
            if (has_variant) {
                val = uniform(jump_prob / not_jump_prob * sum_e_times_b);
                B_prev = sum_e_times_b;
            } else {
                val = uniform(jump_prob / not_jump_prob * B_prev_star);
                B_prev = B_prev_star;
            }
            B_prev_star = uniform(c_g * B_prev);
        }
        const int tcol = (int)uniform(L.sc[SC_TCOL * 64 + jl]);
        if (tcol >= 0 && prm.K_top > 0) {
            // beta of this grid = state + val (:2020-2031); gamma = alpha * beta
            const int tp = fresh_tid(wave), lanep = tp & 63;   // == t, lane (see fresh_tid)
            const int sl = (int)uniform(L.sc[SC_SLOT * 64 + jl]);
            const double2 *av = ain + (size_t)sl * col_vecs;
            const int Ktop = prm.K_top;
            bool to_topk = !prm.fused_topk;
            // beta of element (vector q, half r) of chunk row j
            auto beta_of = [&](auto jc, int q, const double2 *st, int stride) -> double2 {
                constexpr int j = decltype(jc)::value;
                if constexpr (j < NR) return make_double2(b[j][2 * q] + val, b[j][2 * q + 1] + val);
                else { const double2 u = st[q * stride]; return make_double2(u.x + val, u.y + val); }
            };
            if (prm.fused_topk) {
                // ---- pass 1: per-lane maximum -> a lower bound of the K_top-th largest gamma
                // (padding beyond K: alpha is 0 there, so gamma is 0).  Over the FIRST chunk row only: the K_top-th largest of any
                // subset is a lower bound of the K_top-th largest of the whole column, and the bound only has to keep the candidate
                // list short (a sixth of the haplotypes: about six times K_top candidates instead of K_top) -- the alpha column of a
                // thinned grid then crosses HBM 1.16 times instead of twice (the picker was two 0.4 MB column reads per pass and grid).
                double mx = 0.0;
                static_for<NCH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j > 0) return;
                    const int k0 = (j * NT + tp) * 16;
                    if (k0 >= K) return;
                    const double2 *st = j >= NR ? lds_row<NL>(L, j - NR, tp) : nullptr;
                    const int stride = j >= NR ? lds_stride<NL>(j - NR, n_last) : 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double2 a2 = av[alpha_vec_index<8>(j, q, NT, tp)];
                        const double2 b2 = beta_of(jc, q, st, stride);
                        const double g0 = a2.x * b2.x, g1 = a2.y * b2.y;
                        mx = g0 > mx ? g0 : mx;
                        mx = g1 > mx ? g1 : mx;
                    }
                });
                wave_top(mx, Ktop, L.wtop + wave * kMaxTop, lanep);
                __syncthreads();
                double T0;
                {
                    double v = -1.0;
                    if (lanep < nwaves * Ktop) v = L.wtop[(lanep / Ktop) * kMaxTop + (lanep % Ktop)];
                    double m = 0;
                    for (int r = 0; r < Ktop; r++) {
                        m = wave_max(v);
                        const unsigned long long owners = __ballot(v == m);
                        const int first = __ffsll((long long)owners) - 1;
                        if (lanep == first) v = -1.0;
                    }
                    T0 = m > 0 ? m : 0.0;   // fewer than K_top lanes with a positive gamma: everything is a candidate
                }
                // ---- pass 2: candidates >= T0 into LDS
                static_for<NCH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int k0 = (j * NT + tp) * 16;
                    if (k0 >= K) return;
                    const double2 *st = j >= NR ? lds_row<NL>(L, j - NR, tp) : nullptr;
                    const int stride = j >= NR ? lds_stride<NL>(j - NR, n_last) : 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double2 a2 = av[alpha_vec_index<8>(j, q, NT, tp)];
                        const double2 b2 = beta_of(jc, q, st, stride);
                        const double gq[2] = {a2.x * b2.x, a2.y * b2.y};
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            if (gq[r] >= T0 && k0 + 2 * q + r < K) {
                                const int at = atomicAdd(&L.misc[0], 1);
                                if (at < kCandCap) { L.cand_v[at] = gq[r]; L.cand_k[at] = k0 + 2 * q + r; }
                            }
                        }
                    }
                });
                if constexpr (SP) {
                    for (int js = 0; js < NS; js++) {
                        const int k0 = ((NCH + js) * NT + tp) * 16;
                        if (k0 >= K) continue;
                        const double2 *st = spill_row(prm, p, js, tp);
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const double2 a2 = av[alpha_vec_index<8>(NCH + js, q, NT, tp)];
                            const double2 u = st[q * NT];
                            const double gq[2] = {a2.x * (u.x + val), a2.y * (u.y + val)};
#pragma unroll
                            for (int r = 0; r < 2; r++) {
                                if (gq[r] >= T0 && k0 + 2 * q + r < K) {
                                    const int at = atomicAdd(&L.misc[0], 1);
                                    if (at < kCandCap) { L.cand_v[at] = gq[r]; L.cand_k[at] = k0 + 2 * q + r; }
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                const int n_c = L.misc[0];
                if (n_c > kCandCap) {
                    to_topk = true;   // ties beyond the LDS list: k_topk does this grid from the beta column
                    if (tp == 0) prm.top_cnt[(size_t)p * prm.n_thin + tcol] = -1;
                } else if (wave == 0) {
                    // ---- exact selection by one wave: rank = position in (value descending, haplotype ascending) order
                    int32_t *oi = prm.top_idx + ((size_t)p * prm.n_thin + tcol) * prm.top_cap;
                    double *ov = static_cast<double *>(prm.top_val) + ((size_t)p * prm.n_thin + tcol) * prm.top_cap;
                    const int want = Ktop < n_c ? Ktop : n_c;   // the threshold is the want-th largest with multiplicity
                    double *thr_slot = reinterpret_cast<double *>(L.misc + 4);
                    for (int c0 = 0; c0 < n_c; c0 += 64) {
                        const int c = c0 + lanep;
                        const double v = c < n_c ? L.cand_v[c] : -1.0;
                        const int kk = c < n_c ? L.cand_k[c] : 0x7fffffff;
                        int rank = 0;
                        for (int m = 0; m < n_c; m++) {
                            const double vm = L.cand_v[m];
                            const int km = L.cand_k[m];
                            rank += (vm > v || (vm == v && km < kk)) ? 1 : 0;
                        }
                        if (c < n_c && rank == want - 1) *thr_slot = v;   // exactly one candidate holds each rank
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const double thr = want > 0 ? *thr_slot : 0.0;
                    int n_all = 0;
                    for (int c0 = 0; c0 < n_c; c0 += 64) {
                        const int c = c0 + lanep;
                        const double v = c < n_c ? L.cand_v[c] : -1.0;
                        const int kk = c < n_c ? L.cand_k[c] : 0x7fffffff;
                        const bool member = c < n_c && v >= thr;
                        n_all += __popcll(__ballot(member));
                        if (!__any(member)) continue;
                        int rank = 0;
                        for (int m = 0; m < n_c; m++) {
                            const double vm = L.cand_v[m];
                            const int km = L.cand_k[m];
                            rank += (vm > v || (vm == v && km < kk)) ? 1 : 0;
                        }
                        if (member && rank < prm.top_cap) { oi[rank] = kk; ov[rank] = v * not_jump_prob; }
                    }
                    if (lanep == 0) prm.top_cnt[(size_t)p * prm.n_thin + tcol] = n_all;
                }
                __syncthreads();   // the picker's scratch is reused at the next thinned grid
                if (tp == 0) L.misc[0] = 0;
            }
            if (to_topk && prm.beta_thin) {
                double2 *dst = reinterpret_cast<double2 *>(prm.beta_thin) + ((size_t)p * prm.n_thin + tcol) * col_vecs_q;
                static_for<NCH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int k0 = (j * NT + tp) * 16;
                    if (k0 >= K) return;
                    const double2 *st = j >= NR ? lds_row<NL>(L, j - NR, tp) : nullptr;
                    const int stride = j >= NR ? lds_stride<NL>(j - NR, n_last) : 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) dst[alpha_vec_index<8>(j, q, NT, tp)] = beta_of(jc, q, st, stride);   // (entries beyond K are ignored by k_topk)
                });
                if constexpr (SP) {
                    for (int js = 0; js < NS; js++) {
                        if (((NCH + js) * NT + tp) * 16 >= K) continue;
                        const double2 *st = spill_row(prm, p, js, tp);
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const double2 u = st[q * NT];
                            dst[alpha_vec_index<8>(NCH + js, q, NT, tp)] = make_double2(u.x + val, u.y + val);
                        }
                    }
                }
            }
        }
        val_prev = val;
        x_prev = uniform(c_g * not_jump_prob);   // beta *= c_g * sigma_g (:2165-2166), applied when the state is next touched
    }
}


// ---------------------------------------------------------------------------------------------
// backward of a DOSAGE pass with fp64 state (qa_panel_set_dosage_precision(64)): the reference's arithmetic throughout
// (reference-single.cpp:1781-2179 with the lazily normalised alpha of k_fwd64), gamma = alpha * beta histogrammed by haplotype
// code for k_dosage (:2083-2139).  Since the end of round 5 k_fwd64 hands alpha over at every SECOND grid (PassParams::fw_add):
// at an odd grid this kernel fetches the column of the even grid below it and re-forms its own from it (reform_alpha: the forward
// step is elementwise given the grid's two scalars, and the codes and the emission table it needs are the ones the beta update
// decodes and gathers anyway) -- the forward kernel's operations on the forward kernel's values, the same bits.
//
// Bytes: 1 B code + 8 B alpha per cell -- with the forward's 1 + 8 the 18 K G of SURVEY.md 8(d) ("fp64 alpha as in the
// reference"; the forward now writes half of its 8).  At 0.8 GB of alpha per pass the kernel is bound by that stream (one pass
// per compute unit, 256 at a time: 205 GB per launch), not by its arithmetic, so the structure differs from k_bwd64's:
//   * gamma of grid g is formed when the state is next touched anyway -- in the chunk loop that applies grid g's
//     emissions (iteration g - 1), where beta_g = state + val is a by-product and the haplotype codes of grid g are already
//     decoded for the emission look-up: one pass over the state per grid, one decode per cell.  Grid 0 gets an epilogue.
//   * alpha and the codes are fetched ONE CHUNK AHEAD (9 x 16 B per lane in flight, 36 KB per compute unit: enough for the
//     stream at this rate) instead of a grid ahead: 72 registers for two chunks of alpha and codes, which is what fits
//     beside four chunk rows of state.
//   * the histogram: LDS u64 bins, 8 copies per code (lane & 7) -- all the LDS the three state rows leave (16 KB).  A cell
//     adds gamma * sigma_g in fixed point at 2^-51: the double (gamma * sigma_g * 2^51 + 2^52) carries that integer in its
//     mantissa (one multiply, one add, one AND instead of a float -> u64 conversion).  gamma * sigma sums to 1 over a grid
//     (colSums(gamma_t) == 1), so a bin never exceeds 2^51; rounding is to nearest at 2^-52 of the total: 1e-16 per cell,
//     ~1e-13 over K = 50 000 cells (the bar against the oracle is 1e-9).  Integer adds commute: the bins do not depend on
//     the order the waves reach them.
// ---------------------------------------------------------------------------------------------
constexpr int kHistCopiesD = 8;
constexpr int kNStreamD = 7;   // SC_SIG .. SC_C, SC_TCOL (here: the forward pass's rescaling factor of the grid; SC_SLOT holds its addend)
constexpr size_t kLdsFixedD = 2 * kMaxRow * 8 + 2 * 16 * 8 + kNStreamD * 64 * 8 + (size_t)kMaxRow * kHistCopiesD * 8;
inline size_t lds_bytes_d(const Geo64 &g) { return kLdsFixedD + (g.NL > 0 ? ((size_t)(g.NL - 1) * kNT + g.n_last) * 128 : 0); }
// geo64() sizes NR / NL by the ranking kernels' LDS; the dosage backward's fixed part is its own.  The tightest <4, 3> case (seven
// rows, a one-wave last row: K = 49 153 .. 50 176) must fit, or a band of K just below 50 176 would stop working while its
// neighbours still do; wider last rows go to <5, 2> in launch_fb64_dosage.
static_assert(kLdsFixedD + (2 * (size_t)kNT + 64) * 128 <= kLdsMax, "k_bwd64d: one more staged scalar stream does not fit beside three LDS rows");

struct LdsD {
    double *etab;                // [2][256]
    double *red;                 // [2][16]
    double *sc;                  // [kNStreamD][64]
    unsigned long long *hist;    // [256][kHistCopiesD]
    double2 *state;
    __device__ __forceinline__ explicit LdsD(char *smem) {
        etab = reinterpret_cast<double *>(smem);
        red = etab + 2 * kMaxRow;
        sc = red + 2 * 16;
        hist = reinterpret_cast<unsigned long long *>(sc + kNStreamD * 64);
        state = reinterpret_cast<double2 *>(hist + kMaxRow * kHistCopiesD);
    }
};

// eight haplotypes of a chunk: t = x + v is beta of the grid whose emissions are about to be applied; gamma = alpha * t goes
// to the histogram bin of the haplotype's code; then x <- (t * s) * table[code] (k_bwd64's update)
template <bool EMIT>
__device__ __forceinline__ void half_step_dos(double (&x)[8], uint32_t w0, uint32_t w1, const double *et, double v, double s,
                                              const double2 *a, double scale, unsigned long long *hist_lane, bool hist_on) {
    const uint32_t w[2] = {w0, w1};
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {   // look-ups in batches of four, back to back, then the arithmetic
        double e[4];
        uint32_t code[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            code[i] = (w[hh] >> (i * 8)) & 0xffu;
            if (EMIT) e[i] = et[code[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ii = 4 * hh + i;
            const double t = x[ii] + v;
            const double gam = ((ii & 1) ? a[ii >> 1].y : a[ii >> 1].x) * t;
            const double y = gam * scale + 4503599627370496.0;   // 2^52: the mantissa now holds round(gam * scale)
            const unsigned long long bits = (unsigned long long)__double_as_longlong(y) & 0x000fffffffffffffull;
            if (hist_on) atomicAdd(hist_lane + code[i] * kHistCopiesD, bits);
            if (EMIT) x[ii] = (t * s) * e[i];
        }
    }
}

// gamma of the special haplotypes (code 0) of half a chunk -> their own list (:2096-2128); `at` = index of the half's first
// special in the pass's gsp array, k0 the haplotype of its first element
__device__ __forceinline__ void special_gammas(const double (&x_before)[8], uint32_t w0, uint32_t w1, double v, const double2 *a,
                                               double *gsp, int at, int k0, int K) {
    const uint32_t w[2] = {w0, w1};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t code = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
        if (code == 0 && k0 + i < K) {
            gsp[at] = ((i & 1) ? a[i >> 1].y : a[i >> 1].x) * (x_before[i] + v);
            at++;
        }
    }
}

// alpha of half a chunk at an odd grid from the even grid's below it: k_fwd64's step (half_step<false>, special_half, the
// rescaling) on the eight values
__device__ __forceinline__ void reform_alpha(double2 (&a)[4], uint32_t w0, uint32_t w1, const double *et, const double *esp_at, bool sp,
                                             double add, double xs) {
    const uint32_t w[2] = {w0, w1};
    int at = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t code = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
        double y = (((i & 1) ? a[i >> 1].y : a[i >> 1].x) + add) * et[code];
        if (sp && code == 0) {
            y *= esp_at[at];
            at++;
        }
        y *= xs;
        if (i & 1) a[i >> 1].y = y; else a[i >> 1].x = y;
    }
}

template <int NR, int NL, bool SP = false>
__global__ __launch_bounds__(kNT) void k_bwd64d(PassParams prm, int n_last) {
    constexpr int NCH = NR + NL, NT = kNT, nwaves = NT >> 6;
    const int NS = SP ? prm.Kq / kRowHaps - NCH : 0;   // chunk rows streamed through HBM
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsD L(smem);
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t lds_etab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int K = prm.K, G = prm.G;
    const double *emat = static_cast<const double *>(prm.emat) + (size_t)p * G * kMaxRow;
    const double *esp = static_cast<const double *>(prm.esp) + (size_t)p * prm.esp_stride;
    const double *emin = prm.emin + (size_t)p * G;
    const size_t col_vecs = (size_t)prm.alpha_col_elems / 2;
    const double *cvec = prm.c + (size_t)p * G;
    const double2 *ain = reinterpret_cast<const double2 *>(static_cast<const double *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride);
    double *gsp = static_cast<double *>(prm.gsp) + (size_t)p * prm.n_special;
    unsigned long long *mg = static_cast<unsigned long long *>(prm.mg) + (size_t)p * G * kMaxRow;
    const double double_K = uniform((double)K);
    const bool last_row_wave = wave * 64 < n_last;   // the last chunk row holds n_last lanes only
    const bool half_cols = prm.fw_add != nullptr;    // alpha handed over at the even grids only

    double b[NR][16];
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int k0 = (j * NT + t) * 16;
        if (j < NR) {
#pragma unroll
            for (int i = 0; i < 16; i++) b[j][i] = (k0 + i < K) ? 1.0 : 0.0;   // beta(G-1) = 1 (:1866-1870)
        } else if (j - NR < NL - 1 || t < n_last) {
            double2 *st = lds_row<NL>(L, j - NR, t);
            const int stride = lds_stride<NL>(j - NR, n_last);
#pragma unroll
            for (int q = 0; q < 8; q++) st[q * stride] = make_double2(k0 + 2 * q < K ? 1.0 : 0.0, k0 + 2 * q + 1 < K ? 1.0 : 0.0);
        }
    }
    if constexpr (SP) {
        for (int js = 0; js < NS; js++) {
            double2 *st = spill_row(prm, p, js, t);
            const int k0 = ((NCH + js) * NT + t) * 16;
#pragma unroll
            for (int q = 0; q < 8; q++) st[q * NT] = make_double2(k0 + 2 * q < K ? 1.0 : 0.0, k0 + 2 * q + 1 < K ? 1.0 : 0.0);
        }
    }
    for (int i = t; i < kMaxRow * kHistCopiesD; i += NT) L.hist[i] = 0ull;
    if (t < kMaxRow) L.etab[((G - 1) & 1) * kMaxRow + t] = emat[(size_t)(G - 1) * kMaxRow + t];

    // half a chunk (8 haplotypes of row j) of grid gq: its alpha (4 vectors) and, with the first half, the chunk's codes --
    // all unconditional 16-byte loads.  A wave without a chunk in the last row fetches one fixed line instead (no branch around
    // loads: the compiler could not count them).
    uint4 d_nx;
    double2 a_nx[4];
    auto prefetch = [&](int gq, int j, int h, int tt) {
        const bool real = SP || j < NCH - 1 || last_row_wave;   // (streamed geometry: every row is full)
        // (the codes' row pitch covers whole chunk rows: beyond K they are the zero padding, whose emission is 0)
        if (h == 0) d_nx = reinterpret_cast<const uint4 *>(prm.hm + (size_t)gq * prm.Kp + (size_t)j * kRowHaps)[tt];
        // (every second column handed over: grid gq's own when gq is even, else the one of grid gq - 1, from which process() re-forms it)
        const double2 *av = ain + (size_t)(half_cols ? gq >> 1 : gq) * col_vecs;
#pragma unroll
        for (int q = 0; q < 4; q++) a_nx[q] = av[real ? alpha_vec_index<8>(j, 4 * h + q, NT, tt) : (size_t)(q * 64 + (tt & 63))];
    };
    prefetch(G - 1, 0, 0, t);
    __syncthreads();
    // (Round 6, what bounds this kernel: 256 / 128 / 64 passes per launch take 39.0 / 35.9 / 33.6 ms -- a workgroup's own serial
    // progress, i.e. the latency of its half-chunk-ahead loads against 1.2 us of work per half chunk, not the device's bandwidth.
    // Measured and dropped: touching the half chunk TWO steps ahead with one dword per 128-byte line, so that the real loads hit
    // L2 -- three more live registers and the in-order vmcnt waits on the touches: 35 -> 70 spilled registers, 39.0 -> 53 ms.)

    // one pass over the state: gamma of grid gp (its alpha, its codes) into the histogram and, unless gp == 0, the update
    // with grid gp's emissions.  Returns this thread's share of sum(e * beta).
    // `recomp`: gp is an odd grid of a pass whose alpha was handed over at the even grids only -- the vectors fetched are grid
    // gp - 1's, and grid gp's are re-formed from them as k_fwd64 formed them: (a + addend) * e[code], times the special's own
    // emission, times the grid's rescaling factor (1 where the forward pass did not renormalise): the same operations on the
    // same values in the same order, so the same bits.
    auto process = [&](auto emit_c, int gp, const double *et, const int32_t *sp_at, int sp_g, double v, double s, double fs,
                       bool recomp, double f_add, double f_xs) -> double {
        constexpr bool EMIT = decltype(emit_c)::value;
        const double scale = uniform(fs * 2251799813685248.0);   // sigma_gp * 2^51
        double psum = 0;
        uint4 d = make_uint4(0, 0, 0, 0);
        int n_sp = 0;   // specials of the chunk's first half
        static_for<2 * NCH>([&](auto jc) {
            constexpr int j = decltype(jc)::value >> 1, h = decltype(jc)::value & 1;
            const int tt = fresh_tid(wave);
            if (h == 0) d = d_nx;
            double2 a[4];
#pragma unroll
            for (int q = 0; q < 4; q++) a[q] = a_nx[q];
            // the next half chunk (of this grid, or the first of the next grid down) is in flight while this one is consumed
            if constexpr (h == 0) prefetch(gp, j, 1, tt);
            else if constexpr (j + 1 < NCH) prefetch(gp, j + 1, 0, tt);
            else if (SP && NS > 0) prefetch(gp, NCH, 0, tt);   // on to the streamed rows
            else prefetch(gp > 0 ? gp - 1 : 0, 0, 0, tt);
            const int k0 = (j * NT + tt) * 16;
            unsigned long long *hl = L.hist + (tt & (kHistCopiesD - 1));
            const bool on = k0 < K;   // (a chunk wholly beyond K was never stored by the forward)
            const uint32_t w0 = h ? d.z : d.x, w1 = h ? d.w : d.y;
            const bool sp = sp_at && (has_zero_byte(w0) || has_zero_byte(w1));
            auto body = [&](double (&x)[8]) {
                if (h == 0) n_sp = 0;
                int at = 0;
                if (sp) at = sp_at[k0 >> 4] + n_sp;   // position in the pass's (lazily padded) special-emission array
                if (recomp) reform_alpha(a, w0, w1, et, esp + at, sp, f_add, f_xs);
                if (sp) {
                    if (on) special_gammas(x, w0, w1, v, a, gsp, at - 16 * sp_g, k0 + 8 * h, K);
                }
                half_step_dos<EMIT>(x, w0, w1, et, v, s, a, scale, hl, on);
                if (EMIT) {
                    if (sp) n_sp += special_half(x, w0, w1, esp + at);
                    psum += sum8(x);
                }
            };
            if constexpr (j < NR) {
                body(*reinterpret_cast<double (*)[8]>(&b[j][8 * h]));
            } else {
                if (j - NR < NL - 1 || last_row_wave) {
                    double2 *st = lds_row<NL>(L, j - NR, tt) + (size_t)(4 * h) * lds_stride<NL>(j - NR, n_last);
                    const int stride = lds_stride<NL>(j - NR, n_last);
                    double x[8];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const double2 u = st[q * stride];
                        x[2 * q] = u.x;
                        x[2 * q + 1] = u.y;
                    }
                    body(x);
                    if (EMIT) {
#pragma unroll
                        for (int q = 0; q < 4; q++) st[q * stride] = make_double2(x[2 * q], x[2 * q + 1]);
                    }
                }
            }
        });
        if constexpr (SP) {   // the streamed rows: the same half-chunk pipeline, the state read and written in place in HBM
            for (int js = 0; js < NS; js++) {
                const int j = NCH + js;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int tt = fresh_tid(wave);
                    if (h == 0) d = d_nx;
                    double2 a[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) a[q] = a_nx[q];
                    if (h == 0) prefetch(gp, j, 1, tt);
                    else if (js + 1 < NS) prefetch(gp, j + 1, 0, tt);
                    else prefetch(gp > 0 ? gp - 1 : 0, 0, 0, tt);
                    const int k0 = (j * NT + tt) * 16;
                    unsigned long long *hl = L.hist + (tt & (kHistCopiesD - 1));
                    const bool on = k0 < K;
                    const uint32_t w0 = h ? d.z : d.x, w1 = h ? d.w : d.y;
                    const bool sp = sp_at && (has_zero_byte(w0) || has_zero_byte(w1));
                    double2 *st = spill_row(prm, p, js, tt) + (size_t)(4 * h) * NT;
                    double x[8];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const double2 u = st[q * NT];
                        x[2 * q] = u.x;
                        x[2 * q + 1] = u.y;
                    }
                    if (h == 0) n_sp = 0;
                    int at = 0;
                    if (sp) at = sp_at[k0 >> 4] + n_sp;
                    if (recomp) reform_alpha(a, w0, w1, et, esp + at, sp, f_add, f_xs);
                    if (sp) {
                        if (on) special_gammas(x, w0, w1, v, a, gsp, at - 16 * sp_g, k0 + 8 * h, K);
                    }
                    half_step_dos<EMIT>(x, w0, w1, et, v, s, a, scale, hl, on);
                    if (EMIT) {
                        if (sp) n_sp += special_half(x, w0, w1, esp + at);
                        psum += sum8(x);
#pragma unroll
                        for (int q = 0; q < 4; q++) st[q * NT] = make_double2(x[2 * q], x[2 * q + 1]);
                    }
                }
            }
        }
        return psum;
    };
    // fold the eight copies of every bin into mg[gp] and clear them (after the barrier that ends gp's atomics; a barrier
    // of its own before the next grid's begin)
    auto fold = [&](int gp) {
        if (t < kMaxRow) {
            unsigned long long *h = L.hist + t * kHistCopiesD;
            unsigned long long v = 0;
#pragma unroll
            for (int c = 0; c < kHistCopiesD; c++) { v += h[c]; h[c] = 0ull; }
            mg[(size_t)gp * kMaxRow + t] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    double val_prev = 0.0, x_prev = 1.0, sig_prev = 1.0;
    double B_prev = 1, B_prev_star = uniform(double_K * cvec[G - 1] * 1.0);   // (:1871)
    for (int g = G - 1; g >= 0; --g) {
        const int jl = g & 63;
        if (jl == 63 || g == G - 1) {
            __syncthreads();
            if (t < 64) {
                const int gb = g & ~63;
                const int gi = clampi(gb + lane, 0, G - 1), g1 = clampi(gb + lane + 1, 0, G - 1), gs = clampi(gb + lane, 0, G > 1 ? G - 2 : 0);
                L.sc[SC_SIG * 64 + lane] = G > 1 ? prm.sigma[gs] : 1.0;
                L.sc[SC_TM1 * 64 + lane] = G > 1 ? prm.tm1[gs] : 0.0;
                L.sc[SC_EMIN * 64 + lane] = g1 == 1 ? prm.emin_b1[p] : emin[g1];   // of grid + 1 (grid 1: as the backward pass sees it)
                L.sc[SC_SPG * 64 + lane] = (double)prm.sp_gidx[g1];        // of grid + 1
                L.sc[SC_C * 64 + lane] = cvec[gi];
                if (half_cols) {   // the forward pass's addend and rescaling factor, of grid + 1 (the grid process() handles)
                    L.sc[SC_SLOT * 64 + lane] = prm.fw_add[(size_t)p * G + g1];
                    L.sc[SC_TCOL * 64 + lane] = prm.fw_xs[(size_t)p * G + g1];
                }
            }
            __syncthreads();
        }
        const double c_g = uniform(L.sc[SC_C * 64 + jl]);
        double not_jump_prob = 1.0, val = 0.0;
        if (g < G - 1) {
            const int buf = (g + 1) & 1;
            const int tt = fresh_tid(wave);
            dma_table(emat, g, lds_etab, buf ^ 1, wave, tt & 63);   // grid g's table, for iteration g - 1
            const double jump_prob = uniform(L.sc[SC_TM1 * 64 + jl]) / double_K;
            not_jump_prob = uniform(L.sc[SC_SIG * 64 + jl]);
            const double *et = L.etab + buf * kMaxRow;
            const int sp_g = (int)uniform(L.sc[SC_SPG * 64 + jl]);
            const int32_t *sp_at = sp_g >= 0 ? prm.sp_chunk_at + (size_t)sp_g * (prm.Kq >> 4) : nullptr;
            const bool has_variant = uniform(L.sc[SC_EMIN * 64 + jl]) >= 0;
            const bool recomp = half_cols && ((g + 1) & 1);
            const double f_add = half_cols ? uniform(L.sc[SC_SLOT * 64 + jl]) : 0.0, f_xs = half_cols ? uniform(L.sc[SC_TCOL * 64 + jl]) : 1.0;
            const double psum = process(std::true_type{}, g + 1, et, sp_at, sp_g, val_prev, x_prev, sig_prev, recomp, f_add, f_xs);
            // (the table DMA is older than the 5 loads of the half chunk fetched ahead: wait for all but those)
            const double sum_e_times_b = block_sum64<5>(psum, L.red + (g & 1) * 16, wave, lane, nwaves);
            fold(g + 1);
            if (has_variant) {   // (:1945-1982)
                val = uniform(jump_prob / not_jump_prob * sum_e_times_b);
                B_prev = sum_e_times_b;
            } else {
                val = uniform(jump_prob / not_jump_prob * B_prev_star);
                B_prev = B_prev_star;
            }
            B_prev_star = uniform(c_g * B_prev);
        }
        val_prev = val;
        x_prev = uniform(c_g * not_jump_prob);   // beta *= c_g * sigma_g (:2165-2166), applied when the state is next touched
        sig_prev = not_jump_prob;
    }
    // grid 0: gamma only
    {
        const int sp_g = prm.sp_gidx[0];
        const int32_t *sp_at = sp_g >= 0 ? prm.sp_chunk_at + (size_t)sp_g * (prm.Kq >> 4) : nullptr;
        process(std::false_type{}, 0, L.etab, sp_at, sp_g, val_prev, x_prev, sig_prev, false, 0.0, 1.0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fold(0);
    }
}

template <int NR, int NL, bool SP = false>
void launch_dos(const PassParams &prm, const Geo64 &geo, hipStream_t s, hipEvent_t e_mid) {
    const size_t lds = lds_bytes(geo), lds_d = lds_bytes_d(geo);
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd64<NR, NL, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_fwd64<NR, NL, SP>), dim3(prm.P), dim3(kNT), lds, s, prm, geo.n_last);
    QA_HIP(hipGetLastError());
    if (e_mid) QA_HIP(hipEventRecord(e_mid, s));
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bwd64d<NR, NL, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
    hipLaunchKernelGGL((k_bwd64d<NR, NL, SP>), dim3(prm.P), dim3(kNT), lds_d, s, prm, geo.n_last);
    QA_HIP(hipGetLastError());
}

template <int NR, int NL, bool SP = false>
void launch(const PassParams &prm, const Geo64 &geo, hipStream_t s, hipEvent_t e_mid) {
    const size_t lds = lds_bytes(geo);
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd64<NR, NL, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_fwd64<NR, NL, SP>), dim3(prm.P), dim3(kNT), lds, s, prm, geo.n_last);
    QA_HIP(hipGetLastError());
    if (e_mid) QA_HIP(hipEventRecord(e_mid, s));
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bwd64<NR, NL, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_bwd64<NR, NL, SP>), dim3(prm.P), dim3(kNT), lds, s, prm, geo.n_last);
    QA_HIP(hipGetLastError());
}

}  // namespace

namespace qa {

int fb64_chunks(int K) { const Geo64 g = geo64(K); return g.NCH + g.NS; }   // chunk rows in all: on chip + streamed
int fb64_spill_rows(int K) { return geo64(K).NS; }
size_t fb64_lds_bytes(int K) { return lds_bytes(geo64(K)); }

// elements per stored alpha column of the dosage passes: the lane-interleaved layout of the chunk rows in use, no padding to
// whole rows (K = 50 000: 50 176 instead of 57 344 doubles -- 0.80 instead of 0.92 GB per pass over 2 000 grids)
size_t fb64_alpha_col_elems(int K) {
    const Geo64 g = geo64(K);
    if (g.NS > 0) return (size_t)(g.NCH + g.NS) * kRowHaps;   // streamed geometry: whole rows
    return g.NCH ? (size_t)(g.NCH - 1) * kRowHaps + (size_t)g.n_last * 16 : 0;
}
size_t fb64_dos_lds_bytes(int K) { return lds_bytes_d(geo64(K)); }

void launch_fb64_dosage(const void *pass_params, hipStream_t st, hipEvent_t e_mid) {
    const PassParams &prm = *static_cast<const PassParams *>(pass_params);
    Geo64 geo = geo64(prm.K);
    // (the dosage backward's LDS, not the ranking kernels', decides here: a <4, 3> geometry whose three LDS rows do not fit beside
    // k_bwd64d's histogram takes five rows in registers, as the ranking pair does)
    if (geo.NCH == 7 && geo.NS == 0 && geo.NR == 4 && lds_bytes_d(geo) > kLdsMax) { geo.NR = 5; geo.NL = 2; }
    if (geo.NCH == 0 || lds_bytes_d(geo) > kLdsMax) throw std::runtime_error("K exceeds the on-chip capacity of the fp64 dosage kernels");
    if (prm.Kq != (geo.NCH + geo.NS) * kRowHaps) throw std::runtime_error("internal: Kq does not match the fp64 geometry");
    if (geo.NS > 0) {
        if (!prm.spill) throw std::runtime_error("internal: streamed chunk rows without their buffer");
        launch_dos<5, 2, true>(prm, geo, st, e_mid);
        return;
    }
    switch (geo.NR * 10 + geo.NL) {
#ifndef QA_FAST_BUILD
        case 10: launch_dos<1, 0>(prm, geo, st, e_mid); break;
        case 20: launch_dos<2, 0>(prm, geo, st, e_mid); break;
        case 30: launch_dos<3, 0>(prm, geo, st, e_mid); break;
        case 40: launch_dos<4, 0>(prm, geo, st, e_mid); break;
        case 41: launch_dos<4, 1>(prm, geo, st, e_mid); break;
        case 42: launch_dos<4, 2>(prm, geo, st, e_mid); break;
#endif
        case 52: launch_dos<5, 2>(prm, geo, st, e_mid); break;
        case 43: launch_dos<4, 3>(prm, geo, st, e_mid); break;
        default: throw std::runtime_error("fp64 geometry not built");
    }
}

void launch_fb64(const void *pass_params, hipStream_t st, hipEvent_t e_mid) {
    const PassParams &prm = *static_cast<const PassParams *>(pass_params);
    Geo64 geo = geo64(prm.K);
    if (geo.NCH == 0) throw std::runtime_error("K exceeds the on-chip capacity of the fp64 ranking kernels");
    // A seven-row panel (K = 50 000) fits as <4, 3> and as <5, 2>.  The ranking pair is bound by instruction issue (counters:
    // VALU 54 %, everything 77 % of SIMD cycles) and a row that lives in LDS costs a ds_read and a ds_write per two cells beside the
    // table gather, so it takes five rows in registers although that spills a little more: stand-alone, 256 passes, 200 thinned grids, k_fwd64 11.4 -> 10.6 ms, k_bwd64 16.9 -> 16.4
    // (10 thinned grids: 9.0 -> 7.8, 9.5 -> 8.6).  The dosage pair streams alpha through HBM and does not care (31.4 / 41.1 against
    // 32.0 / 40.9 ms): it keeps <4, 3>.  QA_FB64_FOUR_ROWS=1: the ranking pair as before.
    static const bool four = [] { const char *e = getenv("QA_FB64_FOUR_ROWS"); return e && e[0] == '1'; }();
    if (!four && geo.NS == 0 && geo.NCH == 7 && geo.NR == 4) { geo.NR = 5; geo.NL = 2; }
    if (prm.Kq != (geo.NCH + geo.NS) * kRowHaps) throw std::runtime_error("internal: Kq does not match the fp64 geometry");
    if (geo.NS > 0) {
        if (!prm.spill) throw std::runtime_error("internal: streamed chunk rows without their buffer");
        launch<5, 2, true>(prm, geo, st, e_mid);
        return;
    }
    switch (geo.NR * 10 + geo.NL) {
#ifndef QA_FAST_BUILD
        case 10: launch<1, 0>(prm, geo, st, e_mid); break;
        case 20: launch<2, 0>(prm, geo, st, e_mid); break;
        case 30: launch<3, 0>(prm, geo, st, e_mid); break;
        case 40: launch<4, 0>(prm, geo, st, e_mid); break;
        case 41: launch<4, 1>(prm, geo, st, e_mid); break;
        case 42: launch<4, 2>(prm, geo, st, e_mid); break;
#endif
        case 52: launch<5, 2>(prm, geo, st, e_mid); break;
        case 43: launch<4, 3>(prm, geo, st, e_mid); break;
        default: throw std::runtime_error("fp64 geometry not built");
    }
}

}  // namespace qa
