// fullpass_ref.hip -- VALIDATION MODE of the full-panel forward/backward (qa_panel_set_sum_order(panel, 1)): the same
// per-element arithmetic as the production kernels (fullpass64.hip), but every K-wide sum is formed IN THE REFERENCE'S ORDER:
//   * forward  run_total  (QUILT/src/reference-single.cpp:1002-1075): the grid's special haplotypes first, in list order,
//     then k = 0 .. K-1 one after the other (specials contribute an exact 0 there and are subtracted as an exact 0);
//   * grid 0   c(0) = 1 / sum(alphaHat_t_col) (:2347): NOT an explicit loop but Armadillo's sum() of an arma::colvec, i.e.
//     arrayops::accumulate (armadillo_bits/arrayops_meat.hpp; no -ffast-math): the even-indexed k into one accumulator, the
//     odd-indexed k into a second, each in increasing k, acc1 + acc2 at the end (serial_sum_arma).  Armadillo is not in the
//     build image, so this is restated from the library's published source, not observed; mode 2 of qa_panel_set_sum_order
//     adds grid 0 left to right like the explicit loops (what rounds 4-5 did), so that a maintainer with R can tell which is
//     right from one c(0) printed at full precision (oracle/quilt_oracle.h, "Armadillo's sum()");
//   * backward sum_e_times_b (:1899-1955): specials first, then k = 0 .. K-1;
//   * dosage   matched_gammas(dh) += gamma(k), k = 0 .. K-1 (:2083-2091), the specials' terms in list order (:2096-2128),
//     then dh = 0 .. nMaxDH-1 per SNP (:2129-2139).
// The reference is built without -ffast-math (QUILT/src/Makevars), so its compiler may not re-associate these loops: the
// order above IS the reference's arithmetic.  A floating-point sum in a prescribed order cannot be spread over lanes; here
// wave 0 of the pass's workgroup adds the values one at a time (64 coalesced loads, then 64 dependent adds fed by
// v_readlane), which costs ~10 clocks per haplotype and grid -- 20-50x slower than the production kernels.  This mode exists
// to PROVE a statement, not to be fast: with it the device's best-haplotype lists, c, alpha, beta and dosage equal the CPU
// restatement's (oracle/fullpass.c) bit for bit, so every difference between the production mode and the CPU path on
// tie-rich panels is the order of these sums and nothing else (tests/test_sum_order_gpu.py).
//
// Like the production kernels it takes the reference's branch at grid 1 of the BACKWARD pass (:1866-1877: "grid_has_variant"
// is not forced there, unlike the forward pass :964-966; PassParams::emin_b1).
//
// Layout: one workgroup of 256 threads per pass; state (alpha resp. beta, and the gamma column) in plain k order, in LDS
// when 2 K doubles fit, else in the pass's HBM scratch (PassParams::spill); checkpoints / outputs in the lane-interleaved
// layout of the generic kernels (geometry NT = 256) so that k_topk / k_unpermute and the host side read them unchanged.
#include "fullpass_dev.hpp"

namespace {

constexpr int kRT = 256;   // threads per pass == kMaxRow (one emission-table row per thread)

// position of haplotype k in a lane-interleaved column (fullpass_dev.hpp: alpha_vec_index)
__device__ __forceinline__ size_t perm_index(int k, int NT) {
    const int chunk = k >> 4, e = k & 15;
    const int j = chunk / NT, t = chunk % NT;
    return alpha_vec_index<8>(j, e >> 1, NT, t) * 2 + (e & 1);
}

template <int I>
__device__ __forceinline__ double lane_value(double x) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), I);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), I);
    return __hiloint2double(hi, lo);
}
// s <- ((s + x[lane 0]) + x[lane 1]) + ... + x[lane 63]
__device__ __forceinline__ double add_lanes_in_order(double s, double x) {
    static_for<64>([&](auto ic) { s += lane_value<decltype(ic)::value>(x); });
    return s;
}

// wave 0, all 64 lanes: s + v[0] + v[1] + ... + v[K-1], left to right.  SKIP0: haplotypes with code 0 contribute an exact
// zero (the reference adds alphaHat_t_col(k) = (..) * eMatDH_col(0) = 0 for them).  Lanes past K add +0.0: x + 0.0 == x.
template <bool SKIP0>
__device__ double serial_sum(const double *v, const uint8_t *code, int K, double s, int lane) {
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        double x = 0.0;
        if (k < K) {
            x = v[k];
            if (SKIP0 && code[k] == 0) x = 0.0;
        }
        s = add_lanes_in_order(s, x);
    }
    return s;
}
// Armadillo's arrayops::accumulate over v[0 .. K-1]: acc1 = v[0] + v[2] + ..., acc2 = v[1] + v[3] + ... (each left to right; an
// odd K's last element has an even index and so lands in acc1, which is the routine's "tail" rule), result acc1 + acc2.  A
// block of 64 consecutive k starts at an even k, so even lanes feed acc1 and odd lanes acc2: two chains of 32 dependent adds.
__device__ double serial_sum_arma(const double *v, int K, int lane) {
    double acc1 = 0.0, acc2 = 0.0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const double x = k < K ? v[k] : 0.0;   // (lanes past K add +0.0: a + 0.0 == a)
        static_for<32>([&](auto ic) {
            acc1 += lane_value<2 * decltype(ic)::value>(x);
            acc2 += lane_value<2 * decltype(ic)::value + 1>(x);
        });
    }
    return acc1 + acc2;
}
// s + v[list[0]] + v[list[1]] + ... (the grid's special haplotypes, in list order)
__device__ double serial_gather_sum(const double *v, const int32_t *list, int n, double s, int lane) {
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const double x = i < n ? v[list[i]] : 0.0;
        s = add_lanes_in_order(s, x);
    }
    return s;
}

struct GridEm {
    const double *et;        // the grid's emission table (LDS)
    const double *esp_g;     // the grid's special emissions, list order
    const int32_t *sp_k;     // the grid's special list
    int sn;
    __device__ __forceinline__ double at(int k, uint32_t code) const {
        if (code) return et[code];
        const int i = special_lower_bound(sp_k, 0, sn, k);
        return esp_g[i];
    }
};
__device__ __forceinline__ GridEm grid_em(const PassParams &prm, const double *et, const double *esp_pass, int g) {
    const int so = prm.sp_off[g], sn = prm.sp_off[g + 1] - so;
    return GridEm{et, esp_pass + so + (sn > 0 ? 16 * prm.sp_gidx[g] : 0), prm.sp_k + so, sn};   // (k_emat, lazy layout)
}

struct Smem {
    double *et;      // [256]
    double *x;       // [4] broadcast slots
    double *mt;      // [256] matched_gammas
    double *state;   // [2][Kpad] when the state lives in LDS
    __device__ __forceinline__ explicit Smem(char *s) {
        et = reinterpret_cast<double *>(s);
        x = et + kMaxRow;
        mt = x + 4;
        state = mt + kMaxRow;
    }
};
constexpr size_t kSmemFixed = (kMaxRow + 4 + kMaxRow) * 8;

// ---------------------------------------------------------------------------------------------
// forward (reference-single.cpp:2292-2354 grid 0, :935-1129 the rest)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRT) void k_fwd_ro(PassParams prm, int NT, int state_in_lds, int Kpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Smem L(smem);
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int K = prm.K, G = prm.G;
    const double *emat = static_cast<const double *>(prm.emat) + (size_t)p * G * kMaxRow;
    const double *esp = static_cast<const double *>(prm.esp) + (size_t)p * prm.esp_stride;
    const double *emin = prm.emin + (size_t)p * G;
    double *aout = static_cast<double *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride;
    const int32_t *slot = prm.alpha_slot + (size_t)p * G;
    double *state = state_in_lds ? L.state : prm.spill + (size_t)p * prm.spill_pass_stride;
    (void)Kpad;
    const double double_K = (double)K, one_over_K = 1 / (double)K;

    double prev_sum = 1, running_min = 1;
    for (int g = 0; g < G; g++) {
        const uint8_t *code = prm.hm + (size_t)g * prm.Kp;
        __syncthreads();   // the previous grid's readers of the table are done
        L.et[t] = emat[(size_t)g * kMaxRow + t];
        __syncthreads();
        const GridEm E = grid_em(prm, L.et, esp, g);
        const double em = emin[g];
        const bool has_variant = g == 0 || em >= 0;   // (grid 1 is forced: k_emat, :964-966)
        double sig = 1.0, addend = 0.0;
        if (g > 0) {
            sig = prm.sigma[g - 1];
            const double jump_prob = prm.tm1[g - 1] / double_K;
            const double jump_prob_plus = prm.always_normalize ? jump_prob : jump_prob * prev_sum;
            addend = jump_prob_plus / sig;
        }
        if (g == 0) {
            for (int k = t; k < K; k += kRT) state[k] = E.at(k, code[k]) * one_over_K;
        } else if (has_variant) {
            for (int k = t; k < K; k += kRT) state[k] = (addend + state[k]) * E.at(k, code[k]);
        } else {
            for (int k = t; k < K; k += kRT) state[k] = addend + state[k];
        }
        double run_total;
        if (has_variant) {
            __syncthreads();
            if (wave == 0) {
                double s = 0.0;
                if (g > 0) {
                    s = serial_gather_sum(state, E.sp_k, E.sn, s, lane);
                    s = serial_sum<true>(state, code, K, s, lane);
                } else {
                    s = prm.grid0_left_to_right ? serial_sum<false>(state, code, K, s, lane) : serial_sum_arma(state, K, lane);
                }
                if (lane == 0) L.x[0] = s;
            }
            __syncthreads();
            run_total = L.x[0];
        } else {
            run_total = prev_sum / sig;   // (:1078-1088)
        }
        double cg = 1.0;
        if (g > 0) {
            if (has_variant) running_min = running_min * em;
            cg = cg / sig;
        }
        if (g == 0 || prm.always_normalize || running_min < prm.norm_threshold || g == G - 1) {
            const double xs = 1 / run_total;
            for (int k = t; k < K; k += kRT) state[k] *= xs;
            cg = (g == 0) ? xs : cg / run_total;
            run_total = 1;
            running_min = 1;
        }
        prev_sum = run_total;
        if (t == 0) prm.c[(size_t)p * G + g] = cg;
        const int sl = slot[g];
        if (sl >= 0) {
            double *dst = aout + (size_t)sl * prm.alpha_col_elems;
            for (int k = t; k < K; k += kRT) dst[perm_index(k, NT)] = state[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward (reference-single.cpp:1854-2177)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRT) void k_bwd_ro(PassParams prm, int NT, int state_in_lds, int Kpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Smem L(smem);
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int K = prm.K, G = prm.G, T = prm.T;
    const int flags = prm.flags[p];
    const bool want_dosage = (flags & 1) != 0, want_gamma = (flags & 4) != 0, want_beta = (flags & 8) != 0;
    const double *emat = static_cast<const double *>(prm.emat) + (size_t)p * G * kMaxRow;
    const double *esp = static_cast<const double *>(prm.esp) + (size_t)p * prm.esp_stride;
    const double *emin = prm.emin + (size_t)p * G;
    const double *ain = static_cast<const double *>(prm.alpha) + (size_t)p * prm.alpha_pass_stride;
    const int32_t *slot = prm.alpha_slot + (size_t)p * G;
    const double *cvec = prm.c + (size_t)p * G;
    double *state = state_in_lds ? L.state : prm.spill + (size_t)p * prm.spill_pass_stride;
    double *gam = state + Kpad;
    const double double_K = (double)K;
    const double eps = prm.ref_error, ome = 1 - eps;

    for (int k = t; k < K; k += kRT) state[k] = 1.0;   // 1 / not_jump_prob, not_jump_prob = 1 (:1855-1857)
    double not_jump_prob = 1.0, B_prev = 1.0;
    double B_prev_star = double_K * cvec[G - 1] * not_jump_prob;
    for (int g = G - 1; g >= 0; --g) {
        const double c_g = cvec[g];
        if (g < G - 1) {
            const double jump_prob = prm.tm1[g] / double_K;
            not_jump_prob = prm.sigma[g];
            const bool has_variant = (g + 1 == 1 ? prm.emin_b1[p] : emin[g + 1]) >= 0;   // (grid 1 is not forced here: :1866-1877)
            double val;
            if (has_variant) {
                const uint8_t *code1 = prm.hm + (size_t)(g + 1) * prm.Kp;
                __syncthreads();
                L.et[t] = emat[(size_t)(g + 1) * kMaxRow + t];
                __syncthreads();
                const GridEm E = grid_em(prm, L.et, esp, g + 1);
                for (int k = t; k < K; k += kRT) state[k] = state[k] * E.at(k, code1[k]);
                __syncthreads();
                if (wave == 0) {
                    double s = 0.0;
                    s = serial_gather_sum(state, E.sp_k, E.sn, s, lane);
                    s = serial_sum<true>(state, code1, K, s, lane);
                    if (lane == 0) L.x[0] = s;
                }
                __syncthreads();
                const double sum_e_times_b = L.x[0];
                val = jump_prob / not_jump_prob * sum_e_times_b;
                B_prev = sum_e_times_b;
            } else {
                val = jump_prob / not_jump_prob * B_prev_star;
                B_prev = B_prev_star;
            }
            for (int k = t; k < K; k += kRT) state[k] = state[k] + val;
            B_prev_star = c_g * B_prev;
        }
        const uint8_t *code = prm.hm + (size_t)g * prm.Kp;
        const int tcol = prm.thin_col[g];
        const int sl = slot[g];
        if (tcol >= 0 && prm.K_top > 0 && prm.beta_thin) {
            // the (unscaled) beta column goes to k_topk, which forms gamma = alpha * beta and picks (:2020-2031): selection by
            // comparisons only, no arithmetic whose order could matter
            double *dst = static_cast<double *>(prm.beta_thin) + ((size_t)p * prm.n_thin + tcol) * prm.Kq;
            for (int k = t; k < K; k += kRT) dst[perm_index(k, NT)] = state[k];
        }
        if ((want_dosage || want_gamma) && sl >= 0) {
            const double *acol = ain + (size_t)sl * prm.alpha_col_elems;
            for (int k = t; k < K; k += kRT) gam[k] = acol[perm_index(k, NT)] * state[k];
            if (want_gamma) {
                double *dst = static_cast<double *>(prm.gamma_out) + ((size_t)p * G + g) * prm.Kq;
                for (int k = t; k < K; k += kRT) dst[perm_index(k, NT)] = gam[k] * not_jump_prob;
            }
        }
        if (want_dosage && sl >= 0) {
            __syncthreads();
            // matched_gammas(dh) = sum over k in order of gamma(k) [hapMatcher(k, g) == dh], then * not_jump_prob (:2083-2095):
            // thread dh walks the column in k order and adds its own haplotypes' gamma (x + 0.0 == x for the others)
            {
                double m = 0.0;
                const uint32_t mine = (uint32_t)t;
                if (t >= 1 && t < prm.nrow) {
                    for (int k = 0; k < K; k++) m += (code[k] == mine) ? gam[k] : 0.0;
                }
                L.mt[t] = m * not_jump_prob;
            }
            __syncthreads();
            const int s = 32 * g, nLocal = min(32, T - s);
            if (t < nLocal) {
                double d = 0.0;
                const int so = prm.sp_off[g], sn = prm.sp_off[g + 1] - so;
                for (int i = 0; i < sn; i++) {   // (:2096-2128)
                    const double gk = gam[prm.sp_k[so + i]] * not_jump_prob;
                    const uint32_t w = prm.sp_word[so + i];
                    d += ((w >> t) & 1u) ? gk * ome : gk * eps;
                }
                const int32_t *Bg = prm.B + (size_t)g * prm.nMaxDH;
                const double *IEs = prm.IE ? prm.IE + (size_t)(s + t) * prm.nMaxDH : nullptr;
                for (int dh = 0; dh < prm.nMaxDH; dh++) {   // (:2129-2139)
                    const double ie = IEs ? IEs[dh] : ((((uint32_t)Bg[dh] >> t) & 1u) ? ome : eps);
                    d += ie * L.mt[dh + 1];
                }
                prm.dosage[(size_t)p * T + s + t] = d;
            }
        }
        // beta *= c_g * sigma_g (:2165-2166)
        const double x = c_g * not_jump_prob;
        for (int k = t; k < K; k += kRT) state[k] *= x;
        if (want_beta) {
            double *dst = static_cast<double *>(prm.beta_out) + ((size_t)p * G + g) * prm.Kq;
            for (int k = t; k < K; k += kRT) dst[perm_index(k, NT)] = state[k];
        }
    }
}

}  // namespace

namespace qa {

size_t fb_ref_state_doubles(int Kq) { return 2 * (size_t)Kq; }

void launch_fb_ref(const void *pass_params, int NT, hipStream_t st, hipEvent_t e_mid) {
    const PassParams &prm = *static_cast<const PassParams *>(pass_params);
    const int Kpad = prm.Kq;
    size_t lds = kSmemFixed;
    int in_lds = 0;
    if (kSmemFixed + 2 * (size_t)Kpad * 8 <= 144 * 1024) {
        in_lds = 1;
        lds += 2 * (size_t)Kpad * 8;
    }
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd_ro), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bwd_ro), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_fwd_ro, dim3(prm.P), dim3(kRT), lds, st, prm, NT, in_lds, Kpad);
    QA_HIP(hipGetLastError());
    if (e_mid) QA_HIP(hipEventRecord(e_mid, st));
    hipLaunchKernelGGL(k_bwd_ro, dim3(prm.P), dim3(kRT), lds, st, prm, NT, in_lds, Kpad);
    QA_HIP(hipGetLastError());
}

}  // namespace qa
