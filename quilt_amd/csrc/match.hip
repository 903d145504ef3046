// match.hip -- the haplotype search of the msPBWT mode (use_mspbwt = TRUE).
//
// In that mode the reference replaces the full-panel pass by a positional-BWT query: the rounded haploid dosages of a
// Gibbs chain are looked up in `mspbwt_nindices` interleaved indices over the panel's per-grid symbols and the long
// matches found select the next small panel (QUILT/R/mspbwt.R:225-474, `select_new_haps_mspbwt_v3`, calling
// mspbwt::Rcpp_find_good_matches_without_a).  The mspbwt package is not in the reference tree, and on this device an index is
// not needed: the panel's symbol table (hapMatcherR, 1 byte per haplotype and grid) streams through at HBM rate, so the
// search is done by comparing the query with EVERY haplotype.  What is reported has the interface of the reference's
// query -- (index0, start0, len1) triples per index -- and this definition (a stand-in, UNPINNED against mspbwt):
//   * index i covers the grids i, i + n, i + 2n, ... (positions j = 0, 1, ...), as the reference's which_grids
//   * the query's symbol at a grid is the 1-based row of distinctHapsB holding its 32-SNP word (mspbwt::map_Z_to_all_symbols);
//     a word that is not in the grid's dictionary matches nothing, and neither do the panel's special haplotypes (code 0)
//   * per haplotype its longest run of consecutive matching positions (the earliest such run), and of those the
//     `max_matches` longest with at least `min_len` positions; ties at the cut go to the lower haplotype index
//   * reported in haplotype order; the caller orders by length (stable), as mspbwt.R:371 does
// Kernels: k_query_codes (query words -> symbols), k_best_run (one thread per haplotype walks the positions; byte loads
// coalesced over haplotypes: K * G / n bytes per query and index, the whole symbol table once per query), k_pick_runs
// (histogram of the run lengths -> threshold -> ordered emission with a block scan).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "panel.hpp"

namespace {

constexpr int kMaxRunLen = 4096;   // positions per index the length histogram holds

__global__ __launch_bounds__(64) void k_query_codes(const int32_t *Z, const int32_t *B, int G, int nMaxDH, uint8_t *qc) {
    const int g = blockIdx.x, q = blockIdx.y, lane = threadIdx.x;
    const int32_t w = Z[(size_t)q * G + g];
    int code = 0;
    for (int d = lane; d < nMaxDH; d += 64)
        if (!code && B[(size_t)g * nMaxDH + d] == w) code = d + 1;   // the lane's first matching row
    // unused rows of distinctHapsB hold 0, which is also a legal word: the FIRST matching row is the symbol
    int best = code ? code : 0x7fffffff;
    for (int off = 32; off; off >>= 1) best = min(best, __shfl_xor(best, off));
    if (lane == 0) qc[(size_t)q * G + g] = best == 0x7fffffff ? 0 : (uint8_t)best;
}

// four haplotypes per thread (one 4-byte load per grid; Kp is a multiple of 8192, so the loads stay inside the row)
__global__ __launch_bounds__(256) void k_best_run(const uint8_t *hm, int K, int Kp, int G, int n_idx, const uint8_t *qc,
                                                  uint16_t *best_len, uint16_t *best_start) {
    const int k0 = (blockIdx.x * 256 + threadIdx.x) * 4, i = blockIdx.y, q = blockIdx.z;
    if (k0 >= K) return;
    const uint8_t *qq = qc + (size_t)q * G;
    int run[4] = {0, 0, 0, 0}, start[4] = {0, 0, 0, 0}, bl[4] = {0, 0, 0, 0}, bs[4] = {0, 0, 0, 0}, j = 0;
    for (int g = i; g < G; g += n_idx, j++) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(hm + (size_t)g * Kp + k0);
        const uint32_t s = qq[g];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t c = (w >> (8 * b)) & 0xffu;
            const bool m = c != 0 && c == s;
            if (m) {
                if (run[b] == 0) start[b] = j;
                run[b]++;
                if (run[b] > bl[b]) { bl[b] = run[b]; bs[b] = start[b]; }
            } else {
                run[b] = 0;
            }
        }
    }
    const size_t o = ((size_t)q * n_idx + i) * K + k0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (k0 + b < K) {
            best_len[o + b] = (uint16_t)bl[b];
            best_start[o + b] = (uint16_t)bs[b];
        }
    }
}

__global__ __launch_bounds__(256) void k_pick_runs(const uint16_t *best_len, const uint16_t *best_start, int K, int n_idx,
                                                   int min_len, int max_matches, int32_t *match, int32_t *n_match) {
    __shared__ int hist[kMaxRunLen + 1];
    __shared__ int s_T, s_above, s_scan[256], s_base_a, s_base_b;
    const int i = blockIdx.x, q = blockIdx.y, t = threadIdx.x;
    const size_t o = ((size_t)q * n_idx + i) * K;
    for (int x = t; x <= kMaxRunLen; x += 256) hist[x] = 0;
    __syncthreads();
    for (int k = t; k < K; k += 256) atomicAdd(&hist[min((int)best_len[o + k], kMaxRunLen)], 1);
    __syncthreads();
    if (t == 0) {
        // T = the largest length with count(len >= T) >= max_matches, not below min_len; above = count(len > T)
        int cnt = 0, T = kMaxRunLen;
        for (; T > min_len; T--) {
            if (cnt + hist[T] >= max_matches) break;
            cnt += hist[T];
        }
        s_T = T;
        s_above = cnt;
        s_base_a = 0;
        s_base_b = 0;
    }
    __syncthreads();
    const int T = s_T, quota_b = max_matches - s_above;   // ties at T accepted, in haplotype order
    int32_t *out = match + ((size_t)q * n_idx + i) * max_matches * 3;
    for (int k0 = 0; k0 < K; k0 += 256) {
        const int k = k0 + t;
        const int len = k < K ? best_len[o + k] : 0;
        const bool a = len > T && len >= min_len, b = len == T && len >= min_len;
        // exclusive scans of the two flags over the block, in haplotype order
        int va = a ? 1 : 0, vb = b ? 1 : 0;
        s_scan[t] = va | (vb << 16);
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = t >= off ? s_scan[t - off] : 0;
            __syncthreads();
            s_scan[t] += add;
            __syncthreads();
        }
        const int inc = s_scan[t], tot = s_scan[255];
        const int ra = s_base_a + (inc & 0xffff) - va, rb = s_base_b + (inc >> 16) - vb;
        // entries are written in haplotype order: position = number of accepted entries before this one
        const bool take = a || (b && rb < quota_b);
        const int before = ra + min(rb, quota_b);
        if (take && before < max_matches) {
            out[(size_t)before * 3 + 0] = k;
            out[(size_t)before * 3 + 1] = best_start[o + k];
            out[(size_t)before * 3 + 2] = len;
        }
        __syncthreads();
        if (t == 0) { s_base_a += tot & 0xffff; s_base_b += tot >> 16; }
        __syncthreads();
    }
    if (t == 0) n_match[(size_t)q * n_idx + i] = min(s_base_a + min(s_base_b, quota_b), max_matches);
}

}  // namespace

extern "C" {

int qa_find_good_matches(qa_panel_t *panel, int32_t n_query, const int32_t *Zs, int32_t nindices, int32_t min_len,
                         int32_t max_matches, int32_t *match, int32_t *n_match) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!panel || n_query <= 0 || !Zs || nindices < 1 || nindices > panel->G || min_len < 1 || max_matches < 1 || !match ||
        !n_match || (panel->G + nindices - 1) / nindices > kMaxRunLen) {
        qa::set_error("qa_find_good_matches: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(panel->device));
        hipStream_t st = panel->stream;
        const int G = panel->G, K = panel->K;
        // queries in slabs that fit the handle's arena: per query G words + G symbols + 4 B per haplotype and index + results
        const size_t per_q = (size_t)G * 5 + (size_t)nindices * K * 4 + (size_t)nindices * (max_matches * 12 + 4) + 1024;
        qa::GateHold hold;
        hold.acquire(panel->gate(), &panel->arena, 0, 0, /*express=*/true);
        qa::Arena &arena = hold.arena();
        const size_t budget = arena.budget_shared(panel->sharers());
        const int slab = (int)std::max<size_t>(1, std::min<size_t>(n_query, budget / per_q));
        hold.require((size_t)slab * per_q + 4096);
        for (int q0 = 0; q0 < n_query; q0 += slab) {
            const int nq = std::min(slab, n_query - q0);
            arena.reset();
            qa::ABuf<int32_t> d_Z, d_match, d_n;
            qa::ABuf<uint8_t> d_qc;
            qa::ABuf<uint16_t> d_len, d_start;
            for (auto *b : {&d_Z, &d_match, &d_n}) b->arena = &arena;
            d_qc.arena = d_len.arena = d_start.arena = &arena;
            d_Z.ensure((size_t)nq * G); d_qc.ensure((size_t)nq * G);
            d_len.ensure((size_t)nq * nindices * K); d_start.ensure((size_t)nq * nindices * K);
            d_match.ensure((size_t)nq * nindices * max_matches * 3); d_n.ensure((size_t)nq * nindices);
            d_Z.upload(Zs + (size_t)q0 * G, (size_t)nq * G, st);
            hipEvent_t e0, e1;
            QA_HIP(hipEventCreate(&e0)); QA_HIP(hipEventCreate(&e1));
            QA_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_query_codes, dim3(G, nq), dim3(64), 0, st, d_Z.p, panel->B.p, G, panel->nMaxDH, d_qc.p);
            hipLaunchKernelGGL(k_best_run, dim3((K + 1023) / 1024, nindices, nq), dim3(256), 0, st, panel->hm.p, K, panel->Kp, G,
                               nindices, d_qc.p, d_len.p, d_start.p);
            hipLaunchKernelGGL(k_pick_runs, dim3(nindices, nq), dim3(256), 0, st, d_len.p, d_start.p, K, nindices, min_len,
                               max_matches, d_match.p, d_n.p);
            QA_HIP(hipGetLastError());
            QA_HIP(hipEventRecord(e1, st));
            d_match.download(match + (size_t)q0 * nindices * max_matches * 3, (size_t)nq * nindices * max_matches * 3, st);
            d_n.download(n_match + (size_t)q0 * nindices, (size_t)nq * nindices, st);
            QA_HIP(hipStreamSynchronize(st));
            float ms = 0;
            QA_HIP(hipEventElapsedTime(&ms, e0, e1));
            qa::profile_add(qa::PK_MATCH, ms, (double)nq * ((double)K * G + (double)nindices * K * 8), qa::profile_clock_ms(e0), nq);
            QA_HIP(hipEventDestroy(e0)); QA_HIP(hipEventDestroy(e1));
        }
        return (int)QA_OK;
    });
}

}  // extern "C"
