// panel_build.hip -- per-grid dictionary compression of the packed reference panel, on the device.
//
// the producer of every input layout of the hot path.  The reference calls
// STITCH::make_rhb_t_equality (STITCH 1.8.4, not vendored; call sites QUILT/R/quilt-prepare-reference.R:416-428,
// QUILT/R/quilt.R:551-563, QUILT/R/test-drivers.R:398) on the packed panel rhb_t (K x nGrids int32, bit b of word g =
// allele of SNP 32 g + b): per grid the distinct 32-bit words are ranked by descending frequency (ties: ascending
// signed value); the first nMaxDH get the 1-based codes of hapMatcherR and their words fill distinctHapsB; every other
// haplotype gets code 0 and is listed, with its word, in the "special" tables.  Semantics pinned by the reference's
// rebuild-rhb_t round trip (QUILT/tests/testthat/test-unit-reference-single.R:210-309) and, here, bit for bit against
// the host restatement quilt_amd/panel.py:make_rhb_t_equality.
//
// qa_panel_create_from_rhb builds the device-resident panel (panel.hpp layout) straight from rhb_t:
//   k_rank_words   one workgroup per grid: LDS hash table (word -> count) filled with integer LDS atomics, compacted,
//                  sorted by (count desc, signed word asc) with a bitonic network, ranks written back into the table,
//                  then codes for all K haplotypes (coalesced byte stores into hm[g][.]) and the grid's special count
//   k_list_specials  second pass, after the host prefix sum over grids: specials of a grid in ascending k (block scan)
// A grid with more distinct words than the table holds is ranked on the host (std::sort), same rule.
#include "panel.hpp"

#include <algorithm>
#include <memory>
#include <numeric>

namespace qa {
int32_t reference_matrix_search(int val, const int32_t *mat, int nrow, int s1, int e1);
}

namespace {

constexpr int kHT = 8192;            // hash slots per grid (LDS: 8 B each)
constexpr int kMaxDistinct = 4096;   // device path up to this many distinct words per grid (load factor 0.5)
constexpr int kBT = 256;

__device__ __forceinline__ uint32_t hash_word(uint32_t w) {
    w ^= w >> 16; w *= 0x7feb352dU; w ^= w >> 15; w *= 0x846ca68bU; w ^= w >> 16;
    return w & (kHT - 1);
}

// rhb: [G][K] (the R matrix K x nGrids, column-major).  Outputs: hm [G][Kp], B [G][nMaxDH], n_special [G], overflow [G].
__global__ __launch_bounds__(kBT) void k_rank_words(const int32_t *rhb, int K, int Kp, int G, int nMaxDH, uint8_t *hm, int32_t *B,
                                                   int32_t *n_special, int32_t *overflow, unsigned long long *sort_scratch) {
    __shared__ uint32_t keys[kHT];    // 0 = empty (the word 0 is counted apart)
    __shared__ uint32_t vals[kHT];    // count, later rank
    __shared__ int s_n, s_zero, s_over, s_sp;
    const int t = threadIdx.x;
    unsigned long long *srt = sort_scratch + (size_t)blockIdx.x * 2 * kMaxDistinct;   // np2 <= 2 * kMaxDistinct
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        for (int i = t; i < kHT; i += kBT) { keys[i] = 0; vals[i] = 0; }
        if (t == 0) { s_n = 0; s_zero = 0; s_over = 0; s_sp = 0; }
        __syncthreads();
        const uint32_t *col = reinterpret_cast<const uint32_t *>(rhb) + (size_t)g * K;
        // ---- count
        for (int k = t; k < K; k += kBT) {
            const uint32_t w = col[k];
            if (w == 0) { atomicAdd(&s_zero, 1); continue; }
            uint32_t slot = hash_word(w);
            for (int probe = 0; probe < kHT; probe++) {
                const uint32_t old = atomicCAS(&keys[slot], 0u, w);
                if (old == 0) {
                    if (atomicAdd(&s_n, 1) >= kMaxDistinct) s_over = 1;
                    atomicAdd(&vals[slot], 1u);
                    break;
                }
                if (old == w) { atomicAdd(&vals[slot], 1u); break; }
                slot = (slot + 1) & (kHT - 1);
            }
        }
        __syncthreads();
        if (s_over) {   // too many distinct words for the table: the host ranks this grid
            if (t == 0) overflow[g] = 1;
            __syncthreads();
            continue;
        }
        // ---- compact to (count desc, signed word asc) sort keys; the word 0 is entry number n if present
        const int n_tab = s_n;
        __syncthreads();
        if (t == 0) s_n = 0;
        __syncthreads();
        for (int i = t; i < kHT; i += kBT) {
            if (keys[i] != 0) {
                const int at = atomicAdd(&s_n, 1);
                srt[at] = ((unsigned long long)(~vals[i]) << 32) | (keys[i] ^ 0x80000000u);
            }
        }
        __syncthreads();
        int n = n_tab;
        if (s_zero > 0) {
            if (t == 0) srt[n] = ((unsigned long long)(~(uint32_t)s_zero) << 32) | (0u ^ 0x80000000u);
            n++;
        }
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int i = n + t; i < np2; i += kBT) srt[i] = ~0ULL;
        __syncthreads();
        // ---- bitonic sort (ascending)
        for (int k2 = 2; k2 <= np2; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = t; i < np2; i += kBT) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = srt[i], b = srt[l];
                        const bool up = (i & k2) == 0;
                        if ((a > b) == up) { srt[i] = b; srt[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        // ---- ranks back into the table; distinctHapsB
        __shared__ uint32_t zero_rank;
        if (t == 0) zero_rank = 0;
        __syncthreads();
        for (int i = t; i < n; i += kBT) {
            const uint32_t w = (uint32_t)(srt[i] & 0xffffffffu) ^ 0x80000000u;
            const uint32_t rank = (uint32_t)i + 1;
            if (i < nMaxDH) B[(size_t)g * nMaxDH + i] = (int32_t)w;
            if (w == 0) { zero_rank = rank; continue; }
            uint32_t slot = hash_word(w);
            while (keys[slot] != w) slot = (slot + 1) & (kHT - 1);
            vals[slot] = rank;
        }
        for (int i = n + t; i < nMaxDH; i += kBT) B[(size_t)g * nMaxDH + i] = 0;
        __syncthreads();
        // ---- codes
        int mine = 0;
        for (int k = t; k < K; k += kBT) {
            const uint32_t w = col[k];
            uint32_t rank;
            if (w == 0) rank = zero_rank;
            else {
                uint32_t slot = hash_word(w);
                while (keys[slot] != w) slot = (slot + 1) & (kHT - 1);
                rank = vals[slot];
            }
            const uint32_t code = rank <= (uint32_t)nMaxDH ? rank : 0u;
            hm[(size_t)g * Kp + k] = (uint8_t)code;
            mine += code == 0;
        }
        atomicAdd(&s_sp, mine);
        __syncthreads();
        if (t == 0) { n_special[g] = s_sp; overflow[g] = 0; }
        __syncthreads();
    }
}

// specials of each grid in ascending haplotype order: thread t owns a contiguous range of haplotypes
__global__ __launch_bounds__(kBT) void k_list_specials(const int32_t *rhb, const uint8_t *hm, int K, int Kp, int G, const int32_t *sp_off,
                                                      int32_t *sp_k, uint32_t *sp_word) {
    __shared__ int s_cnt[kBT];
    const int t = threadIdx.x;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        if (sp_off[g + 1] == sp_off[g]) continue;   // uniform
        const int per = (K + kBT - 1) / kBT, k0 = t * per, k1 = min(K, k0 + per);
        int mine = 0;
        for (int k = k0; k < k1; k++) mine += hm[(size_t)g * Kp + k] == 0;
        s_cnt[t] = mine;
        __syncthreads();
        int before = 0;
        for (int i = 0; i < t; i++) before += s_cnt[i];
        int at = sp_off[g] + before;
        for (int k = k0; k < k1; k++) {
            if (hm[(size_t)g * Kp + k] == 0) {
                sp_k[at] = k;
                sp_word[at] = (uint32_t)rhb[(size_t)g * K + k];
                at++;
            }
        }
        __syncthreads();
    }
}

// the same rule on the host, for a grid the device table could not hold
void rank_grid_on_host(const int32_t *col, int K, int nMaxDH, uint8_t *codes, int32_t *Bcol) {
    std::vector<uint32_t> w(col, col + K);
    std::vector<uint32_t> srt(w);
    std::sort(srt.begin(), srt.end());
    std::vector<std::pair<uint32_t, uint32_t>> dc;   // (word, count)
    for (size_t i = 0; i < srt.size();) {
        size_t j = i;
        while (j < srt.size() && srt[j] == srt[i]) j++;
        dc.emplace_back(srt[i], (uint32_t)(j - i));
        i = j;
    }
    std::sort(dc.begin(), dc.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) {
        return a.second > b.second || (a.second == b.second && (int32_t)a.first < (int32_t)b.first);
    });
    std::vector<std::pair<uint32_t, uint32_t>> rank_of;   // (word, rank) sorted by word for lookup
    for (size_t i = 0; i < dc.size(); i++) rank_of.emplace_back(dc[i].first, (uint32_t)i + 1);
    std::sort(rank_of.begin(), rank_of.end());
    for (int i = 0; i < nMaxDH; i++) Bcol[i] = i < (int)dc.size() ? (int32_t)dc[i].first : 0;
    for (int k = 0; k < K; k++) {
        auto it = std::lower_bound(rank_of.begin(), rank_of.end(), std::make_pair(w[k], 0u));
        const uint32_t r = it->second;
        codes[k] = (uint8_t)(r <= (uint32_t)nMaxDH ? r : 0);
    }
}

// rhi: [T][K] 0 / 1 alleles (the R matrix K x nSNPs, column-major), SNPs t0 .. t0 + 32 n_grid - 1 of it resident at `rhi`;
// rhb: [G][K].  One thread per (haplotype, grid): bit b of the word = allele at SNP 32 g + b; loads coalesce over k.
__global__ __launch_bounds__(256) void k_pack_rhi(const int32_t *rhi, int K, int T, int t0, int n_grid, int32_t *rhb, int g0) {
    const int k = blockIdx.x * 256 + threadIdx.x, gl = blockIdx.y;
    if (k >= K || gl >= n_grid) return;
    uint32_t w = 0;
#pragma unroll 8
    for (int b = 0; b < 32; b++) {
        const int t = t0 + 32 * gl + b;
        if (t < T && rhi[(size_t)(32 * gl + b) * K + k] != 0) w |= 1u << b;
    }
    rhb[(size_t)(g0 + gl) * K + k] = (int32_t)w;
}

}  // namespace

extern "C" {

int qa_make_rhb_t_from_rhi_t(const int32_t *rhi_t, int32_t K, int32_t nSNPs, int32_t *rhb_t) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!rhi_t || !rhb_t || K <= 0 || nSNPs <= 0) {
        qa::set_error("qa_make_rhb_t_from_rhi_t: bad argument");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        const int G = (nSNPs + 31) / 32;
        // slabs of whole grids, about 256 MB of alleles each: upload, pack, next
        const int grids_per_slab = std::max<int>(1, (int)std::min<int64_t>(G, ((int64_t)64 << 20) / ((int64_t)32 * K)));
        qa::DBuf<int32_t> d_rhi((size_t)grids_per_slab * 32 * K), d_rhb((size_t)G * K);
        hipStream_t st;
        QA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int g0 = 0; g0 < G; g0 += grids_per_slab) {
            const int ng = std::min(grids_per_slab, G - g0);
            const int t0 = 32 * g0, nt = std::min(32 * ng, nSNPs - t0);
            qa::staged_upload(d_rhi.p, rhi_t + (size_t)t0 * K, sizeof(int32_t) * (size_t)nt * K, st);
            hipLaunchKernelGGL(k_pack_rhi, dim3((K + 255) / 256, ng), dim3(256), 0, st, d_rhi.p, K, nSNPs, t0, ng, d_rhb.p, g0);
            QA_HIP(hipGetLastError());
            QA_HIP(hipStreamSynchronize(st));
        }
        qa::staged_download(rhb_t, d_rhb.p, sizeof(int32_t) * (size_t)G * K, st);
        QA_HIP(hipStreamSynchronize(st));
        QA_HIP(hipStreamDestroy(st));
        return QA_OK;
    });
}

int qa_panel_create_from_rhb(const int32_t *rhb_t, int32_t K, int32_t nGrids, int32_t nSNPs, int32_t nMaxDH,
                             const double *transMatRate_t, double ref_error, int32_t use_eMatDH_special_symbols,
                             qa_panel_t **out) {
    if (!out) return QA_ERR_INVALID;
    *out = nullptr;
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!rhb_t || K <= 0 || nGrids <= 0 || nSNPs <= 0 || !transMatRate_t || (nSNPs + 31) / 32 != nGrids) {
        qa::set_error("qa_panel_create_from_rhb: bad argument (nGrids must be ceil(nSNPs / 32))");
        return QA_ERR_INVALID;
    }
    if (nMaxDH <= 0 || nMaxDH > 255) {
        qa::set_error("qa_panel_create_from_rhb: nMaxDH must be 1..255 (hapMatcherR layout)");
        return QA_ERR_UNSUPPORTED;
    }
    return qa::guarded([&] {
        auto *p = new qa_panel();
        std::unique_ptr<qa_panel> guard(p);
        QA_HIP(hipGetDevice(&p->device));
        QA_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        hipStream_t st = p->stream;
        const int G = nGrids;
        p->K = K; p->G = G; p->T = nSNPs; p->nMaxDH = nMaxDH; p->nrow = nMaxDH + 1;
        p->Kp = (K + 8191) / 8192 * 8192;
        p->ref_error = ref_error;
        p->ie_derived = true;
        p->hm.alloc((size_t)G * p->Kp);
        p->hm.zero(st);
        p->B.alloc((size_t)nMaxDH * G);
        qa::DBuf<int32_t> d_rhb((size_t)K * G), d_nsp(G), d_over(G);
        d_rhb.upload(rhb_t, (size_t)K * G, st);
        const int blocks = std::min(G, 512);
        qa::DBuf<unsigned long long> d_srt((size_t)blocks * kMaxDistinct * 2);
        hipLaunchKernelGGL(k_rank_words, dim3(blocks), dim3(kBT), 0, st, d_rhb.p, K, p->Kp, G, nMaxDH, p->hm.p, p->B.p, d_nsp.p,
                           d_over.p, d_srt.p);
        QA_HIP(hipGetLastError());
        std::vector<int32_t> nsp(G), over(G);
        d_nsp.download(nsp.data(), G, st);
        d_over.download(over.data(), G, st);
        // grids the device table could not hold
        std::vector<uint8_t> codes(K);
        std::vector<int32_t> Bcol(nMaxDH);
        for (int g = 0; g < G; g++) {
            if (!over[g]) continue;
            rank_grid_on_host(rhb_t + (size_t)g * K, K, nMaxDH, codes.data(), Bcol.data());
            qa::staged_upload(p->hm.p + (size_t)g * p->Kp, codes.data(), K, st);
            qa::staged_upload(p->B.p + (size_t)g * nMaxDH, Bcol.data(), sizeof(int32_t) * nMaxDH, st);
            nsp[g] = (int32_t)std::count(codes.begin(), codes.end(), (uint8_t)0);
        }
        std::vector<int32_t> off(G + 1, 0);
        for (int g = 0; g < G; g++) off[g + 1] = off[g] + nsp[g];
        p->n_special = off[G];
        p->h_sp_off = off;
        p->sp_off.alloc(G + 1);
        p->sp_off.upload(off.data(), G + 1, st);
        p->sp_k.alloc(std::max<size_t>(off[G], 1));
        p->sp_word.alloc(std::max<size_t>(off[G], 1));
        if (off[G] > 0) {
            hipLaunchKernelGGL(k_list_specials, dim3(blocks), dim3(kBT), 0, st, d_rhb.p, p->hm.p, K, p->Kp, G, p->sp_off.p, p->sp_k.p,
                               p->sp_word.p);
            QA_HIP(hipGetLastError());
            if (use_eMatDH_special_symbols) {
                // without rhb_t at run time the reference decodes a special haplotype's word by its clamped binary search
                // over the special matrix (gibbs-small.cpp:69-105), quirks included: reproduce what it would find
                std::vector<int32_t> sk(off[G]);
                std::vector<uint32_t> sw(off[G]);
                p->sp_k.download(sk.data(), sk.size(), st);
                p->sp_word.download(sw.data(), sw.size(), st);
                const int nrow = off[G];
                std::vector<int32_t> mat((size_t)2 * nrow);   // special matrix: column 0 = k, column 1 = word
                for (int i = 0; i < nrow; i++) { mat[i] = sk[i]; mat[(size_t)nrow + i] = (int32_t)sw[i]; }
                for (int g = 0; g < G; g++)
                    for (int i = off[g]; i < off[g + 1]; i++)
                        sw[i] = (uint32_t)qa::reference_matrix_search(sk[i], mat.data(), nrow, off[g] + 1, off[g + 1]);
                p->sp_word.upload(sw.data(), sw.size(), st);
            }
        }
        p->h_sigma.resize(std::max(G - 1, 1));
        p->h_tm1.resize(std::max(G - 1, 1));
        for (int g = 0; g < G - 1; g++) {
            p->h_sigma[g] = transMatRate_t[2 * (size_t)g];
            p->h_tm1[g] = transMatRate_t[2 * (size_t)g + 1];
        }
        p->sigma.alloc(std::max(G - 1, 1));
        p->sigma.upload(p->h_sigma.data(), std::max(G - 1, 0), st);
        QA_HIP(hipStreamSynchronize(st));
        qa::finish_panel_tables(p);
        *out = guard.release();
        return QA_OK;
    });
}

int qa_panel_export_tables(qa_panel_t *panel, uint8_t *hapMatcherR, int32_t *distinctHapsB, int32_t *special_off,
                           int32_t *special_k, int32_t *special_word, int64_t special_cap) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!panel) return QA_ERR_INVALID;
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(panel->device));
        hipStream_t st = panel->stream;
        const int K = panel->K, G = panel->G;
        if (hapMatcherR) {   // K x nGrids, column-major (the R raw matrix)
            std::vector<uint8_t> row(panel->Kp);
            for (int g = 0; g < G; g++) {
                qa::staged_download(row.data(), panel->hm.p + (size_t)g * panel->Kp, K, st);
                memcpy(hapMatcherR + (size_t)g * K, row.data(), K);
            }
        }
        if (distinctHapsB) panel->B.download(distinctHapsB, (size_t)panel->nMaxDH * G, st);
        if (special_off) memcpy(special_off, panel->h_sp_off.data(), sizeof(int32_t) * (G + 1));
        if (special_k || special_word) {
            if (special_cap < panel->n_special) {
                qa::set_error("qa_panel_export_tables: capacity %lld < %d specials", (long long)special_cap, panel->n_special);
                return (int)QA_ERR_CAPACITY;
            }
            if (special_k) panel->sp_k.download(special_k, panel->n_special, st);
            if (special_word) panel->sp_word.download(reinterpret_cast<uint32_t *>(special_word), panel->n_special, st);
        }
        QA_HIP(hipStreamSynchronize(st));
        return (int)QA_OK;
    });
}

}  // extern "C"
