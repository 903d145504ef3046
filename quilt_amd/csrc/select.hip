// select.hip -- the small-panel re-selection on the device: everything_per_hap_rejig_haps +
// everything_select_good_haps (QUILT/R/functions.R:2161-2170, 2262-2310) behind the full-panel call, so that a round's
// best-haplotype lists never leave the device: only the next which_haps_to_use does.
//
// One wave per chain.  The reference draws with R's sample(); here every random choice is "the n smallest keys of a counter
// stream" (key j of stream `seed` = the library's splitmix64 stream, quilt_amd/rng.py), which a host restatement reproduces
// exactly (quilt_amd/driver.py everything_select_good_haps_dense):
//   1. previously_selected_haplotypes = Ksubset - Knew of the chain's current haplotypes: smallest keys [0, Ksubset), in key order
//   2. rank by rank (best first), the candidates top[label][thinned grid][rank] in label-major, grid order -- R's
//      unlist(sapply(new_haps, ...)) -- that are new (not kept before, not previously selected, first occurrence) are appended
//      until Knew are found; the rank that overshoots is subsampled: smallest keys [2^20, 2^20 + n_new), in key order
//   3. when the ranks up to K_top_matches do not fill Knew the reference goes on to "all entries of all lists" and then to a
//      random draw from the whole panel (functions.R:2278-2300): status 1, the caller's host path does those (rare: it
//      needs the complete, untruncated lists)
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "fullpass_dev.hpp"

namespace {

__device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr uint64_t kSubsampleOffset = 1ull << 20;

// number of keys ordered before key j (ties by index): its position in the stable ascending order
__device__ __forceinline__ int key_rank(const uint64_t *keys, int n, int j) {
    const uint64_t kj = keys[j];
    int r = 0;
    for (int m = 0; m < n; m++) {
        const uint64_t km = keys[m];
        r += (km < kj || (km == kj && m < j)) ? 1 : 0;
    }
    return r;
}

__global__ __launch_bounds__(64) void k_select(qa::SelectParams p) {
    extern __shared__ uint32_t lds[];
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (p.want && !p.want[chain]) {
        if (lane == 0) p.status[chain] = -1;   // no selection asked for
        return;
    }
    const int n_words = (p.K + 31) / 32, n_cand = p.n_label * p.n_thin;
    uint32_t *bits = lds;
    int32_t *newv = reinterpret_cast<int32_t *>(lds + n_words);
    uint64_t *keys = reinterpret_cast<uint64_t *>(lds + ((n_words + n_cand + 1) & ~1));
    const uint64_t seed = p.seed[chain];
    const int32_t *which = p.which + (size_t)chain * p.Ksubset;
    int32_t *out = p.out + (size_t)chain * p.Ksubset;
    const int32_t *top = p.top + (size_t)chain * n_cand * p.top_width;

    for (int w = lane; w < n_words; w += 64) bits[w] = 0;
    for (int j = lane; j < p.Ksubset; j += 64) keys[j] = stream_key(seed, (uint64_t)j);
    __syncthreads();
    const int n_prev = p.Ksubset - p.Knew;
    for (int j = lane; j < p.Ksubset; j += 64) {
        const int r = key_rank(keys, p.Ksubset, j);
        if (r < n_prev) {
            const int v = which[j];   // 1-based
            out[r] = v;
            atomicOr(&bits[(v - 1) >> 5], 1u << ((v - 1) & 31));
        }
    }
    __syncthreads();

    int kept = 0, status = 1;
    const int n_rank = min(p.K_top_matches, p.top_width);
    for (int i = 0; i < n_rank && status == 1; i++) {
        int n_new = 0;
        for (int base = 0; base < n_cand; base += 64) {
            const int c = base + lane;
            const int v = c < n_cand ? top[(size_t)c * p.top_width + i] : -1;   // 0-based, -1 = no entry
            bool valid = v >= 0 && !((bits[v >> 5] >> (v & 31)) & 1u);
            const uint64_t first = __ballot(valid);
            for (int j = 0; j < 63; j++) {
                const int vj = __shfl(v, j);
                if (((first >> j) & 1ull) && lane > j && vj == v) valid = false;
            }
            const uint64_t mask = __ballot(valid);
            if (valid) {
                newv[n_new + __popcll(mask & ((1ull << lane) - 1ull))] = v;
                atomicOr(&bits[v >> 5], 1u << (v & 31));
            }
            n_new += __popcll(mask);
            __syncthreads();
        }
        const int room = p.Knew - kept;
        if (n_new < room) {
            for (int q = lane; q < n_new; q += 64) out[n_prev + kept + q] = newv[q] + 1;
            kept += n_new;
        } else {
            for (int q = lane; q < n_new; q += 64) keys[q] = stream_key(seed, kSubsampleOffset + (uint64_t)q);
            __syncthreads();
            for (int q = lane; q < n_new; q += 64) {
                const int r = key_rank(keys, n_new, q);
                if (r < room) out[n_prev + kept + r] = newv[q] + 1;
            }
            kept = p.Knew;
            status = 0;
        }
        __syncthreads();
    }
    if (lane == 0) p.status[chain] = status;
}

// rows of one launch set's list table (run_passes' S.top_idx / S.top_cnt: [pass][thinned grid][top_cap]) to their places
// in the call-wide table [chain * n_label + label][thinned grid][top_cap]
__global__ __launch_bounds__(64) void k_scatter_lists(const int32_t *src_idx, const int32_t *src_cnt, const int32_t *rows,
                                                      int n_thin, int top_cap, int32_t *dst_idx, int32_t *dst_cnt) {
    const int tc = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
    const size_t s = (size_t)i * n_thin + tc, d = (size_t)rows[i] * n_thin + tc;
    const int n = min(max(src_cnt[s], 0), top_cap);
    if (lane < top_cap) dst_idx[d * top_cap + lane] = lane < n ? src_idx[s * top_cap + lane] : -1;
    if (lane == 0) dst_cnt[d] = src_cnt[s];
}

}  // namespace

namespace qa {

size_t select_lds_bytes(const SelectParams &p) {
    const size_t n_words = (p.K + 31) / 32, n_cand = (size_t)p.n_label * p.n_thin;
    return 4 * ((n_words + n_cand + 1) & ~(size_t)1) + 8 * std::max<size_t>(n_cand, p.Ksubset);
}

// false: the tables of one chain do not fit this device's LDS (160 KB per workgroup on gfx950; K, the thinned grids or Ksubset
// too large) -- nothing is launched and the caller reports status 1 for every chain, which sends the driver down its exact
// host path (the one it takes for truncated lists)
bool launch_select(const SelectParams &p, int n_chain, hipStream_t st) {
    const size_t lds = select_lds_bytes(p);
    int dev = 0, lds_max = 0;
    QA_HIP(hipGetDevice(&dev));
    QA_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    if (lds > (size_t)lds_max) return false;
    QA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_select, dim3(n_chain), dim3(64), lds, st, p);
    QA_HIP(hipGetLastError());
    return true;
}

void launch_scatter_lists(const int32_t *src_idx, const int32_t *src_cnt, const int32_t *rows, int n, int n_thin, int top_cap,
                          int32_t *dst_idx, int32_t *dst_cnt, hipStream_t st) {
    if (n <= 0 || n_thin <= 0) return;
    hipLaunchKernelGGL(k_scatter_lists, dim3(n_thin, n), dim3(64), 0, st, src_idx, src_cnt, rows, n_thin, top_cap, dst_idx,
                       dst_cnt);
    QA_HIP(hipGetLastError());
}

}  // namespace qa
