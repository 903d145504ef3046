// impute.cpp -- the per-sample driver loop behind the C ABI: qa_impute_samples (include/quilt_amd.h).
//
// What it replaces: the body of the reference's per-core loop over a sample range -- get_and_impute_one_sample
// (QUILT/R/functions.R:3-1500) as mclapply calls it (QUILT/R/quilt.R:688-996) -- for method = "diploid" with
// use_mspbwt = FALSE (mode M1: impute_using_everything, functions.R:1922-2157, everything_select_good_haps :2262-2310) or
// use_mspbwt = TRUE (functions.R:784-893, select_new_haps_mspbwt_v3, QUILT/R/mspbwt.R:225-474): the loop nest over Gibbs
// samples and seek iterations, the hand-over of which_haps_to_use, the (i_it > n_burn_in_seek_its) accumulation
// (:999-1020), read confidence (:1615-1660), consensus labels (:1680-1784), the phasing iteration and recast_haps
// (:3180-3209).  Host C++ (threads per device, no HIP here): every K-wide or read-wide piece of arithmetic is one of the
// library's own batched entry points -- qa_gibbs_batch, qa_fullpass_reads_select_batch, qa_rcpp_make_eMatRead_t_hap_major,
// qa_mspbwt_select_new_haps, qa_accumulate_dosage, qa_consensus_read_labels -- reached through a table of function pointers,
// so that the tests can run this very loop on a checker's entry points without a device (qa_impute_samples_backend,
// declared in the private impute_testhook.h -- not in the public header).
//
// How it is arranged for a GPU (the structure quilt_amd/driver.py + workers.py had in Python):
//   * all chains of a launch set of samples advance in lock-step: a round = ONE batched Gibbs call + ONE batched full-panel
//     call (+ selection on the device);
//   * launch sets are software-pipelined: the phasing rounds of set i (one chain per sample) share their launches with the
//     main rounds of set i + 1 (nGibbsSamples chains per sample);
//   * n_handles host threads, each with its own panel handle (stream, scratch), take whole launch sets in turn; what the
//     thread count leaves over goes whole to the first threads (a call with fewer sets than threads is cut across them); the
//     threads' last sets run their phasing rounds together in one launch per round (the thread that drains last runs them).
// Results do not depend on any of this: every (sample, Gibbs sample) owns its random stream (ChainStream below =
// quilt_amd/rng.py::ChainStream), keyed by the GLOBAL sample index.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/quilt_amd.h"
#include "../../include/quilt_amd_io.h"
#include "impute_testhook.h"   // the table of entry points this loop runs over (private: tests fill it with a checker's)

namespace qa { void set_error(const char *fmt, ...); }

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// random streams (quilt_amd/rng.py)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ull;
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t stream_u64(uint64_t seed, uint64_t i) { return mix64(seed + (i + 1) * GOLD); }   // element i (0-based)
constexpr uint64_t SELECT_OFFSET_PREV = 0, SELECT_OFFSET_RANK = 1ull << 20, SELECT_OFFSET_POOL = 1ull << 21;

// indices (into 0 .. n-1) of the m smallest keys of a stream, in key order, ties by index
std::vector<int32_t> keyed_subset(uint64_t seed, int64_t n, int64_t m, uint64_t offset) {
    std::vector<std::pair<uint64_t, int32_t>> k((size_t)std::max<int64_t>(n, 0));
    for (int64_t i = 0; i < n; i++) k[(size_t)i] = {stream_u64(seed, offset + (uint64_t)i), (int32_t)i};
    m = std::max<int64_t>(0, std::min(m, n));
    std::partial_sort(k.begin(), k.begin() + m, k.end());
    std::vector<int32_t> out((size_t)m);
    for (int64_t i = 0; i < m; i++) out[(size_t)i] = k[(size_t)i].second;
    return out;
}

struct ChainStream {
    uint64_t key = 0, ctr = 0;
    ChainStream() = default;
    ChainStream(uint64_t seed, int64_t i_sample, int64_t i_chain) {
        uint64_t k = mix64(seed + GOLD);
        k = mix64(k ^ ((uint64_t)(i_sample + 1) * 0xD1342543DE82EF95ull));
        key = mix64(k ^ ((uint64_t)(i_chain + 1) * 0x2545F4914F6CDD1Dull));
    }
    uint64_t u64() { return stream_u64(key, ctr++); }
    double uniform() { return (double)(u64() >> 11) * (1.0 / 9007199254740992.0); }
    // lo + floor(uniform * span); span as a double (2^63 for the seeds: the product stays below 2^63)
    int64_t integers(int64_t lo, double span) { return lo + (int64_t)std::floor(uniform() * span); }
    // choice(3, p): the number of cumulative masses (normalised by the last) <= uniform, at most 2 (numpy's searchsorted(cdf, u, "right"))
    int choice3(const double (&p)[3]) {
        double cdf[3] = {p[0], p[0] + p[1], (p[0] + p[1]) + p[2]};
        const double last = cdf[2];
        for (double &c : cdf) c /= last;
        const double u = uniform();
        int k = 0;
        while (k < 3 && cdf[k] <= u) k++;
        return k < 2 ? k : 2;
    }
    // choice(n, m, replace = False): the m smallest of the next n keys
    std::vector<int32_t> choice_without_replacement(int64_t n, int64_t m) {
        std::vector<int32_t> out = keyed_subset(key, n, m, ctr);
        ctr += (uint64_t)n;
        return out;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// the entry points the loop calls
// ---------------------------------------------------------------------------------------------------------------------------
struct Failure : std::runtime_error {
    int status;
    Failure(int st, const std::string &m) : std::runtime_error(m), status(st) {}
};

void check(int st, const char *what) {
    if (st == QA_OK) return;
    std::string m = std::string(what) + ": " + qa_last_error();
    throw Failure(st, m);
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Diagnostics (QA_IMPUTE_TRACE=1): the native calls of every host thread -- name, thread, start and end in ms on the gate trace's
// clock (qa_gate_trace), chains -- on stderr when the call returns; shows what of a call lies outside its device hold.
struct CallSpan {
    static bool on() { static const bool v = [] { const char *e = std::getenv("QA_IMPUTE_TRACE"); return e && e[0] == '1'; }(); return v; }
    const char *what; int w, n; double t0;
    CallSpan(const char *what_, int w_, int n_) : what(what_), w(w_), n(n_), t0(on() ? now_s() * 1e3 : 0) {}
    bool done = false;
    void end() { if (on() && !done) std::fprintf(stderr, "[impute-trace] %s thr %d n %d %.1f %.1f\n", what, w, n, t0, now_s() * 1e3); done = true; }
    ~CallSpan() { end(); }
};

// thinned_grid_columns (quilt.R:719-721): R's seq(1, nGrids, length.out = n) used as an index vector, renumbered densely
std::vector<int32_t> thinned_grid_columns(int G, double thin) {
    const int n = std::max(1, (int)std::nearbyint(thin * G));   // R's round(): half to even, as nearbyint
    std::vector<int32_t> cols((size_t)G, -1);
    for (int i = 0; i < n; i++) {
        // numpy.linspace(1, G, n): start + i * step, the last element exactly G
        const double w = (n == 1) ? 1.0 : (i == n - 1 ? (double)G : 1.0 + i * ((double)(G - 1) / (double)(n - 1)));
        const int g = (int)std::floor(w + 1e-9) - 1;
        cols[(size_t)std::min(std::max(g, 0), G - 1)] = i;
    }
    int next = 0;
    for (int g = 0; g < G; g++)
        if (cols[(size_t)g] >= 0) cols[(size_t)g] = next++;
    return cols;
}

struct Reads {   // one sample's reads: views into the caller's arrays
    int32_t R = 0;
    int64_t nb = 0;   // bases
    const int32_t *read_ptr = nullptr, *u = nullptr, *bq = nullptr, *wif = nullptr;
};

struct Chain {
    int sample = 0;       // index within the call
    int i_chain = 0;      // 1 .. nGibbsSamples; nGibbsSamples + 1 = the phasing iteration
    bool phasing = false;
    ChainStream rng;
    std::vector<int32_t> which;    // 1-based
    std::vector<int32_t> labels;
};

struct Batch {
    int lo = 0, hi = 0;   // samples [lo, hi) of the call
    std::vector<Chain> chains;    // main chains, sample-major
    std::vector<Chain> phasing;   // one per sample, once the main rounds are done
    bool done_flag = false;       // set under Tail::mu when another thread has run this batch's phasing rounds
    std::exception_ptr failed;
};

struct Tail {   // quilt_amd/driver.py::PhasingTail
    std::mutex mu;
    std::condition_variable cv;
    int n_active = 0;
    std::vector<Batch *> waiting;
    std::exception_ptr failed;
    void abort(std::exception_ptr e) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) failed = e;
        for (Batch *b : waiting) { b->failed = e; b->done_flag = true; }
        waiting.clear();
        cv.notify_all();
    }
};

struct Ctx {
    const qa_impute_backend_t *be = nullptr;
    bool product = false;   // the table is the library's own (qa_impute_samples): options only those entry points know may be used
    int K = 0, G = 0, T = 0;
    qa_impute_params_t P{};
    int n_burn = 0;
    std::vector<int32_t> blocks;
    std::vector<int32_t> cols;
    int n_thin = 0, top_width = 8;
    int64_t sample_offset = 0;
    const int64_t *sample_index = nullptr;   // params->sample_index
    int64_t global_index(int s) const { return sample_index ? sample_index[s] : sample_offset + s; }
    std::vector<Reads> reads;
    // impute_rare_common: the all-SNP reads, the all-SNP dimensions (T_out = T_all then) and where the common SNPs sit
    const qa_impute_nipt_t *nipt = nullptr;   // method = "nipt": three labels, a fetal fraction per sample
    int nL = 2;
    const qa_impute_rare_common_t *rc = nullptr;
    std::vector<Reads> reads_all;
    std::vector<int32_t> common_at;   // all-SNP index of common SNP j
    int T_out = 0;
    // outputs
    double *dosage = nullptr, *gp_t = nullptr, *phasing_haps = nullptr;
    int32_t *read_labels = nullptr, *nDosage = nullptr;
    const int32_t *read_off = nullptr;
    // params->sample_source: reads[s] / reads_all[s] / label_dst[s] are filled when the set holding s is taken (Worker::new_batch)
    const qa_sample_source_t *source = nullptr;
    std::vector<int32_t *> label_dst;   // where sample s's consensus read labels go
    std::atomic<int64_t> n_underflow_retries{0}, n_full_list_refetches{0}, n_device_selections{0}, n_gibbs_chain_calls{0},
        n_gibbs_launches{0};
    std::mutex stat_mu;
    double t_gibbs = 0, t_fullpass = 0, t_host = 0, t_consensus = 0, t_finish = 0, t_accumulate = 0;
    Tail tail;
    bool use_tail = false;
};

template <typename T>
struct HostBuf {   // grow-only transfer buffer from the backend's allocator (pinned with the product backend)
    const qa_impute_backend_t *be = nullptr;
    T *p = nullptr;
    size_t cap = 0;
    ~HostBuf() { if (p) be->host_free(p); }
    T *get(size_t n) {
        if (n > cap) {
            if (p) { be->host_free(p); p = nullptr; cap = 0; }
            const size_t want = n + n / 8;
            p = static_cast<T *>(be->host_alloc(want * sizeof(T)));
            if (!p) throw Failure(QA_ERR_HIP, std::string("qa_impute_samples: cannot allocate a transfer buffer: ") + qa_last_error());
            cap = want;
        }
        return p;
    }
};

// copies of many pieces into one array, on a few threads (a 2 048-chain Gibbs launch carries ~1.5 GB of per-chain read copies)
void parallel_for(size_t n, int n_thr, const std::function<void(size_t)> &f) {
    n_thr = (int)std::min<size_t>((size_t)std::max(n_thr, 1), std::max<size_t>(n, 1));
    if (n_thr <= 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex mu;
    std::vector<std::thread> th;
    for (int t = 0; t < n_thr; t++)
        th.emplace_back([&] {
            try {
                for (size_t i; (i = next.fetch_add(1)) < n;) f(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!err) err = std::current_exception();
            }
        });
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

int helper_threads(int cap = 8) {
    int c = std::min<int>(cap, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("QA_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) c = std::min(c, v);
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------------------------------
// host logic of the R driver
// ---------------------------------------------------------------------------------------------------------------------------

// everything_select_good_haps (functions.R:2262-2310) on a dense table top[label][thinned grid][rank] of 1-based haplotypes
// (0 = no entry), lists ordered best first; draws keyed as csrc/select.hip keys them
std::vector<int32_t> select_good_haps_dense(int Knew, int K_top_matches, const std::vector<int64_t> &top, int n_label, int n_thin,
                                            int width, const std::vector<int32_t> &prev, int K, uint64_t seed_select) {
    std::vector<int32_t> to_keep;
    std::vector<uint8_t> taken((size_t)K + 1, 0);
    for (int32_t h : prev) taken[(size_t)h] = 1;
    int i = 1;
    bool done = false;
    while (!done) {
        std::vector<int32_t> fresh;
        std::vector<uint8_t> seen;   // first occurrence wins within this rank's candidates
        auto offer = [&](int64_t v) {
            if (v <= 0 || v > K) return;
            if (taken[(size_t)v]) return;
            if (seen.empty()) seen.assign((size_t)K + 1, 0);
            if (seen[(size_t)v]) return;
            seen[(size_t)v] = 1;
            fresh.push_back((int32_t)v);
        };
        if (i <= K_top_matches && i <= width) {
            for (int a = 0; a < n_label; a++)
                for (int b = 0; b < n_thin; b++) offer(top[((size_t)a * n_thin + b) * width + (i - 1)]);
        } else {
            for (size_t j = 0; j < top.size(); j++) offer(top[j]);
            done = true;
        }
        if ((int)fresh.size() < Knew - (int)to_keep.size()) {
            for (int32_t v : fresh) { to_keep.push_back(v); taken[(size_t)v] = 1; }
            i++;
        } else {
            const int toadd = Knew - (int)to_keep.size();
            for (int32_t j : keyed_subset(seed_select, (int64_t)fresh.size(), toadd, SELECT_OFFSET_RANK)) {
                to_keep.push_back(fresh[(size_t)j]);
                taken[(size_t)fresh[(size_t)j]] = 1;
            }
            done = true;
        }
    }
    if ((int)to_keep.size() < Knew) {   // functions.R:2297-2301: the rest at random from the haplotypes not yet in
        std::vector<int32_t> pool;
        for (int h = 1; h <= K; h++)
            if (!taken[(size_t)h]) pool.push_back(h);
        for (int32_t j : keyed_subset(seed_select, (int64_t)pool.size(), Knew - (int64_t)to_keep.size(), SELECT_OFFSET_POOL))
            to_keep.push_back(pool[(size_t)j]);
    }
    if ((int)to_keep.size() != Knew) throw Failure(QA_ERR_INVALID, "Have returned too many haps");
    return to_keep;
}

// recast_haps (functions.R:3180-3209), in place on hd1 / hd2; g = gp_t (3 x T rows)
void recast_haps(double *hd1, double *hd2, const double *g0, const double *g1, const double *g2, int T) {
    for (int t = 0; t < T; t++) {
        const double gt1 = std::nearbyint(hd1[t]) + std::nearbyint(hd2[t]);   // R's round(): half to even
        double mx = g0[t];
        double gt3 = 0;
        if (g1[t] > mx) { gt3 = 1; mx = g1[t]; }
        if (g2[t] > mx) { gt3 = 2; mx = g2[t]; }
        if (gt3 == gt1) continue;
        if (gt3 == 0) { hd1[t] = 0; hd2[t] = 0; }
        else if (gt3 == 2) { hd1[t] = 1; hd2[t] = 1; }
        else {
            const bool first = hd1[t] > hd2[t];
            hd1[t] = first ? 1.0 : 0.0;
            hd2[t] = first ? 0.0 : 1.0;
        }
    }
}

// recast_nipt_haps (functions.R:3214-3287): the phased haplotypes of mother and fetus made to agree with the argmax genotypes;
// mg / fg = mother's and fetus' gp_t (3 x T rows).  In place on hap1 (maternal transmitted), hap2 (maternal untransmitted), hap3
// (paternal transmitted); every output is rounded.
void recast_nipt_haps(double *hap1, double *hap2, double *hap3, const double *mg, const double *fg, int T) {
    for (int t = 0; t < T; t++) {
        int gm = 0, gf = 0;
        double mxA = mg[t], mxB = fg[t];
        for (int i = 1; i <= 2; i++) {
            if (mg[(size_t)i * T + t] > mxA) { gm = i; mxA = mg[(size_t)i * T + t]; }
            if (fg[(size_t)i * T + t] > mxB) { gf = i; mxB = fg[(size_t)i * T + t]; }
        }
        double a = hap1[t], b = hap2[t], c = hap3[t];
        static const int conv[8][5] = {{0, 0, 0, 0, 0}, {0, 1, 0, 0, 1}, {0, 2, 0, 0, 1}, {1, 0, 0, 1, 0}, {1, 2, 1, 0, 1}, {2, 0, 1, 1, 0},
                                       {2, 1, 1, 1, 0}, {2, 2, 1, 1, 1}};
        bool done = false;
        for (const auto &r : conv)
            if (gm == r[0] && gf == r[1]) { a = r[2]; b = r[3]; c = r[4]; done = true; }
        if (!done) {   // mother het, fetus het: keep the call if it is one of the two consistent ones, else round (functions.R:3262-3283)
            const double r1 = std::nearbyint(a), r2 = std::nearbyint(b), r3 = std::nearbyint(c);
            if (r1 == 1 && r2 == 0 && r3 == 0) { a = 1; b = 0; c = 0; }
            else if (r1 == 0 && r2 == 1 && r3 == 1) { a = 0; b = 1; c = 1; }
            else { a = r1; b = r2; c = 1 - a; }
        }
        hap1[t] = std::nearbyint(a); hap2[t] = std::nearbyint(b); hap3[t] = std::nearbyint(c);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// one host thread: its handle, its buffers, its stream of launch sets
// ---------------------------------------------------------------------------------------------------------------------------
// get_initial_read_labels for method = "nipt" (rare_common.R:104-105 with get_read_groupings_given_fetal_fraction_and_cov and
// sample_H_for_NIPT_given_groupings, gibbs-nipt.R:1655-1777, :1796-1849; the statement in quilt_amd/driver.py): e = the all-SNP
// reads' rescaled likelihoods against (hap1, hap2, hap3), [read][3].  A read is grouped by which haplotypes it fits (> 0.5);
// reads that fit all or none follow the label prior, reads that fit exactly one take its label, reads shared by two are split
// between them in the ratio of the label priors (preserve_round) and drawn with those proportions.  Draws in the order of the
// numpy text: one vector for the prior group, then one per pair (1,2), (1,3), (2,3) that has reads.
void initial_read_labels_nipt(const double *e, int R, double ff, ChainStream &rng, std::vector<int32_t> &H) {
    const double frp[3] = {0.5, 0.5 - ff / 2, ff / 2};
    std::vector<uint8_t> m((size_t)R * 3);
    std::vector<int> n_fit((size_t)R, 0);
    for (int r = 0; r < R; r++)
        for (int i = 0; i < 3; i++) {
            const double v = e[(size_t)r * 3 + i];
            const bool fit = !std::isnan(v) && v > 0.5;
            m[(size_t)r * 3 + i] = fit;
            n_fit[(size_t)r] += fit;
        }
    std::fill(H.begin(), H.end(), 0);
    for (int r = 0; r < R; r++)
        if (n_fit[(size_t)r] == 3 || n_fit[(size_t)r] == 0) H[(size_t)r] = 1 + rng.choice3(frp);
    for (int r = 0; r < R; r++)
        if (n_fit[(size_t)r] == 1)
            for (int i = 0; i < 3; i++)
                if (m[(size_t)r * 3 + i]) H[(size_t)r] = i + 1;
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++) {
            std::vector<int> both;
            for (int r = 0; r < R; r++)
                if (n_fit[(size_t)r] == 2 && m[(size_t)r * 3 + i] && m[(size_t)r * 3 + j]) both.push_back(r);
            const int n = (int)both.size();
            if (n == 0) continue;
            const double f1 = frp[i] / (frp[i] + frp[j]);
            // preserve_round(n * c(f1, 1 - f1)) (gibbs-nipt.R:1779-1789): floors, then the largest fractional parts go up
            const double x[2] = {n * f1, n * (1 - f1)};
            double y[2] = {std::floor(x[0]), std::floor(x[1])};
            const int n_up = (int)(std::nearbyint(x[0] + x[1]) - (y[0] + y[1]));
            if (n_up > 0) {
                // numpy: argsort(x - y, stable)[-n_up:] -- of two elements the larger fractional part (ties: the second) first
                const double fr[2] = {x[0] - y[0], x[1] - y[1]};
                const int order[2] = {fr[1] < fr[0] ? 1 : 0, fr[1] < fr[0] ? 0 : 1};   // ascending, stable
                for (int q = 0; q < n_up && q < 2; q++) y[order[1 - q]] += 1;
            }
            const double thr = y[0] / (y[0] + y[1]);
            for (int r : both) H[(size_t)r] = (rng.uniform() < thr) ? i + 1 : j + 1;
        }
}

// A host thread's marshalling and transfer buffers.  With the product's entry points they outlive the call (kept per panel
// handle, freed by qa_impute_release_buffers): a launch set of 2 048 chains carries ~1.5 GB of per-chain read copies and 2 GB
// of pinned dosage rows, and allocating -- pinning, first-touching -- them anew inside every call cost 1.5 s of a 70 s run.
struct WorkerBuffers {
    // marshalling buffers of the Gibbs call (per chain copies of the reads) and of the full-panel call (per sample)
    std::vector<int32_t> g_which, g_read_off, g_read_ptr, g_u, g_bq, g_wif, g_first, g_H, g_uf, g_words;
    std::vector<uint64_t> g_sr, g_ss, seed_sel;
    std::vector<int32_t> f_cs, f_read_off, f_read_ptr, f_u, f_bq, f_H, f_wd, f_wt, f_cnt, f_next, f_status;
    HostBuf<double> dos;          // the round's haploid dosages [chain][label][T]
    HostBuf<double> conf;         // read confidence [reads][K]
    HostBuf<double> dos_all;      // impute_rare_common: the all-SNP round's haploid dosages [chain][2][all SNPs]
    explicit WorkerBuffers(const qa_impute_backend_t *be) { dos.be = be; conf.be = be; dos_all.be = be; }
};
// product backend only: handle -> its thread's buffers.  Never destroyed: a static's destructor would free pinned memory after
// the HIP runtime (and the library's own registry of pinned regions) may be gone; qa_impute_release_buffers frees in time.
std::mutex &buf_mu() { static std::mutex *m = new std::mutex; return *m; }
std::map<void *, std::unique_ptr<WorkerBuffers>> &bufs() {
    static auto *m = new std::map<void *, std::unique_ptr<WorkerBuffers>>;
    return *m;
}

struct Worker {
    Ctx &cx;
    void *handle;
    int w;
    int n_help;
    int n_draw;   // threads for per-chain arithmetic (the copies' n_help is bounded by memory bandwidth, this by cores)
    std::unique_ptr<WorkerBuffers> own;   // a caller-supplied table of entry points: buffers of this call only
    WorkerBuffers &B;
    std::vector<int32_t> &g_which, &g_read_off, &g_read_ptr, &g_u, &g_bq, &g_wif, &g_first, &g_H, &g_uf, &g_words;
    std::vector<uint64_t> &g_sr, &g_ss, &seed_sel;
    std::vector<int32_t> &f_cs, &f_read_off, &f_read_ptr, &f_u, &f_bq, &f_H, &f_wd, &f_wt, &f_cnt, &f_next, &f_status;
    HostBuf<double> &dos, &conf, &dos_all;
    std::function<void()> on_first_launch;
    double t_gibbs = 0, t_fullpass = 0, t_host = 0, t_consensus = 0, t_finish = 0, t_accumulate = 0;

    static WorkerBuffers &buffers_for(Ctx &c, void *h, bool keep, std::unique_ptr<WorkerBuffers> &own) {
        if (!keep) {
            own.reset(new WorkerBuffers(c.be));
            return *own;
        }
        std::lock_guard<std::mutex> g(buf_mu());
        auto &slot = bufs()[h];
        if (!slot) slot.reset(new WorkerBuffers(c.be));
        return *slot;
    }
    const void *rc_handle = nullptr;   // this thread's qa_rare_common_t (impute_rare_common)
    std::vector<double> eh;            // eHapsCurrent_tc of get_initial_read_labels: [chain][all SNPs][2]
    Worker(Ctx &c, void *h, int wi, bool keep)
        : cx(c), handle(h), w(wi), n_help(helper_threads()), n_draw(helper_threads(24)), B(buffers_for(c, h, keep, own)), g_which(B.g_which),
          g_read_off(B.g_read_off), g_read_ptr(B.g_read_ptr), g_u(B.g_u), g_bq(B.g_bq), g_wif(B.g_wif), g_first(B.g_first), g_H(B.g_H),
          g_uf(B.g_uf), g_words(B.g_words), g_sr(B.g_sr), g_ss(B.g_ss), seed_sel(B.seed_sel), f_cs(B.f_cs), f_read_off(B.f_read_off),
          f_read_ptr(B.f_read_ptr), f_u(B.f_u), f_bq(B.f_bq), f_H(B.f_H), f_wd(B.f_wd), f_wt(B.f_wt), f_cnt(B.f_cnt), f_next(B.f_next),
          f_status(B.f_status), dos(B.dos), conf(B.conf), dos_all(B.dos_all) {}

    // ---- the Gibbs call of a round with impute_one_sample's underflow retry (functions.R:2612-2716)
    // rare: the all-SNP call (qa_gibbs_batch_rare_common on the samples' all-SNP reads, labels given, read categories off:
    // impute_one_sample's defaults for it, functions.R:2385-2409); hap_out then [chain][2][all SNPs]
    void gibbs_with_retry(std::vector<Chain *> &ch, const std::vector<std::vector<int32_t>> &starts, bool any_first, bool want_words,
                          double *hap_out, bool rare = false) {
        const auto &P = cx.P;
        const int C = (int)ch.size();
        const int G = cx.G, T = rare ? cx.T_out : cx.T;
        const std::vector<Reads> &RD = rare ? cx.reads_all : cx.reads;
        const int nLh = cx.nL;   // labels of the call's haploid dosages
        std::vector<int> pending((size_t)C);
        for (int i = 0; i < C; i++) pending[(size_t)i] = i;
        std::vector<double> maxdiff((size_t)C, P.maxDifferenceBetweenReads);
        if (want_words) g_words.assign((size_t)C * 3 * G, 0);
        int n_try = 0;
        while (!pending.empty()) {
            std::vector<std::pair<double, std::vector<int>>> groups;   // by maxDifferenceBetweenReads, in order of appearance
            for (int i : pending) {
                auto it = std::find_if(groups.begin(), groups.end(), [&](const auto &g) { return g.first == maxdiff[(size_t)i]; });
                if (it == groups.end()) groups.push_back({maxdiff[(size_t)i], {i}});
                else it->second.push_back(i);
            }
            std::vector<int> nxt;
            for (auto &grp : groups) {
                const std::vector<int> &idx = grp.second;
                const int n = (int)idx.size();
                const bool whole = (n == C && n_try == 0);   // the round's first call writes straight into the round's buffers
                g_read_off.assign((size_t)n + 1, 0);
                std::vector<int64_t> base_off((size_t)n + 1, 0);
                for (int a = 0; a < n; a++) {
                    const Reads &r = RD[(size_t)ch[(size_t)idx[(size_t)a]]->sample];
                    g_read_off[(size_t)a + 1] = g_read_off[(size_t)a] + r.R;
                    base_off[(size_t)a + 1] = base_off[(size_t)a] + r.nb;
                }
                const int64_t totR = g_read_off[(size_t)n], totB = base_off[(size_t)n];
                g_which.resize((size_t)n * P.Ksubset);
                g_read_ptr.resize((size_t)totR + n);
                g_wif.resize((size_t)totR);
                g_H.resize((size_t)totR);
                g_u.resize((size_t)totB);
                g_bq.resize((size_t)totB);
                g_first.resize((size_t)n);
                g_sr.resize((size_t)n);
                g_ss.resize((size_t)n);
                g_uf.assign((size_t)n, 0);
                // the chains of one sample share its reads: on the library's own entry points the bases travel once per sample
                // (qa_gibbs_opts_t.reads_same_as); a caller-supplied table of entry points gets every chain's copy
                std::vector<int32_t> same_as;
                if (cx.product) {
                    same_as.resize((size_t)n);
                    std::map<int, int> first_of;
                    for (int a = 0; a < n; a++) {
                        const int sm = ch[(size_t)idx[(size_t)a]]->sample;
                        auto it = first_of.find(sm);
                        if (it == first_of.end()) it = first_of.emplace(sm, a).first;
                        same_as[(size_t)a] = it->second;
                    }
                }
                parallel_for((size_t)n, n_help, [&](size_t a) {
                    const int i = idx[a];
                    const Chain &c = *ch[(size_t)i];
                    const Reads &r = RD[(size_t)c.sample];
                    std::memcpy(&g_which[a * P.Ksubset], c.which.data(), sizeof(int32_t) * (size_t)P.Ksubset);
                    std::memcpy(&g_read_ptr[(size_t)g_read_off[a] + a], r.read_ptr, sizeof(int32_t) * ((size_t)r.R + 1));
                    std::memcpy(&g_wif[(size_t)g_read_off[a]], r.wif, sizeof(int32_t) * (size_t)r.R);
                    std::memcpy(&g_H[(size_t)g_read_off[a]], starts[(size_t)i].data(), sizeof(int32_t) * (size_t)r.R);
                    if (!same_as.empty() && same_as[a] != (int32_t)a) return;   // (its bases are another chain's)
                    std::memcpy(&g_u[(size_t)base_off[a]], r.u, sizeof(int32_t) * (size_t)r.nb);
                    std::memcpy(&g_bq[(size_t)base_off[a]], r.bq, sizeof(int32_t) * (size_t)r.nb);
                });
                for (int a = 0; a < n; a++) {
                    g_first[(size_t)a] = first_reads[(size_t)idx[(size_t)a]];
                    g_sr[(size_t)a] = seed_reads[(size_t)idx[(size_t)a]];
                    g_ss[(size_t)a] = seed_shards[(size_t)idx[(size_t)a]];
                }
                qa_gibbs_opts_t o{};
                o.Ks = P.Ksubset;
                o.reads_same_as = same_as.empty() ? nullptr : same_as.data();
                std::vector<double> ffc;
                if (cx.nipt) {   // every chain carries its sample's fetal fraction (functions.R:128)
                    ffc.resize((size_t)n);
                    for (int a = 0; a < n; a++) ffc[(size_t)a] = cx.nipt->ff[ch[(size_t)idx[(size_t)a]]->sample];
                }
                o.ff = cx.nipt ? ffc[0] : 0.0;
                o.ff_chain = cx.nipt ? ffc.data() : nullptr;
                o.sample_is_diploid = cx.nipt ? 0 : 1;
                o.Jmax = P.Jmax;
                o.maxDifferenceBetweenReads = grp.first;
                o.rescale_eMatRead_t = 1;
                o.n_gibbs_burn_in_its = P.small_ref_panel_gibbs_iterations;
                o.n_gibbs_sample_its = P.n_gibbs_sample_its;
                o.block_gibbs_iterations = cx.blocks.data();
                o.n_block_gibbs_iterations = (int32_t)cx.blocks.size();
                o.perform_block_gibbs = 1;
                o.do_shard_block_gibbs = cx.nipt ? 0 : 1;   // (functions.R:2552-2556: no shard pass for ff > 0)
                o.gibbs_initialize_iteratively = any_first ? 1 : 0;
                o.disable_read_category_usage = rare ? 1 : 0;
                o.class_sum_cutoff = 0.06;
                o.L_grid = cx.nipt ? (rare ? cx.rc->L_grid_all : cx.nipt->L_grid) : nullptr;
                o.shuffle_bin_radius = cx.nipt ? cx.nipt->shuffle_bin_radius : 5000;
                o.block_gibbs_quantile_prob = 0.95;
                std::vector<int32_t> words_tmp;
                std::vector<double> hap_tmp;
                if (want_words) {
                    if (whole) o.hap_words_out = g_words.data();
                    else { words_tmp.assign((size_t)n * 3 * G, 0); o.hap_words_out = words_tmp.data(); }
                }
                if (hap_out) {
                    if (whole) o.hap_major_out = hap_out;
                    else { hap_tmp.resize((size_t)n * nLh * T); o.hap_major_out = hap_tmp.data(); }
                    o.hap_major_labels = nLh;
                }
                if (on_first_launch) { auto cb = on_first_launch; on_first_launch = nullptr; cb(); }
                cx.n_gibbs_chain_calls += n;
                cx.n_gibbs_launches += 1;
                CallSpan span_g(rare ? "gibbs_rc" : "gibbs", w, n);
                const int st = rare
                    ? cx.be->gibbs_batch_rare_common(handle, rc_handle, &o, n, g_which.data(), g_read_off.data(), g_read_ptr.data(),
                                                     g_u.data(), g_bq.data(), g_wif.data(), nullptr, g_first.data(), nullptr, g_H.data(),
                                                     nullptr, nullptr, nullptr, nullptr, g_uf.data(), nullptr, g_sr.data(), g_ss.data())
                    : cx.be->gibbs_batch(handle, &o, n, g_which.data(), g_read_off.data(), g_read_ptr.data(), g_u.data(),
                                         g_bq.data(), g_wif.data(), nullptr, g_first.data(), nullptr, g_H.data(), nullptr,
                                         nullptr, nullptr, nullptr, g_uf.data(), nullptr, g_sr.data(), g_ss.data());
                span_g.end();
                if (st != QA_OK && st != QA_UNDERFLOW) check(st, rare ? "qa_gibbs_batch_rare_common" : "qa_gibbs_batch");
                for (int a = 0; a < n; a++) {
                    const int i = idx[(size_t)a];
                    if (g_uf[(size_t)a]) {
                        maxdiff[(size_t)i] = std::max(1.0, maxdiff[(size_t)i] / 10);   // functions.R:2704-2715
                        nxt.push_back(i);
                        cx.n_underflow_retries += 1;
                        continue;
                    }
                    Chain &c = *ch[(size_t)i];
                    const int R = RD[(size_t)c.sample].R;
                    // (the all-SNP call's ending labels are not carried on: the next Gibbs sample starts afresh, the phasing
                    // iteration's result is its haplotypes)
                    if (!rare) c.labels.assign(g_H.begin() + g_read_off[(size_t)a], g_H.begin() + g_read_off[(size_t)a] + R);
                    if (!whole) {
                        if (want_words) std::memcpy(&g_words[(size_t)i * 3 * G], &words_tmp[(size_t)a * 3 * G], sizeof(int32_t) * 3 * (size_t)G);
                        if (hap_out) std::memcpy(hap_out + (size_t)i * nLh * T, &hap_tmp[(size_t)a * nLh * T], sizeof(double) * nLh * (size_t)T);
                    }
                }
            }
            pending = nxt;
            n_try++;
            if (n_try > 10 && !pending.empty())
                throw Failure(QA_ERR_INVALID, "There were consecutive underflow problems (functions.R:2710)");
        }
    }

    std::vector<int32_t> first_reads;
    std::vector<uint64_t> seed_reads, seed_shards;

    // ---- complete best-haplotype lists of one chain (the reference's lists hold every haplotype at or above the threshold;
    // the batched call keeps their first top_width entries): make_gl_from_u_bq (reference-single.R:19-42) per label, a thin
    // full-panel pass returning whole lists, ordered per thinned grid as everything_per_hap_rejig_haps does (functions.R:2161-2170)
    std::vector<int64_t> full_lists(const Chain &c, int &width_out) {
        const auto &P = cx.P;
        const int T = cx.T, nL = cx.nL, n_thin = cx.n_thin;
        const Reads &r = cx.reads[(size_t)c.sample];
        std::vector<double> gl((size_t)nL * 2 * T, 1.0);   // per label a 2 x T column-major matrix
        for (int l = 1; l <= nL; l++) {
            double *g = &gl[(size_t)(l - 1) * 2 * T];
            for (int rd = 0; rd < r.R; rd++) {
                if (c.labels[(size_t)rd] != l) continue;
                for (int32_t j = r.read_ptr[rd]; j < r.read_ptr[rd + 1]; j++) {
                    const int32_t q = r.bq[j];
                    if (q == 0) continue;
                    const double eps = std::pow(10.0, -std::fabs((double)q) / 10.0);
                    const double pR = q < 0 ? 1 - eps : eps / 3, pA = q < 0 ? eps / 3 : 1 - eps;
                    g[(size_t)2 * r.u[j]] *= pR;
                    g[(size_t)2 * r.u[j] + 1] *= pA;
                }
            }
            if (P.minGLValue > 0) {
                std::vector<int32_t> to_fix;
                for (int t = 0; t < T; t++)
                    if (g[(size_t)2 * t] < P.minGLValue || g[(size_t)2 * t + 1] < P.minGLValue) to_fix.push_back(t);
                if (!to_fix.empty()) check(qa_Rcpp_make_gl_bound(g, P.minGLValue, to_fix.data(), (int32_t)to_fix.size()), "qa_Rcpp_make_gl_bound");
            }
        }
        std::vector<int32_t> wd((size_t)nL, 0), bptr((size_t)nL * n_thin + 1, 0);
        int64_t cap = (int64_t)nL * n_thin * 16;
        std::vector<int32_t> bidx;
        std::vector<double> bval;
        for (int attempt = 0; attempt < 2; attempt++) {
            bidx.assign((size_t)cap, 0);
            bval.assign((size_t)cap, 0.0);
            const int st = cx.be->fullpass_batch(handle, nL, gl.data(), wd.data(), cx.cols.data(), P.K_top_matches, nullptr, bptr.data(),
                                                 bidx.data(), bval.data(), cap);
            if (st == QA_ERR_CAPACITY && attempt == 0) { cap = bptr.back(); continue; }
            check(st, "qa_fullpass_batch");
            break;
        }
        int width = 1;
        for (int p = 0; p < nL * n_thin; p++) width = std::max(width, bptr[(size_t)p + 1] - bptr[(size_t)p]);
        std::vector<int64_t> top((size_t)nL * n_thin * width, 0);
        std::vector<int> ord;
        for (int p = 0; p < nL * n_thin; p++) {
            const int b0 = bptr[(size_t)p], n = bptr[(size_t)p + 1] - b0;
            ord.resize((size_t)n);
            for (int j = 0; j < n; j++) ord[(size_t)j] = j;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return bval[(size_t)b0 + a] > bval[(size_t)b0 + b]; });
            for (int j = 0; j < n; j++) top[(size_t)p * width + j] = (int64_t)bidx[(size_t)b0 + ord[(size_t)j]] + 1;
        }
        width_out = width;
        return top;
    }

    // ---- one [Gibbs -> full-panel pass per label -> new small panel] round over a set of chains at the same seek iteration.
    // The first n_cur chains are the current launch set's main chains (their dosages are accumulated); the rest are phasing chains.
    bool round(std::vector<Chain *> &ch, int i_it, Batch *cur) {
        const auto &P = cx.P;
        const int C = (int)ch.size(), K = cx.K, T = cx.T, G = cx.G, nL = cx.nL;
        const double t0 = now_s();
        bool any_first = false;
        std::vector<std::vector<int32_t>> starts((size_t)C);
        first_reads.assign((size_t)C, 0);
        seed_reads.assign((size_t)C, 0);
        seed_shards.assign((size_t)C, 0);
        for (int i = 0; i < C; i++) {
            if (cx.reads[(size_t)ch[(size_t)i]->sample].R < 1)
                throw Failure(QA_ERR_INVALID, "a sample without reads cannot be imputed (no read intersects a SNP of the region)");
            any_first |= (i_it == 1) && !ch[(size_t)i]->phasing;
        }
        // (every chain draws from its own stream: the chains' draws -- a keyed subset of the K haplotypes and one label per read
        // in a set's first round, 0.5 ms a chain at K = 50 000 -- are made side by side)
        parallel_for((size_t)C, n_draw, [&](size_t i) {
            Chain &c = *ch[i];
            const int R = cx.reads[(size_t)c.sample].R;
            const bool first = (i_it == 1) && !c.phasing;
            if (first) {   // functions.R:579-585
                c.which = c.rng.choice_without_replacement(K, P.Ksubset);
                std::sort(c.which.begin(), c.which.end());
                for (auto &v : c.which) v += 1;
                starts[i].resize((size_t)R);
                if (cx.nipt) {   // functions.R:586: sample(1:3, nReads, prob = c(0.5, 0.5 - ff / 2, ff / 2))
                    const double ff = cx.nipt->ff[c.sample];
                    const double pr[3] = {0.5, 0.5 - ff / 2, ff / 2};
                    for (int r = 0; r < R; r++) starts[i][(size_t)r] = 1 + c.rng.choice3(pr);
                } else {
                    for (int r = 0; r < R; r++) starts[i][(size_t)r] = (int32_t)c.rng.integers(1, 2.0);
                }
            } else {
                starts[i] = c.labels;
            }
            seed_reads[i] = (uint64_t)c.rng.integers(0, 9223372036854775808.0);
            const int32_t fr = (int32_t)c.rng.integers(0, (double)R);
            first_reads[i] = first ? fr : -1;
            seed_shards[i] = (uint64_t)c.rng.integers(0, 9223372036854775808.0);
        });
        if (!any_first) std::fill(first_reads.begin(), first_reads.end(), 0);
        const double t1 = now_s();
        t_host += t1 - t0;
        const bool return_dosage = i_it > cx.n_burn;
        double *hap = nullptr;
        if (return_dosage) hap = dos.get((size_t)C * nL * T);
        if (P.use_mspbwt) {
            gibbs_with_retry(ch, starts, any_first, true, return_dosage ? hap : nullptr);
            const double t2 = now_s();
            t_gibbs += t2 - t1;
            // functions.R:784-893: the next small panel from the long matches of the call's rounded haploid dosages
            std::vector<int> idx;
            for (int i = 0; i < C; i++)
                if (i_it < P.n_seek_its || cx.rc || (!ch[(size_t)i]->phasing && ch[(size_t)i]->i_chain == P.nGibbsSamples)) idx.push_back(i);
            if (!idx.empty()) {
                std::vector<uint64_t> seeds(idx.size());
                std::vector<int32_t> Zs(idx.size() * nL * (size_t)G), out(idx.size() * (size_t)P.Knew);
                for (size_t a = 0; a < idx.size(); a++) {
                    seeds[a] = (uint64_t)ch[(size_t)idx[a]]->rng.integers(0, 9223372036854775808.0);
                    std::memcpy(&Zs[a * nL * G], &g_words[(size_t)idx[a] * 3 * G], sizeof(int32_t) * nL * (size_t)G);
                }
                check(cx.be->mspbwt_select_new_haps(P.mspbwt_index, (int32_t)idx.size(), nL, Zs.data(), P.mspbwtL, P.mspbwtM, P.Knew,
                                                    seeds.data(), out.data()), "qa_mspbwt_select_new_haps");
                for (size_t a = 0; a < idx.size(); a++)
                    ch[(size_t)idx[a]]->which.assign(out.begin() + a * P.Knew, out.begin() + (a + 1) * P.Knew);
            }
            t_fullpass += now_s() - t2;
        } else {
            gibbs_with_retry(ch, starts, any_first, false, nullptr);
            const double t2 = now_s();
            t_gibbs += t2 - t1;
            // ---- impute_using_everything for every chain, selection behind it
            std::map<int, int> uniq;
            std::vector<int> sample_list;
            f_cs.resize((size_t)C);
            for (int i = 0; i < C; i++) {
                auto it = uniq.find(ch[(size_t)i]->sample);
                if (it == uniq.end()) { it = uniq.emplace(ch[(size_t)i]->sample, (int)sample_list.size()).first; sample_list.push_back(ch[(size_t)i]->sample); }
                f_cs[(size_t)i] = it->second;
            }
            const int nS = (int)sample_list.size();
            f_read_off.assign((size_t)nS + 1, 0);
            std::vector<int64_t> boff((size_t)nS + 1, 0);
            for (int s = 0; s < nS; s++) {
                const Reads &r = cx.reads[(size_t)sample_list[(size_t)s]];
                f_read_off[(size_t)s + 1] = f_read_off[(size_t)s] + r.R;
                boff[(size_t)s + 1] = boff[(size_t)s] + r.nb;
            }
            f_read_ptr.resize((size_t)f_read_off[(size_t)nS] + nS);
            f_u.resize((size_t)boff[(size_t)nS]);
            f_bq.resize((size_t)boff[(size_t)nS]);
            for (int s = 0; s < nS; s++) {
                const Reads &r = cx.reads[(size_t)sample_list[(size_t)s]];
                std::memcpy(&f_read_ptr[(size_t)f_read_off[(size_t)s] + s], r.read_ptr, sizeof(int32_t) * ((size_t)r.R + 1));
                std::memcpy(&f_u[(size_t)boff[(size_t)s]], r.u, sizeof(int32_t) * (size_t)r.nb);
                std::memcpy(&f_bq[(size_t)boff[(size_t)s]], r.bq, sizeof(int32_t) * (size_t)r.nb);
            }
            size_t totH = 0;
            for (int i = 0; i < C; i++) totH += ch[(size_t)i]->labels.size();
            f_H.resize(totH);
            {
                size_t at = 0;
                for (int i = 0; i < C; i++) {
                    std::memcpy(&f_H[at], ch[(size_t)i]->labels.data(), sizeof(int32_t) * ch[(size_t)i]->labels.size());
                    at += ch[(size_t)i]->labels.size();
                }
            }
            f_wd.assign((size_t)C, return_dosage ? 1 : 0);
            f_wt.resize((size_t)C);
            bool any_top = false;
            for (int i = 0; i < C; i++) {
                f_wt[(size_t)i] = (i_it < P.n_seek_its || cx.rc || (!ch[(size_t)i]->phasing && ch[(size_t)i]->i_chain == P.nGibbsSamples)) ? 1 : 0;
                any_top |= f_wt[(size_t)i] != 0;
            }
            seed_sel.resize((size_t)C);
            for (int i = 0; i < C; i++) seed_sel[(size_t)i] = (uint64_t)ch[(size_t)i]->rng.integers(0, 9223372036854775808.0);
            g_which.resize((size_t)C * P.Ksubset);
            for (int i = 0; i < C; i++) std::memcpy(&g_which[(size_t)i * P.Ksubset], ch[(size_t)i]->which.data(), sizeof(int32_t) * (size_t)P.Ksubset);
            f_cnt.assign((size_t)C * nL * cx.n_thin, 0);
            f_next.assign((size_t)C * P.Ksubset, 0);
            f_status.assign((size_t)C, -1);
            const double t3 = now_s();
            t_host += t3 - t2;
            CallSpan span_f("fullpass_select", w, C);
            check(cx.be->fullpass_reads_select_batch(handle, C, nL, nS, f_cs.data(), f_read_off.data(), f_read_ptr.data(), f_u.data(),
                                                     f_bq.data(), f_H.data(), f_wd.data(), f_wt.data(), cx.cols.data(), P.K_top_matches,
                                                     P.minGLValue, hap, cx.top_width, nullptr, nullptr, f_cnt.data(), P.Ksubset, P.Knew,
                                                     g_which.data(), seed_sel.data(), f_next.data(), f_status.data()),
                  "qa_fullpass_reads_select_batch");
            span_f.end();
            const double t4 = now_s();
            t_fullpass += t4 - t3;
            if (return_dosage) {   // functions.R:2072-2075
                std::atomic<bool> bad{false};
                parallel_for((size_t)C, n_help, [&](size_t i) {
                    const double *d = hap + i * nL * T;
                    for (size_t t = 0; t < (size_t)nL * T; t++)
                        if (!(d[t] >= -1e-5 && d[t] <= 1 + 1e-5)) { bad = true; break; }
                });
                if (bad) throw Failure(QA_ERR_INVALID, "Dosage observed outside of range of 0 to 1 on forward-backward full iteration");
            }
            (void)any_top;
            for (int i = 0; i < C; i++) {
                if (!f_wt[(size_t)i]) continue;
                Chain &c = *ch[(size_t)i];
                if (f_status[(size_t)i] == 0) {
                    c.which.assign(f_next.begin() + (size_t)i * P.Ksubset, f_next.begin() + (size_t)(i + 1) * P.Ksubset);
                    cx.n_device_selections += 1;
                    continue;
                }
                // the ranks up to K_top_matches did not yield Knew new haplotypes: every entry of the complete lists, then a
                // random draw from the rest of the panel (functions.R:2278-2300)
                std::vector<int32_t> prev;
                for (int32_t j : keyed_subset(seed_sel[(size_t)i], P.Ksubset, P.Ksubset - P.Knew, SELECT_OFFSET_PREV)) prev.push_back(c.which[(size_t)j]);
                cx.n_full_list_refetches += 1;
                int width = 1;
                std::vector<int64_t> top = full_lists(c, width);
                std::vector<int32_t> sel = select_good_haps_dense(P.Knew, P.K_top_matches, top, nL, cx.n_thin, width, prev, K, seed_sel[(size_t)i]);
                c.which = prev;
                c.which.insert(c.which.end(), sel.begin(), sel.end());
            }
            t_host += now_s() - t4;
        }
        // the phasing chains' haploid dosages are their samples' phased haplotypes (functions.R:1207-1217, before recast_haps)
        if (return_dosage && !cx.rc)
            for (int i = 0; i < C; i++)
                if (ch[(size_t)i]->phasing)
                    std::memcpy(cx.phasing_haps + (size_t)ch[(size_t)i]->sample * nL * T, hap + (size_t)i * nL * T, sizeof(double) * nL * (size_t)T);
        if (return_dosage && cur && !cur->chains.empty() && !cx.rc) {   // functions.R:999-1006 (rare + common: the all-SNP round counts)
            const double ta = now_s();
            const int n_cur = (int)cur->chains.size();
            std::vector<int32_t> cs((size_t)n_cur);
            for (int i = 0; i < n_cur; i++) cs[(size_t)i] = cur->chains[(size_t)i].sample - cur->lo;
            // (functions.R:1009-1016: the fetus = maternal transmitted + paternal transmitted)
            check(cx.be->accumulate_dosage(n_cur, nL, T, hap, cs.data(), cur->hi - cur->lo, cx.dosage + (size_t)cur->lo * T,
                                           cx.gp_t + (size_t)cur->lo * 3 * T, cx.nipt ? cx.nipt->fet_dosage + (size_t)cur->lo * T : nullptr,
                                           cx.nipt ? cx.nipt->fet_gp_t + (size_t)cur->lo * 3 * T : nullptr), "qa_accumulate_dosage");
            for (int i = 0; i < n_cur; i++) cx.nDosage[cur->chains[(size_t)i].sample] += 1;
            t_accumulate += now_s() - ta;
        }
        return return_dosage;
    }

    // impute_final_gibbs_with_rare_common (rare_common.R:109-420), once per Gibbs sample after its seek iterations
    // (functions.R:1042-1098): starting labels from the all-SNP reads against the latest (hap1, hap2) spread over all SNPs
    // (get_initial_read_labels, rare_common.R:61-107: 0.5 at the rare SNPs), then one Gibbs call over ALL SNPs with the
    // haplotypes selected last.  `ch`: the chains of the last round in its order (their dosages are rows of `dos`).
    void rare_common_round(std::vector<Chain *> &ch, Batch *cur) {
        const auto &P = cx.P;
        const int C = (int)ch.size(), T = cx.T, Ta = cx.T_out, nL = cx.nL;
        const double t0 = now_s();
        const double *last = dos.p;   // [chain][label][T] of the last seek iteration
        std::vector<int32_t> read_off((size_t)C + 1, 0);
        for (int i = 0; i < C; i++) read_off[(size_t)i + 1] = read_off[(size_t)i] + cx.reads_all[(size_t)ch[(size_t)i]->sample].R;
        double *lik = conf.get((size_t)read_off[(size_t)C] * nL);
        if (cx.be->make_eMatRead_t_rare_common) {
            // the device spreads the haplotypes over all SNPs itself (0.5 at the rare ones) and reads a sample's all-SNP reads
            // once for all of its chains: no 2.75 GB expansion, no seven copies of the reads (per launch set of 896 chains)
            std::vector<int32_t> samples_of;            // the distinct samples of the chains, in order of first appearance
            std::vector<int32_t> chain_sample((size_t)C);
            {
                std::map<int, int> at;
                for (int i = 0; i < C; i++) {
                    const int sm = ch[(size_t)i]->sample;
                    auto it = at.find(sm);
                    if (it == at.end()) { it = at.emplace(sm, (int)samples_of.size()).first; samples_of.push_back(sm); }
                    chain_sample[(size_t)i] = it->second;
                }
            }
            const int NS = (int)samples_of.size();
            std::vector<int32_t> s_off((size_t)NS + 1, 0);
            std::vector<int64_t> s_boff((size_t)NS + 1, 0);
            for (int i = 0; i < NS; i++) {
                const Reads &r = cx.reads_all[(size_t)samples_of[(size_t)i]];
                s_off[(size_t)i + 1] = s_off[(size_t)i] + r.R;
                s_boff[(size_t)i + 1] = s_boff[(size_t)i] + r.nb;
            }
            g_read_ptr.resize((size_t)s_off[(size_t)NS] + NS);
            g_u.resize((size_t)s_boff[(size_t)NS]);
            g_bq.resize((size_t)s_boff[(size_t)NS]);
            parallel_for((size_t)NS, n_help, [&](size_t i) {
                const Reads &r = cx.reads_all[(size_t)samples_of[i]];
                std::memcpy(&g_read_ptr[(size_t)s_off[i] + i], r.read_ptr, sizeof(int32_t) * ((size_t)r.R + 1));
                std::memcpy(&g_u[(size_t)s_boff[i]], r.u, sizeof(int32_t) * (size_t)r.nb);
                std::memcpy(&g_bq[(size_t)s_boff[i]], r.bq, sizeof(int32_t) * (size_t)r.nb);
            });
            check(cx.be->make_eMatRead_t_rare_common(handle, rc_handle, C, NS, chain_sample.data(), nL, last, s_off.data(), g_read_ptr.data(),
                                                     g_u.data(), g_bq.data(), P.maxDifferenceBetweenReads, 100, 1, lik),
                  "qa_rcpp_make_eMatRead_t_rare_common");
        } else {
            eh.resize((size_t)C * Ta * nL);
            parallel_for((size_t)C, n_help, [&](size_t c) {
                double *e = &eh[c * Ta * nL];
                for (size_t t = 0; t < (size_t)Ta * nL; t++) e[t] = 0.5;
                for (int l = 0; l < nL; l++) {
                    const double *h = last + (c * nL + l) * T;
                    for (int j = 0; j < T; j++) e[(size_t)cx.common_at[(size_t)j] * nL + l] = h[j];
                }
            });
            std::vector<int64_t> boff((size_t)C + 1, 0);
            for (int i = 0; i < C; i++) boff[(size_t)i + 1] = boff[(size_t)i] + cx.reads_all[(size_t)ch[(size_t)i]->sample].nb;
            g_read_ptr.resize((size_t)read_off[(size_t)C] + C);
            g_u.resize((size_t)boff[(size_t)C]);
            g_bq.resize((size_t)boff[(size_t)C]);
            parallel_for((size_t)C, n_help, [&](size_t i) {
                const Reads &r = cx.reads_all[(size_t)ch[i]->sample];
                std::memcpy(&g_read_ptr[(size_t)read_off[i] + i], r.read_ptr, sizeof(int32_t) * ((size_t)r.R + 1));
                std::memcpy(&g_u[(size_t)boff[i]], r.u, sizeof(int32_t) * (size_t)r.nb);
                std::memcpy(&g_bq[(size_t)boff[i]], r.bq, sizeof(int32_t) * (size_t)r.nb);
            });
            // rcpp_make_eMatRead_t as get_initial_read_labels calls it (rare_common.R:82-98): rescaled, Jmax = 100
            check(cx.be->make_eMatRead_t_nsnps(handle, Ta, C, nL, eh.data(), read_off.data(), g_read_ptr.data(), g_u.data(), g_bq.data(),
                                               P.maxDifferenceBetweenReads, 100, 1, lik), "qa_rcpp_make_eMatRead_t_nsnps");
        }
        std::vector<std::vector<int32_t>> starts((size_t)C);
        first_reads.assign((size_t)C, 0);
        seed_reads.assign((size_t)C, 0);
        seed_shards.assign((size_t)C, 0);
        parallel_for((size_t)C, n_draw, [&](size_t i) {   // (every chain draws from its own stream: order between chains is free)
            Chain &c = *ch[(size_t)i];
            const int R = cx.reads_all[(size_t)c.sample].R;
            const double *e = lik + (size_t)read_off[(size_t)i] * nL;
            starts[(size_t)i].resize((size_t)R);
            if (cx.nipt) {
                initial_read_labels_nipt(e, R, cx.nipt->ff[c.sample], c.rng, starts[(size_t)i]);
            } else {
                for (int r = 0; r < R; r++)   // H <- as.integer(runif(nReads) < e[1, ] / colSums(e)) + 1
                    starts[(size_t)i][(size_t)r] = (c.rng.uniform() < e[(size_t)r * 2] / (e[(size_t)r * 2] + e[(size_t)r * 2 + 1])) ? 2 : 1;
            }
            seed_reads[(size_t)i] = (uint64_t)c.rng.integers(0, 9223372036854775808.0);
            seed_shards[(size_t)i] = (uint64_t)c.rng.integers(0, 9223372036854775808.0);
        });
        const double t1 = now_s();
        t_host += t1 - t0;
        if (CallSpan::on()) std::fprintf(stderr, "[impute-trace] rc_prep thr %d n %d %.1f %.1f\n", w, C, t0 * 1e3, t1 * 1e3);
        double *hall = dos_all.get((size_t)C * nL * Ta);
        gibbs_with_retry(ch, starts, false, false, hall, true);
        const double t2 = now_s();
        t_gibbs += t2 - t1;
        for (int i = 0; i < C; i++)
            if (ch[(size_t)i]->phasing)
                std::memcpy(cx.phasing_haps + (size_t)ch[(size_t)i]->sample * nL * Ta, hall + (size_t)i * nL * Ta, sizeof(double) * nL * (size_t)Ta);
        if (cur && !cur->chains.empty()) {   // functions.R:1099-1123
            const int n_cur = (int)cur->chains.size();
            std::vector<int32_t> cs((size_t)n_cur);
            for (int i = 0; i < n_cur; i++) cs[(size_t)i] = cur->chains[(size_t)i].sample - cur->lo;
            check(cx.be->accumulate_dosage(n_cur, nL, Ta, hall, cs.data(), cur->hi - cur->lo, cx.dosage + (size_t)cur->lo * Ta,
                                           cx.gp_t + (size_t)cur->lo * 3 * Ta, cx.nipt ? cx.nipt->fet_dosage + (size_t)cur->lo * Ta : nullptr,
                                           cx.nipt ? cx.nipt->fet_gp_t + (size_t)cur->lo * 3 * Ta : nullptr), "qa_accumulate_dosage");
            for (int i = 0; i < n_cur; i++) cx.nDosage[cur->chains[(size_t)i].sample] += 1;
        }
        t_accumulate += now_s() - t2;
        if (CallSpan::on()) std::fprintf(stderr, "[impute-trace] rc_post thr %d n %d %.1f %.1f\n", w, C, t2 * 1e3, now_s() * 1e3);
    }

    // the reads of one sample as the caller describes them: the checks of the flat form
    static void check_reads(const Reads &r, int s, const char *what) {
        if (r.R < 1) throw Failure(QA_ERR_INVALID, "sample " + std::to_string(s) + " has no " + what + "reads (the reference drops such samples before imputing, functions.R:300-310)");
        if (!r.read_ptr || !r.u || !r.bq || !r.wif) throw Failure(QA_ERR_INVALID, std::string("sample ") + std::to_string(s) + ": missing " + what + "read arrays");
        if (r.read_ptr[0] != 0) throw Failure(QA_ERR_INVALID, std::string(what) + "read_ptr of sample " + std::to_string(s) + " does not start at 0");
    }

    // params->sample_source: samples [lo, hi) from the caller, in order; returns where the range ends (hi, or the first s the
    // source reports as beyond its last sample)
    int acquire_samples(int lo, int hi) {
        for (int s = lo; s < hi; s++) {
            qa_sample_view_t v{};
            const int st = cx.source->acquire(cx.source->ctx, s, &v);
            if (st == QA_END_OF_SAMPLES) return s;
            if (st != QA_OK) throw Failure(st < 0 ? st : QA_ERR_INVALID, std::string("the sample source failed at sample ") + std::to_string(s) + ": " + qa_last_error());
            Reads &r = cx.reads[(size_t)s];
            r.R = v.n_reads; r.read_ptr = v.read_ptr; r.u = v.u; r.bq = v.bq; r.wif = v.wif;
            check_reads(r, s, "");
            r.nb = r.read_ptr[r.R];
            if (!v.read_labels) throw Failure(QA_ERR_INVALID, "the sample source gave sample " + std::to_string(s) + " no place for its read labels");
            cx.label_dst[(size_t)s] = v.read_labels;
            if (cx.rc) {
                Reads &a = cx.reads_all[(size_t)s];
                a.R = v.n_reads_all; a.read_ptr = v.read_ptr_all; a.u = v.u_all; a.bq = v.bq_all; a.wif = v.wif_all;
                check_reads(a, s, "all-SNP ");
                a.nb = a.read_ptr[a.R];
            }
            if (cx.nipt && !(cx.nipt->ff[s] > 0.0 && cx.nipt->ff[s] < 1.0))
                throw Failure(QA_ERR_INVALID, "fetal fraction of sample " + std::to_string(s) + " outside (0, 1)");
        }
        return hi;
    }

    Batch *new_batch(int lo, int hi) {
        if (cx.source) {
            hi = acquire_samples(lo, hi);
            if (hi <= lo) return nullptr;   // the range ended before this set
        }
        Batch *b = new Batch;
        b->lo = lo;
        b->hi = hi;
        const auto &P = cx.P;
        // the set's rows of the accumulators start at zero (here, by the thread that takes the set, beside the other threads'
        // device phases -- zeroing the whole range's 48 bytes per sample and SNP before the first launch kept the device waiting
        // for a second at 2 560 samples); phasing_haps and read_labels are written whole
        {
            const size_t To = (size_t)cx.T_out;
            parallel_for((size_t)(hi - lo), n_help, [&](size_t si) {
                const size_t s2 = (size_t)lo + si;
                std::memset(cx.dosage + s2 * To, 0, sizeof(double) * To);
                std::memset(cx.gp_t + s2 * 3 * To, 0, sizeof(double) * 3 * To);
                if (cx.nipt) {
                    std::memset(cx.nipt->fet_dosage + s2 * To, 0, sizeof(double) * To);
                    std::memset(cx.nipt->fet_gp_t + s2 * 3 * To, 0, sizeof(double) * 3 * To);
                }
                cx.nDosage[s2] = 0;
            });
        }
        b->chains.reserve((size_t)(hi - lo) * P.nGibbsSamples);
        for (int s = lo; s < hi; s++)
            for (int c = 1; c <= P.nGibbsSamples; c++) {
                Chain ch;
                ch.sample = s;
                ch.i_chain = c;
                ch.rng = ChainStream(P.seed, cx.global_index(s), c);
                b->chains.push_back(std::move(ch));
            }
        return b;
    }

    // read confidence per chain and consensus labels (functions.R:1144-1205); one phasing chain per sample.  `hap`: the last
    // round's dosages, the batch's main chains first.
    void start_phasing(Batch &b, const double *hap) {
        const auto &P = cx.P;
        const int n = (int)b.chains.size(), T = cx.T, nL = cx.nL;
        std::vector<int32_t> read_off((size_t)n + 1, 0);
        std::vector<int64_t> boff((size_t)n + 1, 0);
        for (int i = 0; i < n; i++) {
            const Reads &r = cx.reads[(size_t)b.chains[(size_t)i].sample];
            read_off[(size_t)i + 1] = read_off[(size_t)i] + r.R;
            boff[(size_t)i + 1] = boff[(size_t)i] + r.nb;
        }
        g_read_ptr.resize((size_t)read_off[(size_t)n] + n);
        g_u.resize((size_t)boff[(size_t)n]);
        g_bq.resize((size_t)boff[(size_t)n]);
        parallel_for((size_t)n, n_help, [&](size_t i) {
            const Reads &r = cx.reads[(size_t)b.chains[i].sample];
            std::memcpy(&g_read_ptr[(size_t)read_off[i] + i], r.read_ptr, sizeof(int32_t) * ((size_t)r.R + 1));
            std::memcpy(&g_u[(size_t)boff[i]], r.u, sizeof(int32_t) * (size_t)r.nb);
            std::memcpy(&g_bq[(size_t)boff[i]], r.bq, sizeof(int32_t) * (size_t)r.nb);
        });
        double *e = conf.get((size_t)read_off[(size_t)n] * nL);
        // calculate_eMatRead_t_vs_haplotypes (functions.R:2975-3020): not rescaled, Jmax = 1000
        CallSpan span_e("ematread_conf", w, n);
        check(cx.be->make_eMatRead_t_hap_major(handle, T, n, nL, hap, read_off.data(), g_read_ptr.data(), g_u.data(), g_bq.data(),
                                               P.maxDifferenceBetweenReads, 1000, 0, e), "qa_rcpp_make_eMatRead_t_hap_major");
        span_e.end();
        b.phasing.clear();
        b.phasing.resize((size_t)(b.hi - b.lo));
        const int nG = P.nGibbsSamples;
        parallel_for((size_t)(b.hi - b.lo), n_help, [&](size_t si) {
            const int s = b.lo + (int)si;
            const int R = cx.reads[(size_t)s].R;
            std::vector<int32_t> labels((size_t)nG * R);
            std::vector<double> p((size_t)nG * nL * R);
            for (int c = 0; c < nG; c++) {
                const size_t k = si * nG + c;   // chains are sample-major
                std::memcpy(&labels[(size_t)c * R], b.chains[k].labels.data(), sizeof(int32_t) * (size_t)R);
                const double *ek = e + (size_t)read_off[k] * nL;   // [read][label]
                for (int r = 0; r < R; r++)
                    for (int l = 0; l < nL; l++) p[((size_t)c * nL + l) * R + r] = ek[(size_t)r * nL + l];
            }
            Chain ph;
            ph.sample = s;
            ph.i_chain = nG + 1;
            ph.phasing = true;
            ph.rng = ChainStream(P.seed, cx.global_index(s), nG + 1);
            ph.which = b.chains[si * nG + (nG - 1)].which;
            ph.labels.resize((size_t)R);
            if (cx.be->consensus_read_labels(R, nG, labels.data(), p.data(), nL, 0.95, nG, ph.labels.data()) != QA_OK)
                throw Failure(QA_ERR_INVALID, "qa_consensus_read_labels failed");
            std::memcpy(cx.label_dst[(size_t)s], ph.labels.data(), sizeof(int32_t) * (size_t)R);
            b.phasing[si] = std::move(ph);
        });
        b.chains.clear();
        b.chains.shrink_to_fit();
    }

    void finish(Batch &b) {
        const int T = cx.T_out;   // (impute_rare_common: the accumulators and the phasing haplotypes cover all SNPs)
        parallel_for((size_t)(b.hi - b.lo), n_help, [&](size_t si) {
            const int s = b.lo + (int)si;
            const double n = (double)cx.nDosage[s];
            double *d = cx.dosage + (size_t)s * T, *g = cx.gp_t + (size_t)s * 3 * T;
            for (int t = 0; t < T; t++) d[t] /= n;
            for (int t = 0; t < 3 * T; t++) g[t] /= n;
            if (cx.nipt) {   // functions.R:1218-1231, 1313-1317
                double *fd = cx.nipt->fet_dosage + (size_t)s * T, *fg = cx.nipt->fet_gp_t + (size_t)s * 3 * T;
                for (int t = 0; t < T; t++) fd[t] /= n;
                for (int t = 0; t < 3 * T; t++) fg[t] /= n;
                double *h = cx.phasing_haps + (size_t)s * 3 * T;
                recast_nipt_haps(h, h + T, h + 2 * (size_t)T, g, fg, T);
                return;
            }
            double *h = cx.phasing_haps + (size_t)s * 2 * T;
            recast_haps(h, h + T, g, g + T, g + 2 * (size_t)T, T);
        });
        if (cx.P.on_samples_done) cx.P.on_samples_done(cx.P.on_samples_done_ctx, b.lo, b.hi);   // every row of [lo, hi) is final
    }

    // quilt_amd/driver.py::Driver.run_stream
    void run_stream(const std::vector<std::pair<int, int>> &sets) {
        const auto &P = cx.P;
        std::unique_ptr<Batch> prev;
        size_t at = 0;
        bool reported = false;
        std::vector<Batch *> taken;
        try {
            while (true) {
                std::unique_ptr<Batch> cur;
                if (at < sets.size()) {
                    cur.reset(new_batch(sets[at].first, sets[at].second));
                    at = cur ? at + 1 : sets.size();   // (a sample source's range ended: the later sets are beyond it too)
                }
                taken.clear();
                if (!cur && cx.use_tail && !reported) {
                    reported = true;
                    Tail &tl = cx.tail;
                    std::unique_lock<std::mutex> lk(tl.mu);
                    if (tl.failed) std::rethrow_exception(tl.failed);
                    tl.n_active -= 1;
                    if (tl.n_active <= 0) {
                        taken.swap(tl.waiting);
                    } else if (prev) {
                        tl.waiting.push_back(prev.get());
                        Batch *mine = prev.get();
                        tl.cv.wait(lk, [&] { return mine->done_flag; });
                        if (mine->failed) std::rethrow_exception(mine->failed);
                        lk.unlock();
                        const double tf = now_s();
                        finish(*mine);
                        t_finish += now_s() - tf;
                        return;
                    } else {
                        return;
                    }
                }
                if (!cur && !prev && taken.empty()) return;
                std::vector<Chain *> phasing;
                if (prev) for (auto &c : prev->phasing) phasing.push_back(&c);
                for (Batch *b : taken) for (auto &c : b->phasing) phasing.push_back(&c);
                for (int i_it = 1; i_it <= P.n_seek_its; i_it++) {
                    std::vector<Chain *> ch;
                    if (cur) for (auto &c : cur->chains) ch.push_back(&c);
                    ch.insert(ch.end(), phasing.begin(), phasing.end());
                    round(ch, i_it, cur.get());
                }
                if (cx.rc) {   // functions.R:1042-1123: every Gibbs sample (and the phasing iteration) ends with the all-SNP call
                    std::vector<Chain *> ch;
                    if (cur) for (auto &c : cur->chains) ch.push_back(&c);
                    ch.insert(ch.end(), phasing.begin(), phasing.end());
                    rare_common_round(ch, cur.get());
                }
                if (!taken.empty()) {
                    std::lock_guard<std::mutex> g(cx.tail.mu);
                    for (Batch *b : taken) b->done_flag = true;
                    cx.tail.cv.notify_all();
                    taken.clear();
                }
                const double t0 = now_s();
                if (prev) finish(*prev);
                const double t1 = now_s();
                if (cur) start_phasing(*cur, dos.p);
                t_finish += t1 - t0;
                t_consensus += now_s() - t1;
                prev = std::move(cur);
            }
        } catch (...) {
            std::exception_ptr e = std::current_exception();
            if (cx.use_tail) {
                {
                    std::lock_guard<std::mutex> g(cx.tail.mu);
                    for (Batch *b : taken) { b->failed = e; b->done_flag = true; }
                    cx.tail.cv.notify_all();
                }
                cx.tail.abort(e);
                if (!reported) {
                    std::lock_guard<std::mutex> g(cx.tail.mu);
                    cx.tail.n_active -= 1;
                }
            }
            throw;
        }
    }
};

// quilt_amd/sharding.py::get_sample_range (STITCH getSampleRange semantics, quilt.R:691): n items in `parts` contiguous ranges
std::vector<std::pair<int, int>> sample_ranges(int n, int parts) {
    std::vector<std::pair<int, int>> out;
    const int base = n / parts, rem = n % parts;
    int at = 0;
    for (int w = 0; w < parts; w++) {
        const int len = base + (w < rem ? 1 : 0);
        out.push_back({at, at + len});
        at += len;
    }
    return out;
}

int impute_impl(bool keep_buffers, const qa_impute_backend_t *be, void *const *handles, int32_t n_handles, int32_t K, int32_t G, int32_t T,
                const qa_impute_params_t *params, int32_t n_sample, int64_t sample_offset, const int32_t *read_off,
                const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *wif, double *dosage, double *gp_t,
                double *phasing_haps, int32_t *read_labels, int32_t *nDosage, int64_t *stats) {
    const bool flat = !(params && params->sample_source);
    if (!be || !handles || n_handles < 1 || n_handles > 16 || !params || n_sample < 0 ||
        (flat && (!read_off || !read_ptr || !u || !bq || !wif || !read_labels)) || (!flat && !params->sample_source->acquire) ||
        !dosage || !gp_t || !phasing_haps || !nDosage || K < 1 || G < 1 || T < 1) {
        qa::set_error("qa_impute_samples: missing argument");
        return QA_ERR_INVALID;
    }
    Ctx cx;
    cx.be = be;
    cx.product = keep_buffers;
    cx.K = K; cx.G = G; cx.T = T;
    cx.P = *params;
    auto &P = cx.P;
    // QUILT()'s argument handling (quilt.R:248-250, :453-471)
    if (P.n_burn_in_seek_its < 0) P.n_burn_in_seek_its = P.n_seek_its - 1;
    if (K < P.Ksubset) { P.n_seek_its = 1; P.n_burn_in_seek_its = 0; P.Ksubset = K; P.Knew = K; }
    if (P.Knew > P.Ksubset) P.Knew = P.Ksubset;
    if (P.n_seek_its < 1 || P.n_burn_in_seek_its < 0 || P.n_burn_in_seek_its >= P.n_seek_its || P.nGibbsSamples < 1 || P.Ksubset < 1 ||
        P.Knew < 1 || P.K_top_matches < 1 || P.n_block_gibbs_iterations < 0 ||
        (P.n_block_gibbs_iterations > 0 && !P.small_ref_panel_block_gibbs_iterations)) {
        qa::set_error("qa_impute_samples: n_seek_its >= 1, 0 <= n_burn_in_seek_its < n_seek_its, nGibbsSamples / Ksubset / Knew / "
                      "K_top_matches >= 1");
        return QA_ERR_INVALID;
    }
    if (P.use_mspbwt && (!P.mspbwt_index || P.Knew != P.Ksubset || P.mspbwtL < 1 || P.mspbwtL > 64 || P.mspbwtM < 1)) {
        qa::set_error("qa_impute_samples: use_mspbwt needs the panel's index (qa_mspbwt_create), Knew == Ksubset, 1 <= mspbwtL <= 64, mspbwtM >= 1");
        return QA_ERR_INVALID;
    }
    cx.n_burn = P.n_burn_in_seek_its;
    cx.blocks.assign(P.small_ref_panel_block_gibbs_iterations, P.small_ref_panel_block_gibbs_iterations + P.n_block_gibbs_iterations);
    cx.cols = thinned_grid_columns(G, P.heuristic_match_thin);
    cx.n_thin = 0;
    for (int32_t c : cx.cols) cx.n_thin += c >= 0;
    cx.top_width = std::max(8, P.K_top_matches);
    cx.sample_offset = sample_offset;
    cx.sample_index = P.sample_index;
    cx.dosage = dosage; cx.gp_t = gp_t; cx.phasing_haps = phasing_haps; cx.read_labels = read_labels; cx.nDosage = nDosage;
    cx.read_off = read_off;
    cx.source = P.sample_source;
    cx.reads.resize((size_t)n_sample);
    cx.label_dst.assign((size_t)n_sample, nullptr);
    if (flat) {
        int64_t base = 0;
        for (int s = 0; s < n_sample; s++) {
            cx.label_dst[(size_t)s] = read_labels + read_off[s];
            Reads &r = cx.reads[(size_t)s];
            r.R = read_off[s + 1] - read_off[s];
            if (r.R < 1) {
                qa::set_error("qa_impute_samples: sample %d has no reads (the reference drops such samples before imputing, functions.R:300-310)", s);
                return QA_ERR_INVALID;
            }
            r.read_ptr = read_ptr + read_off[s] + s;
            if (r.read_ptr[0] != 0) { qa::set_error("qa_impute_samples: read_ptr of sample %d does not start at 0", s); return QA_ERR_INVALID; }
            r.nb = r.read_ptr[r.R];
            r.u = u + base;
            r.bq = bq + base;
            r.wif = wif + read_off[s];
            base += r.nb;
        }
    }
    cx.T_out = T;
    if (P.nipt) {
        if (P.rare_common && !P.rare_common->L_grid_all) {
            qa::set_error("qa_impute_samples: method = \"nipt\" with impute_rare_common needs rare_common->L_grid_all");
            return QA_ERR_INVALID;
        }
        if (!P.nipt->ff || !P.nipt->L_grid || !P.nipt->fet_dosage || !P.nipt->fet_gp_t) {
            qa::set_error("qa_impute_samples: method = \"nipt\" needs ff, L_grid and the fetus' output arrays");
            return QA_ERR_INVALID;
        }
        for (int s2 = 0; flat && s2 < n_sample; s2++)   // (a sample source: checked as the samples arrive)
            if (!(P.nipt->ff[s2] > 0.0 && P.nipt->ff[s2] < 1.0)) { qa::set_error("qa_impute_samples: fetal fraction of sample %d outside (0, 1)", s2); return QA_ERR_INVALID; }
        cx.nipt = P.nipt;
        cx.nL = 3;
    }
    if (P.rare_common) {
        const qa_impute_rare_common_t &rc = *P.rare_common;
        if (!rc.handles || rc.nSNPs_all < T || rc.nGrids_all != (rc.nSNPs_all + 31) / 32 || !rc.snp_is_common ||
            (flat && (!rc.read_off || !rc.read_ptr || !rc.u || !rc.bq || !rc.wif)) || !be->gibbs_batch_rare_common || !be->make_eMatRead_t_nsnps) {
            qa::set_error("qa_impute_samples: impute_rare_common needs the all-SNP handles, dimensions, flags and reads");
            return QA_ERR_INVALID;
        }
        for (int i = 0; i < n_handles; i++)
            if (!rc.handles[i]) { qa::set_error("qa_impute_samples: one qa_rare_common_t per panel handle"); return QA_ERR_INVALID; }
        cx.rc = &rc;
        cx.T_out = rc.nSNPs_all;
        for (int t = 0; t < rc.nSNPs_all; t++)
            if (rc.snp_is_common[t]) cx.common_at.push_back(t);
        if ((int)cx.common_at.size() != T) {
            qa::set_error("qa_impute_samples: snp_is_common marks %d SNPs, the panel has %d", (int)cx.common_at.size(), T);
            return QA_ERR_INVALID;
        }
        cx.reads_all.resize((size_t)n_sample);
        int64_t base = 0;
        for (int s = 0; flat && s < n_sample; s++) {
            Reads &r = cx.reads_all[(size_t)s];
            r.R = rc.read_off[s + 1] - rc.read_off[s];
            if (r.R < 1) { qa::set_error("qa_impute_samples: sample %d has no all-SNP reads", s); return QA_ERR_INVALID; }
            r.read_ptr = rc.read_ptr + rc.read_off[s] + s;
            if (r.read_ptr[0] != 0) { qa::set_error("qa_impute_samples: all-SNP read_ptr of sample %d does not start at 0", s); return QA_ERR_INVALID; }
            r.nb = r.read_ptr[r.R];
            r.u = rc.u + base;
            r.bq = rc.bq + base;
            r.wif = rc.wif + rc.read_off[s];
            base += r.nb;
        }
    }
    if (n_sample == 0) return QA_OK;   // (the accumulators are zeroed set by set: Worker::new_batch)

    // ---- the plan: launch sets of `per_set` samples; whole sets to the threads in turn, the left-overs cut across them
    const int per_set = P.samples_per_launch_set > 0 ? P.samples_per_launch_set : 256;
    std::vector<std::pair<int, int>> sets;
    for (int lo = 0; lo < n_sample; lo += per_set) sets.push_back({lo, std::min(n_sample, lo + per_set)});
    const int W = n_handles;
    // Sets the thread count leaves over: a Gibbs launch costs nearly the same between 700 and 2 048 chains (a chain's serial
    // time), so a set cut into W parts costs W launches per round where it would cost one -- the left-over sets go WHOLE to the
    // first threads (the others drain, leave their last set's phasing rounds at the meeting point, and the thread that drains
    // last runs all of them together).  Only a call with fewer sets than threads is cut across them (QA_IMPUTE_CUT_LEFTOVERS=1:
    // always, the form of round 3).
    static const bool cut_leftovers = [] { const char *e = std::getenv("QA_IMPUTE_CUT_LEFTOVERS"); return e && e[0] == '1'; }();
    size_t n_whole = (cut_leftovers || sets.size() < (size_t)W) ? sets.size() / W * W : sets.size();
    std::vector<std::vector<std::pair<int, int>>> streams((size_t)W);
    for (size_t i = 0; i < n_whole; i++) streams[i % W].push_back(sets[i]);
    if (n_whole < sets.size()) {
        const int lo = sets[n_whole].first, hi = n_sample;
        auto parts = sample_ranges(hi - lo, W);
        for (int w2 = 0; w2 < W; w2++)
            if (parts[(size_t)w2].second > parts[(size_t)w2].first)
                streams[(size_t)w2].push_back({lo + parts[(size_t)w2].first, lo + parts[(size_t)w2].second});
    }
    cx.use_tail = W > 1 && !P.no_fused_tails;
    cx.tail.n_active = W;

    std::vector<std::unique_ptr<Worker>> workers;
    for (int w2 = 0; w2 < W; w2++) {
        workers.emplace_back(new Worker(cx, handles[w2], w2, keep_buffers));
        if (cx.rc) workers.back()->rc_handle = cx.rc->handles[w2];
    }
    // staggered start: thread w prepares its first launch once thread w - 1 has handed its own to the device
    struct Started { std::mutex mu; std::condition_variable cv; bool set = false; };
    std::vector<Started> started((size_t)W);
    auto signal = [&](int w2) {
        std::lock_guard<std::mutex> g(started[(size_t)w2].mu);
        started[(size_t)w2].set = true;
        started[(size_t)w2].cv.notify_all();
    };
    std::vector<std::exception_ptr> errs((size_t)W);
    std::vector<std::string> err_text((size_t)W);
    std::vector<int> err_status((size_t)W, 0);
    std::vector<int> err_order((size_t)W, -1);   // the order in which the threads failed: the report names the FIRST failure
    std::atomic<int> n_failed{0};
    std::atomic<bool> stagger_timed_out{false};
    auto body = [&](int w2) {
        try {
            if (w2 > 0) {
                std::unique_lock<std::mutex> lk(started[(size_t)w2 - 1].mu);
                // (staggered start: a launch plan, not a dependency -- a predecessor whose first launch takes longer than this is
                // not waited for; results do not depend on it, QA_TIMING says when it happened)
                // (wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer follows; wait_for's
                // pthread_cond_clockwait it does not -- tests/test_tsan_cpu.py.  A clock step only moves this 5 s plan.)
                if (!started[(size_t)w2 - 1].cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds(5),
                                                           [&] { return started[(size_t)w2 - 1].set; }))
                    stagger_timed_out.store(true);
            }
            workers[(size_t)w2]->on_first_launch = [&, w2] { signal(w2); };
            if (be->bind_thread) be->bind_thread(handles[w2]);
            workers[(size_t)w2]->run_stream(streams[(size_t)w2]);
        } catch (const Failure &f) {
            errs[(size_t)w2] = std::current_exception();
            err_order[(size_t)w2] = n_failed.fetch_add(1);
            err_text[(size_t)w2] = f.what();
            err_status[(size_t)w2] = f.status;
        } catch (const std::exception &e) {
            errs[(size_t)w2] = std::current_exception();
            err_order[(size_t)w2] = n_failed.fetch_add(1);
            err_text[(size_t)w2] = e.what();
            err_status[(size_t)w2] = QA_ERR_HIP;
        } catch (...) {
            errs[(size_t)w2] = std::current_exception();
            err_order[(size_t)w2] = n_failed.fetch_add(1);
            err_text[(size_t)w2] = "unknown failure";
            err_status[(size_t)w2] = QA_ERR_HIP;
        }
        signal(w2);
    };
    if (W == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        for (int w2 = 0; w2 < W; w2++) th.emplace_back(body, w2);
        for (auto &t : th) t.join();
    }
    if (stats) {
        stats[0] = cx.n_underflow_retries; stats[1] = cx.n_full_list_refetches; stats[2] = cx.n_device_selections;
        stats[3] = cx.n_gibbs_chain_calls; stats[4] = cx.n_gibbs_launches;
        double tg = 0, tf = 0, th2 = 0, tc = 0, tfi = 0, ta = 0;
        for (auto &wk : workers) { tg += wk->t_gibbs; tf += wk->t_fullpass; th2 += wk->t_host; tc += wk->t_consensus; tfi += wk->t_finish; ta += wk->t_accumulate; }
        stats[5] = (int64_t)(tg * 1e3); stats[6] = (int64_t)(tf * 1e3); stats[7] = (int64_t)(th2 * 1e3); stats[8] = (int64_t)(tc * 1e3);
        stats[9] = (int64_t)(tfi * 1e3); stats[10] = (int64_t)(ta * 1e3);
    }
    if (stagger_timed_out.load() && std::getenv("QA_TIMING"))
        std::fprintf(stderr, "[qa_impute_samples] a host thread's first launch took more than 5 s: the staggered start was skipped for its successor\n");
    {
        // the failure that happened FIRST (a thread that only rethrows a failed tail's exception carries that same exception:
        // text and status are the originator's either way)
        int first = -1;
        for (int w2 = 0; w2 < W; w2++)
            if (errs[(size_t)w2] && (first < 0 || err_order[(size_t)w2] < err_order[(size_t)first])) first = w2;
        if (first >= 0) {
            qa::set_error("qa_impute_samples: %s", err_text[(size_t)first].c_str());
            return err_status[(size_t)first] ? err_status[(size_t)first] : QA_ERR_HIP;
        }
    }
    return QA_OK;
}

// ---- the product's table: the library's own entry points
int be_gibbs(void *h, const qa_gibbs_opts_t *o, int32_t n, const int32_t *which, const int32_t *read_off, const int32_t *read_ptr,
             const int32_t *u, const int32_t *bq, const int32_t *wif, const double *ru, const int32_t *fr, const double *rs, int32_t *H,
             int32_t *Hc, double *hp, double *gm, double *gf, int32_t *uf, double *state, const uint64_t *sr, const uint64_t *ss) {
    return qa_gibbs_batch(static_cast<qa_panel_t *>(h), o, n, which, read_off, read_ptr, u, bq, wif, ru, fr, rs, H, Hc, hp, gm, gf, uf,
                          state, sr, ss);
}
int be_fullpass_select(void *h, int32_t n_chain, int32_t n_label, int32_t n_sample, const int32_t *cs, const int32_t *read_off,
                       const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *H, const int32_t *wd,
                       const int32_t *wt, const int32_t *cols, int32_t Ktop, double minGL, double *dosage, int32_t top_width,
                       int32_t *top_idx, float *top_val, int32_t *top_cnt, int32_t Ksubset, int32_t Knew, const int32_t *which,
                       const uint64_t *seed, int32_t *which_next, int32_t *status) {
    return qa_fullpass_reads_select_batch(static_cast<qa_panel_t *>(h), n_chain, n_label, n_sample, cs, read_off, read_ptr, u, bq, H, wd,
                                          wt, cols, Ktop, minGL, dosage, top_width, top_idx, top_val, top_cnt, Ksubset, Knew, which,
                                          seed, which_next, status);
}
int be_fullpass(void *h, int32_t n_pass, const double *gl, const int32_t *wd, const int32_t *cols, int32_t Ktop, double *dosage,
                int32_t *bptr, int32_t *bidx, double *bval, int64_t cap) {
    return qa_fullpass_batch(static_cast<qa_panel_t *>(h), n_pass, gl, wd, cols, Ktop, dosage, bptr, bidx, bval, cap);
}
int be_ematread(void *h, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps, const int32_t *read_off, const int32_t *read_ptr,
                const int32_t *u, const int32_t *bq, double maxdiff, int32_t Jmax, int32_t rescale, double *out) {
    return qa_rcpp_make_eMatRead_t_hap_major(static_cast<qa_panel_t *>(h), nSNPs, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxdiff,
                                             Jmax, rescale, out);
}
void be_bind(void *h) { (void)qa_panel_bind_thread(static_cast<qa_panel_t *>(h)); }
int be_gibbs_rc(void *h, const void *rc, const qa_gibbs_opts_t *o, int32_t n, const int32_t *which, const int32_t *read_off,
                const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *wif, const double *ru, const int32_t *fr,
                const double *rs, int32_t *H, int32_t *Hc, double *hp, double *gm, double *gf, int32_t *uf, double *state,
                const uint64_t *sr, const uint64_t *ss) {
    return qa_gibbs_batch_rare_common(static_cast<qa_panel_t *>(h), static_cast<const qa_rare_common_t *>(rc), o, n, which, read_off,
                                      read_ptr, u, bq, wif, ru, fr, rs, H, Hc, hp, gm, gf, uf, state, sr, ss);
}
int be_ematread_nsnps(void *h, int32_t nSNPs, int32_t n_chain, int32_t K, const double *eHaps, const int32_t *read_off,
                      const int32_t *read_ptr, const int32_t *u, const int32_t *bq, double maxdiff, int32_t Jmax, int32_t rescale,
                      double *out) {
    return qa_rcpp_make_eMatRead_t_nsnps(static_cast<qa_panel_t *>(h), nSNPs, n_chain, K, eHaps, read_off, read_ptr, u, bq, maxdiff,
                                         Jmax, rescale, out);
}

int be_ematread_rc(void *h, const void *rc, int32_t n_chain, int32_t n_sample, const int32_t *chain_sample, int32_t K,
                   const double *hap_common, const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                   double maxdiff, int32_t Jmax, int32_t rescale, double *out) {
    return qa_rcpp_make_eMatRead_t_rare_common(static_cast<qa_panel_t *>(h), static_cast<const qa_rare_common_t *>(rc), n_chain, n_sample,
                                               chain_sample, K, hap_common, read_off, read_ptr, u, bq, maxdiff, Jmax, rescale, out);
}

const qa_impute_backend_t kProduct = {be_gibbs, be_fullpass_select, be_fullpass, be_ematread, qa_mspbwt_select_new_haps,
                                      qa_accumulate_dosage, qa_consensus_read_labels, qa_host_alloc, qa_host_free, be_bind,
                                      be_gibbs_rc, be_ematread_nsnps, be_ematread_rc};

}   // namespace

extern "C" {

int qa_impute_params_default(qa_impute_params_t *p) {
    if (!p) return QA_ERR_INVALID;
    static const int32_t kBlocks[3] = {3, 6, 9};   // small_ref_panel_block_gibbs_iterations (quilt.R:150), 0-based sweeps
    *p = qa_impute_params_t{};
    p->nGibbsSamples = 7;
    p->n_seek_its = 3;
    p->n_burn_in_seek_its = -1;
    p->Ksubset = 600;
    p->Knew = 600;
    p->K_top_matches = 5;
    p->heuristic_match_thin = 0.1;
    p->small_ref_panel_gibbs_iterations = 20;
    p->n_gibbs_sample_its = 1;
    p->small_ref_panel_block_gibbs_iterations = kBlocks;
    p->n_block_gibbs_iterations = 3;
    p->maxDifferenceBetweenReads = 1e10;
    p->minGLValue = 1e-10;
    p->Jmax = 10000;
    p->seed = 1;
    p->mspbwtL = 3;
    p->mspbwtM = 1;
    return QA_OK;
}

int qa_impute_samples(qa_panel_t *const *panels, int32_t n_panels, const qa_impute_params_t *params, int32_t n_sample,
                      int64_t sample_offset, const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq,
                      const int32_t *wif, double *dosage, double *gp_t, double *phasing_haps, int32_t *read_labels, int32_t *nDosage,
                      int64_t *stats) {
    if (!panels || n_panels < 1 || !panels[0]) {
        qa::set_error("qa_impute_samples: no panel handle");
        return QA_ERR_INVALID;
    }
    if (qa_device_count() < 1) {
        qa::set_error("no usable gfx950 (MI355X) device: libquilt_amd has no CPU fallback");
        return QA_ERR_NO_DEVICE;
    }
    int32_t K = 0, G = 0, T = 0;
    if (qa_panel_get_dims(panels[0], &K, &G, &T) != QA_OK) return QA_ERR_INVALID;
    for (int i = 1; i < n_panels; i++) {
        int32_t k2 = 0, g2 = 0, t2 = 0;
        if (!panels[i] || qa_panel_get_dims(panels[i], &k2, &g2, &t2) != QA_OK || k2 != K || g2 != G || t2 != T) {
            qa::set_error("qa_impute_samples: the handles must be replicas of one panel");
            return QA_ERR_INVALID;
        }
    }
    for (int i = 0; i < n_panels; i++)
        for (int j = 0; j < i; j++)
            if (panels[i] == panels[j]) {
                qa::set_error("qa_impute_samples: every host thread needs a handle of its own");
                return QA_ERR_INVALID;
            }
    return impute_impl(true, &kProduct, reinterpret_cast<void *const *>(panels), n_panels, K, G, T, params, n_sample, sample_offset, read_off,
                       read_ptr, u, bq, wif, dosage, gp_t, phasing_haps, read_labels, nDosage, stats);
}

int qa_impute_release_buffers(void) {
    std::lock_guard<std::mutex> g(buf_mu());
    bufs().clear();
    return QA_OK;
}

int qa_impute_kept_buffers(void) {
    std::lock_guard<std::mutex> g(buf_mu());
    return (int)bufs().size();
}

// qa_panel_destroy's hook (panel.hip): the host thread's buffers kept for this handle go with it -- a caller that creates and
// destroys its handles per call (the shim's qa_impute_sample_range) would otherwise leak gigabytes of pinned memory per call,
// and a later handle allocated at the same address would inherit the stale entry.
extern "C" void qa_impute_drop_handle_buffers(void *handle) {
    std::lock_guard<std::mutex> g(buf_mu());
    bufs().erase(handle);
}

int qa_impute_samples_backend(const qa_impute_backend_t *backend, void *const *handles, int32_t n_handles, int32_t K, int32_t nGrids,
                              int32_t nSNPs, const qa_impute_params_t *params, int32_t n_sample, int64_t sample_offset,
                              const int32_t *read_off, const int32_t *read_ptr, const int32_t *u, const int32_t *bq, const int32_t *wif,
                              double *dosage, double *gp_t, double *phasing_haps, int32_t *read_labels, int32_t *nDosage,
                              int64_t *stats) {
    if (!backend || !backend->gibbs_batch || !backend->fullpass_reads_select_batch || !backend->fullpass_batch ||
        !backend->make_eMatRead_t_hap_major || !backend->mspbwt_select_new_haps || !backend->accumulate_dosage ||
        !backend->consensus_read_labels || !backend->host_alloc || !backend->host_free) {
        qa::set_error("qa_impute_samples_backend: incomplete table");
        return QA_ERR_INVALID;
    }
    return impute_impl(false, backend, handles, n_handles, K, nGrids, nSNPs, params, n_sample, sample_offset, read_off, read_ptr, u, bq, wif,
                       dosage, gp_t, phasing_haps, read_labels, nDosage, stats);
}

}   // extern "C"
