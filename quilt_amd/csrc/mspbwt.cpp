// mspbwt.cpp -- the panel's multi-symbol positional BWT indices and the neighbour scan of the msPBWT mode
// (use_mspbwt = TRUE, QUILT2's default: QUILT2.R:456, :613).  Host code, as BASELINE's north star
// asks ("mspbwt haplotype-subset selection ... stay on the host"): threads over (query, index) pairs.
//
// What it stands in for: `mspbwt::ms_BuildIndices_Algorithm5` (the per-index tables `all_symbols`, `usge_all`, `egs` that
// quilt-prepare-reference stores) and `mspbwt::Rcpp_find_good_matches_without_a` as select_new_haps_mspbwt_v3 calls it
// (QUILT/R/mspbwt.R:297-310: Z, all_symbols, usge_all, egs, pbwtL = mspbwtL, pbwtM = mspbwtM).  The mspbwt package
// (rwdavies/mspbwt, DESCRIPTION: mspbwt >= 0.1.0) is an R dependency that is NOT in the reference tree, so its text cannot be
// followed: PARITY UNPINNED against the package.  What is built is the published algorithm its interface names (Durbin 2014,
// Bioinformatics 30:1266, algorithm 5's insertion-point update, in the multi-symbol form of the QUILT2 paper's Methods),
// exactly as tests/mspbwt_scan.py states it -- that file is this one's specification and its test:
//   * index i covers grids i, i + n, i + 2 n, ... (positions 0, 1, ...); a haplotype's symbol at a grid is its hapMatcherR
//     code (row of the grid's dictionary, 1-based; 0 = not in the dictionary: sorts first, matches nothing)
//   * after position t the haplotypes are in positional prefix order: a stable sort by the symbol at t of the order after
//     t - 1 (identity before position 0)
//   * the query's insertion point f after t: the haplotypes with a smaller symbol at t, plus those before the old f with the
//     query's symbol (a query word that is not in the grid's dictionary has no symbol: f = 0)
//   * the L haplotypes above and the L below f are reported when their match with the query ending at t has >= M positions:
//     (haplotype0, start0, len1); the same (haplotype, start) seen at several positions keeps its longest report
//
// Index layout per (index, position): `a` the order after the position (K int32), `old_of` for each place in that order the
// place its haplotype had in the order before (K int32: increasing within a symbol's block, so the count of the query's
// symbol before the old insertion point -- Durbin's u / the package's usge -- is one binary search in the block), `below`
// the block starts (257 int32).  2 x 4 x K x nGrids bytes over all indices: 0.8 GB for K = 50 000 x 2 000 grids, host memory.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/quilt_amd.h"
#include "../../include/quilt_amd_io.h"

namespace qa { void set_error(const char *fmt, ...); }

struct qa_mspbwt {
    int K = 0, G = 0, nindices = 0, nMaxDH = 0;
    std::vector<uint8_t> sym;   // [K][G] haplotype-major symbols (a match's length is a walk back along one row)
    // per grid the dictionary's (word, row) pairs sorted by word then row: the FIRST row holding a word is its symbol
    // (distinctHapsB is zero padded: a genuine word 0 must map to its own row, the lowest)
    std::vector<std::pair<int32_t, int32_t>> dict;   // [G][nMaxDH]
    struct Index {
        int Tp = 0;
        std::vector<int32_t> a, old_of;   // [Tp][K]
        std::vector<int32_t> below;        // [Tp][257]: below[x] = haplotypes with a symbol < x at the position
    };
    std::vector<Index> idx;
};

namespace {

int n_threads(size_t n_tasks) {
    int c = std::min<int>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("QA_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) c = std::min(c, v);
    }
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)c, n_tasks));
}

template <class F>
void parallel_tasks(size_t n, F &&f) {
    const int nt = n_threads(n);
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    // an exception inside a task (std::bad_alloc: the index is 0.8 GB at K = 50 000) must not escape its thread -- that is
    // std::terminate, and the host is R or Python: the first one is kept, the other tasks are skipped, and it is rethrown on the
    // calling thread, whose callers map it to a status with qa::set_error
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr err;
    std::mutex mu;
    std::vector<std::thread> th;
    auto work = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            try {
                f(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!err) err = std::current_exception();
                failed.store(true);
                return;
            }
        }
    };
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

struct Rep { int32_t k, s0, n; };

// the scan of one query against one index; `zq`: the query's symbols at ALL grids (0: none)
void scan_one(const qa_mspbwt &m, int i_index, const uint8_t *zq, int L, int M, std::vector<Rep> &out) {
    const qa_mspbwt::Index &ix = m.idx[(size_t)i_index];
    const int K = m.K, G = m.G, n = m.nindices;
    out.clear();
    int64_t f = 0;
    // the neighbours of the position before, with their match lengths: a haplotype that stays a neighbour and matches again
    // is one position longer (no walk back)
    int32_t prev_k[2 * 64 + 2], prev_n[2 * 64 + 2], cur_k[2 * 64 + 2], cur_n[2 * 64 + 2];
    int n_prev = 0;
    for (int t = 0; t < ix.Tp; t++) {
        const int g = i_index + t * n;
        const int z = zq[g];
        const int32_t *below = ix.below.data() + (size_t)t * 257;
        if (z == 0) {
            f = 0;
            n_prev = 0;   // nothing matches a query without a symbol: every run ends here
            continue;
        }
        {
            const int32_t *blk = ix.old_of.data() + (size_t)t * K;
            const int32_t *b0 = blk + below[z], *b1 = blk + below[z + 1];
            f = below[z] + (std::lower_bound(b0, b1, (int32_t)f) - b0);
        }
        const int64_t lo = std::max<int64_t>(f - L, 0), hi = std::min<int64_t>(f + L, K);
        const int32_t *a = ix.a.data() + (size_t)t * K;
        int n_cur = 0;
        for (int64_t j = lo; j < hi; j++) {
            const int32_t k = a[j];
            const uint8_t *row = m.sym.data() + (size_t)k * G;
            if (row[g] != z) continue;   // (z != 0 here, so a haplotype outside the dictionary never matches)
            int32_t len = 0;
            for (int q = 0; q < n_prev; q++)
                if (prev_k[q] == k) { len = prev_n[q] + 1; break; }
            if (!len) {
                len = 1;
                for (int gg = g - n; gg >= 0 && zq[gg] != 0 && row[gg] == zq[gg]; gg -= n) len++;
            }
            cur_k[n_cur] = k;
            cur_n[n_cur++] = len;
            if (len >= M) out.push_back({k, t - len + 1, len});
        }
        std::memcpy(prev_k, cur_k, sizeof(int32_t) * (size_t)n_cur);
        std::memcpy(prev_n, cur_n, sizeof(int32_t) * (size_t)n_cur);
        n_prev = n_cur;
    }
    // one row per (haplotype, start): the longest report; rows in (haplotype, start) order
    std::sort(out.begin(), out.end(), [](const Rep &x, const Rep &y) {
        if (x.k != y.k) return x.k < y.k;
        if (x.s0 != y.s0) return x.s0 < y.s0;
        return x.n > y.n;
    });
    size_t w = 0;
    for (size_t r = 0; r < out.size(); r++)
        if (r == 0 || out[r].k != out[r - 1].k || out[r].s0 != out[r - 1].s0) out[w++] = out[r];
    out.resize(w);
}

// the query's symbols: per grid the first dictionary row holding its word (1-based), 0 when none does
void query_symbols(const qa_mspbwt &m, const int32_t *Z, uint8_t *zq) {
    for (int g = 0; g < m.G; g++) {
        const std::pair<int32_t, int32_t> *d0 = m.dict.data() + (size_t)g * m.nMaxDH, *d1 = d0 + m.nMaxDH;
        const auto it = std::lower_bound(d0, d1, std::make_pair(Z[g], (int32_t)-1));
        zq[g] = (it != d1 && it->first == Z[g]) ? (uint8_t)(it->second + 1) : 0;
    }
}

bool scan_args_ok(const qa_mspbwt *m, int32_t n_query, const int32_t *Zs, int32_t L, int32_t M) {
    if (!m || n_query < 0 || (n_query > 0 && !Zs) || L < 1 || L > 64 || M < 1) {
        qa::set_error("msPBWT scan: needs an index, queries, 1 <= mspbwtL <= 64 and mspbwtM >= 1");
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

qa_mspbwt_t *qa_mspbwt_create(int32_t K, int32_t nGrids, const uint8_t *hapMatcherR, int32_t nMaxDH, const int32_t *distinctHapsB,
                              int32_t nindices) {
    if (K < 1 || nGrids < 1 || !hapMatcherR || nMaxDH < 1 || nMaxDH > 255 || !distinctHapsB || nindices < 1 || nindices > nGrids) {
        qa::set_error("qa_mspbwt_create: needs hapMatcherR (K x nGrids, uint8), distinctHapsB (nMaxDH <= 255 rows) and 1 <= nindices <= nGrids");
        return nullptr;
    }
    qa_mspbwt *m = nullptr;
    try {
        m = new qa_mspbwt;
        m->K = K; m->G = nGrids; m->nindices = nindices; m->nMaxDH = nMaxDH;
        m->sym.resize((size_t)K * nGrids);
        for (int g = 0; g < nGrids; g++) {          // R's K x nGrids column-major -> haplotype-major rows
            const uint8_t *col = hapMatcherR + (size_t)g * K;
            for (int k = 0; k < K; k++) {
                if (col[k] > nMaxDH) throw std::runtime_error("qa_mspbwt_create: a hapMatcherR code exceeds nMaxDH");
                m->sym[(size_t)k * nGrids + g] = col[k];
            }
        }
        m->dict.resize((size_t)nGrids * nMaxDH);
        for (int g = 0; g < nGrids; g++) {
            std::pair<int32_t, int32_t> *d = m->dict.data() + (size_t)g * nMaxDH;
            for (int r = 0; r < nMaxDH; r++) d[r] = {distinctHapsB[(size_t)g * nMaxDH + r], r};
            std::sort(d, d + nMaxDH);
        }
        m->idx.resize((size_t)nindices);
        parallel_tasks((size_t)nindices, [&](size_t i) {
            qa_mspbwt::Index &ix = m->idx[i];
            ix.Tp = (nGrids - (int)i + nindices - 1) / nindices;
            ix.a.resize((size_t)ix.Tp * K);
            ix.old_of.resize((size_t)ix.Tp * K);
            ix.below.assign((size_t)ix.Tp * 257, 0);
            std::vector<int32_t> prev((size_t)K);
            for (int k = 0; k < K; k++) prev[(size_t)k] = k;
            for (int t = 0; t < ix.Tp; t++) {
                const uint8_t *col = hapMatcherR + (size_t)((int)i + t * nindices) * K;
                int32_t *below = ix.below.data() + (size_t)t * 257;
                int32_t cnt[257] = {0};
                for (int j = 0; j < K; j++) cnt[col[prev[(size_t)j]] + 1]++;
                for (int x = 1; x <= 256; x++) cnt[x] += cnt[x - 1];
                std::memcpy(below, cnt, sizeof(cnt));
                int32_t *a = ix.a.data() + (size_t)t * K, *old_of = ix.old_of.data() + (size_t)t * K;
                for (int j = 0; j < K; j++) {       // stable: the old order is kept within a symbol
                    const int32_t k = prev[(size_t)j];
                    const int32_t at = cnt[col[k]]++;
                    a[at] = k;
                    old_of[at] = j;
                }
                std::memcpy(prev.data(), a, sizeof(int32_t) * (size_t)K);
            }
        });
    } catch (const std::exception &e) {
        qa::set_error("%s", e.what());
        delete m;
        return nullptr;
    }
    return m;
}

void qa_mspbwt_destroy(qa_mspbwt_t *m) { delete m; }

int64_t qa_mspbwt_bytes(const qa_mspbwt_t *m) {
    if (!m) return 0;
    int64_t b = (int64_t)m->sym.size() + (int64_t)m->dict.size() * 8;
    for (const auto &ix : m->idx) b += (int64_t)(ix.a.size() + ix.old_of.size() + ix.below.size()) * 4;
    return b;
}

// CSR form: row_ptr n_query x nindices + 1; rows (haplotype0, start0, len1) up to cap_rows triples.  Returns the number of
// rows found (the caller re-calls with a larger buffer when it exceeds cap_rows: nothing is written beyond the capacity,
// row_ptr is complete either way), or a negative status.
int64_t qa_mspbwt_find_good_matches(const qa_mspbwt_t *m, int32_t n_query, const int32_t *Zs, int32_t L, int32_t M,
                                    int64_t *row_ptr, int32_t *rows, int64_t cap_rows) {
    if (!scan_args_ok(m, n_query, Zs, L, M) || !row_ptr || (cap_rows > 0 && !rows)) return QA_ERR_INVALID;
    try {
    const int ni = m->nindices;
    std::vector<std::vector<Rep>> found((size_t)n_query * ni);
    std::vector<uint8_t> zq((size_t)n_query * m->G);
    parallel_tasks((size_t)n_query, [&](size_t q) { query_symbols(*m, Zs + q * (size_t)m->G, zq.data() + q * (size_t)m->G); });
    parallel_tasks((size_t)n_query * ni, [&](size_t qi) {
        scan_one(*m, (int)(qi % (size_t)ni), zq.data() + (qi / (size_t)ni) * (size_t)m->G, L, M, found[qi]);
    });
    int64_t total = 0;
    for (size_t qi = 0; qi < found.size(); qi++) {
        row_ptr[qi] = total;
        for (const Rep &r : found[qi]) {
            if (total < cap_rows) { rows[3 * total] = r.k; rows[3 * total + 1] = r.s0; rows[3 * total + 2] = r.n; }
            total++;
        }
    }
    row_ptr[found.size()] = total;
    return total;
    } catch (const std::exception &e) {
        qa::set_error("qa_mspbwt_find_good_matches: %s", e.what());
        return QA_ERR_INVALID;
    }
}

// The scan followed by select_new_haps_mspbwt_v3 (QUILT/R/mspbwt.R:225-474) for every chain of a round: the match tables
// never leave the native side.  Zs: n_chain x n_label x nGrids packed haploid dosages; out: n_chain x Knew, 1-based.
int qa_mspbwt_select_new_haps(const qa_mspbwt_t *m, int32_t n_chain, int32_t n_label, const int32_t *Zs, int32_t L, int32_t M,
                              int32_t Knew, const uint64_t *seed, int32_t *out) {
    if (!scan_args_ok(m, n_chain, Zs, L, M) || n_label < 1 || n_label > 3 || Knew < 1 || Knew > m->K || !seed || !out)
        return QA_ERR_INVALID;
    try {
    const int ni = m->nindices, G = m->G;
    std::atomic<int> status{QA_OK};
    parallel_tasks((size_t)n_chain, [&](size_t c) {
        std::vector<std::vector<Rep>> found((size_t)n_label * ni);
        std::vector<uint8_t> zq((size_t)G);
        size_t mm = 1;
        for (int h = 0; h < n_label; h++) {
            query_symbols(*m, Zs + (c * (size_t)n_label + h) * (size_t)G, zq.data());
            for (int i = 0; i < ni; i++) {
                scan_one(*m, i, zq.data(), L, M, found[(size_t)h * ni + i]);
                mm = std::max(mm, found[(size_t)h * ni + i].size());
            }
        }
        std::vector<int32_t> match((size_t)n_label * ni * mm * 3, 0), n_match((size_t)n_label * ni);
        for (size_t q = 0; q < found.size(); q++) {
            n_match[q] = (int32_t)found[q].size();
            int32_t *dst = match.data() + q * mm * 3;
            for (size_t r = 0; r < found[q].size(); r++) { dst[3 * r] = found[q][r].k; dst[3 * r + 1] = found[q][r].s0; dst[3 * r + 2] = found[q][r].n; }
        }
        const int st = qa_select_new_haps_mspbwt(1, n_label, ni, (int32_t)mm, match.data(), n_match.data(), Knew, m->K, G, seed + c,
                                                 out + c * (size_t)Knew);
        if (st != QA_OK) status = st;
    });
    return status;
    } catch (const std::exception &e) {
        qa::set_error("qa_mspbwt_select_new_haps: %s", e.what());
        return QA_ERR_INVALID;
    }
}

}  // extern "C"
