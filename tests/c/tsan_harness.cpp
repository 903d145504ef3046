// tsan_harness.cpp -- the product's HOST threading under ThreadSanitizer (tests/test_tsan_cpu.py builds and runs it).
//
// csrc/impute.cpp (three host threads taking launch sets in turn, helper threads, the staggered start, the fused tails, the sample
// source, on_samples_done) and csrc/bamrange.cpp (loader threads settling files in order beside the call, formatter pool, count
// sums) are compiled here with -fsanitize=thread and run over a table of TRIVIAL entry points (constant dosages, labels passed
// through; the accumulation and the consensus labels are the library's own host functions): nothing numerical is tested -- the oracle-backed tests do that -- only that the threads' accesses are ordered.  The
// run also checks what is checkable without numerics: flat call == one-by-one call, and the BAM range's bookkeeping.
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/quilt_amd.h"
#include "../../include/quilt_amd_io.h"
#include "../../quilt_amd/csrc/impute_testhook.h"

namespace qa {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace qa
extern "C" const char *qa_last_error(void) { return qa::g_err; }

namespace {

int g_T = 0, g_T_all = 0, g_G = 0;   // SNPs of the panel / of the all-SNP call, grids (the table's functions do not get them everywhere)
std::atomic<long> n_gibbs{0}, n_full{0};

int f_gibbs(void *, const qa_gibbs_opts_t *o, int32_t n, const int32_t *, const int32_t *read_off, const int32_t *, const int32_t *,
            const int32_t *, const int32_t *, const double *, const int32_t *, const double *, int32_t *H, int32_t *, double *, double *,
            double *, int32_t *uf, double *, const uint64_t *, const uint64_t *) {
    n_gibbs += n;
    for (int a = 0; a < n; a++) uf[a] = 0;
    const int nl = o->sample_is_diploid ? 2 : 3;
    for (int r = 0; r < read_off[n]; r++) H[r] = 1 + (H[r] % nl);   // (labels stay in 1 .. n_label)
    if (o->hap_major_out)
        for (size_t i = 0; i < (size_t)n * o->hap_major_labels * g_T; i++) o->hap_major_out[i] = 0.25;
    if (o->hap_words_out)
        for (size_t i = 0; i < (size_t)n * 3 * g_G; i++) o->hap_words_out[i] = (int32_t)i;
    return QA_OK;
}
int f_gibbs_rc(void *, const void *, const qa_gibbs_opts_t *o, int32_t n, const int32_t *, const int32_t *read_off, const int32_t *,
               const int32_t *, const int32_t *, const int32_t *, const double *, const int32_t *, const double *, int32_t *H, int32_t *,
               double *, double *, double *, int32_t *uf, double *, const uint64_t *, const uint64_t *) {
    n_gibbs += n;
    for (int a = 0; a < n; a++) uf[a] = 0;
    const int nl = o->sample_is_diploid ? 2 : 3;
    for (int r = 0; r < read_off[n]; r++) H[r] = 1 + (H[r] % nl);
    if (o->hap_major_out)
        for (size_t i = 0; i < (size_t)n * o->hap_major_labels * g_T_all; i++) o->hap_major_out[i] = 0.25;
    return QA_OK;
}
int f_fullpass_select(void *, int32_t n_chain, int32_t n_label, int32_t, const int32_t *, const int32_t *, const int32_t *, const int32_t *,
                      const int32_t *, const int32_t *, const int32_t *, const int32_t *, const int32_t *, int32_t, double, double *dosage,
                      int32_t, int32_t *, float *, int32_t *, int32_t Ksubset, int32_t, const int32_t *which, const uint64_t *,
                      int32_t *which_next, int32_t *select_status) {
    n_full += n_chain;
    if (dosage)
        for (size_t i = 0; i < (size_t)n_chain * n_label * g_T; i++) dosage[i] = 0.5;
    std::memcpy(which_next, which, sizeof(int32_t) * (size_t)n_chain * Ksubset);
    for (int i = 0; i < n_chain; i++) select_status[i] = 0;
    return QA_OK;
}
int f_fullpass(void *, int32_t, const double *, const int32_t *, const int32_t *, int32_t, double *, int32_t *, int32_t *, double *, int64_t) {
    return QA_ERR_INVALID;   // (the complete-lists branch is not reached: every selection reports status 0)
}
int f_emat(void *, int32_t, int32_t n_chain, int32_t K, const double *, const int32_t *read_off, const int32_t *, const int32_t *,
           const int32_t *, double, int32_t, int32_t, double *e) {
    for (size_t i = 0; i < (size_t)read_off[n_chain] * K; i++) e[i] = 0.5;
    return QA_OK;
}
int f_mspbwt(const qa_mspbwt_t *, int32_t n_chain, int32_t, const int32_t *, int32_t, int32_t, int32_t Knew, const uint64_t *, int32_t *out) {
    for (int a = 0; a < n_chain; a++)
        for (int j = 0; j < Knew; j++) out[(size_t)a * Knew + j] = 1 + j;
    return QA_OK;
}
void *f_alloc(size_t b) { return std::malloc(b ? b : 1); }
int f_free(void *p) { std::free(p); return QA_OK; }

qa_impute_backend_t table() {
    qa_impute_backend_t t{};
    t.gibbs_batch = f_gibbs;
    t.fullpass_reads_select_batch = f_fullpass_select;
    t.fullpass_batch = f_fullpass;
    t.make_eMatRead_t_hap_major = f_emat;
    t.mspbwt_select_new_haps = f_mspbwt;
    t.gibbs_batch_rare_common = f_gibbs_rc;
    t.make_eMatRead_t_nsnps = f_emat;   // (same signature: the all-SNP likelihoods, constant here)
    t.accumulate_dosage = qa_accumulate_dosage;          // (the library's own: host functions of csrc/hostio.cpp)
    t.consensus_read_labels = qa_consensus_read_labels;
    t.host_alloc = f_alloc;
    t.host_free = f_free;
    return t;
}

struct Samples {
    std::vector<int32_t> read_off{0}, read_ptr, u, bq, wif;
    std::vector<std::vector<int32_t>> p_ptr, p_u, p_bq, p_wif, p_labels;   // the same, sample by sample
};
Samples make_samples(int n, int T) {
    Samples S;
    for (int s = 0; s < n; s++) {
        const int R = 30 + (s * 7) % 23;
        std::vector<int32_t> ptr{0}, uu, qq, ww;
        for (int r = 0; r < R; r++) {
            const int snp = (int)(((long)r * T) / R);
            for (int j = 0; j < 2 && snp + j < T; j++) { uu.push_back(snp + j); qq.push_back((r + j) % 2 ? 25 : -25); }
            ptr.push_back((int32_t)uu.size());
            ww.push_back(snp / 32);
        }
        S.read_off.push_back(S.read_off.back() + R);
        S.read_ptr.insert(S.read_ptr.end(), ptr.begin(), ptr.end());
        S.u.insert(S.u.end(), uu.begin(), uu.end());
        S.bq.insert(S.bq.end(), qq.begin(), qq.end());
        S.wif.insert(S.wif.end(), ww.begin(), ww.end());
        S.p_ptr.push_back(ptr); S.p_u.push_back(uu); S.p_bq.push_back(qq); S.p_wif.push_back(ww);
        S.p_labels.emplace_back((size_t)R, 0);
    }
    return S;
}

struct MemSource {
    Samples *S;
    int n;
    std::atomic<int> n_done{0};
    static int acquire(void *ctx, int32_t s, qa_sample_view_t *v) {
        MemSource &M = *static_cast<MemSource *>(ctx);
        if (s >= M.n) return QA_END_OF_SAMPLES;
        v->n_reads = (int32_t)M.S->p_wif[(size_t)s].size();
        v->read_ptr = M.S->p_ptr[(size_t)s].data(); v->u = M.S->p_u[(size_t)s].data(); v->bq = M.S->p_bq[(size_t)s].data();
        v->wif = M.S->p_wif[(size_t)s].data();
        v->read_labels = M.S->p_labels[(size_t)s].data();
        return QA_OK;
    }
    static void done(void *ctx, int32_t lo, int32_t hi) { static_cast<MemSource *>(ctx)->n_done += hi - lo; }
};

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "tsan harness: %s failed (line %d): %s\n", #c, __LINE__, qa_last_error()); return 1; } } while (0)

}  // namespace

int main(int argc, char **argv) {
    // argv: sites file (int32 T, then T x int32 L, T bytes ref, T bytes alt, T x int32 grid), then BAM paths
    const int K = 300, W = 3;
    qa_impute_params_t P;
    qa_impute_params_default(&P);
    P.nGibbsSamples = 2; P.n_seek_its = 2; P.Ksubset = 64; P.Knew = 64; P.n_block_gibbs_iterations = 0;
    P.small_ref_panel_block_gibbs_iterations = nullptr; P.small_ref_panel_gibbs_iterations = 2;
    const qa_impute_backend_t tab = table();
    void *handles[W] = {(void *)1, (void *)2, (void *)3};
    {
        const int T = 640, G = 20, n = 23;
        g_T = T; g_G = G;
        Samples S = make_samples(n, T);
        for (int per_set : {2, 5, 256}) {
            P.samples_per_launch_set = per_set;
            std::vector<double> d((size_t)n * T), g((size_t)n * 3 * T), h((size_t)n * 2 * T);
            std::vector<int32_t> lab((size_t)S.read_off[(size_t)n]), nd((size_t)n);
            int64_t stats[11];
            P.sample_source = nullptr; P.on_samples_done = nullptr;
            REQUIRE(qa_impute_samples_backend(&tab, handles, W, K, G, T, &P, n, 7, S.read_off.data(), S.read_ptr.data(), S.u.data(), S.bq.data(),
                                              S.wif.data(), d.data(), g.data(), h.data(), lab.data(), nd.data(), stats) == QA_OK);
            const int up = n + 4;   // (the source's call: n_sample is an upper bound)
            std::vector<double> d2((size_t)up * T, -1), g2((size_t)up * 3 * T, -1), h2((size_t)up * 2 * T, -1);
            std::vector<int32_t> nd2((size_t)up, 0);
            MemSource M{&S, n};
            const qa_sample_source_t src{&MemSource::acquire, &M};
            P.sample_source = &src; P.on_samples_done = &MemSource::done; P.on_samples_done_ctx = &M;
            REQUIRE(qa_impute_samples_backend(&tab, handles, W, K, G, T, &P, up, 7, nullptr, nullptr, nullptr, nullptr, nullptr, d2.data(),
                                              g2.data(), h2.data(), nullptr, nd2.data(), stats) == QA_OK);
            REQUIRE(M.n_done.load() == n);
            REQUIRE(std::memcmp(d.data(), d2.data(), sizeof(double) * (size_t)n * T) == 0);
            REQUIRE(std::memcmp(g.data(), g2.data(), sizeof(double) * (size_t)n * 3 * T) == 0);
            REQUIRE(std::memcmp(h.data(), h2.data(), sizeof(double) * (size_t)n * 2 * T) == 0);
            for (int s = 0; s < n; s++)
                REQUIRE(std::memcmp(&lab[(size_t)S.read_off[(size_t)s]], S.p_labels[(size_t)s].data(), sizeof(int32_t) * S.p_labels[(size_t)s].size()) == 0);
            REQUIRE(d2[(size_t)n * T] == -1);   // rows beyond the range's end are not touched
        }
    }
    {   // the other modes' branches of the loop: msPBWT selection, three labels (NIPT), the all-SNP round (rare + common)
        const int T = 640, G = 20, n = 11;
        g_T = T; g_G = G;
        Samples S = make_samples(n, T);
        int64_t stats[11];
        P.sample_source = nullptr; P.on_samples_done = nullptr; P.samples_per_launch_set = 2;
        std::vector<int32_t> lab((size_t)S.read_off[(size_t)n]), nd((size_t)n);
        {
            // use_mspbwt with the library's own host index and neighbour scan (csrc/mspbwt.cpp: threaded over indices and chains),
            // called concurrently by the loop's host threads
            std::vector<uint8_t> hm((size_t)K * G);
            std::vector<int32_t> B((size_t)16 * G);
            for (size_t i = 0; i < hm.size(); i++) hm[i] = (uint8_t)(1 + (i * 2654435761u >> 7) % 16);
            for (size_t i = 0; i < B.size(); i++) B[i] = (int32_t)(i * 40503u);
            qa_mspbwt_t *index = qa_mspbwt_create(K, G, hm.data(), 16, B.data(), 2);
            REQUIRE(index != nullptr);
            qa_impute_backend_t tab2 = tab;
            tab2.mspbwt_select_new_haps = qa_mspbwt_select_new_haps;
            qa_impute_params_t Q = P;
            Q.use_mspbwt = 1; Q.mspbwt_index = index; Q.mspbwtL = 3; Q.mspbwtM = 1;
            {
                std::vector<double> d((size_t)n * T), g((size_t)n * 3 * T), h((size_t)n * 2 * T);
                REQUIRE(qa_impute_samples_backend(&tab2, handles, W, K, G, T, &Q, n, 0, S.read_off.data(), S.read_ptr.data(), S.u.data(),
                                                  S.bq.data(), S.wif.data(), d.data(), g.data(), h.data(), lab.data(), nd.data(), stats) == QA_OK);
            }
            qa_mspbwt_destroy(index);
            Q.mspbwt_index = reinterpret_cast<const qa_mspbwt_t *>(0x10);   // (and once with the trivial selection of the table)
            std::vector<double> d((size_t)n * T), g((size_t)n * 3 * T), h((size_t)n * 2 * T);
            REQUIRE(qa_impute_samples_backend(&tab, handles, W, K, G, T, &Q, n, 0, S.read_off.data(), S.read_ptr.data(), S.u.data(), S.bq.data(),
                                              S.wif.data(), d.data(), g.data(), h.data(), lab.data(), nd.data(), stats) == QA_OK);
        }
        {
            qa_impute_params_t Q = P;
            std::vector<double> ff((size_t)n, 0.2), fd((size_t)n * T), fg((size_t)n * 3 * T);
            std::vector<int32_t> Lg((size_t)G);
            for (int i = 0; i < G; i++) Lg[(size_t)i] = 1000 * (i + 1);
            qa_impute_nipt_t nq{ff.data(), Lg.data(), 5000, fd.data(), fg.data()};
            Q.nipt = &nq;
            std::vector<double> d((size_t)n * T), g((size_t)n * 3 * T), h((size_t)n * 3 * T);
            REQUIRE(qa_impute_samples_backend(&tab, handles, W, K, G, T, &Q, n, 0, S.read_off.data(), S.read_ptr.data(), S.u.data(), S.bq.data(),
                                              S.wif.data(), d.data(), g.data(), h.data(), lab.data(), nd.data(), stats) == QA_OK);
        }
        {
            qa_impute_params_t Q = P;
            const int Ta = 2 * T, Ga = (Ta + 31) / 32;
            g_T_all = Ta;
            Samples A = make_samples(n, Ta);
            std::vector<uint8_t> is_common((size_t)Ta, 0);
            for (int t = 0; t < T; t++) is_common[(size_t)2 * t] = 1;
            const qa_rare_common_t *rch[W] = {reinterpret_cast<const qa_rare_common_t *>(0x20), reinterpret_cast<const qa_rare_common_t *>(0x30),
                                              reinterpret_cast<const qa_rare_common_t *>(0x40)};
            qa_impute_rare_common_t rc{rch, Ta, Ga, is_common.data(), A.read_off.data(), A.read_ptr.data(), A.u.data(), A.bq.data(), A.wif.data(), nullptr};
            Q.rare_common = &rc;
            std::vector<double> d((size_t)n * Ta), g((size_t)n * 3 * Ta), h((size_t)n * 2 * Ta);
            REQUIRE(qa_impute_samples_backend(&tab, handles, W, K, G, T, &Q, n, 0, S.read_off.data(), S.read_ptr.data(), S.u.data(), S.bq.data(),
                                              S.wif.data(), d.data(), g.data(), h.data(), lab.data(), nd.data(), stats) == QA_OK);
            REQUIRE(nd[0] == Q.nGibbsSamples);
        }
    }
    if (argc >= 3) {
        FILE *f = std::fopen(argv[1], "rb");
        REQUIRE(f != nullptr);
        int32_t T = 0;
        REQUIRE(std::fread(&T, 4, 1, f) == 1 && T > 0);
        std::vector<int32_t> L((size_t)T), grid((size_t)T);
        std::vector<char> ref((size_t)T), alt((size_t)T);
        REQUIRE(std::fread(L.data(), 4, (size_t)T, f) == (size_t)T && std::fread(ref.data(), 1, (size_t)T, f) == (size_t)T &&
                std::fread(alt.data(), 1, (size_t)T, f) == (size_t)T && std::fread(grid.data(), 4, (size_t)T, f) == (size_t)T);
        std::fclose(f);
        g_T = T;
        const int G = grid[(size_t)T - 1] + 1, n = argc - 2;
        g_G = G;
        qa_bam_range_io_t io{};
        io.chr = "chr20"; io.nSNPs = T; io.L = L.data(); io.ref = ref.data(); io.alt = alt.data(); io.grid = grid.data();
        qa_bam_opts_default(&io.bam);
        io.bam.bqFilter = 1; io.bam.downsampleToCov = 0;
        io.minimum_number_of_sample_reads = 2; io.output_gt_phased_genotypes = 1;
        std::vector<int64_t> idx((size_t)n);
        for (int i = 0; i < n; i++) idx[(size_t)i] = 100 + i;
        P.sample_source = nullptr; P.on_samples_done = nullptr;
        std::string first;
        for (int n_io : {1, 4, 9}) {
            io.n_io_threads = n_io;
            io.discard_sample_arrays = n_io == 4;   // (the rows' pages handed back by the formatter threads)
            P.samples_per_launch_set = n_io == 4 ? 3 : 2;
            qa_bam_range_result_t *res = nullptr;
            REQUIRE(qa_impute_bam_range_backend(&tab, handles, W, K, G, &P, &io, n, argv + 2, idx.data(), nullptr, &res) == QA_OK);
            std::string all;
            int kept = 0;
            for (int i = 0; i < n; i++) {
                const char *buf = nullptr;
                const int64_t *off = nullptr;
                REQUIRE(qa_bam_range_column(res, i, &buf, &off) == QA_OK);
                if (qa_bam_range_imputed(res, i)) { kept++; all.append(buf, (size_t)off[T]); } else REQUIRE(buf == nullptr);
            }
            std::vector<double> af((size_t)T);
            REQUIRE(qa_bam_range_counts(res, nullptr, af.data(), nullptr, nullptr) == QA_OK);
            qa_bam_range_destroy(res);
            REQUIRE(kept >= 2 && kept < n);   // (the test hands over at least one file without reads)
            if (first.empty()) first = all;
            REQUIRE(all == first);             // the text does not depend on the threads
        }
        // damaged files: the loader answers with a status (or reads what is still readable), never with a fault -- the first file cut at
        // 60 lengths and with one byte changed at 400 places, under the sanitizers
        if (const char *scratch = std::getenv("QA_HARNESS_SCRATCH")) {
            std::vector<unsigned char> bytes;
            {
                FILE *fb = std::fopen(argv[2], "rb");
                REQUIRE(fb != nullptr);
                unsigned char buf[65536];
                size_t got;
                while ((got = std::fread(buf, 1, sizeof buf, fb)) > 0) bytes.insert(bytes.end(), buf, buf + got);
                std::fclose(fb);
            }
            REQUIRE(bytes.size() > 1000);
            const std::string mut = std::string(scratch) + "/damaged.bam";
            int n_ok = 0, n_refused = 0;
            uint64_t lcg = 12345;
            for (int it = 0; it < 460; it++) {
                std::vector<unsigned char> m = bytes;
                if (it < 60) {
                    m.resize(bytes.size() * (size_t)it / 60);
                } else {
                    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                    const size_t at = (size_t)((lcg >> 20) % bytes.size());
                    m[at] ^= (unsigned char)(1u << ((lcg >> 12) & 7));
                    if (it % 3 == 0) m[at] = (unsigned char)(lcg >> 40);
                }
                FILE *fm = std::fopen(mut.c_str(), "wb");
                REQUIRE(fm != nullptr);
                if (!m.empty()) REQUIRE(std::fwrite(m.data(), 1, m.size(), fm) == m.size());
                std::fclose(fm);
                qa_sample_reads_t *h = nullptr;
                qa::set_error("%s", "");
                const int st = qa_bam_load_sample_reads(mut.c_str(), "chr20", T, L.data(), ref.data(), alt.data(), grid.data(), &io.bam, &h);
                if (st == QA_OK) {
                    n_ok++;
                    REQUIRE(h != nullptr && qa_sample_reads_n_reads(h) >= 0);
                    qa_sample_reads_destroy(h);
                } else {
                    n_refused++;
                    if (h != nullptr || qa_last_error()[0] == 0)
                        std::fprintf(stderr, "tsan harness: damaged file %d: status %d, handle %p, text '%s'\n", it, st, (void *)h, qa_last_error());
                    REQUIRE(h == nullptr && qa_last_error()[0] != 0);
                }
            }
            std::printf("tsan harness: %d damaged files read, %d refused\n", n_ok, n_refused);
            REQUIRE(n_refused >= 30);   // (every truncation inside the data is refused; a changed byte may be harmless)
        }
        // a damaged INDEX next to a sound file: "any problem with the index just means the sequential scan" -- or a refusal -- never a fault
        if (const char *scratch = std::getenv("QA_HARNESS_SCRATCH")) {
            const char *indexed = std::getenv("QA_HARNESS_INDEXED");
            if (indexed) {
                auto slurp = [](const std::string &path, std::vector<unsigned char> &out) {
                    FILE *fb = std::fopen(path.c_str(), "rb");
                    if (!fb) return false;
                    unsigned char buf[65536];
                    size_t got;
                    while ((got = std::fread(buf, 1, sizeof buf, fb)) > 0) out.insert(out.end(), buf, buf + got);
                    std::fclose(fb);
                    return true;
                };
                auto spill = [](const std::string &path, const std::vector<unsigned char> &b) {
                    FILE *fm = std::fopen(path.c_str(), "wb");
                    if (!fm) return false;
                    const bool ok = b.empty() || std::fwrite(b.data(), 1, b.size(), fm) == b.size();
                    std::fclose(fm);
                    return ok;
                };
                std::vector<unsigned char> bam, bai;
                REQUIRE(slurp(indexed, bam) && slurp(std::string(indexed) + ".bai", bai) && bai.size() > 16);
                const std::string copy = std::string(scratch) + "/indexed.bam";
                REQUIRE(spill(copy, bam));
                qa_bam_opts_t w = io.bam;
                w.chrStart = L[(size_t)T / 2]; w.chrEnd = L[(size_t)T - 1];   // (a window: the index is consulted)
                int sound_reads = -1, n_ok = 0, n_refused = 0;
                uint64_t lcg = 777;
                for (int it = -1; it < 300; it++) {
                    std::vector<unsigned char> m = bai;
                    if (it >= 0 && it < 40) m.resize(bai.size() * (size_t)it / 40);
                    else if (it >= 40) {
                        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                        const size_t at = (size_t)((lcg >> 20) % bai.size());
                        m[at] = (unsigned char)(lcg >> 40);
                        if (it % 2) m[(at + 1) % bai.size()] ^= 0xff;
                    }
                    REQUIRE(spill(copy + ".bai", m));
                    qa_sample_reads_t *h = nullptr;
                    const int st = qa_bam_load_sample_reads(copy.c_str(), "chr20", T, L.data(), ref.data(), alt.data(), grid.data(), &w, &h);
                    if (it < 0) { REQUIRE(st == QA_OK); sound_reads = qa_sample_reads_n_reads(h); REQUIRE(sound_reads > 0); }
                    if (st == QA_OK) { n_ok++; qa_sample_reads_destroy(h); } else { n_refused++; REQUIRE(h == nullptr && qa_last_error()[0] != 0); }
                }
                std::printf("tsan harness: damaged index: %d loads went through, %d refused (sound: %d reads)\n", n_ok, n_refused, sound_reads);
            }
        }
        // an unreadable file: the call fails, the loaders and formatters are joined
        std::vector<const char *> bad(argv + 2, argv + argc);
        bad[(size_t)n / 2] = "/nonexistent/file.bam";
        qa_bam_range_result_t *res = nullptr;
        io.n_io_threads = 4;
        REQUIRE(qa_impute_bam_range_backend(&tab, handles, W, K, G, &P, &io, n, bad.data(), idx.data(), nullptr, &res) != QA_OK && res == nullptr);
    }
    std::printf("tsan harness: ok (%ld chain calls, %ld full-panel chains)\n", n_gibbs.load(), n_full.load());
    return 0;
}
