// bamrange.cpp -- qa_impute_bam_range (include/quilt_amd_io.h): a core's sample range from BAM paths to VCF columns in ONE
// native call.
//
// What it replaces: per sample of the range, get_and_impute_one_sample's own I/O either side of the imputation
// (QUILT/R/functions.R:243-298: STITCH::loadBamAndConvert + load() of the per-sample RData temp file +
// snap_sampleReads_to_grid; :1380-1463: allele counts from the pile-up, eij / fij / max_gen, rcpp_make_column_of_vcf and the
// paste0 assembly of the column) and the range's part of the loop body around it (quilt.R:955-961: the four count arrays
// summed over the samples of the core).  In R these run one sample after the other on the worker that owns the GPU: about a
// sample per second, against the ~40 samples per second the device imputes -- the R loader, not the device, would set the
// throughput a QUILT2.R user sees.  Here
//   1. the BAM files are read by qa_bam_load_sample_reads on n_io_threads host threads (17 ms per 1x sample and thread), in file
//      order and BESIDE the imputation: the first launch set starts as soon as its own files are read;
//   2. the kept samples go through ONE qa_impute_samples call (csrc/impute.cpp) that is handed each sample when the launch set
//      holding it is taken (params->sample_source) -- params->sample_index names every kept sample's GLOBAL index, so a sample
//      dropped for too few reads does not shift the streams of the samples behind it;
//   3. the columns of every finished launch set are formatted by qa_vcf_column_diploid / _nipt on host threads while later sets
//      are on the device (params->on_samples_done), and the four count arrays are summed over the imputed samples in sample
//      order (the order of the reference's loop: floating-point sums).
// Host code only: no HIP here (the device work is inside qa_impute_samples).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <unistd.h>

#include "../../include/quilt_amd.h"
#include "../../include/quilt_amd_io.h"
#include "impute_testhook.h"   // (qa_impute_bam_range_backend: the same host code over a checker's entry points, for tests/)

namespace qa { void set_error(const char *fmt, ...); }

// The imputation's result arrays: 48 bytes per sample and SNP (7.9 GB at 2 560 samples x 64 000 SNPs).  qa_impute_samples does not
// ask for them zeroed (it zeroes a launch set's rows itself, on the thread that takes the set), so they are allocated WITHOUT a
// fill: std::vector's zero fill touched every page on the calling thread before the first launch -- 2.4 s of the 70 s job.
struct RawDoubles {
    std::unique_ptr<double[]> p;
    void alloc(size_t n) { p.reset(new double[std::max<size_t>(n, 1)]); }
    double *data() const { return p.get(); }
};

struct qa_bam_range_result {
    int n = 0, n_kept = 0, T_out = 0, nL = 2;
    bool nipt = false, discarded = false;
    std::vector<uint8_t> imputed;        // per file
    std::vector<int32_t> n_reads;        // per file: reads the loader returned (before the minimum test)
    std::vector<int32_t> slot;           // per file: index among the kept samples, or -1
    std::vector<int32_t> kept;           // per kept sample: its file
    std::vector<std::vector<int32_t>> labels_of;   // per kept sample: its reads' consensus labels
    std::vector<int32_t> nDosage;
    RawDoubles dosage, gp_t, haps, fet_dosage, fet_gp_t;            // kept-major, the layouts of qa_impute_samples
    std::vector<std::vector<char>> col_buf;
    std::vector<std::vector<int64_t>> col_off;
    std::vector<double> infoCount, afCount, hweCount, alleleCount;
    double seconds[4] = {0, 0, 0, 0};    // load, impute (the columns of finished launch sets are formatted beside it), what is left of formatting + counts, whole call
    double format_busy_s = 0;            // thread-seconds spent formatting (most of them inside seconds[1])
    int64_t stats[11] = {0};
    int64_t load_stats[8] = {0};         // the loader's counters summed over the files (qa_sample_reads_stats)
};

namespace {

using Clock = std::chrono::steady_clock;
double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// run f(i) for i in [0, n) on up to n_threads threads; the first failure (status, text) wins
template <class F>
int parallel_for(int n, int n_threads, std::string &err, F f) {
    std::atomic<int> next{0};
    std::atomic<int> status{QA_OK};
    std::mutex mu;
    auto body = [&] {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || status.load() != QA_OK) return;
            std::string e;
            int st;
            try {
                st = f(i, e);
            } catch (const std::exception &ex) {
                st = QA_ERR_INVALID;
                e = ex.what();
            }
            if (st != QA_OK) {
                std::lock_guard<std::mutex> g(mu);
                if (status.load() == QA_OK) { status.store(st); err = e; }
            }
        }
    };
    const int W = std::max(1, std::min(n_threads, n));
    if (W == 1) {
        body();
    } else {
        std::vector<std::thread> th;
        for (int w = 0; w < W; w++) th.emplace_back(body);
        for (auto &t : th) t.join();
    }
    return status.load();
}

struct Loaded {
    std::vector<int32_t> read_ptr, u, bq, wif;
    int32_t R = 0;
};

int load_one(const char *path, const char *chr, int32_t T, const int32_t *L, const char *ref, const char *alt, const int32_t *grid,
             const qa_bam_opts_t *o, Loaded &out, int64_t stats[8], std::string &err) {
    qa_sample_reads_t *h = nullptr;
    const int st = qa_bam_load_sample_reads(path, chr, T, L, ref, alt, grid, o, &h);
    if (st != QA_OK) {
        err = std::string("cannot load ") + path + ": " + qa_last_error();
        return st;
    }
    out.R = qa_sample_reads_n_reads(h);
    const int64_t nb = qa_sample_reads_n_bases(h);
    out.read_ptr.assign((size_t)out.R + 1, 0);
    out.u.resize((size_t)nb);
    out.bq.resize((size_t)nb);
    out.wif.resize((size_t)out.R);
    int32_t dummy = 0;   // (export wants non-null pointers only for what it writes; empty vectors have a null data())
    const int st2 = qa_sample_reads_export(h, out.read_ptr.data(), nb ? out.u.data() : &dummy, nb ? out.bq.data() : &dummy,
                                           out.R ? out.wif.data() : &dummy, nullptr);
    if (stats) qa_sample_reads_stats(h, stats);
    qa_sample_reads_destroy(h);
    if (st2 != QA_OK) err = std::string("cannot export the reads of ") + path;
    return st2;
}

// qa_impute_samples, or the test hook's form of it, on the kept samples
using ImputeFn = std::function<int(const qa_impute_params_t *, int32_t, const int32_t *, const int32_t *, const int32_t *, const int32_t *,
                                   const int32_t *, double *, double *, double *, int32_t *, int32_t *, int64_t *)>;

int bam_range_impl(const ImputeFn &impute, const qa_impute_params_t *params, const qa_bam_range_io_t *io, int32_t n_sample,
                   const char *const *bam_paths, const int64_t *sample_index, const double *ff, qa_bam_range_result_t **out) {
    if (out) *out = nullptr;
    if (!params || !io || n_sample < 0 || (n_sample > 0 && (!bam_paths || !sample_index)) || !out ||
        !io->chr || io->nSNPs < 1 || !io->L || !io->ref || !io->alt || !io->grid) {
        qa::set_error("qa_impute_bam_range: missing argument");
        return QA_ERR_INVALID;
    }
    const bool rare = params->rare_common != nullptr, nipt = params->nipt != nullptr;
    if (rare && (io->nSNPs_all < io->nSNPs || !io->L_all || !io->ref_all || !io->alt_all || !io->grid_all ||
                 params->rare_common->nSNPs_all != io->nSNPs_all)) {
        qa::set_error("qa_impute_bam_range: impute_rare_common needs the all-SNP sites (L_all, ref_all, alt_all, grid_all; nSNPs_all as in "
                      "params->rare_common)");
        return QA_ERR_INVALID;
    }
    if (nipt && n_sample > 0 && !ff) {
        qa::set_error("qa_impute_bam_range: method = \"nipt\" needs one fetal fraction per file");
        return QA_ERR_INVALID;
    }
    for (int i = 0; i < n_sample; i++)
        if (!bam_paths[i]) { qa::set_error("qa_impute_bam_range: bam_paths[%d] is null", i); return QA_ERR_INVALID; }
    const auto t_all = Clock::now();
    // host threads of the loading and of the formatting (each): 16 by default.  Measured at 2 560 files on a 128-core host, both ends
    // beside the imputation: 16 / 32 / 64 threads -> the last file is in after 3.1 / 2.9 / 3.0 s either way (the loading does not
    // scale past 16), the formatters' busy time is 11.5 / 14.6 / 30.3 thread-seconds (they get in each other's way), and the
    // imputation itself takes 61.4 / 61.8 / 63.2 s (they get in ITS host threads' way): 40.3 / 40.0 / 39.2 samples/s.
    int n_io = io->n_io_threads > 0 ? io->n_io_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const int T = io->nSNPs, T_out = rare ? io->nSNPs_all : T, nL = nipt ? 3 : 2;
    const int min_reads = io->minimum_number_of_sample_reads > 0 ? io->minimum_number_of_sample_reads : 1;   // (an empty sample cannot be imputed)
    std::unique_ptr<qa_bam_range_result> R(new qa_bam_range_result);
    R->n = n_sample; R->T_out = T_out; R->nL = nL; R->nipt = nipt;
    R->imputed.assign((size_t)n_sample, 0);
    R->n_reads.assign((size_t)n_sample, 0);
    R->slot.assign((size_t)n_sample, -1);
    R->infoCount.assign((size_t)T_out * 2, 0.0);
    R->afCount.assign((size_t)T_out, 0.0);
    R->hweCount.assign((size_t)T_out * 3, 0.0);
    R->alleleCount.assign((size_t)T_out * 2, 0.0);

    // ---- 1. the reads of every file (functions.R:251-298; with impute_rare_common also over all SNPs, :132-172), on host threads
    // BESIDE the imputation: the files are loaded in order, and the imputation's host threads are handed each sample when the
    // launch set holding it is taken (qa_sample_source_t) -- as the reference's loop reads a sample's BAM at the top of its own
    // iteration.  Whether a file is imputed (functions.R:274-287) is known once it is loaded; the kept samples are numbered in
    // file order as the files before them are settled.
    auto t0 = Clock::now();
    const bool trace = std::getenv("QA_BAM_RANGE_TRACE") != nullptr;   // (phase times on stderr)
    double tr_setup = 0, tr_call = 0, tr_drain = 0, tr_sums = 0;
    std::vector<Loaded> common((size_t)n_sample), all_snps(rare ? (size_t)n_sample : 0);
    std::vector<std::array<int64_t, 8>> lstats((size_t)n_sample);
    std::vector<int64_t> index((size_t)std::max(n_sample, 1));     // per kept sample, filled as the files are settled
    std::vector<double> ffk((size_t)std::max(n_sample, 1));
    R->kept.assign((size_t)n_sample, -1);                          // (entry j is written before kept sample j is handed to anyone; cut to n_kept at the end)
    R->labels_of.resize((size_t)n_sample);
    struct Loader {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<uint8_t> loaded;
        int settled = 0;            // files [0, settled) are loaded and sorted into kept / dropped
        int n_kept = 0;
        int status = QA_OK;
        std::string err;
        std::atomic<int> next{0};
        std::atomic<bool> stop{false};
        double done_s = 0;
    } ld;
    ld.loaded.assign((size_t)n_sample, 0);
    auto loader_body = [&] {
        for (;;) {
            const int i = ld.next.fetch_add(1);
            if (i >= n_sample || ld.stop.load()) return;
            std::string e;
            int s1;
            try {
                s1 = load_one(bam_paths[i], io->chr, T, io->L, io->ref, io->alt, io->grid, &io->bam, common[(size_t)i], lstats[(size_t)i].data(), e);
                // (the all-SNP pile-up only for samples that will be imputed: the minimum test is on the common-SNP reads, functions.R:274)
                if (s1 == QA_OK && rare && common[(size_t)i].R >= min_reads)
                    s1 = load_one(bam_paths[i], io->chr, io->nSNPs_all, io->L_all, io->ref_all, io->alt_all, io->grid_all, &io->bam,
                                  all_snps[(size_t)i], nullptr, e);
            } catch (const std::exception &ex) {
                s1 = QA_ERR_INVALID;
                e = ex.what();
            }
            std::lock_guard<std::mutex> g(ld.mu);
            if (s1 != QA_OK) {
                if (ld.status == QA_OK) { ld.status = s1; ld.err = e; }
                ld.stop.store(true);
                ld.cv.notify_all();
                return;
            }
            ld.loaded[(size_t)i] = 1;
            while (ld.settled < n_sample && ld.loaded[(size_t)ld.settled]) {
                const int f = ld.settled;
                R->n_reads[(size_t)f] = common[(size_t)f].R;
                for (int q = 0; q < 8; q++) R->load_stats[q] += lstats[(size_t)f][(size_t)q];
                bool keep = common[(size_t)f].R >= min_reads;
                if (keep && rare && all_snps[(size_t)f].R < 1) keep = false;   // (cannot happen: every common SNP is among the all-SNP sites)
                if (keep) {
                    const size_t j = (size_t)ld.n_kept;
                    index[j] = sample_index[f];
                    if (nipt) ffk[j] = ff[f];
                    R->labels_of[j].assign((size_t)common[(size_t)f].R, 0);
                    R->slot[(size_t)f] = (int32_t)j;
                    R->imputed[(size_t)f] = 1;
                    R->kept[j] = f;
                    ld.n_kept++;
                }
                ld.settled++;
            }
            if (ld.settled == n_sample) ld.done_s = since(t0);
            ld.cv.notify_all();
        }
    };
    struct Source {
        Loader *ld;
        qa_bam_range_result *R;
        const std::vector<Loaded> *common, *all_snps;
        int n_files;
        bool rare;
        static int acquire(void *ctx, int32_t s, qa_sample_view_t *v) {
            Source &S = *static_cast<Source *>(ctx);
            int f;
            {
                std::unique_lock<std::mutex> lk(S.ld->mu);
                S.ld->cv.wait(lk, [&] { return S.ld->n_kept > s || S.ld->settled == S.n_files || S.ld->status != QA_OK; });
                if (S.ld->status != QA_OK) { qa::set_error("%s", S.ld->err.c_str()); return S.ld->status; }
                if (S.ld->n_kept <= s) return QA_END_OF_SAMPLES;
                f = S.R->kept[(size_t)s];
            }
            const Loaded &c = (*S.common)[(size_t)f];
            static const int32_t none = 0;   // (a read-less base array is never dereferenced; the pointers must not be null)
            v->n_reads = c.R; v->read_ptr = c.read_ptr.data(); v->u = c.u.empty() ? &none : c.u.data(); v->bq = c.bq.empty() ? &none : c.bq.data();
            v->wif = c.wif.data();
            v->read_labels = S.R->labels_of[(size_t)s].data();
            if (S.rare) {
                const Loaded &a = (*S.all_snps)[(size_t)f];
                v->n_reads_all = a.R; v->read_ptr_all = a.read_ptr.data(); v->u_all = a.u.empty() ? &none : a.u.data();
                v->bq_all = a.bq.empty() ? &none : a.bq.data(); v->wif_all = a.wif.data();
            }
            return QA_OK;
        }
    } src{&ld, R.get(), &common, &all_snps, n_sample, rare};
    const qa_sample_source_t source{&Source::acquire, &src};
    std::vector<std::thread> loaders;
    for (int w = 0; w < std::max(1, std::min(n_io, n_sample)); w++) loaders.emplace_back(loader_body);
    auto join_loaders = [&] {
        ld.stop.store(true);
        for (auto &t : loaders) t.join();
        loaders.clear();
    };

    // ---- 2. the ONE call for every chain of every kept sample (n_sample = the files: an upper bound, the source ends the range)
    R->dosage.alloc((size_t)n_sample * T_out);   // (untouched pages of rows that no kept sample takes cost nothing)
    R->gp_t.alloc((size_t)n_sample * 3 * T_out);
    R->haps.alloc((size_t)n_sample * nL * T_out);
    R->nDosage.assign((size_t)std::max(n_sample, 1), 0);
    qa_impute_params_t P = *params;
    P.sample_index = index.data();
    P.sample_source = &source;
    qa_impute_rare_common_t rcq;
    qa_impute_nipt_t nq;
    if (rare) {
        rcq = *params->rare_common;
        rcq.read_off = rcq.read_ptr = rcq.u = rcq.bq = rcq.wif = nullptr;
        P.rare_common = &rcq;
    }
    if (nipt) {
        R->fet_dosage.alloc((size_t)n_sample * T_out);
        R->fet_gp_t.alloc((size_t)n_sample * 3 * T_out);
        nq = *params->nipt;
        nq.ff = ffk.data();
        nq.fet_dosage = R->fet_dosage.data();
        nq.fet_gp_t = R->fet_gp_t.data();
        P.nipt = &nq;
    }
    const int nk_max = n_sample;
    // ---- 3. (beside 2.) per kept sample: its VCF column (functions.R:1408-1463), eij / fij / max_gen and the pile-up's allele counts
    // (:1380-1418).  qa_impute_samples reports every launch set whose samples are final (params->on_samples_done); a pool of host
    // threads formats those samples while later launch sets are still on the device.
    R->col_buf.resize((size_t)nk_max);
    R->col_off.resize((size_t)nk_max);
    std::vector<std::vector<double>> eij((size_t)nk_max), fij((size_t)nk_max), ac((size_t)nk_max);
    std::vector<std::vector<uint8_t>> maxg((size_t)nk_max);
    R->discarded = io->discard_sample_arrays != 0;
    // the whole pages inside [p, p + n): their memory goes back to the system now (the addresses stay valid and read as zeros)
    auto give_back = [](double *p, size_t n) {
        static const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
        const uintptr_t lo = ((uintptr_t)p + page - 1) / page * page, hi = ((uintptr_t)(p + n)) / page * page;
        if (hi > lo) madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_DONTNEED);
    };
    auto format_one = [&](int j, std::string &e) -> int {
        const double *gp = R->gp_t.data() + (size_t)j * 3 * T_out;          // [3][T_out]
        const double *hd = R->haps.data() + (size_t)j * nL * T_out;         // [nL][T_out] == T_out x nL column-major
        std::vector<double> gpc((size_t)3 * T_out), fgc;                    // 3 x T_out column-major, as the column writers take it
        for (int t = 0; t < T_out; t++)
            for (int g = 0; g < 3; g++) gpc[(size_t)3 * t + g] = gp[(size_t)g * T_out + t];
        auto &buf = R->col_buf[(size_t)j];
        auto &off = R->col_off[(size_t)j];
        off.assign((size_t)T_out + 1, 0);
        int64_t need = 0, cap = (int64_t)48 * T_out + 64;
        for (int pass = 0; pass < 2; pass++) {
            buf.assign((size_t)cap, 0);
            int s1;
            if (nipt) {
                if (fgc.empty()) {
                    const double *fg = R->fet_gp_t.data() + (size_t)j * 3 * T_out;
                    fgc.resize((size_t)3 * T_out);
                    for (int t = 0; t < T_out; t++)
                        for (int g = 0; g < 3; g++) fgc[(size_t)3 * t + g] = fg[(size_t)g * T_out + t];
                }
                s1 = qa_vcf_column_nipt(T_out, gpc.data(), fgc.data(), hd, R->dosage.data() + (size_t)j * T_out,
                                        R->fet_dosage.data() + (size_t)j * T_out, buf.data(), cap, off.data(), &need);
            } else {
                s1 = qa_vcf_column_diploid(T_out, gpc.data(), hd, io->output_gt_phased_genotypes, buf.data(), cap, off.data(), &need);
            }
            if (s1 == QA_ERR_CAPACITY && pass == 0) { cap = need; continue; }
            if (s1 != QA_OK) { e = std::string("VCF column of a sample could not be formatted: ") + qa_last_error(); return s1; }
            break;
        }
        buf.resize((size_t)off[(size_t)T_out]);
        buf.shrink_to_fit();
        // eij, fij (functions.R:1399-1400: round(x, 3)), max_gen (STITCH::get_max_gen_rapid: the first maximum), the pile-up's
        // allele counts (increment2N over STITCH::convertScaledBQtoProbs of the reads as loaded, :1382-1398)
        auto &E = eij[(size_t)j]; auto &F = fij[(size_t)j]; auto &M = maxg[(size_t)j]; auto &A = ac[(size_t)j];
        E.resize((size_t)T_out); F.resize((size_t)T_out); M.resize((size_t)T_out); A.assign((size_t)2 * T_out, 0.0);
        for (int t = 0; t < T_out; t++) {
            const double g0 = gp[t], g1 = gp[(size_t)T_out + t], g2 = gp[(size_t)2 * T_out + t];
            E[(size_t)t] = std::nearbyint((g1 + 2 * g2) * 1000.0) / 1000.0;
            F[(size_t)t] = std::nearbyint((g1 + 4 * g2) * 1000.0) / 1000.0;
            M[(size_t)t] = (uint8_t)((g1 > g0) ? ((g2 > g1) ? 2 : 1) : ((g2 > g0) ? 2 : 0));
        }
        const int file = R->kept[(size_t)j];   // (settled before the sample was handed to the imputation)
        const Loaded &s = rare ? all_snps[(size_t)file] : common[(size_t)file];
        double *c1 = A.data(), *c2 = A.data() + T_out;   // sums of P(ref), P(alt) per site, bases in the order they were loaded
        for (size_t b = 0; b < s.u.size(); b++) {
            const int q = s.bq[b];
            const double eps = std::pow(10.0, -std::fabs((double)q) / 10.0);
            c1[s.u[b]] += q < 0 ? 1 - eps : eps / 3;
            c2[s.u[b]] += q < 0 ? eps / 3 : 1 - eps;
        }
        // the sample is final: its reads are not needed again (released here, on this thread, not in one sweep at the end)
        if (R->discarded) {   // nor are its result rows, for a caller that asked for columns, labels and counts only
            give_back(R->dosage.data() + (size_t)j * T_out, (size_t)T_out);
            give_back(R->gp_t.data() + (size_t)j * 3 * T_out, (size_t)3 * T_out);
            give_back(R->haps.data() + (size_t)j * nL * T_out, (size_t)nL * T_out);
            if (nipt) {
                give_back(R->fet_dosage.data() + (size_t)j * T_out, (size_t)T_out);
                give_back(R->fet_gp_t.data() + (size_t)j * 3 * T_out, (size_t)3 * T_out);
            }
        }
        common[(size_t)file] = Loaded();
        if (rare) all_snps[(size_t)file] = Loaded();
        return (int)QA_OK;
    };
    // The range's sums, per SNP over the samples IN SAMPLE ORDER, as the reference's loop adds them (quilt.R:955-961: floating-point
    // sums, so the order is part of the result).  Launch sets finish nearly in order: whichever formatter completes the next sample
    // in line adds it -- and every formatted sample behind it -- to the sums and releases its per-SNP vectors, beside the device
    // work; nothing is left to sum, and 2 MB per sample less to hand back, when the call ends.
    struct Sums {
        std::mutex mu;
        std::vector<uint8_t> formatted;
        int next = 0;
    } sums;
    sums.formatted.assign((size_t)nk_max, 0);
    auto add_in_order = [&](int j_done) {
        std::lock_guard<std::mutex> g(sums.mu);
        sums.formatted[(size_t)j_done] = 1;
        double *i0 = R->infoCount.data(), *i1 = i0 + T_out, *af = R->afCount.data(), *hw = R->hweCount.data();
        double *a0 = R->alleleCount.data(), *a1 = a0 + T_out;
        while (sums.next < nk_max && sums.formatted[(size_t)sums.next]) {
            const size_t j = (size_t)sums.next++;
            const double *E = eij[j].data(), *F = fij[j].data(), *c1 = ac[j].data(), *c2 = c1 + T_out;
            const uint8_t *M = maxg[j].data();
            for (int t = 0; t < T_out; t++) {
                i0[t] += E[t];
                i1[t] += F[t] - E[t] * E[t];
                af[t] += E[t] / 2;
                hw[(size_t)M[t] * T_out + t] += 1;
                a0[t] += c2[t];              // per_sample_alleleCount = cbind(c2, c1 + c2) (functions.R:1398)
                a1[t] += c1[t] + c2[t];
            }
            std::vector<double>().swap(eij[j]); std::vector<double>().swap(fij[j]); std::vector<double>().swap(ac[j]);
            std::vector<uint8_t>().swap(maxg[j]);
        }
    };
    struct Pool {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<int> queue;
        size_t head = 0;
        bool closed = false;
        int status = QA_OK;
        std::string err;
        double busy_s = 0;
    } pool;
    auto pool_body = [&] {
        for (;;) {
            int j;
            {
                std::unique_lock<std::mutex> lk(pool.mu);
                pool.cv.wait(lk, [&] { return pool.head < pool.queue.size() || pool.closed; });
                if (pool.head >= pool.queue.size()) return;
                j = pool.queue[pool.head++];
            }
            const auto tj = Clock::now();
            std::string e;
            int s1;
            try { s1 = format_one(j, e); } catch (const std::exception &ex) { s1 = QA_ERR_INVALID; e = ex.what(); }
            if (s1 == QA_OK) add_in_order(j);
            std::lock_guard<std::mutex> g(pool.mu);
            pool.busy_s += since(tj);
            if (s1 != QA_OK && pool.status == QA_OK) { pool.status = s1; pool.err = e; }
        }
    };
    struct Hook {
        Pool *pool;
        static void done(void *ctx, int32_t lo, int32_t hi) {
            Pool *p = static_cast<Hook *>(ctx)->pool;
            {
                std::lock_guard<std::mutex> g(p->mu);
                for (int j = lo; j < hi; j++) p->queue.push_back(j);
            }
            p->cv.notify_all();
        }
    } hook{&pool};
    P.on_samples_done = &Hook::done;
    P.on_samples_done_ctx = &hook;
    std::vector<std::thread> formatters;
    for (int w = 0; w < std::max(1, std::min(n_io, nk_max)); w++) formatters.emplace_back(pool_body);
    auto close_pool = [&] {
        { std::lock_guard<std::mutex> g(pool.mu); pool.closed = true; }
        pool.cv.notify_all();
        for (auto &t : formatters) t.join();
    };
    tr_setup = since(t0);
    int st = QA_OK;
    std::string err;
    if (n_sample > 0)
        st = impute(&P, n_sample, nullptr, nullptr, nullptr, nullptr, nullptr, R->dosage.data(), R->gp_t.data(), R->haps.data(), nullptr,
                    R->nDosage.data(), R->stats);
    join_loaders();
    if (ld.status != QA_OK) { close_pool(); qa::set_error("qa_impute_bam_range: %s", ld.err.c_str()); return ld.status; }
    if (st != QA_OK) { close_pool(); return st; }   // (qa_last_error holds qa_impute_samples' text)
    const int nk = R->n_kept = ld.n_kept;
    R->kept.resize((size_t)nk);
    R->seconds[0] = ld.done_s;
    R->seconds[1] = since(t0);
    tr_call = R->seconds[1] - tr_setup;
    t0 = Clock::now();
    close_pool();   // (what is still queued when the device work ends: the last launch sets' samples)
    tr_drain = since(t0);
    if (pool.status != QA_OK) { qa::set_error("qa_impute_bam_range: %s", pool.err.c_str()); return pool.status; }
    if ((int)pool.queue.size() != nk) { qa::set_error("qa_impute_bam_range: %d of %d samples were reported final", (int)pool.queue.size(), nk); return QA_ERR_INVALID; }
    if (sums.next != nk) { qa::set_error("qa_impute_bam_range: %d of %d samples were added to the range's sums", sums.next, nk); return QA_ERR_INVALID; }
    R->format_busy_s = pool.busy_s;
    R->seconds[2] = since(t0);
    tr_sums = R->seconds[2] - tr_drain;
    R->seconds[3] = since(t_all);
    if (trace)
        std::fprintf(stderr, "qa_impute_bam_range: %d files (%d kept), %d host threads: last file loaded at %.3f s (beside the imputation), setup %.3f, "
                     "qa_impute_samples %.3f, formatters' drain %.3f (busy %.3f thread-s), sums %.3f, whole call %.3f\n", n_sample, nk, n_io,
                     R->seconds[0], tr_setup, tr_call, tr_drain, pool.busy_s, tr_sums, R->seconds[3]);
    *out = R.release();
    return QA_OK;
}

}  // namespace

extern "C" {

int qa_impute_bam_range(qa_panel_t *const *panels, int32_t n_panels, const qa_impute_params_t *params, const qa_bam_range_io_t *io,
                        int32_t n_sample, const char *const *bam_paths, const int64_t *sample_index, const double *ff,
                        qa_bam_range_result_t **out) {
    if (!panels || n_panels < 1 || !panels[0]) {
        if (out) *out = nullptr;
        qa::set_error("qa_impute_bam_range: no panel handle");
        return QA_ERR_INVALID;
    }
    return bam_range_impl(
        [&](const qa_impute_params_t *P, int32_t n, const int32_t *ro, const int32_t *rp, const int32_t *u, const int32_t *bq, const int32_t *wif,
            double *dosage, double *gp_t, double *haps, int32_t *labels, int32_t *nDosage, int64_t *stats) {
            return qa_impute_samples(panels, n_panels, P, n, 0, ro, rp, u, bq, wif, dosage, gp_t, haps, labels, nDosage, stats);
        },
        params, io, n_sample, bam_paths, sample_index, ff, out);
}

// test hook (impute_testhook.h): the same host code -- loader, kept-sample bookkeeping, formatting, counts -- with the imputation
// running over a checker's entry points instead of the device
int qa_impute_bam_range_backend(const qa_impute_backend_t *backend, void *const *handles, int32_t n_handles, int32_t K, int32_t nGrids,
                                const qa_impute_params_t *params, const qa_bam_range_io_t *io, int32_t n_sample, const char *const *bam_paths,
                                const int64_t *sample_index, const double *ff, qa_bam_range_result_t **out) {
    if (!backend || !handles || !io) {
        if (out) *out = nullptr;
        qa::set_error("qa_impute_bam_range_backend: missing argument");
        return QA_ERR_INVALID;
    }
    const int32_t T = io->nSNPs;
    return bam_range_impl(
        [&](const qa_impute_params_t *P, int32_t n, const int32_t *ro, const int32_t *rp, const int32_t *u, const int32_t *bq, const int32_t *wif,
            double *dosage, double *gp_t, double *haps, int32_t *labels, int32_t *nDosage, int64_t *stats) {
            return qa_impute_samples_backend(backend, handles, n_handles, K, nGrids, T, P, n, 0, ro, rp, u, bq, wif, dosage, gp_t, haps, labels,
                                             nDosage, stats);
        },
        params, io, n_sample, bam_paths, sample_index, ff, out);
}

int32_t qa_bam_range_n_samples(const qa_bam_range_result_t *r) { return r ? r->n : 0; }
int32_t qa_bam_range_n_snps(const qa_bam_range_result_t *r) { return r ? r->T_out : 0; }
int32_t qa_bam_range_imputed(const qa_bam_range_result_t *r, int32_t i) { return (r && i >= 0 && i < r->n) ? r->imputed[(size_t)i] : 0; }
int32_t qa_bam_range_n_reads(const qa_bam_range_result_t *r, int32_t i) { return (r && i >= 0 && i < r->n) ? r->n_reads[(size_t)i] : 0; }

int qa_bam_range_column(const qa_bam_range_result_t *r, int32_t i, const char **buf, const int64_t **off) {
    if (!r || i < 0 || i >= r->n || !buf || !off) return QA_ERR_INVALID;
    const int j = r->slot[(size_t)i];
    *buf = j < 0 ? nullptr : r->col_buf[(size_t)j].data();
    *off = j < 0 ? nullptr : r->col_off[(size_t)j].data();
    return QA_OK;
}

int qa_bam_range_sample(const qa_bam_range_result_t *r, int32_t i, const double **dosage, const double **gp_t, const double **phasing_haps,
                        const double **fet_dosage, const double **fet_gp_t, const int32_t **read_labels, int32_t *n_labels, int32_t *nDosage) {
    if (!r || i < 0 || i >= r->n) return QA_ERR_INVALID;
    const int j = r->slot[(size_t)i];
    const size_t T = (size_t)r->T_out;
    if (dosage) *dosage = (j < 0 || r->discarded) ? nullptr : r->dosage.data() + (size_t)j * T;
    if (gp_t) *gp_t = (j < 0 || r->discarded) ? nullptr : r->gp_t.data() + (size_t)j * 3 * T;
    if (phasing_haps) *phasing_haps = (j < 0 || r->discarded) ? nullptr : r->haps.data() + (size_t)j * r->nL * T;
    if (fet_dosage) *fet_dosage = (j < 0 || !r->nipt || r->discarded) ? nullptr : r->fet_dosage.data() + (size_t)j * T;
    if (fet_gp_t) *fet_gp_t = (j < 0 || !r->nipt || r->discarded) ? nullptr : r->fet_gp_t.data() + (size_t)j * 3 * T;
    if (read_labels) *read_labels = j < 0 ? nullptr : r->labels_of[(size_t)j].data();
    if (n_labels) *n_labels = j < 0 ? 0 : (int32_t)r->labels_of[(size_t)j].size();
    if (nDosage) *nDosage = j < 0 ? 0 : r->nDosage[(size_t)j];
    return QA_OK;
}

int qa_bam_range_counts(const qa_bam_range_result_t *r, double *infoCount, double *afCount, double *hweCount, double *alleleCount) {
    if (!r) return QA_ERR_INVALID;
    const size_t T = (size_t)r->T_out;
    if (infoCount) std::memcpy(infoCount, r->infoCount.data(), sizeof(double) * 2 * T);
    if (afCount) std::memcpy(afCount, r->afCount.data(), sizeof(double) * T);
    if (hweCount) std::memcpy(hweCount, r->hweCount.data(), sizeof(double) * 3 * T);
    if (alleleCount) std::memcpy(alleleCount, r->alleleCount.data(), sizeof(double) * 2 * T);
    return QA_OK;
}

void qa_bam_range_timings(const qa_bam_range_result_t *r, double seconds[4], int64_t impute_stats[11], int64_t load_stats[8]) {
    if (!r) return;
    if (seconds) std::memcpy(seconds, r->seconds, sizeof r->seconds);
    if (impute_stats) std::memcpy(impute_stats, r->stats, sizeof r->stats);
    if (load_stats) std::memcpy(load_stats, r->load_stats, sizeof r->load_stats);
}

void qa_bam_range_destroy(qa_bam_range_result_t *r) { delete r; }

}  // extern "C"
