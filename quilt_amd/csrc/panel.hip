// panel.hip -- library state, error reporting and panel upload.
#include <chrono>
#include "panel.hpp"

#include <mutex>

#include <algorithm>
#include <cmath>
#include <memory>

namespace qa {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static int usable_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

bool device_ready() {
    static thread_local int cached = -1;
    if (cached < 0) cached = usable_devices() > 0 ? 1 : 0;
    if (!cached) set_error("no usable gfx950 (MI355X) device: libquilt_amd has no CPU fallback");
    return cached == 1;
}

DeviceGate &device_gate(int device) {
    // never destroyed: a static's destructor would run after the HIP runtime may already be gone (hipFree at exit); the
    // arena itself is freed by gate_user when the last handle that opted in goes
    static DeviceGate *const gates = new DeviceGate[16];
    return gates[device >= 0 && device < 16 ? device : 0];
}
// handles that opted in; the last one out frees the device-wide arena
void gate_user(int device, int delta) {
    DeviceGate &g = device_gate(device);
    std::lock_guard<std::mutex> lk(g.mu);
    g.users += delta;
    if (g.users <= 0 && g.active == 0 && g.queue.empty()) {
        g.users = 0;
        if (g.arena.base) { (void)hipFree(g.arena.base); g.arena.base = nullptr; g.arena.cap = g.arena.off = 0; }
    }
}
double GateHold::now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the reference's clamped halving search over rows s1..e1 (1-based) of the special matrix
// (QUILT/src/gibbs-small.cpp:69-105), quirks included: this is how the device tables inherit
// exactly the word the reference would decode.
int32_t reference_matrix_search(int val, const int32_t *mat, int nrow, int s1, int e1) {
    int nori = e1 - s1 + 1;
    if (nori == 1) return 0;
    int n = nori, i = n / 2;
    n /= 4;
    for (int c = 0; c < 100; c++) {
        int32_t key = mat[s1 - 1 + i];
        if (key == val) return mat[(size_t)nrow + s1 - 1 + i];
        i += (key < val) ? n : -n;
        n = std::max(n / 2, 1);
        i = std::min(std::max(i, 0), nori - 1);
    }
    return mat[(size_t)nrow + s1];
}

struct ProfileSlot {
    double ms = 0, bytes = 0, units = 0, serial = 0, workgroups = 0;
    long long launches = 0;
    std::vector<std::pair<double, double>> busy;   // [start, end) of every launch, ms since the process's reference event
};
static ProfileSlot g_profile[PK_COUNT];
static std::mutex g_profile_mutex;   // several host threads (one panel handle each) may share the device

static hipEvent_t g_ref_event = nullptr;

// ms between the process-wide reference event and a completed event (any stream)
double profile_clock_ms(hipEvent_t ev) {
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    if (!g_ref_event) {
        if (hipEventCreate(&g_ref_event) != hipSuccess) return 0;
        (void)hipEventRecord(g_ref_event, nullptr);
        (void)hipEventSynchronize(g_ref_event);
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, g_ref_event, ev) != hipSuccess) return 0;
    return ms;
}

void profile_add(int kernel, double ms, double alg_bytes, double start_ms, double units, double serial, double workgroups) {
    if (kernel < 0 || kernel >= PK_COUNT) return;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    g_profile[kernel].ms += ms;
    g_profile[kernel].bytes += alg_bytes;
    g_profile[kernel].units += units;
    g_profile[kernel].serial += serial;
    g_profile[kernel].workgroups += workgroups;
    g_profile[kernel].launches += 1;
    if (start_ms >= 0) g_profile[kernel].busy.emplace_back(start_ms, start_ms + ms);
}

static const char *const kProfileNames[PK_COUNT] = {
    "k_emat", "k_fwd", "k_bwd", "k_dosage", "k_ematread", "k_gibbs", "k_happrobs", "k_fwd64", "k_bwd64", "k_topk",
    "k_fwd<double>", "k_bwd<double>", "k_gibbs3", "k_block3", "k_select", "k_best_run", "k_fwd64d", "k_bwd64d",
    "k_gibbs<10, 1, true>"};   // (the sampler's 256-register build, two chains per SIMD: different code, priced by itself)

// sp_gidx / sp_chunk_at (see panel.hpp): one thread per (grid with specials, 16-haplotype chunk), lower bound of the
// chunk's first haplotype in the grid's ascending special list, shifted into the padded per-pass layout
__global__ void k_special_chunk_index(const int32_t *sp_off, const int32_t *sp_k, const int32_t *sp_grids, int n_chunks,
                                      int32_t *chunk_at) {
    const int i = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int g = sp_grids[i];
    int lo = sp_off[g], hi = sp_off[g + 1];
    const int k0 = c * 16;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sp_k[mid] < k0) lo = mid + 1; else hi = mid;
    }
    chunk_at[(size_t)i * n_chunks + c] = lo + 16 * i;
}

void finish_panel_tables(qa_panel *p) {
    const int G = p->G;
    std::vector<int32_t> gidx(G, -1), grids;
    for (int g = 0; g < G; g++)
        if (p->h_sp_off[g + 1] > p->h_sp_off[g]) { gidx[g] = (int32_t)grids.size(); grids.push_back(g); }
    p->n_sp_grids = (int)grids.size();
    p->sp_gidx.alloc(G);
    p->sp_gidx.upload(gidx.data(), G, p->stream);
    // chunks up to the largest padded K any kernel geometry uses (rows of 8192 haplotypes: fullpass64.hip)
    const int n_chunks = (p->K + 8191) / 8192 * 512;
    p->sp_chunk_at.alloc(std::max<size_t>((size_t)p->n_sp_grids * n_chunks, 1));
    if (p->n_sp_grids > 0) {
        DBuf<int32_t> d_grids(grids.size());
        d_grids.upload(grids.data(), grids.size(), p->stream);
        hipLaunchKernelGGL(k_special_chunk_index, dim3((n_chunks + 255) / 256, p->n_sp_grids), dim3(256), 0, p->stream,
                           p->sp_off.p, p->sp_k.p, d_grids.p, n_chunks, p->sp_chunk_at.p);
        QA_HIP(hipGetLastError());
        QA_HIP(hipStreamSynchronize(p->stream));
    }
    p->tm1.alloc(std::max(G - 1, 1));
    p->tm1.upload(p->h_tm1.data(), std::max(G - 1, 0), p->stream);
    QA_HIP(hipStreamSynchronize(p->stream));
}

}  // namespace qa

extern "C" {

int qa_profile_count(void) { return qa::PK_COUNT; }

const char *qa_profile_name(int32_t kernel) {
    return (kernel >= 0 && kernel < qa::PK_COUNT) ? qa::kProfileNames[kernel] : "";
}

int qa_profile_get_workgroups(int32_t kernel, double *workgroups) {
    if (kernel < 0 || kernel >= qa::PK_COUNT) return QA_ERR_INVALID;
    std::lock_guard<std::mutex> lock(qa::g_profile_mutex);
    if (workgroups) *workgroups = qa::g_profile[kernel].workgroups;
    return QA_OK;
}

int qa_profile_get_work(int32_t kernel, double *units, double *serial) {
    if (kernel < 0 || kernel >= qa::PK_COUNT) return QA_ERR_INVALID;
    std::lock_guard<std::mutex> lock(qa::g_profile_mutex);
    if (units) *units = qa::g_profile[kernel].units;
    if (serial) *serial = qa::g_profile[kernel].serial;
    return QA_OK;
}

void *qa_host_alloc(size_t bytes) {
    if (!qa::device_ready()) return nullptr;
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
        qa::set_error("qa_host_alloc: cannot pin %zu bytes of host memory", bytes);
        return nullptr;
    }
    qa::pinned_registry().add(p, bytes);
    return p;
}

int qa_host_free(void *p) {
    if (!p || !qa::pinned_registry().remove(p)) {
        qa::set_error("qa_host_free: not a qa_host_alloc buffer");
        return QA_ERR_INVALID;
    }
    (void)hipHostFree(p);
    return QA_OK;
}

int qa_selftest_copy_rate(int32_t to_device, size_t bytes, int32_t mode, double *ms) {
    // diagnostic: one transfer of `bytes` between a device buffer and a pinned host buffer; mode 0 = the library's copy
    // kernel writing / reading the host buffer directly, 1 = hipMemcpyAsync, 2 = the staged path into pageable memory
    if (!qa::device_ready() || !ms || bytes == 0) return QA_ERR_INVALID;
    try {
        void *d = nullptr, *h = nullptr;
        QA_HIP(hipMalloc(&d, bytes));
        std::vector<char> pageable;
        if (mode == 2) { pageable.resize(bytes); h = pageable.data(); }
        else QA_HIP(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        memset(h, 1, bytes);
        hipStream_t st;
        QA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            QA_HIP(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            if (mode == 0) {
                if (to_device) qa::stage_copy(d, h, bytes, st); else qa::stage_copy(h, d, bytes, st);
                QA_HIP(hipGetLastError());
            } else if (mode == 1) {
                QA_HIP(hipMemcpyAsync(to_device ? d : h, to_device ? h : d, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
            } else {
                if (to_device) qa::staged_upload(d, h, bytes, st); else qa::staged_download(h, d, bytes, st);
            }
            QA_HIP(hipStreamSynchronize(st));
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        *ms = best;
        (void)hipStreamDestroy(st);
        (void)hipFree(d);
        if (mode != 2) (void)hipHostFree(h);
        return QA_OK;
    } catch (const std::exception &e) {
        qa::set_error("%s", e.what());
        return QA_ERR_HIP;
    }
}

int qa_profile_reset(void) {
    std::lock_guard<std::mutex> lock(qa::g_profile_mutex);
    for (auto &s : qa::g_profile) s = qa::ProfileSlot();
    return QA_OK;
}

int qa_profile_get(int32_t kernel, double *ms, int64_t *launches, double *alg_bytes) {
    if (kernel < 0 || kernel >= qa::PK_COUNT) return QA_ERR_INVALID;
    std::lock_guard<std::mutex> lock(qa::g_profile_mutex);
    if (ms) *ms = qa::g_profile[kernel].ms;
    if (launches) *launches = qa::g_profile[kernel].launches;
    if (alg_bytes) *alg_bytes = qa::g_profile[kernel].bytes;
    return QA_OK;
}

int qa_profile_get_busy(int32_t kernel, double *busy_ms) {
    if (kernel < 0 || kernel >= qa::PK_COUNT || !busy_ms) return QA_ERR_INVALID;
    std::lock_guard<std::mutex> lock(qa::g_profile_mutex);
    auto iv = qa::g_profile[kernel].busy;
    std::sort(iv.begin(), iv.end());
    double tot = 0, cs = 0, ce = -1;
    for (auto &x : iv) {
        if (x.first > ce) { if (ce > cs) tot += ce - cs; cs = x.first; ce = x.second; }
        else ce = std::max(ce, x.second);
    }
    if (ce > cs) tot += ce - cs;
    *busy_ms = tot;
    return QA_OK;
}

int qa_abi_version(void) { return 5; }   // 5: qa_impute_params_t.sample_index, qa_impute_bam_range, qa_panel_set_sum_order mode 2 (round 6); 4: qa_gibbs_opts_t.reads_same_as, qa_panel_set_sum_order, qa_rcpp_make_eMatRead_t_rare_common (round 5)

const char *qa_last_error(void) { return qa::g_err; }

int qa_device_count(void) { return qa::usable_devices(); }

int qa_set_device(int device) {
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(device));
        return QA_OK;
    });
}

int qa_panel_create(const qa_panel_desc_t *d, qa_panel_t **out) {
    if (!out) return QA_ERR_INVALID;
    *out = nullptr;
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    if (!d || d->K <= 0 || d->nGrids <= 0 || d->nSNPs <= 0 || d->nMaxDH <= 0 || !d->distinctHapsB ||
        !d->transMatRate_t || !d->eMatDH_special_grid_which || (!d->hapMatcherR && !d->hapMatcher)) {
        qa::set_error("qa_panel_create: missing panel table");
        return QA_ERR_INVALID;
    }
    if (d->nMaxDH > 255) {
        qa::set_error("qa_panel_create: nMaxDH = %d > 255 is not supported (hapMatcherR layout only)", d->nMaxDH);
        return QA_ERR_UNSUPPORTED;
    }
    if ((d->nSNPs + 31) / 32 != d->nGrids) {
        qa::set_error("qa_panel_create: nGrids must be ceil(nSNPs / 32)");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        auto *p = new qa_panel();
        std::unique_ptr<qa_panel> guard(p);
        QA_HIP(hipGetDevice(&p->device));
        QA_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        const int K = d->K, G = d->nGrids;
        p->K = K; p->G = G; p->T = d->nSNPs; p->nMaxDH = d->nMaxDH; p->nrow = d->nMaxDH + 1;
        p->Kp = (K + 8191) / 8192 * 8192;   // whole 512-lane x 16-haplotype chunk rows (zero padded): see panel.hpp
        p->ref_error = d->ref_error;
        // hapMatcher -> uint8 [G][Kp]
        p->hm.alloc((size_t)G * p->Kp);
        p->hm.zero(p->stream);
        std::vector<uint8_t> tmp;
        const uint8_t *src = d->hapMatcherR;
        if (!src) {
            tmp.resize((size_t)K * G);
            for (size_t i = 0; i < (size_t)K * G; i++) {
                int32_t v = d->hapMatcher[i];
                if (v < 0 || v > 255) throw std::runtime_error("hapMatcher value outside 0..255");
                tmp[i] = (uint8_t)v;
            }
            src = tmp.data();
        }
        QA_HIP(hipMemcpy2DAsync(p->hm.p, p->Kp, src, K, K, G, hipMemcpyHostToDevice, p->stream));
        p->B.alloc((size_t)d->nMaxDH * G);
        p->B.upload(d->distinctHapsB, (size_t)d->nMaxDH * G, p->stream);
        // specials
        std::vector<int32_t> off(G + 1, 0), sk;
        std::vector<uint32_t> sw;
        for (int g = 0; g < G; g++) {
            off[g] = (int32_t)sk.size();
            if (d->eMatDH_special_grid_which[g] <= 0) continue;
            if (d->use_eMatDH_special_symbols) {
                if (!d->eMatDH_special_matrix || !d->eMatDH_special_matrix_helper)
                    throw std::runtime_error("special symbols requested without the special matrix");
                int s1 = d->eMatDH_special_matrix_helper[g];
                int e1 = d->eMatDH_special_matrix_helper[(size_t)G + g];
                for (int r = s1; r <= e1; r++) {
                    int k = d->eMatDH_special_matrix[r - 1];
                    sk.push_back(k);
                    sw.push_back((uint32_t)qa::reference_matrix_search(
                        k, d->eMatDH_special_matrix, d->eMatDH_special_matrix_nrow, s1, e1));
                }
            } else {
                if (!d->rhb_t) throw std::runtime_error("special haplotypes need rhb_t or the special matrix");
                for (int k = 0; k < K; k++) {
                    if (src[(size_t)K * g + k] == 0) {
                        sk.push_back(k);
                        sw.push_back((uint32_t)d->rhb_t[(size_t)K * g + k]);
                    }
                }
            }
        }
        off[G] = (int32_t)sk.size();
        p->n_special = (int)sk.size();
        p->h_sp_off = off;
        p->sp_off.alloc(G + 1);
        p->sp_off.upload(off.data(), G + 1, p->stream);
        p->sp_k.alloc(std::max<size_t>(sk.size(), 1));
        p->sp_word.alloc(std::max<size_t>(sw.size(), 1));
        p->sp_k.upload(sk.data(), sk.size(), p->stream);
        p->sp_word.upload(sw.data(), sw.size(), p->stream);
        // transitions
        p->h_sigma.resize(std::max(G - 1, 1));
        p->h_tm1.resize(std::max(G - 1, 1));
        for (int g = 0; g < G - 1; g++) {
            p->h_sigma[g] = d->transMatRate_t[2 * (size_t)g];
            p->h_tm1[g] = d->transMatRate_t[2 * (size_t)g + 1];
        }
        p->sigma.alloc(std::max(G - 1, 1));
        p->sigma.upload(p->h_sigma.data(), std::max(G - 1, 0), p->stream);
        // distinctHapsIE: by construction (STITCH make_rhb_t_equality) the per-SNP expansion of
        // distinctHapsB with ref_error / 1 - ref_error.  Verify; keep a device copy only if not.
        p->ie_derived = true;
        if (d->distinctHapsIE) {
            const double lo = d->ref_error, hi = 1 - d->ref_error;
            for (int t = 0; t < d->nSNPs && p->ie_derived; t++) {
                int g = t / 32, b = t % 32;
                for (int r = 0; r < d->nMaxDH; r++) {
                    uint32_t w = (uint32_t)d->distinctHapsB[(size_t)d->nMaxDH * g + r];
                    double want = ((w >> b) & 1u) ? hi : lo;
                    if (d->distinctHapsIE[(size_t)d->nMaxDH * t + r] != want) { p->ie_derived = false; break; }
                }
            }
            if (!p->ie_derived) {
                p->IE.alloc((size_t)d->nMaxDH * d->nSNPs);
                p->IE.upload(d->distinctHapsIE, (size_t)d->nMaxDH * d->nSNPs, p->stream);
            }
        }
        QA_HIP(hipStreamSynchronize(p->stream));
        qa::finish_panel_tables(p);
        *out = guard.release();
        return QA_OK;
    });
}

extern "C" void qa_impute_drop_handle_buffers(void *handle);   // impute.cpp: the per-handle buffers of qa_impute_samples

void qa_panel_destroy(qa_panel_t *panel) {
    if (panel) qa_impute_drop_handle_buffers(panel);   // (pinned host memory: freed while the runtime is alive)
    if (panel && panel->exclusive) {
        qa::drop_pass_scratch(panel);
        panel->exclusive = false;
        (void)hipSetDevice(panel->device);
        qa::gate_user(panel->device, -1);
    }
    delete panel;
}

int qa_panel_set_device_share(qa_panel_t *panel, int32_t n_sharers) {
    if (!panel || n_sharers < 1 || n_sharers > 16) {
        qa::set_error("qa_panel_set_device_share: n_sharers must be 1..16");
        return QA_ERR_INVALID;
    }
    panel->share = n_sharers;
    return QA_OK;
}

int qa_panel_set_exclusive(qa_panel_t *panel, int32_t on) {
    if (!panel) {
        qa::set_error("qa_panel_set_exclusive: null handle");
        return QA_ERR_INVALID;
    }
    if ((on != 0) != panel->exclusive) {
        qa::drop_pass_scratch(panel);   // its buffers are views of the arena in use
        panel->exclusive = on != 0;
        qa::gate_user(panel->device, on ? 1 : -1);
    }
    return QA_OK;
}

int qa_gate_stats(int32_t device, double out[7]) {
    if (!out || device < 0 || device >= 16) return QA_ERR_INVALID;
    qa::DeviceGate &g = qa::device_gate(device);
    std::lock_guard<std::mutex> lk(g.mu);
    out[0] = g.held_ms; out[1] = g.wait_ms; out[2] = (double)g.n_holds;
    out[3] = g.excl_ms; out[4] = g.slot_ms; out[5] = (double)g.n_shared; out[6] = (double)g.slots_total;
    return QA_OK;
}

int qa_gate_stats_reset(int32_t device) {
    if (device < 0 || device >= 16) return QA_ERR_INVALID;
    qa::DeviceGate &g = qa::device_gate(device);
    std::lock_guard<std::mutex> lk(g.mu);
    g.held_ms = g.wait_ms = g.excl_ms = g.slot_ms = 0; g.n_holds = g.n_shared = g.slots_total = 0;
    return QA_OK;
}

// the admission rules on a gate of its own, without touching a device (no arena bytes are asked for)
int qa_gate_selftest(void) {
    return qa::guarded([&] {
        qa::DeviceGate gate;
        std::vector<int> order;
        std::mutex om;
        auto note = [&](int who) { std::lock_guard<std::mutex> lk(om); order.push_back(who); };
        auto queued = [&](size_t n) {
            for (int spin = 0; spin < 20000; spin++) {
                { std::lock_guard<std::mutex> lk(gate.mu); if (gate.queue.size() >= n) return true; }
                std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
            return false;
        };
        qa::Arena own;
        qa::GateHold a;
        a.acquire(&gate, &own, 600, 0);                 // a Gibbs launch on 600 SIMDs
        qa::GateHold fits;
        fits.acquire(&gate, &own, 400, 0);              // one that fits beside it: admitted at once
        if (gate.active != 2 || gate.used_slots != 1000) throw std::runtime_error("gate selftest: co-running launches");
        fits.release();
        std::thread tb([&] { qa::GateHold h; h.acquire(&gate, &own, 1024, 0); note(2); });          // does not fit: waits
        if (!queued(1)) throw std::runtime_error("gate selftest: queue");
        std::thread tc([&] { qa::GateHold h; h.acquire(&gate, &own, 0, 0); note(3); });             // a full-panel set, behind it
        if (!queued(2)) throw std::runtime_error("gate selftest: queue");
        std::thread td([&] { qa::GateHold h; h.acquire(&gate, &own, 100, 0); note(4); });           // would fit now, but is not first
        if (!queued(3)) throw std::runtime_error("gate selftest: queue");
        std::thread te([&] { qa::GateHold h; h.acquire(&gate, &own, 0, 0, true); note(5); });       // express: before all of them
        if (!queued(4)) throw std::runtime_error("gate selftest: queue");
        if (gate.active != 1) throw std::runtime_error("gate selftest: a later launch overtook the queue");
        a.release();
        tb.join(); tc.join(); td.join(); te.join();
        if (order != std::vector<int>{5, 2, 3, 4}) throw std::runtime_error("gate selftest: admission order");
        if (gate.active != 0 || gate.used_slots != 0 || !gate.queue.empty() || gate.n_holds != 6)
            throw std::runtime_error("gate selftest: bookkeeping");
        return (int)QA_OK;
    });
}

int qa_gate_trace(int32_t device, int32_t on, double *rows, int32_t cap_rows) {
    if (device < 0 || device >= 16 || cap_rows < 0 || (cap_rows && !rows)) return QA_ERR_INVALID;
    qa::DeviceGate &g = qa::device_gate(device);
    std::lock_guard<std::mutex> lk(g.mu);
    const int n = (int)std::min<size_t>(g.trace.size(), (size_t)cap_rows);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 6; j++) rows[(size_t)i * 6 + j] = g.trace[i][j];
    const int total = (int)g.trace.size();
    if (rows || !on) g.trace.clear();
    g.tracing = on != 0;
    return total;
}

int qa_panel_set_cu_partition(qa_panel_t *panel, int32_t index, int32_t count) {
    if (!panel || count < 1 || count > 8 || index < 0 || index >= count) {
        qa::set_error("qa_panel_set_cu_partition: need 0 <= index < count <= 8");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(panel->device));
        if (panel->gibbs_stream) { QA_HIP(hipStreamDestroy(panel->gibbs_stream)); panel->gibbs_stream = nullptr; }
        if (count == 1) return (int)QA_OK;
        hipDeviceProp_t prop;
        QA_HIP(hipGetDeviceProperties(&prop, panel->device));
        const int n_cu = prop.multiProcessorCount;
        const int lo = (int)((long)n_cu * index / count), hi = (int)((long)n_cu * (index + 1) / count);
        std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
        for (int cu = lo; cu < hi; cu++) mask[cu >> 5] |= 1u << (cu & 31);
        QA_HIP(hipExtStreamCreateWithCUMask(&panel->gibbs_stream, (uint32_t)mask.size(), mask.data()));
        return (int)QA_OK;
    });
}

int qa_panel_set_pass_priority(qa_panel_t *panel, int32_t on) {
    if (!panel) {
        qa::set_error("qa_panel_set_pass_priority: null handle");
        return QA_ERR_INVALID;
    }
    return qa::guarded([&] {
        QA_HIP(hipSetDevice(panel->device));
        if (panel->pass_stream) { QA_HIP(hipStreamDestroy(panel->pass_stream)); panel->pass_stream = nullptr; }
        if (!on) return (int)QA_OK;
        int least = 0, greatest = 0;
        QA_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        QA_HIP(hipStreamCreateWithPriority(&panel->pass_stream, hipStreamNonBlocking, greatest));
        return (int)QA_OK;
    });
}

int qa_panel_set_ranking_precision(qa_panel_t *panel, int32_t bits) {
    if (!panel || (bits != 32 && bits != 64)) {
        qa::set_error("qa_panel_set_ranking_precision: bits must be 32 or 64");
        return QA_ERR_INVALID;
    }
    panel->rank_fp64 = bits == 64;
    return QA_OK;
}

int qa_panel_set_dosage_precision(qa_panel_t *panel, int32_t bits) {
    if (!panel || (bits != 32 && bits != 64)) {
        qa::set_error("qa_panel_set_dosage_precision: bits must be 32 or 64");
        return QA_ERR_INVALID;
    }
    panel->dosage_fp64 = bits == 64;
    return QA_OK;
}

int qa_panel_set_sum_order(qa_panel_t *panel, int32_t reference_order) {
    if (!panel || reference_order < 0 || reference_order > 2) {
        qa::set_error("qa_panel_set_sum_order: reference_order must be 0, 1 or 2");
        return QA_ERR_INVALID;
    }
    panel->sum_order_ref = reference_order != 0;
    panel->sum_order_grid0_ltr = reference_order == 2;
    return QA_OK;
}

int qa_panel_get_dims(const qa_panel_t *panel, int32_t *K, int32_t *nGrids, int32_t *nSNPs) {
    if (!panel) {
        qa::set_error("qa_panel_get_dims: null handle");
        return QA_ERR_INVALID;
    }
    if (K) *K = panel->K;
    if (nGrids) *nGrids = panel->G;
    if (nSNPs) *nSNPs = panel->T;
    return QA_OK;
}

int qa_panel_bind_thread(const qa_panel_t *panel) {
    if (!panel) {
        qa::set_error("qa_panel_bind_thread: null handle");
        return QA_ERR_INVALID;
    }
    if (!qa::device_ready()) return QA_ERR_NO_DEVICE;
    return hipSetDevice(panel->device) == hipSuccess ? QA_OK : QA_ERR_HIP;
}

int qa_Rcpp_make_gl_bound(double *gl, double minGLValue, const int32_t *to_fix, int32_t n_to_fix) {
    // reference-single.cpp:68-94; O(n_to_fix) host arithmetic, not worth a launch
    if (!gl || (n_to_fix > 0 && !to_fix)) return QA_ERR_INVALID;
    for (int i = 0; i < n_to_fix; i++) {
        double *p = gl + 2 * (size_t)to_fix[i];
        double a = p[0], b = p[1];
        if (a > b) { b = b / a; a = 1; if (b < minGLValue) b = minGLValue; }
        else       { a = a / b; b = 1; if (a < minGLValue) a = minGLValue; }
        p[0] = a; p[1] = b;
    }
    return QA_OK;
}

}  // extern "C"
