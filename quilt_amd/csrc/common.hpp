// common.hpp -- shared host-side plumbing of libquilt_amd (error state, device buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <condition_variable>
#include <deque>
#include <thread>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/quilt_amd.h"

namespace qa {

void set_error(const char *fmt, ...);
bool device_ready();

// per-kernel accumulators (HIP-event time on the launch stream, launches, algorithmic HBM bytes)
enum ProfileKernel { PK_EMAT = 0, PK_FWD, PK_BWD, PK_DOSAGE, PK_EMATREAD, PK_GIBBS, PK_HAPPROBS, PK_FWD64, PK_BWD64, PK_TOPK,
                     PK_FWD64G, PK_BWD64G, PK_GIBBS3, PK_BLOCK3, PK_SELECT, PK_MATCH, PK_FWD64D, PK_BWD64D, PK_GIBBS_LEAN, PK_COUNT };
// units / serial: work units of the launch (Gibbs: read visits + grid steps over all chains) and the length of its serial
// chain (Gibbs: read visits + grid steps of the longest chain), for rates other than bytes per second
// workgroups: of the launch (a Gibbs launch: one per chain) -- what the counters' bytes per workgroup are scaled by
void profile_add(int kernel, double ms, double alg_bytes, double start_ms = -1, double units = 0, double serial = 0,
                 double workgroups = 0);
double profile_clock_ms(hipEvent_t completed_event);

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define QA_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            char _b[512];                                                                    \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                     __FILE__, __LINE__);                                                    \
            throw qa::HipError(_b);                                                          \
        }                                                                                    \
    } while (0)

// Host <-> device transfers go through a pinned, device-visible staging buffer (one per host thread) and are carried
// out by k_stage_copy below, a copy kernel with one-wave workgroups.  The runtime's own copies (pageable or pinned
// alike in this environment) are blit kernels with 512-thread workgroups, and those cannot be placed on a CU while Gibbs
// waves hold the register files of two of its SIMDs: a 100 MB upload then waits for the other host thread's whole
// Gibbs launch (0.6 s, seen in the rocprofv3 trace of round 1).  One-wave workgroups fit on the SIMDs left free.
template <typename V>
static __global__ __launch_bounds__(64) void k_stage_copy(V *dst, const V *src, size_t n, const unsigned char *src_tail,
                                                          unsigned char *dst_tail, int n_tail) {
    const size_t stride = (size_t)gridDim.x * 64;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
template <typename V>
inline void stage_copy_as(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const size_t n = bytes / sizeof(V);
    const int n_tail = (int)(bytes - n * sizeof(V));
    const int blocks = (int)std::min<size_t>(std::max<size_t>((n + 63) / 64, 1), 2048);
    hipLaunchKernelGGL(k_stage_copy<V>, dim3(blocks), dim3(64), 0, s, static_cast<V *>(dst), static_cast<const V *>(src), n,
                       static_cast<const unsigned char *>(src) + n * sizeof(V), static_cast<unsigned char *>(dst) + n * sizeof(V),
                       n_tail);
}
// QA_COPY_ENGINE=sdma: measurement switch -- the same transfers through the runtime's hipMemcpyAsync (the DMA engines where the
// runtime uses them for pinned memory) instead of the one-wave copy kernel; DESIGN.md 5 has what it measured
inline bool copy_by_runtime() {
    static const bool v = [] { const char *e = getenv("QA_COPY_ENGINE"); return e && !strcmp(e, "sdma"); }();
    return v;
}
inline void stage_copy(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (copy_by_runtime()) {
        (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s);
        return;
    }
    const uintptr_t a = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src);
    if ((a & 15) == 0) stage_copy_as<uint4>(dst, src, bytes, s);
    else if ((a & 7) == 0) stage_copy_as<uint2>(dst, src, bytes, s);
    else if ((a & 3) == 0) stage_copy_as<uint32_t>(dst, src, bytes, s);
    else stage_copy_as<unsigned char>(dst, src, bytes, s);
}
// Two pinned staging buffers per host thread (double buffering) and the events that say when the device is done with them.
struct PinnedStage {
    char *p = nullptr;
    size_t cap = 0;   // bytes per buffer
    hipEvent_t ev[2] = {nullptr, nullptr};
    ~PinnedStage() {
        if (p) (void)hipHostFree(p);
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    }
    char *get(size_t bytes) {
        if (bytes > cap) {
            if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
            const size_t want = std::max<size_t>(bytes, size_t(64) << 20);
            if (hipHostMalloc((void **)&p, 2 * want, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; return nullptr; }
            cap = want;
        }
        for (auto &e : ev)
            if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        return p;
    }
    char *buf(int i) const { return p + (size_t)i * cap; }
};
inline PinnedStage &pinned_stage() {
    thread_local PinnedStage s;
    return s;
}
constexpr size_t kStagePiece = size_t(64) << 20;

// Host buffers handed out by qa_host_alloc (include/quilt_amd.h): pinned and device-visible, so a transfer from / to one
// needs no staging -- the copy kernel reads / writes it directly.  (Callers with buffers of their own, e.g. R vectors, get
// the staged path.)
struct PinnedRegistry {
    std::mutex mu;
    std::map<uintptr_t, size_t> regions;   // base -> bytes
    void add(void *p, size_t n) { std::lock_guard<std::mutex> g(mu); regions[reinterpret_cast<uintptr_t>(p)] = n; }
    bool remove(void *p) { std::lock_guard<std::mutex> g(mu); return regions.erase(reinterpret_cast<uintptr_t>(p)) > 0; }
    bool covers(const void *p, size_t n) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        std::lock_guard<std::mutex> g(mu);
        if (regions.empty()) return false;
        auto it = regions.upper_bound(a);
        if (it == regions.begin()) return false;
        --it;
        return a >= it->first && a + n <= it->first + it->second;
    }
};
inline PinnedRegistry &pinned_registry() {
    static PinnedRegistry r;
    return r;
}

// Host threads one native call may use for its per-chain host work (input validation, tables, block definition): at most
// 16, the machine's hardware threads, or QA_HOST_THREADS when the caller sets it (a launcher that runs several ranks and
// several host threads per rank divides the cores between them: bench.py does).
inline int host_threads_cap() {
    static const int cap = [] {
        int c = std::min<int>(16, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = getenv("QA_HOST_THREADS")) {
            const int v = atoi(e);
            if (v >= 1) c = std::min(c, v);
        }
        return c;
    }();
    return cap;
}

// pageable <-> pinned copies of tens of MB run at one core's memcpy rate (~8 GB/s), a sixth of what PCIe moves: split them
inline void par_memcpy(void *dst, const void *src, size_t n) {
    constexpr size_t kMin = size_t(4) << 20;
    if (n < 2 * kMin) { memcpy(dst, src, n); return; }
    const int nt = (int)std::min<size_t>(4, n / kMin);
    const size_t part = ((n / nt) + 63) & ~size_t(63);
    std::vector<std::thread> th;
    for (int i = 1; i < nt; i++) {
        const size_t off = (size_t)i * part, len = i == nt - 1 ? n - off : part;
        th.emplace_back([=] { memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len); });
    }
    memcpy(dst, src, std::min(part, n));
    for (auto &t : th) t.join();
}

// host -> device; complete (stream-synchronised) on return.  Pieces alternate between the two staging buffers: the host fills
// one while the device drains the other.
inline void staged_upload(void *dev, const void *host, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    if (pinned_registry().covers(host, bytes)) {   // a qa_host_alloc buffer: no staging
        stage_copy(dev, host, bytes, s);
        QA_HIP(hipGetLastError());
        QA_HIP(hipStreamSynchronize(s));
        return;
    }
    PinnedStage &ps = pinned_stage();
    if (!ps.get(std::min(bytes, kStagePiece))) {   // no pinned memory: fall back to the runtime's own path
        QA_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    int i = 0;
    for (size_t off = 0; off < bytes; off += kStagePiece, i++) {
        const size_t n = std::min(kStagePiece, bytes - off);
        const int b = i & 1;
        if (i >= 2) QA_HIP(hipEventSynchronize(ps.ev[b]));   // the copy that last read this buffer
        par_memcpy(ps.buf(b), static_cast<const char *>(host) + off, n);
        stage_copy(static_cast<char *>(dev) + off, ps.buf(b), n, s);
        QA_HIP(hipGetLastError());
        QA_HIP(hipEventRecord(ps.ev[b], s));
    }
    QA_HIP(hipStreamSynchronize(s));
}
// device -> host, after everything queued on the stream; complete on return
inline void staged_download(void *host, const void *dev, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    if (pinned_registry().covers(host, bytes)) {   // a qa_host_alloc buffer: no staging
        stage_copy(host, dev, bytes, s);
        QA_HIP(hipGetLastError());
        QA_HIP(hipStreamSynchronize(s));
        return;
    }
    PinnedStage &ps = pinned_stage();
    if (!ps.get(std::min(bytes, kStagePiece))) {
        QA_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        QA_HIP(hipStreamSynchronize(s));
        return;
    }
    const size_t n_piece = (bytes + kStagePiece - 1) / kStagePiece;
    auto launch = [&](size_t i) {
        const size_t off = i * kStagePiece, n = std::min(kStagePiece, bytes - off);
        stage_copy(ps.buf((int)(i & 1)), static_cast<const char *>(dev) + off, n, s);
        QA_HIP(hipGetLastError());
        QA_HIP(hipEventRecord(ps.ev[i & 1], s));
    };
    launch(0);
    for (size_t i = 0; i < n_piece; i++) {
        const size_t off = i * kStagePiece, n = std::min(kStagePiece, bytes - off);
        QA_HIP(hipEventSynchronize(ps.ev[i & 1]));
        if (i + 1 < n_piece) launch(i + 1);            // the device fills the other buffer while this one is copied out
        par_memcpy(static_cast<char *>(host) + off, ps.buf((int)(i & 1)), n);
    }
    QA_HIP(hipStreamSynchronize(s));
}

// RAII device buffer
template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    explicit DBuf(size_t n_) { alloc(n_); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    DBuf(DBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf &operator=(DBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DBuf() { release(); }
    void alloc(size_t n_) {
        release();
        n = n_;
        if (n) QA_HIP(hipMalloc((void **)&p, n * sizeof(T)));
    }
    // grow-only with slack: hipFree / hipMalloc synchronise the whole device (every stream, the other host threads' launches
    // included), so a buffer whose size wanders from call to call must not be reallocated for every new maximum
    void ensure(size_t n_) { if (n_ > n) alloc(n_ + n_ / 4 + 64); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void upload(const T *h, size_t cnt, hipStream_t s = nullptr) { staged_upload(p, h, cnt * sizeof(T), s); }
    void download(T *h, size_t cnt, hipStream_t s = nullptr) const { staged_download(h, p, cnt * sizeof(T), s); }
    void zero(hipStream_t s = nullptr) {
        if (n) QA_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

// One device arena per panel handle, shared by every launch set (the Gibbs sampler and the full-panel pass never
// run concurrently): bump-allocated per call, grown only when a call needs more.  Avoids both hipMalloc churn
// (a 100 GB hipMalloc costs seconds) and double-counting of differently shaped scratch buffers.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    bool owned = true;        // false: a slice of another arena (GateHold): fixed size, not freed
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena() { if (base && owned) (void)hipFree(base); }
    void reset() { off = 0; }
    // Share of the device's free memory the launch sets are planned against.  The rest (12 %: 34 GB of 288) is headroom for
    // what lives outside the arena: the handles' own per-thread buffers (read tables, labels, lists: tens of MB each), the
    // panel replicas (0.1 GB each) and the runtime's own allocations.  QA_ARENA_FRACTION (0.1 .. 0.95) overrides it.
    static double fraction() {
        static const double f = [] {
            if (const char *e = getenv("QA_ARENA_FRACTION")) {
                const double v = atof(e);
                if (v >= 0.1 && v <= 0.95) return v;
            }
            return 0.88;
        }();
        return f;
    }
    // bytes this arena may grow to: what is free now plus what it already holds, with headroom
    size_t budget() const {
        size_t free_b = 0, total_b = 0;
        QA_HIP(hipMemGetInfo(&free_b, &total_b));
        return (size_t)((free_b + cap) * fraction());
    }
    // the same for a handle that shares the device with `share` - 1 others: an equal part of the device at most, but not a
    // fraction of what is free NOW (the others' arenas are already out of `free`: dividing that again by `share` counted
    // them twice and kept launch sets a fifth smaller than the memory allows)
    size_t budget_shared(int share) const {
        size_t free_b = 0, total_b = 0;
        QA_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t mine = (size_t)((free_b + cap) * fraction()), part = (size_t)(total_b * fraction() / (share > 0 ? share : 1));
        return mine < part ? mine : part;
    }
    void require(size_t bytes) {
        if (bytes <= cap) return;
        if (!owned) throw std::runtime_error("device arena slice too small (internal sizing error)");
        if (base) { QA_HIP(hipFree(base)); base = nullptr; cap = 0; }
        const size_t want = (bytes + (size_t(1) << 28) - 1) >> 28 << 28;   // 256 MiB granules
        QA_HIP(hipMalloc((void **)&base, want));
        cap = want;
        off = 0;
    }
    void *take(size_t bytes) {
        const size_t a = (off + 255) & ~size_t(255);
        if (a + bytes > cap) throw std::runtime_error("device arena exhausted (internal sizing error)");
        off = a + bytes;
        return base + a;
    }
};

// Device phases (qa_panel_set_exclusive): the launch sets of the handles that opted in take the device in turn.
//   * a full-panel launch set (one workgroup per compute unit) holds it EXCLUSIVELY;
//   * Gibbs launches (one wave per chain, a SIMD each) hold as many of the 1 024 SIMD slots as they have waves: launches
//     that fit together run together (the 128 phasing chains of one batch beside the 896 main chains of another), a launch
//     that does not fit waits for the phase to end.
// Admission is first come, first served, with one exception: EXPRESS holds (short exclusive jobs: the haplotype search of
// the msPBWT mode, tens of ms) go before the queued launch sets.  Without it a thread's search waited behind every other
// thread's second-long Gibbs launch, all threads finished their rounds together and did their host work together with the
// device idle (27 % of the use_mspbwt bench).  Whoever is next blocks everyone behind it (a waiting
// full-panel set is never overtaken by later Gibbs launches).  Why phases at all: both kinds of launch sets are bound by
// the same HBM stream when they fill the chip, so overlapping them gains nothing, while a full-panel workgroup needs a
// whole compute unit and used to wait for one while another thread's Gibbs waves took the SIMDs one by one (62 against
// 11 ms per launch); and every thread sized its launches for a fraction of the memory (512 chains / 200 passes instead of
// 1 024 / 256).  ONE scratch arena serves all holders: an exclusive holder has all of it, Gibbs launches get disjoint
// slices; it is allocated once, at the size the launch sets are planned against (Arena::budget), so that no launch set
// of a later, larger shape has to re-allocate it (a 7 s stall when that happened inside a timed region).  The other host
// threads do their host-side work (marshalling, the R-level logic between the native calls) meanwhile.
struct DeviceGate {
    static constexpr int kSlots = 1024;   // SIMDs of the device (256 CUs x 4)
    struct Waiter {
        int slots = 0;
        size_t need = 0, offset = 0;
        bool exclusive = false, express = false, admitted = false, failed = false;
        std::string error;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Waiter *> queue;
    int active = 0, used_slots = 0, users = 0;
    bool active_exclusive = false;
    size_t bump = 0;          // bytes of the arena handed to the Gibbs launches of the running phase
    Arena arena;
    double held_ms = 0;       // accumulated time with at least one holder (qa_gate_stats)
    double wait_ms = 0;       // accumulated time callers spent queueing
    double excl_ms = 0;       // of held_ms: an exclusive holder (full-panel launch sets)
    double slot_ms = 0;       // sum over Gibbs holds of (SIMD slots x duration): / (held_ms - excl_ms) = mean slots in use
    double t_busy_from = 0;
    uint64_t n_holds = 0, n_shared = 0, slots_total = 0;
    // qa_gate_trace: one row per finished hold -- request, admit, mark (the holder's GateHold::mark(), e.g. its last
    // kernel done), release [ms, one clock], SIMD slots (0: exclusive), holder's thread
    bool tracing = false;
    std::vector<std::array<double, 6>> trace;

    // the whole planning budget at once (see above); called with `mu` held, by a thread whose current device is the gate's
    void grow_arena(size_t need) {
        if (need <= arena.cap) return;
        const size_t all = arena.budget();
        arena.require(std::max(need, all));
    }
    // admit whoever can go now (with `mu` held)
    void pump(double now) {
        for (;;) {
            if (queue.empty()) return;
            auto it = queue.begin();
            for (auto e = queue.begin(); e != queue.end(); ++e)
                if ((*e)->express) { it = e; break; }
            Waiter *w = *it;
            if (w->exclusive) {
                if (active != 0) return;
                active_exclusive = true;
            } else {
                if (active_exclusive || used_slots + w->slots > kSlots) return;
                if (active != 0 && bump + w->need > arena.cap) return;   // (the first holder of a phase may grow the arena)
                if (active == 0) {
                    try {
                        grow_arena(w->need);
                    } catch (const std::exception &e) {   // its acquire() reports it; the others go on
                        w->failed = w->admitted = true;
                        w->error = e.what();
                        queue.erase(it);
                        continue;
                    }
                    bump = 0;
                }
                w->offset = bump;
                bump += w->need;
                used_slots += w->slots;
            }
            if (active == 0) t_busy_from = now;
            active++;
            w->admitted = true;
            queue.erase(it);
        }
    }
};
DeviceGate &device_gate(int device);
void gate_user(int device, int delta);

// RAII hold of the device (or of `slots` SIMD slots of it) with the scratch arena that goes with it.  Without a gate (a
// handle that did not opt in) the hold is a no-op around the handle's own arena.
struct GateHold {
    DeviceGate *g = nullptr;
    Arena *own = nullptr;
    Arena view;               // slice of the gate's arena (shared holds)
    int slots = 0;
    bool exclusive = false;
    double queued_ms = 0, t_in = 0, t_mark = 0;
    static double now_ms();
    void mark() { if (g && g->tracing) t_mark = now_ms(); }
    GateHold() = default;
    GateHold(const GateHold &) = delete;
    GateHold &operator=(const GateHold &) = delete;
    ~GateHold() { release(); }
    // slots_ == 0: the whole device and the whole arena (which the holder may grow); else `slots_` SIMD slots and `bytes` of
    // arena.  express: a short exclusive job that goes before the queued launch sets.
    void acquire(DeviceGate *gate, Arena *own_arena, int slots_ = 0, size_t bytes = 0, bool express = false) {
        own = own_arena;
        if (!gate || g) return;
        const double t0 = now_ms();
        slots = std::min(slots_, (int)DeviceGate::kSlots);
        exclusive = slots == 0;
        DeviceGate::Waiter w;
        w.slots = slots;
        w.need = (bytes + 4095) & ~size_t(4095);
        w.exclusive = exclusive;
        w.express = express && exclusive;
        std::unique_lock<std::mutex> lk(gate->mu);
        gate->queue.push_back(&w);
        gate->pump(t0);
        if (!w.admitted) gate->cv.wait(lk, [&] { return w.admitted; });
        if (w.failed) {
            lk.unlock();
            gate->cv.notify_all();
            throw std::runtime_error("device arena: " + w.error);
        }
        if (!exclusive) {
            view.base = gate->arena.base + w.offset;
            view.cap = w.need;
            view.off = 0;
            view.owned = false;
        }
        g = gate;
        t_in = now_ms();
        queued_ms = t_in - t0;
        gate->wait_ms += queued_ms;
    }
    Arena &arena() { return g ? (exclusive ? g->arena : view) : *own; }
    // at least `bytes` of arena for this hold (an exclusive holder grows the device-wide arena: to the planning budget)
    void require(size_t bytes) {
        if (g && exclusive) {
            std::lock_guard<std::mutex> lk(g->mu);
            g->grow_arena(bytes);
        } else {
            arena().require(bytes);
        }
    }
    void release() {
        if (!g) return;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->active--;
            if (exclusive) g->active_exclusive = false; else g->used_slots -= slots;
            const double t_out = now_ms();
            if (g->active == 0) { g->bump = 0; g->held_ms += t_out - g->t_busy_from; }
            if (exclusive) g->excl_ms += t_out - t_in;
            else { g->slot_ms += (double)slots * (t_out - t_in); g->n_shared++; g->slots_total += (uint64_t)slots; }
            g->n_holds++;
            if (g->tracing && g->trace.size() < (1u << 20))
                g->trace.push_back({t_in - queued_ms, t_in, t_mark ? t_mark : t_out, t_out, (double)slots,
                                    (double)(std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xffffff)});
            g->pump(t_out);
        }
        g->cv.notify_all();
        g = nullptr;
        view.base = nullptr;
        view.cap = 0;
    }
};

// typed view carved from an arena; same surface as DBuf for the call sites
template <typename T>
struct ABuf {
    Arena *arena = nullptr;
    T *p = nullptr;
    size_t n = 0;
    void ensure(size_t n_) {   // a fresh carve per call (the arena is reset by the entry point)
        n = n_;
        p = n ? static_cast<T *>(arena->take(n * sizeof(T))) : nullptr;
    }
    void upload(const T *h, size_t cnt, hipStream_t s = nullptr) { staged_upload(p, h, cnt * sizeof(T), s); }
    void download(T *h, size_t cnt, hipStream_t s = nullptr) const { staged_download(h, p, cnt * sizeof(T), s); }
};

// translate exceptions at the C boundary
template <typename F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const HipError &e) {
        set_error("%s", e.what());
        return QA_ERR_HIP;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return QA_ERR_INVALID;
    }
}

}  // namespace qa
