// common.hpp -- shared host-side plumbing of libquilt_amd (error state, device buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/quilt_amd.h"

namespace qa {

void set_error(const char *fmt, ...);
bool device_ready();

// per-kernel accumulators (HIP-event time on the launch stream, launches, algorithmic HBM bytes)
enum ProfileKernel { PK_EMAT = 0, PK_FWD, PK_BWD, PK_POST, PK_EMATREAD, PK_GIBBS, PK_HAPPROBS, PK_FWD64, PK_BWD64, PK_COUNT };
void profile_add(int kernel, double ms, double alg_bytes);

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
    void ensure(size_t n_) { if (n_ > n) alloc(n_); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void upload(const T *h, size_t cnt, hipStream_t s = nullptr) {
        if (cnt) QA_HIP(hipMemcpyAsync(p, h, cnt * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *h, size_t cnt, hipStream_t s = nullptr) const {
        if (cnt) QA_HIP(hipMemcpyAsync(h, p, cnt * sizeof(T), hipMemcpyDeviceToHost, s));
    }
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
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena() { if (base) (void)hipFree(base); }
    void reset() { off = 0; }
    // bytes this arena may grow to: what is free now plus what it already holds, with headroom
    size_t budget() const {
        size_t free_b = 0, total_b = 0;
        QA_HIP(hipMemGetInfo(&free_b, &total_b));
        return (size_t)((free_b + cap) * 0.88);
    }
    void require(size_t bytes) {
        if (bytes <= cap) return;
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
    void upload(const T *h, size_t cnt, hipStream_t s = nullptr) {
        if (cnt) QA_HIP(hipMemcpyAsync(p, h, cnt * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *h, size_t cnt, hipStream_t s = nullptr) const {
        if (cnt) QA_HIP(hipMemcpyAsync(h, p, cnt * sizeof(T), hipMemcpyDeviceToHost, s));
    }
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
