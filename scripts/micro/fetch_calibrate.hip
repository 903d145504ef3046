// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access shapes this library's
// kernels use (MI355X_MICROARCH.md, "HBM": "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read
// (16 B/lane) ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel streams a buffer much larger than the 256 MiB Infinity Cache exactly once and prints the bytes it
// requested; scripts/calibrate_fetch.sh runs the binary under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
// passes) and divides.
//   rd16      global_load_dwordx4, lanes contiguous (1 KiB per wave instruction): k_fwd64 / k_bwd64 codes, alpha vectors, the
//             sampler's pattern packs
//   rd8       global_load_dwordx2, lanes contiguous (512 B per wave instruction)
//   rd8buf    raw_buffer_load_b64, lanes contiguous, 10 instructions per 5 120-byte-pitched column of which 4 800 bytes are
//             read: the sampler's state columns (gibbs_dev.hpp ldm: Ks = 600 rows of a 640-row pitch)
//   rd4       global_load_dword, lanes contiguous (256 B per wave instruction)
//   rd1gather one byte per lane from a random 64-byte-aligned place each (k_ematread's gather of panel codes): requested bytes
//             are 1 per lane; the counter shows what a line per code costs
//   wr16 / wr8 / wr8buf   the same shapes as stores
//   hipcc -O3 --offload-arch=gfx950 fetch_calibrate.hip -o fetch_calibrate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void rd16(const uint4 *src, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void rd8(const uint2 *src, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = src[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void rd4(const uint32_t *src, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// one wave per workgroup; column = 640 doubles of pitch, 600 read (lanes 24..63 of row 9 are out of the buffer's range)
__global__ __launch_bounds__(64) void rd8buf(const double *src, size_t n_col, uint32_t *sink) {
    const int lane = threadIdx.x;
    double acc = 0;
    for (size_t c = blockIdx.x; c < n_col; c += gridDim.x) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src + c * 640), 0, 600 * 8, 0x00020000);
#pragma unroll
        for (int i = 0; i < 10; i++) acc += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (lane + 64 * i) * 8, 0, 0));
    }
    if (acc == 1.2345) sink[0] = 1;
}
__global__ void rd1gather(const uint8_t *src, size_t n_lines, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (i * 0x9E3779B97F4A7C15ull >> 20) % n_lines;   // a pseudo-random 64-byte line per lane
        acc ^= src[line * 64 + (i & 63)];
    }
    if (acc == 0x5Au) sink[0] = acc;   // (a byte-sized constant: against a wider one the compiler proves the branch dead and drops the loads)
}
__global__ void wr16(uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void wr8(uint2 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_uint2((uint32_t)i, 1);
}
__global__ __launch_bounds__(64) void wr8buf(double *dst, size_t n_col) {
    const int lane = threadIdx.x;
    for (size_t c = blockIdx.x; c < n_col; c += gridDim.x) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst + c * 640, 0, 600 * 8, 0x00020000);
#pragma unroll
        for (int i = 0; i < 10; i++)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, (double)(c + i)), r,
                                                  (lane + 64 * i) * 8, 0, 0);
    }
}

int main() {
    const size_t bytes = (size_t)4 << 30;   // 4 GiB: 16 x the Infinity Cache
    void *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int nb = 256 * 8, nt = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timed = [&](const char *name, double req_bytes, auto launch) {
        CK(hipEventRecord(a));
        launch();
        CK(hipGetLastError());
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("CAL %s requested_bytes %.0f ms %.3f GBps %.1f\n", name, req_bytes, ms, req_bytes / 1e6 / ms);
    };
    timed("rd16", (double)bytes, [&] { hipLaunchKernelGGL(rd16, dim3(nb), dim3(nt), 0, 0, (const uint4 *)buf, bytes / 16, sink); });
    timed("rd8", (double)bytes, [&] { hipLaunchKernelGGL(rd8, dim3(nb), dim3(nt), 0, 0, (const uint2 *)buf, bytes / 8, sink); });
    timed("rd4", (double)bytes, [&] { hipLaunchKernelGGL(rd4, dim3(nb), dim3(nt), 0, 0, (const uint32_t *)buf, bytes / 4, sink); });
    const size_t n_col = bytes / 5120;
    timed("rd8buf", (double)n_col * 4800, [&] { hipLaunchKernelGGL(rd8buf, dim3(256 * 16), dim3(64), 0, 0, (const double *)buf, n_col, sink); });
    const size_t n_g = (size_t)1 << 28;
    timed("rd1gather", (double)n_g, [&] { hipLaunchKernelGGL(rd1gather, dim3(nb), dim3(nt), 0, 0, (const uint8_t *)buf, bytes / 64, n_g, sink); });
    timed("wr16", (double)bytes, [&] { hipLaunchKernelGGL(wr16, dim3(nb), dim3(nt), 0, 0, (uint4 *)buf, bytes / 16); });
    timed("wr8", (double)bytes, [&] { hipLaunchKernelGGL(wr8, dim3(nb), dim3(nt), 0, 0, (uint2 *)buf, bytes / 8); });
    timed("wr8buf", (double)n_col * 4800, [&] { hipLaunchKernelGGL(wr8buf, dim3(256 * 16), dim3(64), 0, 0, (double *)buf, n_col); });
    CK(hipDeviceSynchronize());
    return 0;
}
