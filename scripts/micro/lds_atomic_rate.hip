// Developer microbenchmark: throughput of LDS atomics (f32 / u32 / u64) vs plain read-modify-write, 256-bin histogram
// with 32 bank-private copies, one 448-thread workgroup per CU.   hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(448) void k(const uint8_t *codes, float *out, int iters) {
    extern __shared__ char smem[];
    float *hf = reinterpret_cast<float *>(smem);
    unsigned *hu = reinterpret_cast<unsigned *>(smem);
    unsigned long long *hl = reinterpret_cast<unsigned long long *>(smem);
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 256 * 32 * 2; i += blockDim.x) hu[i] = 0;
    __syncthreads();
    uint4 d = reinterpret_cast<const uint4 *>(codes)[blockIdx.x * blockDim.x + t];
    float g = 1e-5f * (t + 1);
    for (int it = 0; it < iters; it++) {
        const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t code = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
            if (MODE == 0) atomicAdd(&hf[code * 32 + (lane & 31)], g);
            if (MODE == 1) atomicAdd(&hu[code * 32 + (lane & 31)], (unsigned)(g * 2147483648.f));
            if (MODE == 2) atomicAdd(&hl[code * 32 + (lane & 31)], (unsigned long long)(g * 4.6e18f));
            if (MODE == 3) hf[code * 32 + (lane & 31)] += g;                     // racy RMW: rate reference only
            if (MODE == 4) atomicAdd(&hf[code * 64 + lane], g);                  // 64 copies (128 KB would be needed: wraps)
            if (MODE == 5) __hip_atomic_fetch_add(&hf[code * 32 + (lane & 31)], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        d.x = d.x * 1664525u + 1013904223u; d.y ^= d.x >> 3; d.z += d.y; d.w ^= d.z << 1;
    }
    __syncthreads();
    if (t < 256) out[blockIdx.x * 256 + t] = hf[t * 32];
}

int main() {
    const int nb = 256, nt = 448, iters = 2000;
    std::vector<uint8_t> h((size_t)nb * nt * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(1 + (i * 2654435761u >> 13) % 255);
    uint8_t *d; float *o;
    hipMalloc(&d, h.size()); hipMalloc(&o, nb * 256 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char *name, size_t lds) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int r = 0; r < 2; r++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(nt), lds, 0, d, o, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r) printf("%-28s %8.2f ms  -> %.2f clk per lane-op per CU (2.4 GHz), %.1f us per 50k-element grid\n", name, ms,
                          ms * 1e-3 * 2.4e9 / ((double)iters * nt * 16), ms * 1e3 / iters * 50000.0 / (nt * 16));
        }
    };
    run(k<0>, "ds_add_f32", 65536);
    run(k<1>, "ds_add_u32", 65536);
    run(k<2>, "ds_add_u64", 131072);
    run(k<3>, "plain RMW (racy)", 65536);
    run(k<4>, "ds_add_f32 64 copies", 65536 * 2);
    run(k<5>, "ds_add_f32 wave scope", 65536);
    return 0;
}
