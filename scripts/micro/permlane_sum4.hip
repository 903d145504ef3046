// Micro-test: four-value wave sum with v_permlane32_swap + v_permlane16_swap packing (see wsum4 in quilt_amd/csrc/gibbs_dev.hpp).
// hipcc --offload-arch=gfx950 -O3 permlane_sum4.hip -o permlane_sum4 && ./permlane_sum4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pack32(double a, double b) {   // lanes 0..31: a(l) + a(l + 32); lanes 32..63: b(l - 32) + b(l)
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__global__ void k(const double *in, double *out) {
    const double a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
    const double ab = pack32(a, b), cd = pack32(c, d);
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(ab), __double2loint(cd), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(ab), __double2hiint(cd), false, false);
    double v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    v += dpp_get<0x111, 0xf>(v);
    v += dpp_get<0x112, 0xf>(v);
    v += dpp_get<0x114, 0xf>(v);
    v += dpp_get<0x118, 0xf>(v);
    out[threadIdx.x] = v;
}
int main() {
    double h[256], ho[64], *in, *o, s[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4; q++)
        for (int i = 0; i < 64; i++) { h[64 * q + i] = (q + 1) * 1000 + i * (q + 1) + 0.25; s[q] += h[64 * q + i]; }
    hipMalloc(&in, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    printf("lanes 15 31 47 63 = %.2f %.2f %.2f %.2f\nexpected a b c d   = %.2f %.2f %.2f %.2f\n", ho[15], ho[31], ho[47], ho[63], s[0], s[1], s[2], s[3]);
    return 0;
}
