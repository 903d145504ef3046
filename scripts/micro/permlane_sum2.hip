// Micro-test: two-value wave sum with v_permlane32_swap packing (see wsum2 in quilt_amd/csrc/gibbs_dev.hpp).
// hipcc --offload-arch=gfx950 -O3 permlane_sum2.hip -o permlane_sum2 && ./permlane_sum2
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
__global__ void k(const double *a, const double *b, double *out, double *dbg) {
    double x = a[threadIdx.x], y = b[threadIdx.x];
    int xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
    auto r1 = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
    auto r2 = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
    double p = __hiloint2double(r2[0], r1[0]), q = __hiloint2double(r2[1], r1[1]);
    dbg[threadIdx.x] = p; dbg[64 + threadIdx.x] = q;
    double v = p + q;
    v += dpp_get<0x111, 0xf>(v);
    v += dpp_get<0x112, 0xf>(v);
    v += dpp_get<0x114, 0xf>(v);
    v += dpp_get<0x118, 0xf>(v);
    v += dpp_get<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    out[threadIdx.x] = v;
}
int main() {
    double ha[64], hb[64], ho[64], hd[128], *a, *b, *o, *d;
    double sa = 0, sb = 0;
    for (int i = 0; i < 64; i++) { ha[i] = i + 0.25; hb[i] = 1000 + 3 * i; sa += ha[i]; sb += hb[i]; }
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&o, 512); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, d);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    printf("p: lanes 0,1,31,32,33,63 = %g %g %g %g %g %g\n", hd[0], hd[1], hd[31], hd[32], hd[33], hd[63]);
    printf("q: lanes 0,1,31,32,33,63 = %g %g %g %g %g %g\n", hd[64], hd[65], hd[95], hd[96], hd[97], hd[127]);
    printf("lane31 %g lane63 %g ; expected sum(a) %g sum(b) %g\n", ho[31], ho[63], sa, sb);
    return 0;
}
