// Does v_mfma_f32_16x16x4_f32 run concurrently with f32 VALU work of ANOTHER wave on the same SIMD?
// One 512-thread block on one CU = 2 waves per SIMD.  Mode 0: all waves VALU; 1: all waves MFMA;
// 2: waves 0-3 VALU + waves 4-7 MFMA (one of each per SIMD).  Compare kernel times.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode, unsigned long long *t)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 1 || (mode == 2 && wave >= 4) || (mode == 3 && wave >= 4);
    const bool idle = (mode == 3 && wave < 4) || (mode == 4 && wave >= 4);
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (!idle) {
        if (do_mfma) {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc[u & 3], 0, 0, 0);
        } else {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[wave] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *t, h[8];
    hipMalloc(&out, 4096); hipMalloc(&t, 64);
    const char *names[] = {"all 8 waves VALU (128 fma/iter)", "all 8 waves MFMA (8 mfma/iter)", "4 VALU + 4 MFMA waves", "4 MFMA waves alone", "4 VALU waves alone"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, 4000, mode, t); hipDeviceSynchronize(); }
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("%-36s per-wave time (us):", names[mode]);
        for (int w = 0; w < 8; ++w) printf(" %6.1f", h[w] / 100.0);
        printf("\n");
    }
    return 0;
}
