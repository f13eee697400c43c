// Micro-benchmark: VALU issue cost of v_fma_f32 / v_pk_fma_f32 / v_sqrt_f32 on gfx950 at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *cyc)
{
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 1.f}; }
    const float m = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
                if (MODE == 1) p[i] = p[i] * m + c;
                if (MODE == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
                if (MODE == 3) a[i] = a[i] + c;
                if (MODE == 4) { if (u == 0) a[i] = __builtin_amdgcn_sqrtf(a[i]); else a[i] = __builtin_fmaf(a[i], m, c); }  // 1 sqrt : 3 fma
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[threadIdx.x >> 6] = t1 - t0; cyc[16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); }
}
int main()
{
    float *out; unsigned long long *cyc, h[32];
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 32 * 8);
    const int iters = 2000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_sqrt_f32", "v_add_f32", "sqrt:fma 1:3"};
    for (int mode = 0; mode < 5; ++mode)
        for (int threads : {64, 256, 512, 1024}) {  // one block on one CU: 0.25 / 1 / 2 / 4 waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 32 * 8, hipMemcpyDeviceToHost);
            const double n = double(iters) * 32;  // instructions per wave
            const int nw = threads / 64;
            unsigned long long mx = 0, mn = ~0ull; double sum = 0;
            for (int w = 0; w < nw; ++w) { mx = h[w] > mx ? h[w] : mx; mn = h[w] < mn ? h[w] : mn; sum += h[w]; }
            const double wps = threads / 256.0 < 1 ? 1 : threads / 256.0;
            printf("%-14s waves/SIMD %-4g  ticks/instr per wave: min %.2f mean %.2f max %.2f   -> SIMD ticks per instr (max/wps) %.2f\n",
                   names[mode], threads / 256.0, mn / n, sum / nw / n, mx / n, mx / n / wps);
        }
    return 0;
}
