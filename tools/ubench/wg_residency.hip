// Micro-benchmark: how many 384-thread workgroups of a given register and LDS footprint does a gfx950 CU keep resident?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/wg_residency.hip -o ab/wg_residency && ab/wg_residency
// 512 workgroups (two per CU) record s_memrealtime at their start and spin ~20 us; if both of a CU's workgroups are resident
// together all 512 start within a microsecond, otherwise half of them start after the first half has finished.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int NV>
__global__ __launch_bounds__(768) void k(unsigned long long *st, float *out, int spin)
{
    extern __shared__ float lds[];
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = v[0];
    for (int it = 0; it < spin; ++it)
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v[i]));
    float s = 0;
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x + 1) % blockDim.x];
    if (threadIdx.x == 0) st[blockIdx.x] = t0;
}

template <int NV>
static void run(const char *name, size_t lds_bytes, unsigned long long *d_st, float *d_out)
{
    const int blocks = 512;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<NV>, 384, lds_bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(384), lds_bytes, 0, d_st, d_out, 300);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), d_st, blocks * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int late = 0;
    double mx = 0;
    for (auto t : h) {
        const double us = (t - t0) / 100.0;
        late += us > 5.0;
        mx = std::max(mx, us);
    }
    std::printf("%-10s LDS %6zu B  runtime occupancy %d blocks/CU  workgroups starting > 5 us late: %3d of %d  (latest %.1f us)\n", name, lds_bytes, occ, late, blocks, mx);
}

int main()
{
    unsigned long long *d_st;
    float *d_out;
    (void)hipMalloc(&d_st, 512 * 8);
    (void)hipMalloc(&d_out, 512 * 384 * 4);
    for (size_t kb : {16, 40, 60, 64, 65, 72, 79, 80}) {
        run<100>("~110 VGPR", kb * 1024, d_st, d_out);
        run<150>("~160 VGPR", kb * 1024, d_st, d_out);
    }
    return 0;
}
