// Micro-benchmark (round 5): does gfx950 skip the 16-lane service groups of a ds_read_b128 whose lanes are all masked off?
// A wave64 ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32;
// MI355X guide, LDS table).  The transposing exchanges of the 2048- / 4096-point kernels read with HALF the lanes active per
// phase (lanes k1 < 16, then k1 >= 16: lanes 0-15 and 32-47 -- eight active lanes in every group).  If a group with no active lane
// costs no LDS cycle, a lane mapping that makes each phase's readers whole groups would halve those reads' cycles.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_exec_groups.hip -o tools/ubench/bin/lds_exec_groups && tools/ubench/bin/lds_exec_groups
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

// mode 0: all lanes; 1: lanes 0-15 and 32-47 (the kernels' phases); 2: service groups 0 and 2 whole ({0-3,12-15,20-27} + 32);
// 3: lanes 0-31 (groups 0 and 1 whole)
__device__ bool active_lane(int mode, int lane)
{
    const int l = lane & 31;
    const bool g0 = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
    if (mode == 0) return true;
    if (mode == 1) return l < 16;
    if (mode == 2) return g0;
    return lane < 32;
}

__global__ void k(float *out, int mode, int iters, unsigned long long *cycles)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = static_cast<float>(i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // conflict-free: lane l reads float4 slot (l + 17 * i) of the wave's 4 KB window
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>(lds)) + 4096u * wave + 16u * lane;
    v4f acc = {0, 0, 0, 0};
    const bool act = active_lane(mode, lane);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (act) {
        for (int it = 0; it < iters; ++it) {
            v4f r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 1024 % 3072));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += r[i];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x)] = acc.x + acc.y + acc.z + acc.w;
    if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    const int blocks = 256, iters = 4000;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, blocks * 1024 * 4);
    hipMalloc(&cyc, blocks * 8);
    const char *names[4] = {"all 64 lanes", "lanes 0-15 + 32-47 (8 of every group)", "groups 0 and 2 whole (32 lanes)", "lanes 0-31 (groups 0 and 1 whole)"};
    for (int threads : {256, 768}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, mode, 10, cyc);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, mode, iters, cyc);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : h) s += static_cast<double>(v);
            const double per = s / blocks / iters / 8.0;  // cycles per ds_read_b128 per wave, as the first wave sees them
            std::printf("%d waves per CU, %-40s %.2f cycles per wave-instruction -> %.2f LDS cycles per instruction across the CU's waves\n",
                        threads / 64, names[mode], per, per / (threads / 64));
        }
    }
    return 0;
}
