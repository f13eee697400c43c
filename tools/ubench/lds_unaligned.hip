// Micro-benchmark (round 4): does gfx950 serve ds_read_b128 / ds_read_b64 at dword-aligned (not 16 / 8-byte aligned) LDS addresses,
// and at what cost?  The mel stages read their P taps as ds_read2_b32 pairs because the compiler may only assume dword
// alignment; one ds_read_b128 per four taps would halve the LDS instructions of that stage.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_unaligned.hip -o tools/ubench/bin/lds_unaligned && tools/ubench/bin/lds_unaligned
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// every lane reads 4 (2) consecutive floats starting at dword index base + lane * stride + shift
template <int W>
__global__ void k(float *out, int stride, int shift, int iters, unsigned long long *cycles)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = static_cast<float>(i);
    __syncthreads();
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>(lds)) + 4u * ((threadIdx.x * stride + shift) & 8191);
    v4f acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        v4f r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (W == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 64));
            if (W == 2) {
                v2f t;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(i * 64));
                r[i] = v4f{t.x, t.y, 0, 0};
            }
            if (W == 1) {  // the same four floats as two ds_read2_b32
                v2f a, b;
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a) : "v"(addr), "n"(i * 16), "n"(i * 16 + 1));
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(b) : "v"(addr), "n"(i * 16 + 2), "n"(i * 16 + 3));
                r[i] = v4f{a.x, a.y, b.x, b.y};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += r[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 0] = acc.x;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 1] = acc.y;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 2] = acc.z;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 3] = acc.w;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int W>
void run(const char *name, int stride, int shift)
{
    const int blocks = 256, threads = 512, iters = 2000;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 16);
    hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(threads), 0, 0, out, stride, shift, 1, cyc);
    hipDeviceSynchronize();
    std::vector<float> h(threads * 4);
    hipMemcpy(h.data(), out, threads * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < threads; ++t) {
        const int b = (t * stride + shift) & 8191;
        for (int c = 0; c < (W == 2 ? 2 : 4); ++c) {
            float want = 0;
            for (int i = 0; i < 8; ++i) want += static_cast<float>(b + i * 16 + c);
            if (h[t * 4 + c] != want) ++bad;
        }
    }
    hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(threads), 0, 0, out, stride, shift, iters, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(blocks);
    hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto c : hc) s += static_cast<double>(c);
    std::printf("%-14s stride %2d shift %d: %s, %.1f cycles per batch of 8 reads (8 waves per CU)\n", name, stride, shift, bad ? "WRONG DATA" : "data ok",
                s / blocks / iters);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int shift = 0; shift < 4; ++shift) run<4>("ds_read_b128", 4, shift);  // lanes 16 bytes apart: contiguous
    for (int shift = 0; shift < 4; ++shift) run<4>("ds_read_b128", 5, shift);  // lanes 20 bytes apart: every alignment at once
    for (int shift = 0; shift < 2; ++shift) run<2>("ds_read_b64", 2, shift);
    for (int shift = 0; shift < 2; ++shift) run<2>("ds_read_b64", 3, shift);
    for (int shift = 0; shift < 2; ++shift) run<1>("2x ds_read2_b32", 4, shift);
    for (int shift = 0; shift < 2; ++shift) run<1>("2x ds_read2_b32", 5, shift);
    return 0;
}
