// Micro-benchmark (round 2): what one SIMD of gfx950 sustains in VALU instructions per cycle at 1/2/3/4 waves per SIMD,
// with s_memtime calibrated against s_memrealtime (100 MHz) in the same run.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
// Modes use inline assembly so that the operand kinds are exactly what the label says:
//   fma1v   v_fma_f32 d, d, s, s        one VGPR source (what round 1 measured)
//   fma3v   v_fma_f32 d, a, b, d        three distinct VGPR sources (what the kernels issue)
//   add2v   v_add_f32 d, a, d
//   mul2v   v_mul_f32 d, a, b           (d written only)
//   pkfma   v_pk_fma_f32 d, a, b, d     three VGPR pairs
//   sqrt    v_sqrt_f32 d, d
//   log     v_log_f32 d, d
//   dppadd  v_add_f32_dpp d, a, d  row_mirror
//   fft16   the kernels' radix-16 register butterfly (ss_fft_reg.h), i.e. the real instruction mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../mfcc-rust_amd/csrc/ss_fft_reg.h"

typedef float v2f __attribute__((ext_vector_type(2)));

enum { FMA1V, FMA3V, ADD2V, MUL2V, PKFMA, SQRT, LOG, DPPADD, FFT16, NMODES };
static const char *kNames[NMODES] = {"fma1v", "fma3v", "add2v", "mul2v", "pkfma", "sqrt", "log", "dppadd", "fft16"};

struct Stamp {
    unsigned long long t0, t1, r0, r1;
};

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, Stamp *st)
{
    float a[8], b[8], c[8];
    v2f pa[8], pb[8], pc[8];
    float2 z[16];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3f + i;
        b[i] = 1.0f + 1e-6f * (threadIdx.x + i);
        c[i] = 1e-3f * i;
        pa[i] = v2f{a[i], a[i] + 1.f};
        pb[i] = v2f{b[i], b[i]};
        pc[i] = v2f{c[i], c[i]};
    }
    for (int i = 0; i < 16; ++i) z[i] = make_float2(threadIdx.x * 1e-3f + i, 1.f - i * 0.01f);
    const float m = 1.0001f;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == FFT16) {
            ss::fft_reg<16>(z);
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(z[i].x), "+v"(z[i].y));
            continue;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == FMA1V) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a[i]) : "s"(m));
                if (MODE == FMA3V) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[(i + u) & 7]));
                if (MODE == ADD2V) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == MUL2V) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[(i + u) & 7]));
                if (MODE == PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(pb[i]), "v"(pc[(i + u) & 7]));
                if (MODE == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (MODE == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if (MODE == DPPADD) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i].x + pa[i].y;
    for (int i = 0; i < 16; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 16 + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};
}

template <int MODE>
static void run(int blocks, int threads, int iters, float *out, Stamp *dst, std::vector<Stamp> &h)
{
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, dst);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), dst, sizeof(Stamp) * 16 * blocks, hipMemcpyDeviceToHost);
}

int main(int argc, char **argv)
{
    int fft_instr = 148;  // VALU instructions of one fft_reg<16> (pass -DFFT16_INSTR or read from the disassembly)
    if (argc > 1) fft_instr = atoi(argv[1]);
    float *out;
    Stamp *dst;
    const int maxb = 1024;
    (void)hipMalloc(&out, 1024 * maxb * 4);
    (void)hipMalloc(&dst, sizeof(Stamp) * 16 * maxb);
    std::vector<Stamp> h(16 * maxb);
    const int iters = 4000;
    printf("# s_memtime ticks; MHz = ticks / (s_memrealtime ticks / 100 MHz); 'simd cyc/instr' = slowest wave's ticks / (instr per wave * waves per SIMD)\n");
    printf("%-7s %-6s %-10s %-12s %-12s %-12s %-10s\n", "mode", "blocks", "waves/SIMD", "wave cyc/ins", "simd cyc/ins", "memtime MHz", "wall us");
    for (int blocks : {1, 256}) {
        for (int mode = 0; mode < NMODES; ++mode) {
            for (int threads : {64, 256, 512, 768, 1024}) {
                switch (mode) {
                case FMA1V: run<FMA1V>(blocks, threads, iters, out, dst, h); break;
                case FMA3V: run<FMA3V>(blocks, threads, iters, out, dst, h); break;
                case ADD2V: run<ADD2V>(blocks, threads, iters, out, dst, h); break;
                case MUL2V: run<MUL2V>(blocks, threads, iters, out, dst, h); break;
                case PKFMA: run<PKFMA>(blocks, threads, iters, out, dst, h); break;
                case SQRT: run<SQRT>(blocks, threads, iters, out, dst, h); break;
                case LOG: run<LOG>(blocks, threads, iters, out, dst, h); break;
                case DPPADD: run<DPPADD>(blocks, threads, iters, out, dst, h); break;
                case FFT16: run<FFT16>(blocks, threads, iters, out, dst, h); break;
                }
                const double n = double(iters) * (mode == FFT16 ? fft_instr : 32);
                const int nw = threads / 64;
                double mx = 0, sum = 0, mhz = 0, wall = 0;
                int cnt = 0;
                for (int bl = 0; bl < blocks; ++bl)
                    for (int w = 0; w < nw; ++w) {
                        const Stamp &s = h[bl * 16 + w];
                        const double t = double(s.t1 - s.t0), r = double(s.r1 - s.r0);
                        mx = t > mx ? t : mx;
                        sum += t;
                        mhz += t / (r / 100.0);
                        wall = r / 100.0 > wall ? r / 100.0 : wall;
                        ++cnt;
                    }
                const double wps = threads / 256.0 < 1 ? 1 : threads / 256.0;
                printf("%-7s %-6d %-10.2f %-12.2f %-12.2f %-12.0f %-10.1f\n", kNames[mode], blocks, threads / 256.0, sum / cnt / n, mx / n / wps, mhz / cnt, wall);
            }
        }
    }
    return 0;
}
