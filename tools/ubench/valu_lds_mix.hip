// Micro-benchmark (round 2): do VALU issue and LDS traffic overlap on a gfx950 CU, or do their times add?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_lds_mix.hip -o /tmp/valu_lds_mix && /tmp/valu_lds_mix
// Every wave loops over one block of NV independent v_fma_f32 (three VGPR sources) and NL conflict-free LDS
// instructions whose results nothing depends on until the s_waitcnt at the end of the block.  The block is timed with
// s_memtime for NV only, NL only and both; "overlap" = (t_valu + t_lds - t_both) / min(t_valu, t_lds): 1 = the shorter
// one hides completely, 0 = the times add.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

enum { RD128, RD64, RD32, WR128, WR32, NKINDS };
static const char *kKind[NKINDS] = {"ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_write_b128", "ds_write_b32"};

struct Stamp {
    unsigned long long t0, t1;
};

template <int KIND, int NV, int NL>
__global__ __launch_bounds__(1024) void k(float *out, int iters, Stamp *st)
{
    __shared__ v4f lds[1024 * 2];
    float a[8], b[8], c[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3f + i;
        b[i] = 1.0f + 1e-6f * (threadIdx.x + i);
        c[i] = 1e-3f * i;
    }
    lds[threadIdx.x] = v4f{a[0], a[1], a[2], a[3]};
    lds[threadIdx.x + 1024] = v4f{a[0], a[1], a[2], a[3]};
    __syncthreads();
    // lane l of a wave reads its own 16 / 8 / 4 bytes: consecutive lanes, consecutive addresses (conflict-free for every width)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>(lds)) + wave * 2048 + lane * (KIND == RD128 || KIND == WR128 ? 16 : KIND == RD64 ? 8 : 4);
    v4f r4[4] = {};
    float2 r2[4] = {};
    float r1[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int STEPS = NL > 0 ? NL : 1;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s < NL) {
                if (KIND == RD128) asm volatile("ds_read_b128 %0, %1" : "=v"(r4[s & 3]) : "v"(base));
                if (KIND == RD64) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[s & 3]) : "v"(base));
                if (KIND == RD32) asm volatile("ds_read_b32 %0, %1" : "=v"(r1[s & 3]) : "v"(base));
                if (KIND == WR128) asm volatile("ds_write_b128 %0, %1" ::"v"(base), "v"(r4[s & 3]));
                if (KIND == WR32) asm volatile("ds_write_b32 %0, %1" ::"v"(base), "v"(r1[s & 3]));
            }
#pragma unroll
            for (int i = 0; i < NV / STEPS; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 7]) : "v"(b[i & 7]), "v"(c[(i + s) & 7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += r4[i].x + r4[i].w + r2[i].x + r2[i].y + r1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) st[blockIdx.x * 16 + wave] = Stamp{t0, t1};
}

static float *g_out;
static Stamp *g_st;
static std::vector<Stamp> g_h;

template <int KIND, int NV, int NL>
static double run(int threads, int iters)
{
    const int blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, NV, NL>), dim3(blocks), dim3(threads), 0, 0, g_out, iters, g_st);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(g_h.data(), g_st, sizeof(Stamp) * 16 * blocks, hipMemcpyDeviceToHost);
    double sum = 0;
    int cnt = 0;
    for (int bl = 0; bl < blocks; ++bl)
        for (int w = 0; w < threads / 64; ++w) {
            sum += double(g_h[bl * 16 + w].t1 - g_h[bl * 16 + w].t0);
            ++cnt;
        }
    return sum / cnt / iters;  // s_memtime ticks per block of the loop, averaged over the waves
}

template <int KIND, int NV, int NL>
static void line(int threads)
{
    const int iters = 2000;
    const double tv = run<KIND, NV, 0>(threads, iters), tl = run<KIND, 0, NL>(threads, iters), tb = run<KIND, NV, NL>(threads, iters);
    const double mn = tv < tl ? tv : tl;
    printf("%-14s %-6d %-4d %-4d %-10.1f %-10.1f %-10.1f %-10.1f %-8.2f\n", kKind[KIND], threads / 64, NV, NL, tv, tl, tb, tv + tl, (tv + tl - tb) / mn);
}

int main()
{
    (void)hipMalloc(&g_out, 1024 * 256 * 4);
    (void)hipMalloc(&g_st, sizeof(Stamp) * 16 * 256);
    g_h.resize(16 * 256);
    printf("# 256 workgroups (one per CU); ticks of s_memtime per loop block, mean over the waves\n");
    printf("%-14s %-6s %-4s %-4s %-10s %-10s %-10s %-10s %-8s\n", "lds op", "waves", "NV", "NL", "valu only", "lds only", "both", "sum", "overlap");
    for (int threads : {256, 512, 1024}) {
        line<RD128, 32, 1>(threads);
        line<RD128, 32, 2>(threads);
        line<RD128, 32, 4>(threads);
        line<RD128, 64, 4>(threads);
        line<RD64, 32, 4>(threads);
        line<RD32, 32, 4>(threads);
        line<RD32, 32, 8>(threads);
        line<WR128, 32, 2>(threads);
        line<WR128, 32, 4>(threads);
        line<WR32, 32, 8>(threads);
    }
    return 0;
}
