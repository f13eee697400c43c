// Semantics check of v_permlane16_swap_b32 vdst, src (inline asm): prints both results per lane for vdst = lane id, src = 100 + lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
    int a = threadIdx.x, b = 100 + threadIdx.x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    o[threadIdx.x] = a;
    o[64 + threadIdx.x] = b;
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 2; ++r) {
        printf("%s:", r ? "src " : "vdst");
        for (int l : {0, 1, 15, 16, 17, 31, 32, 47, 48, 63}) printf(" l%d=%d", l, h[64 * r + l]);
        printf("\n");
    }
    return 0;
}
