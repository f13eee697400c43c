// Semantics check of __builtin_amdgcn_permlane32_swap(vdst, src): prints r[0], r[1] per lane for vdst = src = lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
    const unsigned x = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0: lane0=%d lane1=%d lane31=%d lane32=%d lane33=%d lane63=%d\n", h[0], h[1], h[31], h[32], h[33], h[63]);
    printf("r1: lane0=%d lane1=%d lane31=%d lane32=%d lane33=%d lane63=%d\n", h[64], h[65], h[95], h[96], h[97], h[127]);
    return 0;
}
