// row_bcast:15 + row_shr:1 = "lane n reads lane n-1" across the two 16-lane rows of a 32-lane group
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    int v = threadIdx.x + 100;
    int x = -1;
    x = __builtin_amdgcn_update_dpp(x, v, 0x142, 0xA, 0xF, false);
    x = __builtin_amdgcn_update_dpp(x, v, 0x111, 0xF, 0xF, false);
    out[threadIdx.x] = x;
}
int main()
{
    int *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const int want = (i % 32 == 0) ? -1 : i - 1 + 100;
        if (h[i] != want) { bad++; printf("lane %d: %d (want %d)\n", i, h[i], want); }
    }
    printf(bad ? "MISMATCH\n" : "ok: lanes 16 and 48 read lanes 15 and 47, lanes 0 and 32 keep the old value\n");
    return bad != 0;
}
