// lds_direct_probe.hip -- where do the bytes of global_load_lds_dwordx4 (gfx950) land?  Lanes 0, 1 of every 16-lane
// group load four uint4 each, straight into LDS at M0 = base + 32 * n: the walker's region-cache layout
// (lane T's piece n at byte 16 * T + 32 * n, lanes that are masked off write nothing).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_direct_probe tools/lds_direct_probe.hip && tools/lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const uint4 *g, uint4 *out)
{
    __shared__ uint4 buf[128];
    typedef __attribute__((address_space(3))) void LdsV;
    typedef __attribute__((address_space(1))) const void GlobV;
    for (int i = threadIdx.x; i < 128; i += blockDim.x) buf[i] = make_uint4(0xdeadu, 0xdeadu, 0xdeadu, 0xdeadu);
    __syncthreads();
    if ((threadIdx.x & 15) < 2) {
#pragma unroll
        for (int n = 0; n < 4; n++)
            __builtin_amdgcn_global_load_lds((GlobV *)(g + threadIdx.x * 8 + n), (LdsV *)(buf + 2 * n), 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 128; i += blockDim.x) out[i] = buf[i];
}

int main()
{
    std::vector<uint4> h(64 * 8);
    for (int t = 0; t < 64; t++)
        for (int n = 0; n < 8; n++) h[t * 8 + n] = make_uint4(t, n, 100 * t + n, 7);
    uint4 *g, *o;
    hipMalloc(&g, h.size() * sizeof(uint4));
    hipMalloc(&o, 128 * sizeof(uint4));
    hipMemcpy(g, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, o);
    std::vector<uint4> r(128);
    if (hipMemcpy(r.data(), o, 128 * sizeof(uint4), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    int bad = 0;
    for (int i = 0; i < 128; i++) {
        // expected: slot i = T + 2n for an active lane T (T & 15 < 2), n < 4
        bool hit = false;
        for (int T = 0; T < 64 && !hit; T++)
            if ((T & 15) < 2)
                for (int n = 0; n < 4; n++)
                    if (T + 2 * n == i) {
                        hit = true;
                        if (r[i].x != (unsigned)T || r[i].y != (unsigned)n) { bad++; printf("slot %d: got lane %u piece %u, want %d %d\n", i, r[i].x, r[i].y, T, n); }
                    }
        if (!hit && r[i].x != 0xdeadu) { bad++; printf("slot %d written (%u %u), should be untouched\n", i, r[i].x, r[i].y); }
    }
    printf("%s\n", bad ? "MISMATCH" : "global_load_lds_dwordx4: lane T, piece n at 16 T + 32 n, masked lanes write nothing: OK");
    return bad != 0;
}
