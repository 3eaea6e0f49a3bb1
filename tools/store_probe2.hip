// store_probe2.hip -- closer to the main launch than store_probe.hip: three blocks of 256 threads per CU, 48 KB of
// LDS per block, the stores' addresses in 64-bit VGPR pairs, the time inside the flush read with s_memtime like
// -DGACT_STAMPS_FLUSH does.  Per variant: clocks per trip, clocks inside the flush.
//   hipcc --offload-arch=gfx950 -O3 -o build/store_probe2 tools/store_probe2.hip && build/store_probe2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(256, 3) void probe(int iters, int valu_blocks, uint32_t seed, uint4 *ws,
                                                                     unsigned long long *cycles, unsigned long long *flush, uint32_t *sink)
{
    __shared__ uint32_t lds[12 * 1024];                 // 48 KB
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = seed * (k + 3) + threadIdx.x;
    uint32_t x = seed | 1;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    lds[threadIdx.x] = seed;
    uint4 *p = ws + (size_t)wave * 27520 + lane;        // 440,320 bytes per wave, rows of 128 uint4
    __syncthreads();
    unsigned long long fl = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        for (int b = 0; b < valu_blocks; b++) {
            if (MODE & 8) {
                uint32_t w;
                asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((uint32_t)(lane * 2 + (b & 7) * 128)) : "memory");
                x ^= w & 0x10000u;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_add_u32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x));
        }
        const unsigned long long f0 = __builtin_amdgcn_s_memtime();
        if (MODE & 1) {
            uint32_t w[16];
#pragma unroll
            for (int k = 0; k < 16; k++) w[k] = __builtin_amdgcn_perm(a[k], a[(k + 1) & 15], 0x05040100u);
            p[0] = make_uint4(w[0], w[1], w[2], w[3]);
            p[64] = make_uint4(w[4], w[5], w[6], w[7]);
            p[128] = make_uint4(w[8], w[9], w[10], w[11]);
            p[192] = make_uint4(w[12], w[13], w[14], w[15]);
            if (MODE & 2) p += 256; else if ((it & 31) == 31) p -= 0;
            if ((MODE & 2) && (it & 31) == 31) p -= 32 * 256;
            // the registers the stores read, written again right behind them (what a kernel short of registers does
            // with its temporaries): by a vector instruction (16), by an LDS read's return (32)
            if (MODE & 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[k]) : "v"(x));
            }
            if (MODE & 32) {
#pragma unroll
                for (int k = 0; k < 16; k += 4) asm volatile("ds_read_b32 %0, %1" : "+v"(w[k]) : "v"((uint32_t)(lane * 4)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (MODE & 48) {
#pragma unroll
                for (int k = 0; k < 16; k++) x ^= w[k] & 0x10000u;
            }
        }
        if (MODE & 4) {                                 // the same bytes through LDS writes
            uint4 *q = reinterpret_cast<uint4 *>(lds) + (threadIdx.x >> 6) * 256 + lane;
            q[0] = make_uint4(a[0], a[1], a[2], a[3]); q[64] = make_uint4(a[4], a[5], a[6], a[7]);
            q[128] = make_uint4(a[8], a[9], a[10], a[11]); q[192] = make_uint4(a[12], a[13], a[14], a[15]);
        }
        fl += __builtin_amdgcn_s_memtime() - f0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) r ^= a[k];
    if (r == 0x12345678u) sink[0] = r + lds[lane];
    if (lane == 0) { cycles[wave] = t1 - t0; flush[wave] = fl; }
}

template <class K> void run(K kern, const char *name, int valu_blocks, uint4 *ws, bool &first)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 3, iters = 400;
    unsigned long long *d, *f;
    uint32_t *sink;
    (void)hipMalloc(&d, (size_t)blocks * 4 * 8); (void)hipMalloc(&f, (size_t)blocks * 4 * 8); (void)hipMalloc(&sink, 4);
    std::vector<unsigned long long> h((size_t)blocks * 4), g((size_t)blocks * 4);
    double best = 1e30, bf = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, valu_blocks, 77u + rep, ws, d, f, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(g.data(), f, g.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end()); std::sort(g.begin(), g.end());
        if (rep && (double)h[h.size() / 2] < best) { best = (double)h[h.size() / 2]; bf = (double)g[g.size() / 2]; }
    }
    (void)hipFree(d); (void)hipFree(f); (void)hipFree(sink);
    printf("%s  {\"variant\": \"%s\", \"vector_instructions_per_trip\": %d, \"clocks_per_trip\": %.0f, \"clocks_inside_the_flush\": %.0f}",
           first ? "" : ",\n", name, valu_blocks * 32, best / iters, bf / iters);
    first = false;
}

int main()
{
    uint4 *ws;
    if (hipMalloc(&ws, (size_t)256 * 12 * 440320 + (1 << 20)) != hipSuccess) { fprintf(stderr, "no memory\n"); return 1; }
    printf("{\"note\": \"3 blocks of 256 threads per CU (three waves per SIMD), 48 KB LDS per block; a trip = N independent vector instructions, then the flush; "
           "s_memtime, median over waves\",\n \"rows\": [\n");
    bool first = true;
    for (int vb : {44}) {
        run(probe<0>, "nothing in the flush", vb, ws, first);
        run(probe<1>, "16 v_perm_b32 + 4 x global_store_dwordx4 (1 KB each), the same 4 KB every trip", vb, ws, first);
        run(probe<3>, "16 v_perm_b32 + 4 x global_store_dwordx4, advancing 4 KB per trip (32 trips, then again)", vb, ws, first);
        run(probe<4>, "4 x ds_write_b128", vb, ws, first);
        run(probe<8>, "nothing in the flush; an LDS read waited for per 32 instructions", vb, ws, first);
        run(probe<11>, "stores advancing + the LDS reads", vb, ws, first);
        run(probe<3 + 16>, "stores advancing, their data registers rewritten by v_add_u32 right behind them", vb, ws, first);
        run(probe<3 + 32>, "stores advancing, four of their data registers rewritten by ds_read_b32 right behind them", vb, ws, first);
        run(probe<11 + 16>, "stores advancing + the LDS reads + data registers rewritten by v_add_u32", vb, ws, first);
    }
    printf("\n ]}\n");
    return 0;
}
