// store_probe.hip -- what does a 16-byte-per-lane global store cost a wave that is otherwise issuing vector instructions?
//
// The main launch stores four uint4 per lane every eight DP steps (~1,400 vector instructions): 0.3 % of its
// instructions.  Taking the stores out (results wrong, timing experiment) made the launch 17 % shorter, and it did
// not matter where the bytes went (every flush on the same 512 bytes: same time as the real footprint).  This probe
// reproduces the pattern in isolation: W waves per SIMD, each running `valu` independent vector instructions and then
// `nst` stores, again and again; s_memtime per wave, median.  Variants: store width, how the 64 lanes' addresses are
// laid out (one contiguous KB / four 256-byte pieces 55 KB apart like the engine's four tile groups), an LDS write
// instead, a wait for the stores before going on.
//   hipcc --offload-arch=gfx950 -O3 -o build/store_probe tools/store_probe.hip && build/store_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(256) void probe(int iters, int valu_blocks, uint32_t seed, uint4 *ws, size_t wave_stride_u4,
                                                                  unsigned long long *cycles, uint32_t *sink)
{
    __shared__ uint4 lds[4 * 64 * 4];
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = seed * (k + 3) + threadIdx.x;
    uint32_t x = seed | 1;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // MODE 1, 3, 4, 5: four groups of 16 lanes, 256 contiguous bytes each, groups 55,040 bytes apart (the engine's layout)
    // MODE 2: 64 lanes contiguous (1 KB)
    uint4 *base = ws + (size_t)wave * wave_stride_u4 + ((MODE == 2 || MODE == 9) ? lane : (lane >> 4) * 3440 + (lane & 15));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        for (int b = 0; b < valu_blocks; b++) {
            if (MODE == 7 || MODE == 8 || MODE == 9) {
                // a DP step's own LDS read: its result is waited for before the step's arithmetic, like the pass's ref row
                uint32_t w;
                asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((uint32_t)(lane * 2 + (b & 7) * 128)) : "memory");
                x ^= w & 0x10000u;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_add_u32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x));
        }
        uint4 *p = base + (size_t)(it & 31) * 32;
        const uint4 v0 = make_uint4(a[0], a[1], a[2], a[3]), v1 = make_uint4(a[4], a[5], a[6], a[7]);
        const uint4 v2 = make_uint4(a[8], a[9], a[10], a[11]), v3 = make_uint4(a[12], a[13], a[14], a[15]);
        if (MODE == 1 || MODE == 2 || MODE == 5 || MODE == 7 || MODE == 9) { p[0] = v0; p[16] = v1; p[1720] = v2; p[1736] = v3; }
        if (MODE == 3) { reinterpret_cast<uint2 *>(p)[0] = make_uint2(a[0], a[1]); reinterpret_cast<uint2 *>(p)[32] = make_uint2(a[2], a[3]);
                         reinterpret_cast<uint2 *>(p)[64] = make_uint2(a[4], a[5]); reinterpret_cast<uint2 *>(p)[96] = make_uint2(a[6], a[7]); }
        if (MODE == 4) { lds[(threadIdx.x >> 6) * 256 + lane] = v0; lds[(threadIdx.x >> 6) * 256 + 64 + lane] = v1;
                         lds[(threadIdx.x >> 6) * 256 + 128 + lane] = v2; lds[(threadIdx.x >> 6) * 256 + 192 + lane] = v3; }
        if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 6) { p[0] = v0; }                                    // one store instead of four
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) r ^= a[k];
    if (r == 0x12345678u) sink[0] = r + lds[lane].x;
    if (lane == 0) cycles[wave] = t1 - t0;
}

template <class K> double run(K kern, int waves_per_simd, int iters, int valu_blocks, uint4 *ws, size_t stride)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * waves_per_simd;
    unsigned long long *d;
    uint32_t *sink;
    (void)hipMalloc(&d, (size_t)blocks * 4 * 8);
    (void)hipMalloc(&sink, 4);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, valu_blocks, 77u + rep, ws, stride, d, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        if (rep) best = std::min(best, (double)h[h.size() / 2]);
    }
    (void)hipFree(d); (void)hipFree(sink);
    return best / iters;
}

int main()
{
    const size_t stride = 4 * 3440;                    // uint4 per wave: four groups x 55,040 bytes
    uint4 *ws;
    if (hipMalloc(&ws, (size_t)256 * 8 * 4 * stride * 16 + (1 << 20)) != hipSuccess) { fprintf(stderr, "no memory\n"); return 1; }
    const char *names[] = {"no store", "4 x dwordx4, four 256-B pieces per instruction (the engine's)", "4 x dwordx4, 1 KB contiguous per instruction",
                           "4 x dwordx2 (half the bytes)", "4 x ds_write_b128 instead", "4 x dwordx4 + s_waitcnt vmcnt(0)", "1 x dwordx4",
                           "the engine's stores + an LDS read waited for in every 32-instruction block", "no store + the same LDS reads",
                           "1 KB contiguous stores + the same LDS reads"};
    printf("{\"note\": \"shader clocks per loop trip of one wave: `valu` x 32 independent vector instructions (v_add_u32, v_pk_max_i16 alternating), then the stores; "
           "median over waves\",\n \"rows\": [\n");
    bool first = true;
    for (int w : {3}) {
        for (int vb : {8, 44}) {                       // 256 and 1,408 vector instructions between flushes (the engine: ~1,400)
            double base = 0;
            for (int mode = 0; mode < 10; mode++) {
                double c = 0;
                switch (mode) {
                    case 0: c = run(probe<0>, w, 400, vb, ws, stride); break;
                    case 1: c = run(probe<1>, w, 400, vb, ws, stride); break;
                    case 2: c = run(probe<2>, w, 400, vb, ws, stride); break;
                    case 3: c = run(probe<3>, w, 400, vb, ws, stride); break;
                    case 4: c = run(probe<4>, w, 400, vb, ws, stride); break;
                    case 5: c = run(probe<5>, w, 400, vb, ws, stride); break;
                    case 6: c = run(probe<6>, w, 400, vb, ws, stride); break;
                    case 7: c = run(probe<7>, w, 400, vb, ws, stride); break;
                    case 8: c = run(probe<8>, w, 400, vb, ws, stride); break;
                    case 9: c = run(probe<9>, w, 400, vb, ws, stride); break;
                }
                if (mode == 0) base = c;
                printf("%s  {\"waves_per_simd\": %d, \"vector_instructions_per_trip\": %d, \"variant\": \"%s\", \"clocks_per_trip\": %.0f, \"added_by_the_stores\": %.0f}",
                       first ? "" : ",\n", w, vb * 32, names[mode], c, c - base);
                first = false;
            }
        }
    }
    printf("\n ]}\n");
    return 0;
}
