// valu_probe.hip -- issue-rate microbenchmarks for the instruction types the GACT kernels use.
// hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip && tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BODY(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(int iters, int seed, int *sink)             \
    {                                                                                        \
        uint32_t a[16];                                                                      \
        _Pragma("unroll") for (int k = 0; k < 16; k++) a[k] = seed + k + threadIdx.x;       \
        const uint32_t inc = seed | 1;                                                       \
        uint32_t z = threadIdx.x >> 10;                                                      \
        for (int it = 0; it < iters; it++) {                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; k++) { ASM }                          \
        }                                                                                    \
        uint32_t r = z;                                                                      \
        _Pragma("unroll") for (int k = 0; k < 16; k++) r ^= a[k];                           \
        if (r == 0x7fffffff) sink[0] = r;                                                    \
    }

// each body = 2 VALU instructions per k
BODY(k_add_max, asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %2" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_pk_add_max, asm volatile("v_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_pk_add_s, asm volatile("v_pk_add_i16 %0, %0, %1\n\tv_pk_add_i16 %0, %0, %2" : "+v"(a[k]) : "s"(inc), "s"(seed));)
BODY(k_pk_mad, asm volatile("v_pk_mad_i16 %0, %0, %1, %2\n\tv_pk_mad_i16 %0, %0, %2, %1" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_pk_minu, asm volatile("v_pk_min_u16 %0, %0, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_cmp_addc, uint64_t m; asm volatile("v_cmp_ge_i32 %1, %0, %2\n\tv_addc_co_u32 %0, %1, %0, %0, %1" : "+v"(a[k]), "=&s"(m) : "v"(inc));)
BODY(k_sdwa_addc, uint64_t m; asm volatile("v_cmp_ge_i16_sdwa %1, %0, %2 src0_sel:WORD_1 src1_sel:WORD_1\n\tv_addc_co_u32 %0, %1, %0, %0, %1" : "+v"(a[k]), "=&s"(m) : "v"(inc));)
BODY(k_sdwa_cmp2, uint64_t m; asm volatile("v_cmp_ge_i16_sdwa %1, %0, %2 src0_sel:WORD_1 src1_sel:WORD_1\n\tv_cmp_eq_u16_sdwa %1, %0, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a[k]), "=&s"(m) : "v"(inc)); z ^= (uint32_t)m;)
BODY(k_cmp2, uint64_t m; asm volatile("v_cmp_ge_i32 %1, %0, %2\n\tv_cmp_eq_u32 %1, %0, %2" : "+v"(a[k]), "=&s"(m) : "v"(inc)); z ^= (uint32_t)m;)
BODY(k_addc2, asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[k]) : "v"(inc) : "vcc");)
BODY(k_max3_dpp, asm volatile("v_max3_i32 %0, %0, %1, %2\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_perm2, asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_perm_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_perm_add, asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_lshr_bfe, asm volatile("v_lshrrev_b32 %0, %1, %0\n\tv_bfe_u32 %0, %0, %1, 2" : "+v"(a[k]) : "v"(inc), "v"(seed));)
BODY(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_eq_u32 vcc, %0, %2" : "+v"(a[k]) : "v"(inc), "v"(seed) : "vcc");)

template <class K> void run(const char *name, K kern, int waves_per_simd)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int *sink; hipMalloc(&sink, 4);
    const int iters = 4096;
    const int blocks = p.multiProcessorCount * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, 1234 + rep, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * iters * 16.0 * 2.0;      // wave-instructions
    const double per_simd_per_s = instr / (best * 1e-3) / (p.multiProcessorCount * 4.0);
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f G wave-instr/s/SIMD -> %.2f cycles/instr at 2.4 GHz\n", name,
           waves_per_simd, best, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    hipFree(sink);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8}) {
        run("add_max", k_add_max, w);
        run("pk_add_max", k_pk_add_max, w);
        run("pk_add_sgpr", k_pk_add_s, w);
        run("pk_mad", k_pk_mad, w);
        run("pk_minu_xor", k_pk_minu, w);
        run("cmp_addc", k_cmp_addc, w);
        run("sdwa_addc", k_sdwa_addc, w);
        run("sdwa_cmp2", k_sdwa_cmp2, w);
        run("cmp2", k_cmp2, w);
        run("addc2", k_addc2, w);
        run("max3_dpp", k_max3_dpp, w);
        run("cndmask_cmp", k_cndmask, w);
        run("perm2", k_perm2, w);
        run("perm_pk_add", k_perm_add, w);
        run("lshr_bfe", k_lshr_bfe, w);
        printf("\n");
    }
    return 0;
}
