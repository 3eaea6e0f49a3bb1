// issue_probe.hip -- how often can one SIMD of gfx950 issue the vector instructions the GACT kernels are made of?
//
// MI355X_MICROARCH.md describes the CU as four SIMD-32 units, a wave64 v_fma_f32 taking 2 cycles (4 for a wave
// alone); the round-1 probe and the main kernel's counters said 4 cycles per wave64 integer instruction.  This probe
// settles it for the instruction mix that matters here, by resident waves per SIMD (1..8), for streams of
// independent instructions (16 accumulators in rotation) and for dependent chains (every instruction reads the
// result of the one before), timed with s_memtime inside the wave (shader clocks, no launch overhead, no assumed
// frequency).  Per cell of the table: cycles between two issues of ONE wave, and (divided by the resident waves)
// the SIMD's issue interval.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe tools/issue_probe.hip && tools/issue_probe > table.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_ACC 16
#define PROBE(NAME, INDEP_ASM, DEP_ASM)                                                               \
    template <bool DEP> __global__ __launch_bounds__(256) void NAME(int iters, uint32_t seed,         \
                                                                    unsigned long long *cycles, uint32_t *sink) \
    {                                                                                                 \
        uint32_t a[N_ACC];                                                                            \
        _Pragma("unroll") for (int k = 0; k < N_ACC; k++) a[k] = seed * (k + 3) + threadIdx.x;        \
        uint32_t x = seed | 1; const uint32_t y = seed ^ 0x00030003u;                                          \
        (void)x; (void)y;                                                                             \
        __syncthreads();                                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < iters; it++) {                                                          \
            if (DEP) { _Pragma("unroll") for (int k = 0; k < N_ACC; k++) { DEP_ASM } }                \
            else     { _Pragma("unroll") for (int k = 0; k < N_ACC; k++) { INDEP_ASM } }              \
        }                                                                                             \
        asm volatile("s_nop 0" ::: "memory");                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        uint32_t r = 0;                                                                               \
        _Pragma("unroll") for (int k = 0; k < N_ACC; k++) r ^= a[k];                                  \
        if (r == 0x12345678u) sink[0] = r;                                                            \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;  \
    }

// register banks (VGPR number mod 4): the same instruction with every source in one bank and in different banks
PROBE(p_bank_max3_same, if (k == 0) asm volatile("v_pk_maximum3_f16 v40, v20, v24, v28\n\tv_pk_maximum3_f16 v41, v20, v24, v28\n\tv_pk_maximum3_f16 v42, v20, v24, v28\n\tv_pk_maximum3_f16 v43, v20, v24, v28\n\tv_pk_maximum3_f16 v44, v20, v24, v28\n\tv_pk_maximum3_f16 v45, v20, v24, v28\n\tv_pk_maximum3_f16 v46, v20, v24, v28\n\tv_pk_maximum3_f16 v47, v20, v24, v28\n\tv_pk_maximum3_f16 v48, v20, v24, v28\n\tv_pk_maximum3_f16 v49, v20, v24, v28\n\tv_pk_maximum3_f16 v50, v20, v24, v28\n\tv_pk_maximum3_f16 v51, v20, v24, v28\n\tv_pk_maximum3_f16 v52, v20, v24, v28\n\tv_pk_maximum3_f16 v53, v20, v24, v28\n\tv_pk_maximum3_f16 v54, v20, v24, v28\n\tv_pk_maximum3_f16 v55, v20, v24, v28" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_pk_maximum3_f16 v40, v20, v24, v28\n\tv_pk_maximum3_f16 v41, v20, v24, v28\n\tv_pk_maximum3_f16 v42, v20, v24, v28\n\tv_pk_maximum3_f16 v43, v20, v24, v28\n\tv_pk_maximum3_f16 v44, v20, v24, v28\n\tv_pk_maximum3_f16 v45, v20, v24, v28\n\tv_pk_maximum3_f16 v46, v20, v24, v28\n\tv_pk_maximum3_f16 v47, v20, v24, v28\n\tv_pk_maximum3_f16 v48, v20, v24, v28\n\tv_pk_maximum3_f16 v49, v20, v24, v28\n\tv_pk_maximum3_f16 v50, v20, v24, v28\n\tv_pk_maximum3_f16 v51, v20, v24, v28\n\tv_pk_maximum3_f16 v52, v20, v24, v28\n\tv_pk_maximum3_f16 v53, v20, v24, v28\n\tv_pk_maximum3_f16 v54, v20, v24, v28\n\tv_pk_maximum3_f16 v55, v20, v24, v28" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_max3_diff, if (k == 0) asm volatile("v_pk_maximum3_f16 v40, v20, v21, v22\n\tv_pk_maximum3_f16 v41, v20, v21, v22\n\tv_pk_maximum3_f16 v42, v20, v21, v22\n\tv_pk_maximum3_f16 v43, v20, v21, v22\n\tv_pk_maximum3_f16 v44, v20, v21, v22\n\tv_pk_maximum3_f16 v45, v20, v21, v22\n\tv_pk_maximum3_f16 v46, v20, v21, v22\n\tv_pk_maximum3_f16 v47, v20, v21, v22\n\tv_pk_maximum3_f16 v48, v20, v21, v22\n\tv_pk_maximum3_f16 v49, v20, v21, v22\n\tv_pk_maximum3_f16 v50, v20, v21, v22\n\tv_pk_maximum3_f16 v51, v20, v21, v22\n\tv_pk_maximum3_f16 v52, v20, v21, v22\n\tv_pk_maximum3_f16 v53, v20, v21, v22\n\tv_pk_maximum3_f16 v54, v20, v21, v22\n\tv_pk_maximum3_f16 v55, v20, v21, v22" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_pk_maximum3_f16 v40, v20, v21, v22\n\tv_pk_maximum3_f16 v41, v20, v21, v22\n\tv_pk_maximum3_f16 v42, v20, v21, v22\n\tv_pk_maximum3_f16 v43, v20, v21, v22\n\tv_pk_maximum3_f16 v44, v20, v21, v22\n\tv_pk_maximum3_f16 v45, v20, v21, v22\n\tv_pk_maximum3_f16 v46, v20, v21, v22\n\tv_pk_maximum3_f16 v47, v20, v21, v22\n\tv_pk_maximum3_f16 v48, v20, v21, v22\n\tv_pk_maximum3_f16 v49, v20, v21, v22\n\tv_pk_maximum3_f16 v50, v20, v21, v22\n\tv_pk_maximum3_f16 v51, v20, v21, v22\n\tv_pk_maximum3_f16 v52, v20, v21, v22\n\tv_pk_maximum3_f16 v53, v20, v21, v22\n\tv_pk_maximum3_f16 v54, v20, v21, v22\n\tv_pk_maximum3_f16 v55, v20, v21, v22" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_pkmax_same, if (k == 0) asm volatile("v_pk_max_i16 v40, v20, v24\n\tv_pk_max_i16 v41, v20, v24\n\tv_pk_max_i16 v42, v20, v24\n\tv_pk_max_i16 v43, v20, v24\n\tv_pk_max_i16 v44, v20, v24\n\tv_pk_max_i16 v45, v20, v24\n\tv_pk_max_i16 v46, v20, v24\n\tv_pk_max_i16 v47, v20, v24\n\tv_pk_max_i16 v48, v20, v24\n\tv_pk_max_i16 v49, v20, v24\n\tv_pk_max_i16 v50, v20, v24\n\tv_pk_max_i16 v51, v20, v24\n\tv_pk_max_i16 v52, v20, v24\n\tv_pk_max_i16 v53, v20, v24\n\tv_pk_max_i16 v54, v20, v24\n\tv_pk_max_i16 v55, v20, v24" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_pk_max_i16 v40, v20, v24\n\tv_pk_max_i16 v41, v20, v24\n\tv_pk_max_i16 v42, v20, v24\n\tv_pk_max_i16 v43, v20, v24\n\tv_pk_max_i16 v44, v20, v24\n\tv_pk_max_i16 v45, v20, v24\n\tv_pk_max_i16 v46, v20, v24\n\tv_pk_max_i16 v47, v20, v24\n\tv_pk_max_i16 v48, v20, v24\n\tv_pk_max_i16 v49, v20, v24\n\tv_pk_max_i16 v50, v20, v24\n\tv_pk_max_i16 v51, v20, v24\n\tv_pk_max_i16 v52, v20, v24\n\tv_pk_max_i16 v53, v20, v24\n\tv_pk_max_i16 v54, v20, v24\n\tv_pk_max_i16 v55, v20, v24" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_pkmax_diff, if (k == 0) asm volatile("v_pk_max_i16 v40, v20, v21\n\tv_pk_max_i16 v41, v20, v21\n\tv_pk_max_i16 v42, v20, v21\n\tv_pk_max_i16 v43, v20, v21\n\tv_pk_max_i16 v44, v20, v21\n\tv_pk_max_i16 v45, v20, v21\n\tv_pk_max_i16 v46, v20, v21\n\tv_pk_max_i16 v47, v20, v21\n\tv_pk_max_i16 v48, v20, v21\n\tv_pk_max_i16 v49, v20, v21\n\tv_pk_max_i16 v50, v20, v21\n\tv_pk_max_i16 v51, v20, v21\n\tv_pk_max_i16 v52, v20, v21\n\tv_pk_max_i16 v53, v20, v21\n\tv_pk_max_i16 v54, v20, v21\n\tv_pk_max_i16 v55, v20, v21" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_pk_max_i16 v40, v20, v21\n\tv_pk_max_i16 v41, v20, v21\n\tv_pk_max_i16 v42, v20, v21\n\tv_pk_max_i16 v43, v20, v21\n\tv_pk_max_i16 v44, v20, v21\n\tv_pk_max_i16 v45, v20, v21\n\tv_pk_max_i16 v46, v20, v21\n\tv_pk_max_i16 v47, v20, v21\n\tv_pk_max_i16 v48, v20, v21\n\tv_pk_max_i16 v49, v20, v21\n\tv_pk_max_i16 v50, v20, v21\n\tv_pk_max_i16 v51, v20, v21\n\tv_pk_max_i16 v52, v20, v21\n\tv_pk_max_i16 v53, v20, v21\n\tv_pk_max_i16 v54, v20, v21\n\tv_pk_max_i16 v55, v20, v21" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_perm_same, if (k == 0) asm volatile("v_perm_b32 v40, v20, v24, v28\n\tv_perm_b32 v41, v20, v24, v28\n\tv_perm_b32 v42, v20, v24, v28\n\tv_perm_b32 v43, v20, v24, v28\n\tv_perm_b32 v44, v20, v24, v28\n\tv_perm_b32 v45, v20, v24, v28\n\tv_perm_b32 v46, v20, v24, v28\n\tv_perm_b32 v47, v20, v24, v28\n\tv_perm_b32 v48, v20, v24, v28\n\tv_perm_b32 v49, v20, v24, v28\n\tv_perm_b32 v50, v20, v24, v28\n\tv_perm_b32 v51, v20, v24, v28\n\tv_perm_b32 v52, v20, v24, v28\n\tv_perm_b32 v53, v20, v24, v28\n\tv_perm_b32 v54, v20, v24, v28\n\tv_perm_b32 v55, v20, v24, v28" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_perm_b32 v40, v20, v24, v28\n\tv_perm_b32 v41, v20, v24, v28\n\tv_perm_b32 v42, v20, v24, v28\n\tv_perm_b32 v43, v20, v24, v28\n\tv_perm_b32 v44, v20, v24, v28\n\tv_perm_b32 v45, v20, v24, v28\n\tv_perm_b32 v46, v20, v24, v28\n\tv_perm_b32 v47, v20, v24, v28\n\tv_perm_b32 v48, v20, v24, v28\n\tv_perm_b32 v49, v20, v24, v28\n\tv_perm_b32 v50, v20, v24, v28\n\tv_perm_b32 v51, v20, v24, v28\n\tv_perm_b32 v52, v20, v24, v28\n\tv_perm_b32 v53, v20, v24, v28\n\tv_perm_b32 v54, v20, v24, v28\n\tv_perm_b32 v55, v20, v24, v28" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_perm_diff, if (k == 0) asm volatile("v_perm_b32 v40, v20, v21, v22\n\tv_perm_b32 v41, v20, v21, v22\n\tv_perm_b32 v42, v20, v21, v22\n\tv_perm_b32 v43, v20, v21, v22\n\tv_perm_b32 v44, v20, v21, v22\n\tv_perm_b32 v45, v20, v21, v22\n\tv_perm_b32 v46, v20, v21, v22\n\tv_perm_b32 v47, v20, v21, v22\n\tv_perm_b32 v48, v20, v21, v22\n\tv_perm_b32 v49, v20, v21, v22\n\tv_perm_b32 v50, v20, v21, v22\n\tv_perm_b32 v51, v20, v21, v22\n\tv_perm_b32 v52, v20, v21, v22\n\tv_perm_b32 v53, v20, v21, v22\n\tv_perm_b32 v54, v20, v21, v22\n\tv_perm_b32 v55, v20, v21, v22" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_perm_b32 v40, v20, v21, v22\n\tv_perm_b32 v41, v20, v21, v22\n\tv_perm_b32 v42, v20, v21, v22\n\tv_perm_b32 v43, v20, v21, v22\n\tv_perm_b32 v44, v20, v21, v22\n\tv_perm_b32 v45, v20, v21, v22\n\tv_perm_b32 v46, v20, v21, v22\n\tv_perm_b32 v47, v20, v21, v22\n\tv_perm_b32 v48, v20, v21, v22\n\tv_perm_b32 v49, v20, v21, v22\n\tv_perm_b32 v50, v20, v21, v22\n\tv_perm_b32 v51, v20, v21, v22\n\tv_perm_b32 v52, v20, v21, v22\n\tv_perm_b32 v53, v20, v21, v22\n\tv_perm_b32 v54, v20, v21, v22\n\tv_perm_b32 v55, v20, v21, v22" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_add_same, if (k == 0) asm volatile("v_add_u32 v40, v20, v24\n\tv_add_u32 v41, v20, v24\n\tv_add_u32 v42, v20, v24\n\tv_add_u32 v43, v20, v24\n\tv_add_u32 v44, v20, v24\n\tv_add_u32 v45, v20, v24\n\tv_add_u32 v46, v20, v24\n\tv_add_u32 v47, v20, v24\n\tv_add_u32 v48, v20, v24\n\tv_add_u32 v49, v20, v24\n\tv_add_u32 v50, v20, v24\n\tv_add_u32 v51, v20, v24\n\tv_add_u32 v52, v20, v24\n\tv_add_u32 v53, v20, v24\n\tv_add_u32 v54, v20, v24\n\tv_add_u32 v55, v20, v24" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_add_u32 v40, v20, v24\n\tv_add_u32 v41, v20, v24\n\tv_add_u32 v42, v20, v24\n\tv_add_u32 v43, v20, v24\n\tv_add_u32 v44, v20, v24\n\tv_add_u32 v45, v20, v24\n\tv_add_u32 v46, v20, v24\n\tv_add_u32 v47, v20, v24\n\tv_add_u32 v48, v20, v24\n\tv_add_u32 v49, v20, v24\n\tv_add_u32 v50, v20, v24\n\tv_add_u32 v51, v20, v24\n\tv_add_u32 v52, v20, v24\n\tv_add_u32 v53, v20, v24\n\tv_add_u32 v54, v20, v24\n\tv_add_u32 v55, v20, v24" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_add_diff, if (k == 0) asm volatile("v_add_u32 v40, v20, v21\n\tv_add_u32 v41, v20, v21\n\tv_add_u32 v42, v20, v21\n\tv_add_u32 v43, v20, v21\n\tv_add_u32 v44, v20, v21\n\tv_add_u32 v45, v20, v21\n\tv_add_u32 v46, v20, v21\n\tv_add_u32 v47, v20, v21\n\tv_add_u32 v48, v20, v21\n\tv_add_u32 v49, v20, v21\n\tv_add_u32 v50, v20, v21\n\tv_add_u32 v51, v20, v21\n\tv_add_u32 v52, v20, v21\n\tv_add_u32 v53, v20, v21\n\tv_add_u32 v54, v20, v21\n\tv_add_u32 v55, v20, v21" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_add_u32 v40, v20, v21\n\tv_add_u32 v41, v20, v21\n\tv_add_u32 v42, v20, v21\n\tv_add_u32 v43, v20, v21\n\tv_add_u32 v44, v20, v21\n\tv_add_u32 v45, v20, v21\n\tv_add_u32 v46, v20, v21\n\tv_add_u32 v47, v20, v21\n\tv_add_u32 v48, v20, v21\n\tv_add_u32 v49, v20, v21\n\tv_add_u32 v50, v20, v21\n\tv_add_u32 v51, v20, v21\n\tv_add_u32 v52, v20, v21\n\tv_add_u32 v53, v20, v21\n\tv_add_u32 v54, v20, v21\n\tv_add_u32 v55, v20, v21" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_pkmad_same, if (k == 0) asm volatile("v_pk_mad_u16 v40, v20, v24, v28\n\tv_pk_mad_u16 v41, v20, v24, v28\n\tv_pk_mad_u16 v42, v20, v24, v28\n\tv_pk_mad_u16 v43, v20, v24, v28\n\tv_pk_mad_u16 v44, v20, v24, v28\n\tv_pk_mad_u16 v45, v20, v24, v28\n\tv_pk_mad_u16 v46, v20, v24, v28\n\tv_pk_mad_u16 v47, v20, v24, v28\n\tv_pk_mad_u16 v48, v20, v24, v28\n\tv_pk_mad_u16 v49, v20, v24, v28\n\tv_pk_mad_u16 v50, v20, v24, v28\n\tv_pk_mad_u16 v51, v20, v24, v28\n\tv_pk_mad_u16 v52, v20, v24, v28\n\tv_pk_mad_u16 v53, v20, v24, v28\n\tv_pk_mad_u16 v54, v20, v24, v28\n\tv_pk_mad_u16 v55, v20, v24, v28" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_pk_mad_u16 v40, v20, v24, v28\n\tv_pk_mad_u16 v41, v20, v24, v28\n\tv_pk_mad_u16 v42, v20, v24, v28\n\tv_pk_mad_u16 v43, v20, v24, v28\n\tv_pk_mad_u16 v44, v20, v24, v28\n\tv_pk_mad_u16 v45, v20, v24, v28\n\tv_pk_mad_u16 v46, v20, v24, v28\n\tv_pk_mad_u16 v47, v20, v24, v28\n\tv_pk_mad_u16 v48, v20, v24, v28\n\tv_pk_mad_u16 v49, v20, v24, v28\n\tv_pk_mad_u16 v50, v20, v24, v28\n\tv_pk_mad_u16 v51, v20, v24, v28\n\tv_pk_mad_u16 v52, v20, v24, v28\n\tv_pk_mad_u16 v53, v20, v24, v28\n\tv_pk_mad_u16 v54, v20, v24, v28\n\tv_pk_mad_u16 v55, v20, v24, v28" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
PROBE(p_bank_pkmad_diff, if (k == 0) asm volatile("v_pk_mad_u16 v40, v20, v21, v22\n\tv_pk_mad_u16 v41, v20, v21, v22\n\tv_pk_mad_u16 v42, v20, v21, v22\n\tv_pk_mad_u16 v43, v20, v21, v22\n\tv_pk_mad_u16 v44, v20, v21, v22\n\tv_pk_mad_u16 v45, v20, v21, v22\n\tv_pk_mad_u16 v46, v20, v21, v22\n\tv_pk_mad_u16 v47, v20, v21, v22\n\tv_pk_mad_u16 v48, v20, v21, v22\n\tv_pk_mad_u16 v49, v20, v21, v22\n\tv_pk_mad_u16 v50, v20, v21, v22\n\tv_pk_mad_u16 v51, v20, v21, v22\n\tv_pk_mad_u16 v52, v20, v21, v22\n\tv_pk_mad_u16 v53, v20, v21, v22\n\tv_pk_mad_u16 v54, v20, v21, v22\n\tv_pk_mad_u16 v55, v20, v21, v22" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");, if (k == 0) asm volatile("v_pk_mad_u16 v40, v20, v21, v22\n\tv_pk_mad_u16 v41, v20, v21, v22\n\tv_pk_mad_u16 v42, v20, v21, v22\n\tv_pk_mad_u16 v43, v20, v21, v22\n\tv_pk_mad_u16 v44, v20, v21, v22\n\tv_pk_mad_u16 v45, v20, v21, v22\n\tv_pk_mad_u16 v46, v20, v21, v22\n\tv_pk_mad_u16 v47, v20, v21, v22\n\tv_pk_mad_u16 v48, v20, v21, v22\n\tv_pk_mad_u16 v49, v20, v21, v22\n\tv_pk_mad_u16 v50, v20, v21, v22\n\tv_pk_mad_u16 v51, v20, v21, v22\n\tv_pk_mad_u16 v52, v20, v21, v22\n\tv_pk_mad_u16 v53, v20, v21, v22\n\tv_pk_mad_u16 v54, v20, v21, v22\n\tv_pk_mad_u16 v55, v20, v21, v22" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");)
// independent: accumulator k only; dependent: accumulator 0 again and again
PROBE(p_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_max_i32, asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_pk_add_i16, asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_pk_max_i16, asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_pk_mad_u16, asm volatile("v_pk_mad_u16 %0, %0, 4, %1 op_sel_hi:[1,0,1]" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_mad_u16 %0, %0, 4, %1 op_sel_hi:[1,0,1]" : "+v"(a[0]) : "v"(x));)
PROBE(p_perm_b32, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_and_or_b32, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_bitop3_b32, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_mov_dpp_shr1, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(a[0]));)
PROBE(p_and_b32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_or_b32, asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_xor_b32, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_max_u32, asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_lshlrev_b32, asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[k]));, asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[0]));)
PROBE(p_lshl_or_b32, asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_add3_u32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(x) : "vcc");, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0]) : "v"(x) : "vcc");)
PROBE(p_pk_add_u16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_pk_add_sgpr, asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "s"(x));, asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[0]) : "s"(x));)
PROBE(p_add_sgpr, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "s"(x));, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "s"(x));)
PROBE(p_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
// the GACT score step's own mix, one slot: perm, add, max, max / max, add, max, max (8 instructions per k)
PROBE(p_lin_slot,
      asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1\n\t"
                   "v_pk_max_i16 %0, %0, %2\n\tv_pk_add_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2" : "+v"(a[k]) : "v"(x), "v"(y));,
      asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1\n\t"
                   "v_pk_max_i16 %0, %0, %2\n\tv_pk_add_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2" : "+v"(a[0]) : "v"(x), "v"(y));)

// mixed streams: does the order of fast-class (v_add_u32) and slow-class (v_pk_max_i16) instructions matter?
//   alternating: add, max, add, max ... each on its own accumulator (independent) / each reading the one before
//   (dependent: add -> max -> add ...);  blocks: eight adds, then eight maxes
PROBE(p_alt_add_max,
      asm volatile("v_add_u32 %0, %0, %2\n\tv_pk_max_i16 %1, %1, %2" : "+v"(a[k]), "+v"(a[(k + 8) % N_ACC]) : "v"(x));,
      asm volatile("v_add_u32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_blk_add_max,
      if (k < 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x)); else asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x));,
      if (k < 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(x)); else asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[1]) : "v"(x));)
// the chain of the linear-gap step: sub -> max -> sub -> max; one chain / two chains side by side
PROBE(p_chain_sub_max,
      asm volatile("v_sub_u32 %0, %1, %2\n\tv_sub_u32 %3, %4, %2\n\tv_pk_max_i16 %1, %0, %2\n\tv_pk_max_i16 %4, %3, %2"
                   : "+v"(a[2]), "+v"(a[0]), "+v"(x) , "+v"(a[3]), "+v"(a[1]));,
      asm volatile("v_sub_u32 %0, %1, %2\n\tv_pk_max_i16 %1, %0, %2\n\tv_sub_u32 %0, %1, %2\n\tv_pk_max_i16 %1, %0, %2"
                   : "+v"(a[2]), "+v"(a[0]), "+v"(x));)

// round 3: the half-precision maxima that order positive int16 like integers, and the score step built on them
PROBE(p_pk_maximum3_f16, asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_pk_max_f16, asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_pk_max_u16, asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
PROBE(p_max3_i32, asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_max3_i16, asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));, asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));)
PROBE(p_pk_sub_i16, asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x));, asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x));)
// perm, add, max3 | sub, max   (5 instructions per k)
PROBE(p_lin_slot5,
      asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_add_u32 %0, %0, %1\n\tv_pk_maximum3_f16 %0, %0, %2, %1\n\t"
                   "v_sub_u32 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x), "v"(y));,
      asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_add_u32 %0, %0, %1\n\tv_pk_maximum3_f16 %0, %0, %2, %1\n\t"
                   "v_sub_u32 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x), "v"(y));)
// perm, add, max, max | sub, max   (6 instructions per k: what the pass was before the fold)
PROBE(p_lin_slot6,
      asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_add_u32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1\n\t"
                   "v_sub_u32 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(x), "v"(y));,
      asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_add_u32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1\n\t"
                   "v_sub_u32 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x), "v"(y));)

struct Cell { double wave_cycles, simd_cycles; };

template <class K> Cell run(K kern, int waves_per_simd, int instr_per_k)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int iters = 2048;
    const int blocks = p.multiProcessorCount * waves_per_simd;        // 256 threads = 4 waves = one per SIMD of a CU
    unsigned long long *d_cycles;
    uint32_t *sink;
    (void)hipMalloc(&d_cycles, (size_t)blocks * 4 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 4);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, 1234u + rep, d_cycles, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d_cycles, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        if (rep) best = std::min(best, med);
    }
    (void)hipFree(d_cycles);
    (void)hipFree(sink);
    Cell c;
    c.wave_cycles = best / ((double)iters * N_ACC * instr_per_k);
    c.simd_cycles = c.wave_cycles / waves_per_simd;
    return c;
}

#define ROW(NAME, KERN, IPK)                                                                           \
    for (int dep = 0; dep < 2; dep++) {                                                                \
        printf("%s  {\"instruction\": \"%s\", \"stream\": \"%s\", \"cycles\": {", first ? "" : ",\n", NAME, dep ? "dependent" : "independent"); \
        first = false;                                                                                 \
        for (int w = 1; w <= 8; w++) {                                                                 \
            const Cell c = dep ? run(KERN<true>, w, IPK) : run(KERN<false>, w, IPK);                   \
            printf("%s\"%d\": [%.2f, %.2f]", w > 1 ? ", " : "", w, c.wave_cycles, c.simd_cycles);      \
        }                                                                                              \
        printf("}}");                                                                                  \
    }

int main(int argc, char **argv)
{
    const bool only_new = argc > 1;          // any argument: the rows added in round 3 only
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d,\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    printf(" \"note\": \"cycles[w] = [shader clocks between two issues of one wave, the same divided by w = the SIMD's issue "
           "interval], w = resident waves per SIMD; s_memtime inside the wave, median over all waves\",\n \"rows\": [\n");
    bool first = true;
    ROW("v_pk_maximum3_f16, sources v20, v24, v28 (one VGPR bank)", p_bank_max3_same, 16)
    ROW("v_pk_maximum3_f16, sources v20, v21, v22 (three banks)", p_bank_max3_diff, 16)
    ROW("v_pk_max_i16, sources v20, v24 (one VGPR bank)", p_bank_pkmax_same, 16)
    ROW("v_pk_max_i16, sources v20, v21 (two banks)", p_bank_pkmax_diff, 16)
    ROW("v_perm_b32, sources v20, v24, v28 (one VGPR bank)", p_bank_perm_same, 16)
    ROW("v_perm_b32, sources v20, v21, v22 (three banks)", p_bank_perm_diff, 16)
    ROW("v_add_u32, sources v20, v24 (one VGPR bank)", p_bank_add_same, 16)
    ROW("v_add_u32, sources v20, v21 (two banks)", p_bank_add_diff, 16)
    ROW("v_pk_mad_u16, sources v20, v24, v28 (one VGPR bank)", p_bank_pkmad_same, 16)
    ROW("v_pk_mad_u16, sources v20, v21, v22 (three banks)", p_bank_pkmad_diff, 16)
    ROW("v_pk_maximum3_f16", p_pk_maximum3_f16, 1)
    ROW("v_pk_max_f16", p_pk_max_f16, 1)
    ROW("v_pk_max_u16", p_pk_max_u16, 1)
    ROW("v_max3_i32", p_max3_i32, 1)
    ROW("v_max3_i16", p_max3_i16, 1)
    ROW("v_pk_sub_i16", p_pk_sub_i16, 1)
    ROW("linear-gap slot, 5 instructions (perm, add, maximum3 | sub, max)", p_lin_slot5, 5)
    ROW("linear-gap slot, 6 instructions (perm, add, max, max | sub, max)", p_lin_slot6, 6)
    if (only_new) { printf("\n ]}\n"); return 0; }
    ROW("v_add_u32", p_add_u32, 1)
    ROW("v_max_i32", p_max_i32, 1)
    ROW("v_pk_add_i16", p_pk_add_i16, 1)
    ROW("v_pk_max_i16", p_pk_max_i16, 1)
    ROW("v_pk_mad_u16", p_pk_mad_u16, 1)
    ROW("v_perm_b32", p_perm_b32, 1)
    ROW("v_and_or_b32", p_and_or_b32, 1)
    ROW("v_bitop3_b32", p_bitop3_b32, 1)
    ROW("v_mov_b32_dpp row_shr:1", p_mov_dpp_shr1, 1)
    ROW("v_and_b32", p_and_b32, 1)
    ROW("v_or_b32", p_or_b32, 1)
    ROW("v_xor_b32", p_xor_b32, 1)
    ROW("v_sub_u32", p_sub_u32, 1)
    ROW("v_max_u32", p_max_u32, 1)
    ROW("v_lshlrev_b32", p_lshlrev_b32, 1)
    ROW("v_lshl_or_b32", p_lshl_or_b32, 1)
    ROW("v_add3_u32", p_add3_u32, 1)
    ROW("v_mad_u32_u24", p_mad_u32_u24, 1)
    ROW("v_cndmask_b32", p_cndmask, 1)
    ROW("v_pk_add_u16", p_pk_add_u16, 1)
    ROW("v_pk_add_i16 (SGPR operand)", p_pk_add_sgpr, 1)
    ROW("v_add_u32 (SGPR operand)", p_add_sgpr, 1)
    ROW("v_fma_f32", p_fma_f32, 1)
    ROW("GACT linear-gap slot (perm, 2 add, 5 max)", p_lin_slot, 8)
    ROW("v_add_u32, v_pk_max_i16 alternating", p_alt_add_max, 2)
    ROW("8 x v_add_u32 then 8 x v_pk_max_i16", p_blk_add_max, 1)
    ROW("sub -> max chains: independent = two side by side, dependent = one", p_chain_sub_max, 4)
    printf("\n ]}\n");
    return 0;
}
