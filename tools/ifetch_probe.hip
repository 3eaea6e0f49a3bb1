// ifetch_probe.hip -- does the length of a straight-line loop body change what a SIMD issues?  tools/issue_probe.hip times
// the instruction mix of the linear-gap pass (perm, add, maximum3 | sub, max per column slot) in a loop of 80 instructions;
// the pass's own hot loops are 1,700 (eight pointer steps) and 300 instructions long, 13 KB and 2.5 KB of code, run by up to
// twelve waves of a CU at different places, and the two CUs of a pair share one instruction cache.  Same instructions, same
// registers, bodies of 80 x {1, 8, 32, 128} instructions (0.6 KB .. 80 KB), 1-4 resident waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ifetch_probe tools/ifetch_probe.hip && tools/ifetch_probe > table.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_ACC 16
template <int UNROLL, bool DEP>
__global__ __launch_bounds__(256) void probe(int iters, uint32_t seed, unsigned long long *cycles, uint32_t *sink)
{
    uint32_t a[N_ACC];
#pragma unroll
    for (int k = 0; k < N_ACC; k++) a[k] = seed * (k + 3) + threadIdx.x;
    const uint32_t x = seed | 1, y = seed ^ 0x00030003u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int k = 0; k < N_ACC; k++) {
                asm volatile("v_perm_b32 %0, %1, %2, %0\n\tv_add_u32 %0, %0, %1\n\tv_pk_maximum3_f16 %0, %0, %2, %1\n\t"
                             "v_sub_u32 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %1" : "+v"(a[DEP ? 0 : k]) : "v"(x), "v"(y));
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < N_ACC; k++) r ^= a[k];
    if (r == 0x12345678u) sink[0] = r;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <class K> double run(K kern, int waves_per_simd, int unroll)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int iters = std::max(8, 2048 / unroll);
    const int blocks = p.multiProcessorCount * waves_per_simd;
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
        if (rep) best = std::min(best, (double)h[h.size() / 2]);
    }
    (void)hipFree(d_cycles);
    (void)hipFree(sink);
    return best / ((double)iters * unroll * N_ACC * 5) / waves_per_simd;          // the SIMD's issue interval, cycles per instruction
}

#define ROWS(U)                                                                                                      \
    for (int dep = 0; dep < 2; dep++) {                                                                              \
        printf("%s  {\"body_instructions\": %d, \"body_bytes\": %d, \"stream\": \"%s\", \"simd_cycles_per_instruction\": {", first ? "" : ",\n", \
               U * N_ACC * 5, U * N_ACC * (8 + 4 + 8 + 4 + 8), dep ? "dependent" : "independent");                    \
        first = false;                                                                                               \
        for (int w = 1; w <= 4; w++) printf("%s\"%d\": %.2f", w > 1 ? ", " : "", w, dep ? run(probe<U, true>, w, U) : run(probe<U, false>, w, U)); \
        printf("}}");                                                                                                \
    }

int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    printf("{\"device\": \"%s\", \"note\": \"linear-gap slot (perm, add, maximum3 | sub, max), 16 accumulators; cycles between two issues of a SIMD, by resident waves per SIMD\",\n \"rows\": [\n", p.gcnArchName);
    bool first = true;
    ROWS(1) ROWS(8) ROWS(32) ROWS(128)
    printf("\n ]}\n");
    return 0;
}
