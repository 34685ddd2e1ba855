// ubench_valu.hip -- issue cost of the VALU instructions the trace kernel is
// made of, measured on the GPU it runs on: every SIMD holds 8 waves, each wave
// runs a long unrolled stream of ONE instruction kind on 8 independent chains.
// Prints cycles per wave-instruction per SIMD (chip clock from wall time is
// assumed 2.4 GHz; compare rows with each other, not with absolutes).
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITERS 20000
#define CHAINS 8

template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    uint32_t x[CHAINS]; float f[CHAINS]; v2f g[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { x[c] = seed + threadIdx.x * 977u + c; f[c] = 1.0f + (threadIdx.x + c) * 1e-3f; g[c] = v2f{f[c], f[c] + 1.f}; }
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) f[c] = __builtin_fmaf(f[c], 1.0001f, 0.5f);
            if (OP == 1) { uint64_t p = (uint64_t)0xD256D193u * x[c]; x[c] = (uint32_t)(p >> 32) ^ (uint32_t)p; }   // mad_u64 + xor
            if (OP == 2) x[c] = __umulhi(x[c], 0xD256D193u);
            if (OP == 3) x[c] = x[c] * 0xD256D193u;
            if (OP == 4) f[c] = __builtin_amdgcn_sinf(f[c]);
            if (OP == 5) f[c] = __builtin_amdgcn_rcpf(f[c]);
            if (OP == 6) x[c] = x[c] ^ (x[c] >> 7);           // 2 full-rate int ops
            if (OP == 7) x[c] = __umul24(x[c], 0x56D193) + 3;   // mad_u32_u24
            if (OP == 9) g[c] = __builtin_elementwise_fma(g[c], v2f{1.0001f, 0.9999f}, v2f{0.5f, 0.25f});
            if (OP == 10) g[c] = g[c] * v2f{1.0001f, 0.9999f};
            if (OP == 11) { f[c] = __builtin_fmaf(f[c], 1.0001f, 0.5f); x[c] = (x[c] >> 3) + 77u; }
            if (OP == 8) f[c] = (f[c] > 1.5f) ? f[c] * 0.5f : f[c] + 0.25f;      // cmp+cndmask+mul+add
        }
    }
    uint32_t r = 0;
    for (int c = 0; c < CHAINS; ++c) r += x[c] + __float_as_uint(f[c]) + __float_as_uint(g[c].x + g[c].y);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP> void run(const char *name, double ops_per_chain_iter)
{
    const int blocks = 256 * 8;      // 8 blocks of 4 waves per CU = 8 waves per SIMD
    uint32_t *d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD = 8 waves * N_ITERS * CHAINS * ops
    double winst = 8.0 * N_ITERS * CHAINS * ops_per_chain_iter;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc / winst);
    hipFree(d);
}

int main()
{
    run<0>("v_fma_f32", 1);
    run<1>("v_mad_u64_u32 (+1 xor)", 1);
    run<2>("v_mul_hi_u32", 1);
    run<3>("v_mul_lo_u32", 1);
    run<4>("v_sin_f32", 1);
    run<5>("v_rcp_f32", 1);
    run<6>("shift+xor (2 ops)", 2);
    run<7>("v_mad_u32_u24", 1);
    run<9>("v_pk_fma_f32", 1);
    run<10>("v_pk_mul_f32", 1);
    run<11>("v_fma_f32 + int lshr/add (3 ops)", 3);
    run<8>("cmp+cndmask+mul+add (4 ops)", 4);
    return 0;
}
