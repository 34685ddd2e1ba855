// ubench_slab.hip -- issue cost of the instruction forms of the spatial index's box test (trace_bvh_kernel's walk step)
// and of the two whole tests: six binary16 bounds converted, subtracted, multiplied, min/max-ed (round 1-3) against
// centre / half-extent bounds fed to v_fma_mix_f32 (f16 sources converted inside the FMA).  Same harness as
// ubench_issue.hip: every SIMD holds WAVES waves, cycles per wave-instruction per SIMD from wall time at 2.4 GHz.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_slab ubench_slab.hip ; run: ./ubench_slab [waves per SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint64_t u64;
#define N_ITERS 12000

#define REP8(fmt) \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                 : "v"(a), "v"(b), "v"(h) : "vcc", "s20", "s21")
// %8 %9 f32 VGPRs, %10 a VGPR holding two halves
#define F_FMA(n)        "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define F_MIX_LO(n)     "v_fma_mix_f32 %" #n ", %10, %8, %" #n " op_sel_hi:[1,0,0]\n"
#define F_MIX_HI(n)     "v_fma_mix_f32 %" #n ", %10, %8, %" #n " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define F_MIX_ABS(n)    "v_fma_mix_f32 %" #n ", -%10, |%8|, %" #n " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define F_MIX_F32(n)    "v_fma_mix_f32 %" #n ", %9, %8, %" #n "\n"
#define F_CVT_LO(n)     "v_cvt_f32_f16 %" #n ", %10\n"
#define F_CVT_HI(n)     "v_cvt_f32_f16_sdwa %" #n ", %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define F_MAX3(n)       "v_max3_f32 %" #n ", %8, %9, %" #n "\n"
#define F_MIN(n)        "v_min_f32 %" #n ", %8, %" #n "\n"
#define F_MIX_FMA(n)    "v_fma_mix_f32 %" #n ", %10, %8, %" #n " op_sel_hi:[1,0,0]\nv_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define F_MIX_MAX3(n)   "v_fma_mix_f32 %" #n ", %10, %8, %" #n " op_sel_hi:[1,0,0]\nv_max3_f32 %" #n ", %8, %9, %" #n "\n"
#define F_LSHL(n)       "v_lshlrev_b32 %" #n ", 4, %" #n "\n"
#define F_BFE(n)        "v_bfe_u32 %" #n ", %8, %" #n ", 1\n"
#define F_XOR(n)        "v_xor_b32 %" #n ", 1, %" #n "\n"
#define F_NOT(n)        "v_not_b32 %" #n ", %" #n "\n"

#define KERNEL(NAME, FMT)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, u64 *cyc)                           \
    {                                                                                           \
        float a = 1.0001f + threadIdx.x * 1e-6f, b = 0.5f;                                      \
        uint32_t h = 0x3c003800u + threadIdx.x;                                                 \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, \
              r6 = r0 + 6, r7 = r0 + 7;                                                         \
        for (int it = 0; it < N_ITERS; ++it) { REP8(FMT); REP8(FMT); REP8(FMT); REP8(FMT); }    \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;            \
    }
#define ALL(X) X(k_fma, F_FMA, 1) X(k_mix_lo, F_MIX_LO, 1) X(k_mix_hi, F_MIX_HI, 1) X(k_mix_abs, F_MIX_ABS, 1) X(k_mix_f32, F_MIX_F32, 1) \
    X(k_cvt_lo, F_CVT_LO, 1) X(k_cvt_hi, F_CVT_HI, 1) X(k_max3, F_MAX3, 1) X(k_min, F_MIN, 1) X(k_mix_fma, F_MIX_FMA, 2) \
    X(k_mix_max3, F_MIX_MAX3, 2) X(k_lshl, F_LSHL, 1) X(k_bfe, F_BFE, 1) X(k_xor, F_XOR, 1) X(k_not, F_NOT, 1)
#define DEF(NAME, FMT, OPS) KERNEL(NAME, FMT)
ALL(DEF)

// the two whole box tests, 4 per loop trip, independent of each other (o = origin, i = 1/d, oi = o/d, bounds in n0..n2)
#define SLAB_OLD \
    "v_cvt_f32_f16 v40, %3\n v_cvt_f32_f16 v41, %4\n v_cvt_f32_f16 v42, %5\n" \
    "v_cvt_f32_f16_sdwa v43, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n" \
    "v_cvt_f32_f16_sdwa v44, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n" \
    "v_cvt_f32_f16_sdwa v45, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n" \
    "v_sub_f32 v40, v40, %6\n v_sub_f32 v41, v41, %7\n v_sub_f32 v42, v42, %8\n v_sub_f32 v43, v43, %6\n v_sub_f32 v44, v44, %7\n v_sub_f32 v45, v45, %8\n" \
    "v_mul_f32 v40, %9, v40\n v_mul_f32 v41, %10, v41\n v_mul_f32 v42, %11, v42\n v_mul_f32 v43, %9, v43\n v_mul_f32 v44, %10, v44\n v_mul_f32 v45, %11, v45\n" \
    "v_min_f32 v46, v40, v43\n v_max_f32 v40, v40, v43\n v_min_f32 v47, v41, v44\n v_max_f32 v41, v41, v44\n v_min_f32 v48, v42, v45\n v_max_f32 v42, v42, v45\n" \
    "v_max_f32 v46, 0, v46\n v_min_f32 v40, v40, %12\n v_max3_f32 v46, v47, v48, v46\n v_min3_f32 v40, v41, v42, v40\n" \
    "v_cmp_le_f32 s[20:21], v46, v40\n s_or_b64 %0, %0, s[20:21]\n"
#define SLAB_NEW \
    "v_fma_mix_f32 v40, %3, %9, -%6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v41, %3, %10, -%7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v42, %4, %11, -%8 op_sel_hi:[1,0,0]\n" \
    "v_fma_mix_f32 v43, -%4, |%9|, v40 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v44, -%5, |%10|, v41 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v45, -%5, |%11|, v42 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
    "v_fma_mix_f32 v40, %4, |%9|, v40 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v41, %5, |%10|, v41 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v42, %5, |%11|, v42 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
    "v_max3_f32 v43, v43, v44, 0\n v_min3_f32 v40, v40, v41, %12\n v_max_f32 v43, v43, v45\n v_min_f32 v40, v40, v42\n" \
    "v_cmp_le_f32 s[20:21], v43, v40\n s_or_b64 %0, %0, s[20:21]\n"
#define SLAB_KERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out, u64 *cyc)                           \
    {                                                                                           \
        const float t = threadIdx.x;                                                            \
        uint32_t n0 = 0x3c003800u + threadIdx.x, n1 = 0x40003a00u + threadIdx.x, n2 = 0x42004100u;  \
        float ox = t * .01f, oy = 1.f - t * .02f, oz = .5f, ix = 1.f / (t + .5f), iy = -2.f, iz = 3.f, bt = 100.f; \
        u64 m = 0;                                                                              \
        for (int it = 0; it < N_ITERS; ++it) {                                                  \
            asm volatile(BODY BODY BODY BODY : "+s"(m), "+v"(n0), "+v"(n1) : "v"(n0), "v"(n1), "v"(n2), "v"(ox), "v"(oy), "v"(oz), \
                         "v"(ix), "v"(iy), "v"(iz), "v"(bt) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "s20", "s21"); \
        }                                                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = (float)m;                                         \
    }
SLAB_KERNEL(k_slab_old, SLAB_OLD)
SLAB_KERNEL(k_slab_new, SLAB_NEW)

static float *d_out;
static u64 *d_cyc;
template <typename K> void run(K kern, const char *name, double winst_per_trip, int waves_per_simd, const char *unit)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)waves_per_simd * N_ITERS * winst_per_trip;
    printf("%-12s waves/SIMD=%d %8.3f ms  %7.2f cycles per %s per SIMD (wall, 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / n, unit);
    fflush(stdout);
}
int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 8;
    hipMalloc(&d_out, 256 * 16 * 256 * 4);
    hipMalloc(&d_cyc, 256 * 16 * 4 * 8);
#define RUN(NAME, FMT, OPS) run(NAME, #NAME, 32.0 * OPS, w, "wave-instruction");
    ALL(RUN)
    run(k_slab_old, "slab_old", 4, w, "box test (30 VALU)");
    run(k_slab_new, "slab_new", 4, w, "box test (14 VALU)");
    return 0;
}
