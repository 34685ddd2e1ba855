// ubench_issue.hip -- issue cost of single gfx950 VALU instruction FORMS (operand
// kinds and encodings included), written in inline asm so that the instruction
// under test is exactly the one named.  Every SIMD holds WAVES waves; each wave
// runs a long stream of one form over 8 independent destination registers.
// Prints cycles per wave-instruction per SIMD from s_memtime-free wall time
// (clock assumed 2.4 GHz: compare rows with each other).
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_issue ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint64_t u64;

#define N_ITERS 12000

// 8 instructions per REP8, destination registers %0..%7 rotate
#define REP8(fmt) \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                 : "s"(s0), "s"(s1), "v"(a), "v"(b), "s"(m64) : "vcc", "s20", "s21")

// operands: %N dst/chain, %8 %9 SGPR scalars, %10 %11 VGPR inputs, %12 SGPR pair
#define F_FMA_VVV(n)    "v_fma_f32 %" #n ", %10, %11, %" #n "\n"
#define F_FMA_SVV(n)    "v_fma_f32 %" #n ", %8, %11, %" #n "\n"
#define F_FMAC_S(n)     "v_fmac_f32 %" #n ", %8, %11\n"
#define F_FMAC_V(n)     "v_fmac_f32 %" #n ", %10, %11\n"
#define F_MUL_S(n)      "v_mul_f32 %" #n ", %8, %" #n "\n"
#define F_MUL_V(n)      "v_mul_f32 %" #n ", %10, %" #n "\n"
#define F_SUBREV_S(n)   "v_subrev_f32 %" #n ", %8, %" #n "\n"
#define F_CMP_VCC(n)    "v_cmp_lt_f32 vcc, %" #n ", %10\n"
#define F_CMP_SGPR(n)   "v_cmp_lt_f32 s[20:21], %" #n ", %10\n"
#define F_CMPU_SGPR(n)  "v_cmp_ne_u32 s[20:21], 3, %" #n "\n"
#define F_CLASS(n)      "v_cmp_class_f32 s[20:21], %" #n ", %11\n"
#define F_CND_VCC(n)    "v_cndmask_b32 %" #n ", %" #n ", %10, vcc\n"
#define F_CND_SGPR(n)   "v_cndmask_b32 %" #n ", %" #n ", %10, %12\n"
#define F_MIN_F32(n)    "v_min_f32 %" #n ", %10, %" #n "\n"
#define F_MAX_I32(n)    "v_max_i32 %" #n ", %10, %" #n "\n"
#define F_MIN3_U32(n)   "v_min3_u32 %" #n ", %10, %11, %" #n "\n"
#define F_MED3_F32(n)   "v_med3_f32 %" #n ", %10, %11, %" #n "\n"
#define F_AND_OR(n)     "v_and_or_b32 %" #n ", %" #n ", %10, %11\n"
#define F_BITOP3(n)     "v_bitop3_b32 %" #n ", %" #n ", %8, %11 bitop3:0x96\n"
#define F_LSHL_OR(n)    "v_lshl_or_b32 %" #n ", %" #n ", 6, %10\n"
#define F_ADD_U32(n)    "v_add_u32 %" #n ", %10, %" #n "\n"
#define F_LSHR(n)       "v_lshrrev_b32 %" #n ", 8, %" #n "\n"
#define F_MOV(n)        "v_mov_b32 %" #n ", %10\n"
#define F_CVT(n)        "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define F_SIN(n)        "v_sin_f32 %" #n ", %" #n "\n"
#define F_RCP(n)        "v_rcp_f32 %" #n ", %" #n "\n"
#define F_SQRT(n)       "v_sqrt_f32 %" #n ", %" #n "\n"
#define F_MULHI(n)      "v_mul_hi_u32 %" #n ", %8, %" #n "\n"
#define F_MULLO(n)      "v_mul_lo_u32 %" #n ", %8, %" #n "\n"
#define F_MUL24(n)      "v_mul_u32_u24 %" #n ", %10, %" #n "\n"
#define F_MULHI24(n)    "v_mul_hi_u32_u24 %" #n ", %10, %" #n "\n"
#define F_MAD24(n)      "v_mad_u32_u24 %" #n ", %10, %" #n ", %11\n"
#define F_ADDC(n)       "v_addc_co_u32 %" #n ", vcc, 0, %" #n ", vcc\n"
#define F_DPP(n)        "v_add_f32_dpp %" #n ", %10, %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
// mixes: one transcendental followed by k plain FMAs (does the pipe overlap them?)
#define F_SIN_3FMA(n)   "v_sin_f32 %" #n ", %" #n "\nv_fma_f32 %" #n ", %10, %11, %" #n "\nv_fma_f32 %" #n ", %10, %11, %" #n "\nv_fma_f32 %" #n ", %10, %11, %" #n "\n"
#define F_SIN_1FMA(n)   "v_sin_f32 %" #n ", %" #n "\nv_fma_f32 %" #n ", %10, %11, %" #n "\n"
#define F_CMP_CND(n)    "v_cmp_lt_f32 vcc, %" #n ", %10\nv_cndmask_b32 %" #n ", %" #n ", %11, vcc\n"
#define F_CMPS_CNDS(n)  "v_cmp_lt_f32 s[20:21], %" #n ", %10\nv_cndmask_b32 %" #n ", %" #n ", %11, s[20:21]\n"


#define F_ADD_F32(n)    "v_add_f32 %" #n ", %10, %" #n "\n"
#define F_SUB_F32(n)    "v_sub_f32 %" #n ", %10, %" #n "\n"
#define F_MAX_F32(n)    "v_max_f32 %" #n ", %10, %" #n "\n"
#define F_AND_B32(n)    "v_and_b32 %" #n ", %10, %" #n "\n"
#define F_XOR_B32(n)    "v_xor_b32 %" #n ", %10, %" #n "\n"
#define F_OR_B32(n)     "v_or_b32 %" #n ", %10, %" #n "\n"
#define F_LSHL(n)       "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define F_SUB_U32(n)    "v_sub_u32 %" #n ", %10, %" #n "\n"
#define F_MUL_INL(n)    "v_mul_f32 %" #n ", 2.0, %" #n "\n"
#define F_MUL_LIT(n)    "v_mul_f32 %" #n ", 0x3f9d70a4, %" #n "\n"
#define F_FMA_NEG(n)    "v_fma_f32 %" #n ", -%10, %11, %" #n "\n"
#define F_FMA_LIT(n)    "v_fmaak_f32 %" #n ", %10, %" #n ", 0x3f9d70a4\n"
#define F_PK_FMA(n)     "v_pk_fma_f32 %" #n ", %13, %14, %" #n "\n"
#define F_PK_MUL(n)     "v_pk_mul_f32 %" #n ", %13, %" #n "\n"
#define F_PK_ADD(n)     "v_pk_add_f32 %" #n ", %13, %" #n "\n"
#define F_ADD3(n)       "v_add3_u32 %" #n ", %10, %11, %" #n "\n"
#define F_LSHL_ADD(n)   "v_lshl_add_u32 %" #n ", %" #n ", 2, %10\n"
#define F_XAD(n)        "v_xad_u32 %" #n ", %" #n ", %10, %11\n"
#define F_MAD_F32_2S(n) "v_fma_f32 %" #n ", %8, %11, %" #n "\n"
#define F_CND_VV(n)     "v_cndmask_b32 %" #n ", %10, %11, vcc\n"
#define F_FMA_VVV_MIX(n) "v_fma_f32 %" #n ", %10, %11, %" #n "\nv_cmp_lt_f32 s[20:21], %" #n ", %10\n"
#define F_FMA_S_FMA_V(n) "v_fma_f32 %" #n ", %8, %11, %" #n "\nv_fma_f32 %" #n ", %10, %11, %" #n "\n"


#define F_FMAS_CMP(n)   "v_fma_f32 %" #n ", %8, %11, %" #n "\nv_cmp_lt_f32 s[20:21], %" #n ", %10\n"
#define F_FMAS_CND(n)   "v_fma_f32 %" #n ", %8, %11, %" #n "\nv_cndmask_b32 %" #n ", %" #n ", %10, %12\n"
#define F_CMP_BITOP(n)  "v_cmp_lt_f32 s[20:21], %" #n ", %10\nv_bitop3_b32 %" #n ", %" #n ", %10, %11 bitop3:0x96\n"
#define F_SIN_CMP(n)    "v_sin_f32 %" #n ", %" #n "\nv_cmp_lt_f32 s[20:21], %" #n ", %10\n"
#define F_SIN_FMAS(n)   "v_sin_f32 %" #n ", %" #n "\nv_fma_f32 %" #n ", %8, %11, %" #n "\n"
#define F_SIN_MULV(n)   "v_sin_f32 %" #n ", %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\n"
#define F_SIN_3MULV(n)  "v_sin_f32 %" #n ", %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\n"
#define F_CMP_MULV(n)   "v_cmp_lt_f32 s[20:21], %" #n ", %10\nv_mul_f32 %" #n ", %10, %" #n "\n"
#define F_CMP_3MULV(n)  "v_cmp_lt_f32 s[20:21], %" #n ", %10\nv_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\n"
#define F_MULS_MULV(n)  "v_mul_f32 %" #n ", %8, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\n"
#define F_MULS_3MULV(n) "v_mul_f32 %" #n ", %8, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %10, %" #n "\n"
#define F_MAD64ish(n)   "v_mul_hi_u32 %" #n ", %10, %" #n "\nv_cmp_lt_f32 s[20:21], %" #n ", %10\n"
#define F_FMA_SELF(n)   "v_fma_f32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define F_FMA_2R(n)     "v_fma_f32 %" #n ", %10, %10, %" #n "\n"
#define F_CMP_CND_MULV2(n) "v_cmp_lt_f32 s[20:21], %" #n ", %10\nv_mul_f32 %" #n ", %10, %" #n "\nv_cndmask_b32 %" #n ", %" #n ", %11, s[20:21]\nv_mul_f32 %" #n ", %10, %" #n "\n"

#define KERNEL(NAME, FMT)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, float s0, float s1, u64 m64, u64 *cyc) \
    {                                                                                           \
        float a = 1.0001f + threadIdx.x * 1e-6f, b = 0.5f;                                      \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, \
              r6 = r0 + 6, r7 = r0 + 7;                                                         \
        const u64 t0 = __builtin_readcyclecounter();                                            \
        for (int it = 0; it < N_ITERS; ++it) { REP8(FMT); REP8(FMT); REP8(FMT); REP8(FMT); }    \
        const u64 t1 = __builtin_readcyclecounter();                                            \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;            \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;          \
    }

#define REP8P(fmt) \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                 : "s"(s0), "s"(s1), "v"(a.x), "v"(b.x), "s"(m64), "v"(a), "v"(b) : "vcc", "s20", "s21")
#define KERNELP(NAME, FMT)                                                                      \
    __global__ __launch_bounds__(256) void NAME(float *out, float s0, float s1, u64 m64, u64 *cyc) \
    {                                                                                           \
        typedef float v2f __attribute__((ext_vector_type(2)));                                  \
        v2f a = {1.0001f + threadIdx.x * 1e-6f, 0.999f}, b = {0.5f, 0.25f};                      \
        v2f r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, \
              r6 = r0 + 6, r7 = r0 + 7;                                                         \
        const u64 t0 = __builtin_readcyclecounter();                                            \
        for (int it = 0; it < N_ITERS; ++it) { REP8P(FMT); REP8P(FMT); REP8P(FMT); REP8P(FMT); } \
        const u64 t1 = __builtin_readcyclecounter();                                            \
        v2f r = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                          \
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;                                        \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;          \
    }

// v_mad_u64_u32 needs a 64-bit destination: its own kernel
__global__ __launch_bounds__(256) void k_mad64(float *out, float s0f, float s1, u64 m64, u64 *cyc)
{
    uint32_t s0 = __float_as_uint(s0f);
    u64 r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    uint32_t a = threadIdx.x * 977u + 1;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_mad_u64_u32 %0, s[20:21], %4, %5, 0\n"
                         "v_mad_u64_u32 %1, s[20:21], %4, %5, 0\n"
                         "v_mad_u64_u32 %2, s[20:21], %4, %5, 0\n"
                         "v_mad_u64_u32 %3, s[20:21], %4, %5, 0\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "s"(s0) : "s20", "s21");
    }
    const u64 t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

#define ALL(X)                                                                                         \
    X(k_fma_vvv, F_FMA_VVV, 1) X(k_fma_svv, F_FMA_SVV, 1) X(k_fmac_s, F_FMAC_S, 1) X(k_fmac_v, F_FMAC_V, 1) \
    X(k_mul_s, F_MUL_S, 1) X(k_mul_v, F_MUL_V, 1) X(k_subrev_s, F_SUBREV_S, 1)                          \
    X(k_cmp_vcc, F_CMP_VCC, 1) X(k_cmp_sgpr, F_CMP_SGPR, 1) X(k_cmpu_sgpr, F_CMPU_SGPR, 1) X(k_class, F_CLASS, 1) \
    X(k_cnd_vcc, F_CND_VCC, 1) X(k_cnd_sgpr, F_CND_SGPR, 1)                                             \
    X(k_min_f32, F_MIN_F32, 1) X(k_max_i32, F_MAX_I32, 1) X(k_min3_u32, F_MIN3_U32, 1) X(k_med3_f32, F_MED3_F32, 1) \
    X(k_and_or, F_AND_OR, 1) X(k_bitop3, F_BITOP3, 1) X(k_lshl_or, F_LSHL_OR, 1) X(k_add_u32, F_ADD_U32, 1) \
    X(k_lshr, F_LSHR, 1) X(k_mov, F_MOV, 1) X(k_cvt, F_CVT, 1) X(k_sin, F_SIN, 1) X(k_rcp, F_RCP, 1) X(k_sqrt, F_SQRT, 1) \
    X(k_mulhi, F_MULHI, 1) X(k_mullo, F_MULLO, 1) X(k_mul24, F_MUL24, 1) X(k_mulhi24, F_MULHI24, 1) X(k_mad24, F_MAD24, 1) \
    X(k_addc, F_ADDC, 1) X(k_dpp, F_DPP, 1)                                                             \
    X(k_add_f32, F_ADD_F32, 1) X(k_sub_f32, F_SUB_F32, 1) X(k_max_f32, F_MAX_F32, 1) X(k_and_b32, F_AND_B32, 1) \
    X(k_xor_b32, F_XOR_B32, 1) X(k_or_b32, F_OR_B32, 1) X(k_lshl, F_LSHL, 1) X(k_sub_u32, F_SUB_U32, 1) \
    X(k_mul_inl, F_MUL_INL, 1) X(k_mul_lit, F_MUL_LIT, 1) X(k_fma_neg, F_FMA_NEG, 1) X(k_fma_lit, F_FMA_LIT, 1) \
    X(k_add3, F_ADD3, 1) X(k_lshl_add, F_LSHL_ADD, 1) X(k_cnd_vv, F_CND_VV, 1) \
    X(k_fma_cmp_mix, F_FMA_VVV_MIX, 2) X(k_fma_s_fma_v, F_FMA_S_FMA_V, 2) \
    X(k_fmas_cmp, F_FMAS_CMP, 2) X(k_fmas_cnd, F_FMAS_CND, 2) X(k_cmp_bitop, F_CMP_BITOP, 2) X(k_sin_cmp, F_SIN_CMP, 2) \
    X(k_sin_fmas, F_SIN_FMAS, 2) X(k_sin_mulv, F_SIN_MULV, 2) X(k_sin_3mulv, F_SIN_3MULV, 4) X(k_cmp_mulv, F_CMP_MULV, 2) \
    X(k_cmp_3mulv, F_CMP_3MULV, 4) X(k_muls_mulv, F_MULS_MULV, 2) X(k_muls_3mulv, F_MULS_3MULV, 4) X(k_mulhi_cmp, F_MAD64ish, 2) \
    X(k_fma_self, F_FMA_SELF, 1) X(k_fma_2r, F_FMA_2R, 1) X(k_cmp_cnd_2mulv, F_CMP_CND_MULV2, 4) \
    X(k_sin_3fma, F_SIN_3FMA, 4) X(k_sin_1fma, F_SIN_1FMA, 2) X(k_cmp_cnd, F_CMP_CND, 2) X(k_cmps_cnds, F_CMPS_CNDS, 2)

#define DEF(NAME, FMT, OPS) KERNEL(NAME, FMT)
ALL(DEF)
#define ALLP(X) X(k_pk_fma, F_PK_FMA, 1) X(k_pk_mul, F_PK_MUL, 1) X(k_pk_add, F_PK_ADD, 1)
#define DEFP(NAME, FMT, OPS) KERNELP(NAME, FMT)
ALLP(DEFP)


// packed-fp32 ops next to single ops: four pair accumulators (%0-%3) and four single ones (%4-%7)
#define P_FMA(n)      "v_pk_fma_f32 %" #n ", %13, %14, %" #n "\n"
#define P_FMA_BC(n)   "v_pk_fma_f32 %" #n ", %13, %14, %" #n " op_sel_hi:[0,1,1]\n"
#define P_FMA_BC1(n)  "v_pk_fma_f32 %" #n ", %13, %14, %" #n " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
#define P_MUL_BC(n)   "v_pk_mul_f32 %" #n ", %13, %" #n " op_sel_hi:[0,1]\n"
#define P_NONE(n)     ""
#define S_MULV(n)     "v_mul_f32 %" #n ", %10, %" #n "\n"
#define S_2MULV(n)    "v_mul_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %11, %" #n "\n"
#define S_FMAV(n)     "v_fma_f32 %" #n ", %10, %11, %" #n "\n"
#define S_FMAS(n)     "v_fma_f32 %" #n ", %8, %11, %" #n "\n"
#define S_CMP(n)      "v_cmp_lt_f32 s[20:21], %" #n ", %10\n"
#define S_CND(n)      "v_cndmask_b32 %" #n ", %" #n ", %10, %12\n"
#define S_RCP(n)      "v_rcp_f32 %" #n ", %" #n "\n"
#define S_NONE(n)     ""
#define REP4M(pf, sf) \
    asm volatile(pf(0) sf(4) pf(1) sf(5) pf(2) sf(6) pf(3) sf(7) \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) \
                 : "s"(s0), "s"(s1), "v"(a.x), "v"(b.x), "s"(m64), "v"(a), "v"(b) : "vcc", "s20", "s21")
#define KERNELM(NAME, PF, SF)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float s0, float s1, u64 m64, u64 *cyc) \
    {                                                                                           \
        typedef float v2f __attribute__((ext_vector_type(2)));                                  \
        v2f a = {1.0001f + threadIdx.x * 1e-6f, 0.999f}, b = {0.5f, 0.25f};                      \
        v2f p0 = {(float)threadIdx.x, 1.f}, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3;              \
        float q0 = threadIdx.x + 4, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;                      \
        const u64 t0 = __builtin_readcyclecounter();                                            \
        for (int it = 0; it < N_ITERS; ++it) {                                                  \
            REP4M(PF, SF); REP4M(PF, SF); REP4M(PF, SF); REP4M(PF, SF);                          \
            REP4M(PF, SF); REP4M(PF, SF); REP4M(PF, SF); REP4M(PF, SF); }                        \
        const u64 t1 = __builtin_readcyclecounter();                                            \
        v2f r = p0 + p1 + p2 + p3;                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + q0 + q1 + q2 + q3;                    \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;          \
    }
// OPS = instructions per (pf, sf) pair
#define ALLM(X) X(m_pk_only, P_FMA, S_NONE, 1) X(m_pk_bc, P_FMA_BC, S_NONE, 1) X(m_pk_bc1, P_FMA_BC1, S_NONE, 1) \
    X(m_pkmul_bc, P_MUL_BC, S_NONE, 1) X(m_pk_mulv, P_FMA, S_MULV, 2) X(m_pk_2mulv, P_FMA, S_2MULV, 3) \
    X(m_pk_fmav, P_FMA, S_FMAV, 2) X(m_pk_fmas, P_FMA, S_FMAS, 2) X(m_pk_cmp, P_FMA, S_CMP, 2) X(m_pk_cnd, P_FMA, S_CND, 2) \
    X(m_pk_rcp, P_FMA, S_RCP, 2) X(m_pkbc_mulv, P_FMA_BC, S_MULV, 2) X(m_fmav_only, P_NONE, S_FMAV, 1)
#define DEFM(NAME, PF, SF, OPS) KERNELM(NAME, PF, SF)
ALLM(DEFM)

static float *d_out;
static u64 *d_cyc;
template <typename K> void run(K kern, const char *name, double ops, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;        // 4-wave blocks: waves_per_simd blocks per CU
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.5f, 0.25f, (u64)0x5555555555555555ull, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.5f, 0.25f, (u64)0x5555555555555555ull, d_cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    static u64 h[256 * 16 * 4];
    hipMemcpy(h, d_cyc, sizeof(u64) * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks * 4; ++i) mean += (double)h[i]; mean /= blocks * 4;
    const double winst = (double)waves_per_simd * N_ITERS * 32.0 * ops;
    printf("%-14s waves/SIMD=%d %8.3f ms  %6.2f cyc/winst/SIMD by wall@2.4GHz  %6.2f by s_memtime (x100MHz ticks: %.0f per wave)\n",
           name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / winst, mean * 24.0 / (N_ITERS * 32.0 * ops) / waves_per_simd, mean);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 8;
    hipMalloc(&d_out, 256 * 16 * 256 * 4);
    hipMalloc(&d_cyc, 256 * 16 * 4 * 8);
#define RUN(NAME, FMT, OPS) run(NAME, #NAME, OPS, w);
    ALL(RUN)
    ALLP(RUN)
#define RUNM(NAME, PF, SF, OPS) run(NAME, #NAME, OPS, w);
    ALLM(RUNM)
    run(k_mad64, "k_mad64", 1, w);
    hipFree(d_out);
    return 0;
}
