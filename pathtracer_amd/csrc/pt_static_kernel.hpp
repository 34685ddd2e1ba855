// pt_static_kernel.hpp -- trace_static_kernel, the default kernel for scenes
// of <= 8 objects (object list in the kernel arguments).  A header of its own
// because it is compiled twice: ahead of time by pt_kernels.hip, for any scene,
// and at run time by pt_jit.hip (hiprtc) with PT_JIT defined and ONE scene's
// geometry as compile-time constants (pt_scene_specialise, include/pt.h).
#pragma once

#include "pt_internal.hpp"

#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

namespace pt {

#ifdef __HIPCC_RTC__
struct true_type  { static constexpr bool value = true; };
struct false_type { static constexpr bool value = false; };
#else
using true_type = std::true_type;
using false_type = std::false_type;
#endif

// The segment counter of one pt_reset epoch is SEG_LANES counters in separate
// 128-byte lines, picked by workgroup: atomics on ONE address retire at ~80 M/s
// in L2 (8,100 workgroups = 0.1 ms, measured on an otherwise empty launch).
__device__ __forceinline__ unsigned long long *seg_lane(unsigned long long *base)
{
    return base + (blockIdx.x & (SEG_LANES - 1)) * SEG_LANE_STRIDE;
}

// The drain gate (TraceArgs::gate): the last workgroup in dispatch order announces that the launch has started it.
__device__ __forceinline__ void signal_gate(const TraceArgs &a)
{
    if (a.gate && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0)
        __hip_atomic_store(a.gate, a.gate_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the per-pixel fp64 primary segment, through the spatial index when the scene has one
template <typename STK = int>
__device__ __forceinline__ PrimaryHit primary_hit(const TraceArgs &a, int pi, int pj, STK *stack = nullptr, int stride = 0)
{
    if (a.n_nodes > 0 && stack)
        return primary_hit_f64_bvh(a.prim, a.bvh_nodes, a.bvh_prims, a.n_unbounded, a.height, a.width,
                                   a.camera_depth, pi, pj, stack, stride);
    return primary_hit_f64(a.prim, a.n_obj, a.height, a.width, a.camera_depth, pi, pj);
}

// ---------------------------------------------------------------------------
// trace kernel for small scenes (object list in the kernel arguments), the
// default for <= 8 objects.  Same two phases and the same work sharing as
// trace_steal_kernel; what differs is the instruction diet, shaped by the issue
// rules measured with tools/ubench_issue.hip on gfx950:
//   * a VALU instruction with an SGPR operand issues once per 4 cycles, an
//     all-VGPR add/mul/fma/logic op once per 2, and compares, selects, min/max,
//     converts, integer multiplies and three-operand integer ops once per 4 on a
//     port of their own; the three limits overlap, so the loop is laid out to
//     load them evenly: n_hat stays in SGPRs (den = v.n_hat), n_raw and A move to
//     VGPRs, and the Philox round keys and multiplier are VGPRs too;
//   * object identity is carried as the byte offset 16*(k+1) of the object's
//     material record (0 = the "miss" record), so the closest-hit search selects
//     LDS addresses directly from inline constants: no index arithmetic;
//   * the material tag m0.w orders the cases (light e > 0, miss 0, diffuse -1,
//     reflective -2): "path ends here" is one compare and the emitted radiance
//     max(e, 0) one integer max, with no select on hit/miss;
//   * sample index and bounce live in one counter (gs << 6 | bounce) that is
//     also the Philox counter word and the loop's exit test;
//   * (4-object list) identity and control decisions are lane masks kept on the
//     scalar unit: _lastStruck is one mask per object, updated from the accept
//     masks of the search with s_and/s_or/s_andn2; where such a mask exists,
//     values are moved in under exec (v_mov_b32, plain issue class) instead of
//     being selected (v_cndmask_b32, second-port class).
// ---------------------------------------------------------------------------
// MASKS: the _lastStruck exclusion arrives as one lane mask per object
// (excl[k] = lanes whose last-struck object is k) and every accept decision is
// returned as a lane mask (okm[k]); the caller derives the next masks from them
// on the scalar unit, so the search spends no vector compare on identity.
// best <- (OFF, key) in the lanes of `mask`: two plain-class moves under exec
// instead of two second-port selects
#ifdef PT_JIT
// The scene this translation unit is compiled for (pt_jit.hip writes the three macros):
// 12 floats per object, (n_raw, A) (n_hat, r^2 | +inf | -r) (centre, 0), padded to PT_JIT_NS like launch_trace_ns pads.
__device__ constexpr float JIT_GEOM[PT_JIT_NS][12] = PT_JIT_GEOM;

// fma chain `a + x1*m1 + x2*m2 + x3*m3` in this order, terms with a zero coefficient c left out
__device__ __forceinline__ float jit_chain(float a, float c1, float x1, float m1, float c2, float x2, float m2,
                                           float c3, float x3, float m3)
{
    float r = a;
    if (c1 != 0.0f) r = fmaf(x1, m1, r);
    if (c2 != 0.0f) r = fmaf(x2, m2, r);
    if (c3 != 0.0f) r = fmaf(x3, m3, r);
    return r;
}
// `x1*c1 + x2*c2 + x3*c3` in this order (first term a product, then fmas), zero coefficients left out
__device__ __forceinline__ float jit_dot(float c1, float x1, float c2, float x2, float c3, float x3)
{
    float r = 0.0f;
    bool any = false;
    if (c1 != 0.0f) { r = x1 * c1; any = true; }
    if (c2 != 0.0f) { r = any ? fmaf(x2, c2, r) : x2 * c2; any = true; }
    if (c3 != 0.0f) { r = any ? fmaf(x3, c3, r) : x3 * c3; any = true; }
    return r;
}
#endif

// Lane mask (SGPR pair) -> per-lane condition.  The builtin costs nothing (the condition stays the SGPR pair);
// hiprtc may be an older compiler than the one that built the library (a process that loaded PyTorch's bundled
// ROCm compiles with that one), hence the spelled-out forms.
__device__ __forceinline__ bool lane_of(unsigned long long m)
{
#if __has_builtin(__builtin_amdgcn_inverse_ballot_w64)
    return __builtin_amdgcn_inverse_ballot_w64(m);
#else
    int x;
    asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(x) : "s"(m));
    return x != 0;
#endif
}
// (m ? a : b) per lane, a and b in VGPRs
__device__ __forceinline__ uint32_t pick(unsigned long long m, uint32_t a, uint32_t b)
{
#if __has_builtin(__builtin_amdgcn_inverse_ballot_w64)
    return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b;
#else
    uint32_t x;
    asm("v_cndmask_b32 %0, %2, %1, %3" : "=v"(x) : "v"(a), "v"(b), "s"(m));
    return x;
#endif
}
// (m ? 16 : 0) per lane
__device__ __forceinline__ int pick_16_0(unsigned long long m)
{
#if __has_builtin(__builtin_amdgcn_inverse_ballot_w64)
    return __builtin_amdgcn_inverse_ballot_w64(m) ? 16 : 0;
#else
    int x;
    asm("v_cndmask_b32 %0, 0, 16, %1" : "=v"(x) : "s"(m));
    return x;
#endif
}

template <int OFF>
__device__ __forceinline__ void take_under(unsigned long long mask, int &bestp, uint32_t &best_key, uint32_t key)
{
    unsigned long long saved;
    asm("s_and_saveexec_b64 %[sv], %[m]\n\t"
        "v_mov_b32 %[bp], %[off]\n\t"
        "v_mov_b32 %[bk], %[key]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [bp] "+v"(bestp), [bk] "+v"(best_key), [sv] "=&s"(saved)
        : [m] "s"(mask), [off] "n"(OFF), [key] "v"(key));
}

// g <- ds_add_rtn_u32(LDS[addr + OFF], val) in the lanes of `mask` (the others keep g); waits for the result
template <int OFF>
__device__ __forceinline__ uint32_t lds_add_rtn_under(unsigned long long mask, uint32_t addr, uint32_t val, uint32_t g)
{
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "ds_add_rtn_u32 %[g], %[a], %[v] offset:%[off]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [g] "+v"(g), [sv] "=&s"(saved)
                 : [m] "s"(mask), [a] "v"(addr), [v] "v"(val), [off] "n"(OFF) : "memory");
    return g;
}
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t addr)
{
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
}
__device__ __forceinline__ void lds_write_u32(uint32_t addr, uint32_t x)
{
    *reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr) = x;
}

template <int NS, bool SPH, bool MASKS>
__device__ __forceinline__ void closest_static(const StaticGeom<NS> &sg, const float4 *__restrict__ rect, f3 o, f3 v, int lastp,
                                               const unsigned long long *excl, unsigned long long *okm,
                                               int &bestp, float &best_t)
{
    // The bits of t order the candidates as unsigned integers: finite positive t
    // below bits(+Inf); t <= -0, NaN and +Inf at or above it.  t = +0 alone would
    // pass (0 < t is strict, src/Object.hs:159), so the product is formed as
    // fma(num, 1/den, -denorm_min): +0 becomes negative, every other value is
    // unchanged by the rounding.  A strict '<' keeps the earlier object on ties.
    uint32_t best_key = 0x7f800000u;
    bestp = 0;
#ifdef PT_JIT
    float rdens[NS];
#endif
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#ifdef PT_JIT
        const float4 c0 = make_float4(JIT_GEOM[k][0], JIT_GEOM[k][1], JIT_GEOM[k][2], JIT_GEOM[k][3]);
        // (-n_raw, A): in VGPRs, as in the general kernel, for up to 8 objects; beyond that 4 registers per object are
        // more than the kernel has, and the numbers ride in the instructions as literals (v_fmamk/v_fmaak_f32)
        const float4 g0 = NS <= 8 ? sg.g[k][0] : make_float4(-c0.x, -c0.y, -c0.z, c0.w);
        const float4 g1 = make_float4(JIT_GEOM[k][4], JIT_GEOM[k][5], JIT_GEOM[k][6], JIT_GEOM[k][7]);
        const float4 g2 = make_float4(JIT_GEOM[k][8], JIT_GEOM[k][9], JIT_GEOM[k][10], JIT_GEOM[k][11]);
#else
        const float4 g0 = sg.g[k][0], g1 = sg.g[k][1], g2 = sg.g[k][2];
#endif
        // A - o.n_raw with the dot folded into A (g0.xyz holds -n_raw here, see the kernel's set-up): the
        // same three roundings as distance_from's fma(-o, n, .) chain
#ifdef PT_JIT
        // This scene's numbers are compile-time constants here (after unrolling): a term whose coefficient is
        // zero is not computed -- adding its +-0 would not change the sum --, a coefficient of +-1 is a sign,
        // and the plane/disk/sphere choice below is decided by the compiler.  Same roundings as the general form.
        const float num = jit_chain(g0.w, c0.x, o.x, g0.x, c0.y, o.y, g0.y, c0.z, o.z, g0.z);
#else
        const float num = fmaf(o.z, g0.z, fmaf(o.y, g0.y, fmaf(o.x, g0.x, g0.w)));
        const float den = fmaf(v.z, g1.z, fmaf(v.y, g1.y, v.x * g1.x));
#endif
#ifdef PT_JIT
        // The reciprocal is taken of v.(+-n_hat), the sign chosen so that the first non-zero coefficient is positive,
        // and the sign goes on the numerator (a free source modifier of the fma; v_rcp_f32 is odd in its argument
        // and every rounding is symmetric in sign, so t has the same bits).  An earlier object with the same or
        // the opposite unit normal (facing walls; the default scene's back wall and its light) then has the very
        // same denominator, and its reciprocal serves: one quarter-rate instruction less.
        // (Written as -(1/den), hipcc folds the negation into the reciprocal's operand and issues a second v_rcp_f32.)
        const bool flipped = g1.x < 0.0f || (g1.x == 0.0f && (g1.y < 0.0f || (g1.y == 0.0f && g1.z < 0.0f)));
        const float ux = flipped ? -g1.x : g1.x, uy = flipped ? -g1.y : g1.y, uz = flipped ? -g1.z : g1.z;
        int shared = -1;
#pragma unroll
        for (int j = 0; j < k; ++j) {
            const bool same = JIT_GEOM[j][4] == g1.x && JIT_GEOM[j][5] == g1.y && JIT_GEOM[j][6] == g1.z;
            const bool opposite = JIT_GEOM[j][4] == -g1.x && JIT_GEOM[j][5] == -g1.y && JIT_GEOM[j][6] == -g1.z;
            const bool real = g1.x != 0.0f || g1.y != 0.0f || g1.z != 0.0f;       // (not a padding record)
            if (shared < 0 && real && (same || opposite)) shared = j;
        }
        const float rden = shared < 0 ? __builtin_amdgcn_rcpf(jit_dot(ux, v.x, uy, v.y, uz, v.z)) : rdens[shared];
        rdens[k] = rden;
        float t = flipped ? fmaf(-num, rden, -__builtin_bit_cast(float, 1u)) : fmaf(num, rden, -__builtin_bit_cast(float, 1u));
#else
        float t = fmaf(num, __builtin_amdgcn_rcpf(den), -__builtin_bit_cast(float, 1u));
#endif
        if constexpr (SPH) {
            if (g1.w < 0.0f)                                       // wave-uniform: Sphere (extension)
                t = fmaf(sphere_t(mk3(g2.x, g2.y, g2.z), -g1.w, o, v), 1.0f, -__builtin_bit_cast(float, 1u));
        }
        const uint32_t key = __float_as_uint(t);
        // 0 < t < best, not _lastStruck (src/Lib.hs:86-90)
        bool ok = false;
        unsigned long long m = 0;
        if constexpr (MASKS) m = __builtin_amdgcn_ballot_w64(key < best_key) & ~excl[k];
        else ok = (key < best_key) & (16 * (k + 1) != lastp);
        if (g1.w != __builtin_inff() && !(SPH && g1.w < 0.0f)) {   // wave-uniform: Disk (src/Object.hs:165-171)
            const float dx = fmaf(t, v.x, o.x) - g2.x;
            const float dy = fmaf(t, v.y, o.y) - g2.y;
            const float dz = fmaf(t, v.z, o.z) - g2.z;
            const bool in = inside_form<SPH>(rect, k, g1.w, dx, dy, dz);
            if constexpr (MASKS) m &= __builtin_amdgcn_ballot_w64(in);
            else ok = ok & in;
        }
        if constexpr (MASKS) {
            okm[k] = m;
            switch (k) {                                           // k is a constant after unrolling
            case 0:                                                // selects from constants: they also initialise
                bestp = pick_16_0(m); best_key = pick(m, key, 0x7f800000u);
                break;
            case 1: take_under<32>(m, bestp, best_key, key); break;
            case 2: take_under<48>(m, bestp, best_key, key); break;
            case 3: take_under<64>(m, bestp, best_key, key); break;
            case 4: take_under<80>(m, bestp, best_key, key); break;
            case 5: take_under<96>(m, bestp, best_key, key); break;
            case 6: take_under<112>(m, bestp, best_key, key); break;
            default: take_under<128>(m, bestp, best_key, key); break;
            }
        } else {
            bestp = ok ? 16 * (k + 1) : bestp;
            best_key = ok ? key : best_key;
        }
    }
    best_t = __uint_as_float(best_key);
}

// One wave per workgroup: the finest dispatch grain, no barriers that wait on
// other waves (measured 2 % faster than 256 at 1024 spp, 1 % at 128 spp).
constexpr int STATIC_BLOCK = 64;

template <int NS, bool SPEC>
__global__ __launch_bounds__(STATIC_BLOCK) void trace_static_kernel(TraceArgs a, StaticGeom<NS> sg)
{
    static_assert(NS > 0, "object list by value");
    constexpr int BLOCK = STATIC_BLOCK;
    constexpr int R = NS + 1;                       // material records per field: miss + NS objects
    // lane masks of "the primary object is k" and "the last-struck object is k" (closest_static, MASKS)
#ifdef PT_JIT
    // (8 objects too: their geometry is in the code, not in 64 SGPRs, so the masks fit; 16 objects' masks -- three
    // SGPR pairs each -- do not: the 9-16-object scene kernels keep identity in a VGPR, as the general 8-object one)
    constexpr bool MASKS = NS <= 8;
#else
    constexpr bool MASKS = NS <= 4;
#endif
    // Per pixel of the wave, one record of three vectors: (P0, record offset), (colour0, e0), (d0, next) -- next, MASKS:
    // the sharing phase's counter of the pixel, the next global sample index not started -- and, MASKS, the radiance
    // handed over by helpers in s_hand at a quarter of the record's offset, so that ONE per-lane LDS address (raddr)
    // names everything of the pixel a lane works on.  48-byte records: the 16 lanes a ds_read_b128 serves per pass
    // start at 12 banks' distance, all different bank quads; with the sums inside a 64-byte record they started 16
    // banks apart, four lanes per quad -- SQ_LDS_BANK_CONFLICT was 30 % of SQ_LDS_IDX_ACTIVE, 1.3 % before and after.
    constexpr int RV = 3;
    __shared__ float4   s_mat[4 * R];               // field f of object k at [f*R + k + 1]; [f*R] = miss
    __shared__ uint32_t s_next[MASKS ? 1 : BLOCK];  // (!MASKS) per pixel: next global sample index not started
    __shared__ float4   s_pix[BLOCK * RV];
    __shared__ float    s_acc[MASKS ? 1 : BLOCK * 3];   // (!MASKS) per pixel: radiance gathered in phase 2
    __shared__ float    s_hand[MASKS ? BLOCK * 3 : 1];  // (MASKS) the same, addressed as (raddr - rbase) / 4
    const int tid = threadIdx.x;
    signal_gate(a);
    for (int q = tid; q < 4 * R; q += BLOCK) {
        const int f = q / R, r = q - f * R;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r > 0 && r - 1 < a.n_obj) {
            v = a.mat[(r - 1) * 4 + f];
            if (f == 0) v.w = v.w > 0.f ? v.w : (v.w < 0.f ? -2.f : -1.f);   // light e | diffuse -1 | reflective -2
        }
        s_mat[q] = v;                               // miss record: colour 0, tag 0
    }
    auto matp = [&](int p, int f) -> float4 {       // p = 16*(k+1): byte offset of the record within a field
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_mat) + p + f * R * 16);
    };
    {   // -n_raw and A to VGPRs (see the header comment; negated so that A - o.n_raw is one fma chain)
        auto fv = [](float &x) { x = __uint_as_float(to_vgpr(__float_as_uint(x))); };
        if constexpr (NS <= MAX_STATIC_OBJECTS) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                sg.g[k][0].x = -sg.g[k][0].x; sg.g[k][0].y = -sg.g[k][0].y; sg.g[k][0].z = -sg.g[k][0].z;
                fv(sg.g[k][0].x); fv(sg.g[k][0].y); fv(sg.g[k][0].z); fv(sg.g[k][0].w);
            }
        }
    }

    const int wbase = tid & ~63;
    const int npix = a.height * a.width;
    __syncthreads();                                       // s_mat is staged
    unsigned long long wave_segs = 0;                      // lane 0: the wave's segment count
    const int pixel0 = blockIdx.x * BLOCK;
    const bool in_canvas = pixel0 + tid < npix;
    const int pc = in_canvas ? pixel0 + tid : 0;
    const uint32_t B = (uint32_t)a.max_bounces;
    const uint32_t Bm1 = B - 1u;                    // 1 <= maxBounces (pt_create): the depth cap as one compare
    const uint32_t stride = a.sample_stride;
    // this workgroup's slice of the launch's samples (TraceArgs); slices beyond the first start their sums at 0
    // in a plane of their own
    const int slice = blockIdx.y;
    const int n_samples = a.slice_n[slice];
    const uint32_t sample_begin = a.sample_begin + (uint32_t)a.slice_j0[slice] * stride;
    float *const canvas = slice == 0 ? a.canvas : a.partial + (size_t)(slice - 1) * npix * 3;
    const bool fresh = slice == 0 ? a.fresh != 0 : true;
    const uint32_t gs_end = sample_begin + (uint32_t)n_samples * stride;   // first sample index beyond this slice
    const uint32_t ctr_end = gs_end << 6;

    unsigned long long segs = 0;
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (in_canvas && !fresh) acc = mk3(canvas[3 * (size_t)pc + 0], canvas[3 * (size_t)pc + 1], canvas[3 * (size_t)pc + 2]);

    // The primary segment (src/Lib.hs:41-51, :81-98) has no jitter and uses no randomness: it is the same for every
    // sample and every launch of this camera and scene.  primary_fill_kernel (pt_kernels.hip) computes it once per
    // scene in fp64, in the reference's operation order, and leaves point, direction and record offset in a.pix.
    f3 cd0 = mk3(0.f, 0.f, 0.f), cP0;
    int cidxp;
    {
        // (P0, record offset) (d0, -): a diffuse scene's kernel never reads the primary direction -- its bounces do not
        // depend on the incoming ray (src/Lib.hs:129-132) -- and fetches the first 16 bytes only
        const float4 q0 = a.pix[(size_t)pc];                   // two planes of 16 bytes per pixel: whole lines of the first only
        cP0 = mk3(q0.x, q0.y, q0.z); cidxp = __float_as_int(q0.w);
        if constexpr (SPEC) {
            const float4 q1 = a.pix[(size_t)npix + (size_t)pc];
            cd0 = mk3(q1.x, q1.y, q1.z);
        }
    }
    f3 ccol = mk3(1.f, 1.f, 1.f);
    float ce = 0.f;
    bool traced = in_canvas && cidxp != 0 && B >= 2;      // else every sample is black (src/Lib.hs:68,72)
    if (in_canvas && cidxp != 0) {
        const float4 m0 = matp(cidxp, 0);
        ce = m0.w; ccol = mk3(m0.x, m0.y, m0.z);
        if (ce > 0.f) {                                    // the primary ray sees a light (src/Lib.hs:74)
            for (int s = 0; s < n_samples; ++s) { acc.x += ce; acc.y += ce; acc.z += ce; }
            traced = false;
        }
    }
    s_pix[tid * RV + 0] = make_float4(cP0.x, cP0.y, cP0.z, __int_as_float(cidxp));
    s_pix[tid * RV + 1] = make_float4(ccol.x, ccol.y, ccol.z, ce);
    s_pix[tid * RV + 2] = make_float4(cd0.x, cd0.y, cd0.z, 0.f);
    // LDS byte address of the current pixel's restart record (MASKS: read under exec at every restart, so that
    // the seven constants do not occupy registers through the loop -- 8 waves per SIMD instead of 7)
    const uint32_t rbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&s_pix[0];
    uint32_t raddr = rbase + (uint32_t)tid * (RV * 16);

    uint32_t pre_hi, pre_lo;                               // Philox's first product M * pixel, hoisted out of the loop
    {
        const uint64_t pre = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)pc;
        pre_hi = (uint32_t)(pre >> 32); pre_lo = (uint32_t)pre;
    }
    f3 o = cP0, d = cd0, thr = ccol;
    int lastp = cidxp;
    unsigned long long prim_is[NS], excl[NS];
    auto primary_masks = [&]() {
#pragma unroll
        for (int k = 0; k < NS; ++k) prim_is[k] = __builtin_amdgcn_ballot_w64(cidxp == 16 * (k + 1));
    };
    primary_masks();
#pragma unroll
    for (int k = 0; k < NS; ++k) excl[k] = prim_is[k];
    uint32_t ctr = (sample_begin << 6) | 1u;               // (global sample << 6) | segments searched on this path
    bool alive = traced;                                   // has a path in flight
    const PhiloxRegs pregs = philox_regs(a.seed);

    auto scatter = [&](int idxp, float e, f3 din, uint32_t c1, f3 at) -> f3 {
        uint32_t r0, r1;
        philox2x32_10_pre_v(pre_hi, pre_lo, c1, pregs, r0, r1);
        if constexpr (SPEC) {
            const float4 m3 = matp(idxp, 3);
            f3 nrm = mk3(m3.y, m3.z, m3.w);
            const bool sphere = m3.x != m3.x;              // NaN marks the Sphere extension: normal at the hit point
            if (sphere) {
                const float4 m1 = matp(idxp, 1);
                nrm = mk3((at.x - m1.x) * m1.w, (at.y - m1.y) * m1.w, (at.z - m1.z) * m1.w);
            }
            if (e < -1.5f)                                 // _reflective (src/Lib.hs:110)
                return scatter_specular(din, nrm, u_theta(r0), u_phi(r1), NOISE_REV);
            if (sphere) return scatter_diffuse_about(nrm, r0, r1);
        }
        return scatter_diffuse_bits(matp(idxp, 1), matp(idxp, 2), matp(idxp, 3).x, r0, r1);   // src/Lib.hs:111
    };
    // MASKS: the second half of a trip.  The continuing path's values for every lane, then -- for the lanes of `tm`,
    // whose path ended and restarts as sample c1r >> 6 of the pixel at raddr -- the restart values read in under the
    // lane mask; scatter; the next trip's _lastStruck masks.
    auto finish_masks = [&](unsigned long long tm, int kp, float t, const float4 &m0, uint32_t c1, uint32_t c1r,
                            const unsigned long long *okm) {
        int idxp; float e; f3 din, hp;
        // The continuing path's values for every lane, then the restart values read in under the lane
        // mask from the pixel's LDS record: two ds_read_b128 and one v_mov_b32 where selects would be eight
        // v_cndmask_b32 (second-port class; tools/ubench_issue.hip) from seven registers held through the loop
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 H = {fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z), __int_as_float(kp)};   // march (src/Object.hs:151-153)
        f4 T = {thr.x * m0.x, thr.y * m0.y, thr.z * m0.z, m0.w};                                // pixel * color (src/Lib.hs:79)
        unsigned long long saved;
        if constexpr (SPEC) {                          // the specular bounce also reads the incoming direction
            f4 D = {d.x, d.y, d.z, 0.f};
            asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                         "ds_read_b128 %[H], %[ad]\n\t"
                         "ds_read_b128 %[T], %[ad] offset:16\n\t"
                         "ds_read_b128 %[D], %[ad] offset:32\n\t"
                         "v_mov_b32 %[c], %[cr]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [H] "+v"(H), [T] "+v"(T), [D] "+v"(D), [c] "+v"(c1), [sv] "=&s"(saved)
                         : [m] "s"(tm), [ad] "v"(raddr), [cr] "v"(c1r) : "memory");
            din = mk3(D.x, D.y, D.z);
            thr = mk3(T.x, T.y, T.z); e = T.w;
        } else {
            // (the diffuse bounce does not read the tag: a 96-bit read straight into the throughput's registers;
            // with the tag as a fourth element hipcc formed the tuple elsewhere and copied it back, 3 v_mov)
            typedef float f3v __attribute__((ext_vector_type(3)));
            f3v T3 = {thr.x * m0.x, thr.y * m0.y, thr.z * m0.z};
            asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                         "ds_read_b128 %[H], %[ad]\n\t"
                         "ds_read_b96 %[T], %[ad] offset:16\n\t"
                         "v_mov_b32 %[c], %[cr]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [H] "+v"(H), [T] "+v"(T3), [c] "+v"(c1), [sv] "=&s"(saved)
                         : [m] "s"(tm), [ad] "v"(raddr), [cr] "v"(c1r) : "memory");
            din = d;
            thr = mk3(T3.x, T3.y, T3.z); e = m0.w;
        }
        hp = mk3(H.x, H.y, H.z); idxp = __float_as_int(H.w);
        d = scatter(idxp, e, din, c1, hp);
        o = hp; lastp = idxp; ctr = c1 + 1u;
        // next trip's _lastStruck masks: the primary object where the path restarts, else this
        // trip's winner = the LAST object that was accepted
        unsigned long long later = 0;
#pragma unroll
        for (int k = NS - 1; k >= 0; --k) {
            const unsigned long long won = okm[k] & ~later;
            later |= okm[k];
            excl[k] = (tm & prim_is[k]) | (won & ~tm);
        }
    };
    // One segment.  `draw(terminal, c1, c1r)` decides what a finished path does next
    // and returns in c1r the counter word (sample << 6 | 0) of a restarted path.
    auto trip = [&](auto all_alive, auto &&draw) {
        int kp; float t;
        unsigned long long okm[NS];
        closest_static<NS, SPEC, MASKS>(sg, a.rect, o, d, lastp, excl, okm, kp, t);
        const bool live = decltype(all_alive)::value ? true : alive;
        if (!decltype(all_alive)::value) segs += live ? 1 : 0;            // phase 1 counts its trips wave-uniformly
        const float4 m0 = matp(kp, 0);
        // miss (tag 0, src/Lib.hs:72), light (tag e > 0, :74) or the depth cap (:68) end the path
        bool terminal;
        unsigned long long tm = 0;
        if constexpr (MASKS) {                             // the same decision as a lane mask (see below)
            tm = __builtin_amdgcn_ballot_w64(m0.w >= 0.f) | __builtin_amdgcn_ballot_w64((ctr & 63u) >= Bm1);
            if (!decltype(all_alive)::value) tm &= __builtin_amdgcn_ballot_w64(live);
            terminal = lane_of(tm);
        } else {
            terminal = live && ((m0.w >= 0.f) || ((ctr & 63u) >= Bm1));
        }
        {                                                  // emittance * pixel (src/Lib.hs:74); max(tag, 0) is 0 unless a light
            float E = __int_as_float(max(__float_as_int(m0.w), 0));
            if (!decltype(all_alive)::value) E = live ? E : 0.f;
            acc.x = fmaf(E, thr.x, acc.x);
            acc.y = fmaf(E, thr.y, acc.y);
            acc.z = fmaf(E, thr.z, acc.z);
        }
        uint32_t c1 = ctr, c1r = ctr;
        draw(terminal, c1, c1r);                           // may clear or set `terminal` (= restart from cP0)
        if constexpr (MASKS) {
            if (!decltype(all_alive)::value) tm = __builtin_amdgcn_ballot_w64(terminal);   // phase 2's draw rewrites it
        }
        if constexpr (MASKS) {
            finish_masks(tm, kp, t, m0, c1, c1r, okm);
        } else {
            int idxp; float e; f3 din, hp;
            c1 = terminal ? c1r : c1;
            idxp = terminal ? cidxp : kp;
            e    = terminal ? ce : m0.w;
            din  = terminal ? cd0 : d;
            hp = mk3(terminal ? cP0.x : fmaf(t, d.x, o.x),                        // march (src/Object.hs:151-153)
                     terminal ? cP0.y : fmaf(t, d.y, o.y),
                     terminal ? cP0.z : fmaf(t, d.z, o.z));
            thr = mk3(terminal ? ccol.x : thr.x * m0.x,                            // pixel * color (src/Lib.hs:79)
                      terminal ? ccol.y : thr.y * m0.y,
                      terminal ? ccol.z : thr.z * m0.z);
            d = scatter(idxp, e, din, c1, hp);
            o = hp; lastp = idxp; ctr = c1 + 1u;
        }
    };

    if (alive) d = scatter(cidxp, ce, cd0, ctr - 1u, cP0);

    // ---- phase 1: every lane on its own pixel --------------------------------
    const unsigned long long everyone = __ballot(true);
    const uint32_t next_sample = (stride << 6) - 63u;      // (ctr | 63) + next_sample = (sample + stride) << 6
    if (__ballot(alive) == everyone) {
        uint32_t trips = 0;                                // wave-uniform
        do {
            trip(true_type{}, [&](bool &terminal, uint32_t &c1, uint32_t &c1r) {
                c1r = (c1 | 63u) + next_sample;
            });
            alive = ctr < ctr_end;                         // the sample the lane is on belongs to this launch
            trips += 1;
        } while (__ballot(alive) == everyone);
        segs += trips;
    }
    // A lane that has just left phase 1 restarted a path that does not exist
    // (its sample index is gs_end); it enters phase 2 pathless.

    // ---- phase 2: the wave shares what is left -------------------------------
    unsigned long long nseg2 = 0;                          // wave-uniform: searches of phase 2 (MASKS)
    if constexpr (MASKS) {
        // Everything of this phase is a lane mask on the scalar unit or lives in the pixel's LDS record, so a trip
        // costs what a phase-1 trip does plus one LDS atomic under exec, a shift and a compare -- unless a lane
        // needs a pixel to help, which takes the wave through the adoption loop below.
        //   alive_m : lanes with a path in flight.  A lane without one keeps executing the trip on whatever its
        //             registers hold, with a throughput of zero (it adds nothing) and outside every mask.
        //   draw    : a lane whose path ended takes the next sample of the pixel it works on with ds_add_rtn_u32 on
        //             the record's counter ([2].w), under exec.
        //   adopt   : lanes left without a sample pick -- wave-uniformly, s_ff1 on the ballot of pixels that still
        //             have unstarted samples -- a pixel v, draw from its counter, hand what they gathered to their
        //             old pixel's record ([3].xyz, ds_add_f32) and point raddr at v's record; Philox's pixel product
        //             and the primary-object masks of the movers are wave-uniform values (one pixel), set from SGPRs.
        const uint32_t own = rbase + (uint32_t)tid * (RV * 16);
        lds_write_u32(own + 44, alive ? (ctr >> 6) + stride : gs_end);
        s_hand[tid * 3 + 0] = 0.f; s_hand[tid * 3 + 1] = 0.f; s_hand[tid * 3 + 2] = 0.f;
        const uint32_t hbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&s_hand[0];
        auto hand_addr = [&](uint32_t rec) { return hbase + ((rec - rbase) >> 2); };    // 48-byte records, 12-byte sums
        unsigned long long alive_m = __builtin_amdgcn_ballot_w64(alive);
        thr = mk3(alive ? thr.x : 0.f, alive ? thr.y : 0.f, alive ? thr.z : 0.f);
        const uint32_t stride_v = to_vgpr(stride);
        bool dry = false;                                  // wave-uniform: no pixel has an unstarted sample
        if (alive_m != 0ull) do {
            int kp; float t;
            unsigned long long okm[NS];
            closest_static<NS, SPEC, true>(sg, a.rect, o, d, lastp, excl, okm, kp, t);
            nseg2 += (unsigned)__builtin_popcountll(alive_m);
            const float4 m0 = matp(kp, 0);
            // miss (tag 0, src/Lib.hs:72), light (tag e > 0, :74) or the depth cap (:68) end the path
            unsigned long long tm = (__builtin_amdgcn_ballot_w64(m0.w >= 0.f) |
                                     __builtin_amdgcn_ballot_w64((ctr & 63u) >= Bm1)) & alive_m;
            {                                              // emittance * pixel (src/Lib.hs:74); a pathless lane's pixel is 0
                const float E = __int_as_float(max(__float_as_int(m0.w), 0));
                acc.x = fmaf(E, thr.x, acc.x);
                acc.y = fmaf(E, thr.y, acc.y);
                acc.z = fmaf(E, thr.z, acc.z);
            }
            const uint32_t c1 = ctr;
            uint32_t g = lds_add_rtn_under<44>(tm, raddr, stride_v, c1);       // next sample of the lane's current pixel
            unsigned long long want = (__builtin_amdgcn_ballot_w64(g >= gs_end) & tm) | (everyone & ~alive_m);
            unsigned long long got = 0;
            if (want != 0ull && !dry) {
                do {
                    const unsigned long long avail = __builtin_amdgcn_ballot_w64(lds_read_u32(own + 44) < gs_end);
                    if (avail == 0ull) { dry = true; break; }
                    const uint32_t v = (uint32_t)__builtin_ctzll(avail);               // wave-uniform choice of pixel to help
                    const uint32_t rv = rbase + v * (RV * 16);
                    const uint32_t g2 = lds_add_rtn_under<44>(want, to_vgpr(rv), stride_v, g);
                    const unsigned long long gotnow = __builtin_amdgcn_ballot_w64(g2 < gs_end) & want;
                    const unsigned long long mv = gotnow & __builtin_amdgcn_ballot_w64(raddr != rv);
                    if (mv != 0ull) {                      // hand over what was gathered, adopt pixel v
                        const uint64_t pre = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)(pixel0 + (int)v);
                        const uint32_t phi = (uint32_t)(pre >> 32), plo = (uint32_t)pre;
                        unsigned long long saved;
                        const uint32_t ha = hand_addr(raddr);
                        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                                     "ds_add_f32 %[ha], %[ax]\n\t"
                                     "ds_add_f32 %[ha], %[ay] offset:4\n\t"
                                     "ds_add_f32 %[ha], %[az] offset:8\n\t"
                                     "v_mov_b32 %[ra], %[rv]\n\t"
                                     "v_mov_b32 %[ph], %[sph]\n\t"
                                     "v_mov_b32 %[pl], %[spl]\n\t"
                                     "s_nop 1\n\t"
                                     "v_mov_b32 %[ax], 0\n\t"
                                     "v_mov_b32 %[ay], 0\n\t"
                                     "v_mov_b32 %[az], 0\n\t"
                                     "s_mov_b64 exec, %[sv]"
                                     : [ax] "+v"(acc.x), [ay] "+v"(acc.y), [az] "+v"(acc.z), [ra] "+v"(raddr),
                                       [ph] "+v"(pre_hi), [pl] "+v"(pre_lo), [sv] "=&s"(saved)
                                     : [m] "s"(mv), [rv] "s"(rv), [sph] "s"(phi), [spl] "s"(plo), [ha] "v"(ha) : "memory");
                        const int idx_v = __builtin_amdgcn_readfirstlane((int)lds_read_u32(rv + 12));
#pragma unroll
                        for (int k = 0; k < NS; ++k)
                            prim_is[k] = (prim_is[k] & ~mv) | (idx_v == 16 * (k + 1) ? mv : 0ull);
                    }
                    g = pick(gotnow, g2, g);
                    got |= gotnow;
                    want &= ~gotnow;
                } while (want != 0ull);
            }
            // want: lanes that asked and got nothing -- no path from here on
            const unsigned long long dying = want & alive_m;
            tm = (tm & ~want) | got;                       // restart from the (current) pixel's primary hit
            alive_m = (alive_m & ~want) | got;
            finish_masks(tm, kp, t, m0, c1, g << 6, okm);
            if (dying != 0ull) {
                const bool gone = lane_of(dying);
                thr = mk3(gone ? 0.f : thr.x, gone ? 0.f : thr.y, gone ? 0.f : thr.z);
            }
        } while (alive_m != 0ull);
        {   // what the lane still holds goes to the record of the pixel it worked on last
            asm volatile("ds_add_f32 %[ha], %[ax]\n\t"
                         "ds_add_f32 %[ha], %[ay] offset:4\n\t"
                         "ds_add_f32 %[ha], %[az] offset:8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : : [ha] "v"(hand_addr(raddr)), [ax] "v"(acc.x), [ay] "v"(acc.y), [az] "v"(acc.z) : "memory");
        }
    } else {
        s_next[tid] = alive ? (ctr >> 6) + stride : gs_end;
        s_acc[tid * 3 + 0] = 0.f; s_acc[tid * 3 + 1] = 0.f; s_acc[tid * 3 + 2] = 0.f;
        __syncthreads();
        int cur = tid;                                         // the pixel (workgroup-relative) this lane works on
        bool dry = false;                                      // wave-uniform: no pixel has an unstarted sample
        if (__ballot(alive) != 0ull || !dry) do {
            trip(false_type{}, [&](bool &terminal, uint32_t &c1, uint32_t &c1r) {
                uint32_t g = gs_end;
                if (terminal) g = atomicAdd(&s_next[cur], stride);           // next sample of my current pixel
                const bool want = terminal || !alive;          // lanes that need a (new) path
                bool needy = want && !(terminal && g < gs_end);
                const bool adopting = !dry && __ballot(needy) != 0ull;
                while (!dry && __ballot(needy) != 0ull) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    const bool has = __hip_atomic_load(&s_next[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gs_end;
                    const unsigned long long avail = __ballot(has);      // pixels of this wave with unstarted samples
                    if (avail == 0ull) { dry = true; break; }
                    const int v = wbase + (__ffsll((long long)avail) - 1);   // wave-uniform choice of pixel to help
                    if (needy) {
                        g = atomicAdd(&s_next[v], stride);
                        if (g < gs_end) {
                            if (v != cur) {                    // hand over what was gathered, adopt pixel v
                                atomicAdd(&s_acc[cur * 3 + 0], acc.x);
                                atomicAdd(&s_acc[cur * 3 + 1], acc.y);
                                atomicAdd(&s_acc[cur * 3 + 2], acc.z);
                                acc = mk3(0.f, 0.f, 0.f);
                                cur = v;
                                raddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&s_pix[v * RV];
                                const float4 p0 = s_pix[v * RV + 0];
                                cidxp = __float_as_int(p0.w);
                                if constexpr (!MASKS) {        // the select form keeps the restart constants in registers
                                    const float4 p1 = s_pix[v * RV + 1], p2 = s_pix[v * RV + 2];
                                    cP0 = mk3(p0.x, p0.y, p0.z); ccol = mk3(p1.x, p1.y, p1.z); ce = p1.w;
                                    cd0 = mk3(p2.x, p2.y, p2.z);
                                }
                                const uint64_t pre = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)(pixel0 + v);
                                pre_hi = (uint32_t)(pre >> 32); pre_lo = (uint32_t)pre;
                            }
                            needy = false;
                        }
                    }
                }
                if (MASKS && adopting) primary_masks();        // lanes may have moved to another pixel
                alive = !want || !needy;                       // path continues, or a sample was drawn
                terminal = want && !needy;                     // restart from the (current) pixel's primary hit
                c1r = g << 6;
            });
        } while (__ballot(alive) != 0ull || !dry);

        atomicAdd(&s_acc[cur * 3 + 0], acc.x);
        atomicAdd(&s_acc[cur * 3 + 1], acc.y);
        atomicAdd(&s_acc[cur * 3 + 2], acc.z);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) segs += __shfl_down(segs, off, 64);
    wave_segs += segs + nseg2;
    __syncthreads();
    if (in_canvas) {
        if constexpr (MASKS) {
            canvas[3 * (size_t)pc + 0] = s_hand[tid * 3 + 0];
            canvas[3 * (size_t)pc + 1] = s_hand[tid * 3 + 1];
            canvas[3 * (size_t)pc + 2] = s_hand[tid * 3 + 2];
        } else {
            canvas[3 * (size_t)pc + 0] = s_acc[tid * 3 + 0];
            canvas[3 * (size_t)pc + 1] = s_acc[tid * 3 + 1];
            canvas[3 * (size_t)pc + 2] = s_acc[tid * 3 + 2];
        }
    }
    // one global atomic per workgroup for the segment counter: 32,400 same-address
    // atomics (one per wave) serialise in L2 when a short launch's workgroups
    // retire together
    if (tid == 0 && wave_segs) atomicAdd(seg_lane(a.segments), wave_segs);
}

#ifdef PT_JIT
template __global__ void trace_static_kernel<PT_JIT_NS, PT_JIT_SPEC != 0>(TraceArgs, StaticGeom<PT_JIT_NS>);
#endif

} // namespace pt
