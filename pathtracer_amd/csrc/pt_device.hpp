// pt_device.hpp -- device-side maths of the radiance loop for gfx950 (wave64).
//
// fp32 VALU restatement of the reference's per-segment work, designed for the
// CDNA4 issue model rather than translated from the Haskell:
//   * Triple.hs vector algebra is inlined scalar fp32 (src/Triple.hs:47-69);
//   * ray/plane/disk intersection uses per-object constants folded at upload
//     (src/Object.hs:156-177);
//   * the trig-of-inverse-trig chain of rotateRel (src/Util.hs:114-145) is
//     replaced by its algebraic form: for a diffuse bounce the whole rotation
//     is a per-object 3x3 matrix, for a specular bounce it is built from the
//     mirror direction's components with one rsqrt;
//   * random draws come from Philox2x32-10 keyed by (pixel, sample, bounce).
// The per-pixel primary segment alone is fp64 and follows the reference's
// operation order, so primary-hit indices are bit-exact (src/Lib.hs:41-51,
// :81-98).
// File:line citations are relative to the reference repository root.
#pragma once

#ifndef __HIPCC_RTC__          // hiprtc brings its own runtime header and the fixed-width integer types
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef signed char int8_t;       typedef unsigned char uint8_t;
typedef short int16_t;            typedef unsigned short uint16_t;
typedef int int32_t;              typedef unsigned int uint32_t;
typedef long int64_t;             typedef unsigned long uint64_t;
typedef unsigned long uintptr_t;
#endif

namespace pt {

// ---- device scene layout --------------------------------------------------
// Geometry record, 12 floats (3 x float4), read with a wave-uniform index:
//   g0 = (n_raw.x, n_raw.y, n_raw.z, A)        A  = point . n_raw
//   g1 = (n_hat.x, n_hat.y, n_hat.z, r2)       r2 = radius^2, +inf for a plane
//   g2 = (c.x, c.y, c.z, 0)                    disk centre
// so that  t = (A - o.n_raw) / (v.n_hat)  ==  ((p-o).n) / (v.normalize n)
// (src/Object.hs:176-177, raw normal above and unit normal below kept).
// Sphere (extension, no reference counterpart: src/Object.hs:191-194 is a TODO):
//   g0 = 0, g1 = (0, 0, 0, -r^2)  -- a NEGATIVE g1.w marks the form --, g2 = (c.xyz, 0)
constexpr int GEOM_FLOATS = 12;
// Material record, 16 floats (4 x float4), read with a per-lane index:
//   m0 = (color.r, color.g, color.b, e)   e > 0: emittance (light)
//                                         e = 0: diffuse, e = -1: reflective
//   m1 = (M00, M01, M02, M10)             M = |n| * Rz(phi') * Ry(theta'),
//   m2 = (M11, M12, M20, M21)             (theta', phi') = toSphericalCoords n
//   m3 = (M22, n_hat.x, n_hat.y, n_hat.z) (src/Util.hs:114-145)
// so a diffuse bounce reads m1, m2 (two 16-B LDS reads) and one dword of m3,
// and only a specular bounce needs the unit normal.
// Sphere: m1 = (c.xyz, 1/r), m3.x = NaN (the marker): the normal depends on the
// hit point, so the bounce builds its rotation per hit (scatter_about).
constexpr int MAT_FLOATS = 16;
// fp64 record for the primary pass, 11 doubles:
//   point[3], n_raw[3], n_hat[3] (normalize n, src/Triple.hs:66-69), radius, kind
constexpr int PRIM_DOUBLES = 19;     // centre 3, normal 3, unit normal 3, radius, kind, Rectangle: u 3, w 3, half width, half height

constexpr int MAX_STATIC_OBJECTS = 8;
constexpr int MAX_JIT_OBJECTS = 16;             // scenes of 9-16 objects have a by-value kernel only as a scene kernel (pt_jit.hip)

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 fma3(float s, f3 a, f3 b) { return mk3(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }

// ---- Philox2x32-10 ----------------------------------------------------------
// Salmon, Moraes, Dror, Shaw, SC'11.  One call = the two uniforms one bounce
// needs (randomRangeList gen [(0,a),(0,380)], src/Lib.hs:127,132).  The key
// schedule is wave-uniform, so the ten round keys live in SGPRs.
__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key,
                                              uint32_t &o0, uint32_t &o1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t prod = (uint64_t)0xD256D193u * (uint64_t)c0;
        const uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
        // three-input xor in one VALU op (v_bitop3_b32, truth table 0x96)
        c0 = __builtin_amdgcn_bitop3_b32(hi, key + (uint32_t)r * 0x9E3779B9u, c1, 0x96);
        c1 = lo;
    }
    o0 = c0; o1 = c1;
}

// The same generator with the first round's product M * c0 supplied by the
// caller: the trace kernel's c0 is the pixel index, constant for as long as a
// lane works on one pixel, so that multiply leaves the per-segment loop.
// Round keys and multiplier held in VGPRs (wave-uniform values in vector
// registers): a VALU instruction that reads an SGPR can issue only once per
// 4 cycles on gfx950 against 2 for an all-VGPR one (tools/ubench_issue.hip), and
// the trace loop already spends that budget on the object list.
struct PhiloxRegs { uint32_t k[10]; uint32_t m; };
__device__ __forceinline__ uint32_t to_vgpr(uint32_t x)
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}
__device__ __forceinline__ PhiloxRegs philox_regs(uint32_t key)
{
    PhiloxRegs p;
#pragma unroll
    for (int r = 0; r < 10; ++r) p.k[r] = to_vgpr(key + (uint32_t)r * 0x9E3779B9u);
    p.m = to_vgpr(0xD256D193u);
    return p;
}
__device__ __forceinline__ void philox2x32_10_pre_v(uint32_t hi0, uint32_t lo0, uint32_t c1, const PhiloxRegs &p,
                                                    uint32_t &o0, uint32_t &o1)
{
    uint32_t c0 = __builtin_amdgcn_bitop3_b32(hi0, p.k[0], c1, 0x96);
    c1 = lo0;
#pragma unroll
    for (int r = 1; r < 10; ++r) {
        // one v_mad_u64_u32 per round, spelled out: left to itself hipcc widens
        // the multiplier to 64 bits inside the work-sharing loop and emits two
        uint64_t prod, carry;                          // the carry-out goes to a scratch SGPR pair, not vcc
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(prod), "=s"(carry) : "v"(c0), "v"(p.m));
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(prod >> 32), p.k[r], c1, 0x96);
        c1 = (uint32_t)prod;
    }
    o0 = c0; o1 = c1;
}

__device__ __forceinline__ void philox2x32_10_pre(uint32_t hi0, uint32_t lo0, uint32_t c1, uint32_t key,
                                                  uint32_t &o0, uint32_t &o1)
{
    uint32_t c0 = __builtin_amdgcn_bitop3_b32(hi0, key, c1, 0x96);
    c1 = lo0;
#pragma unroll
    for (int r = 1; r < 10; ++r) {
        const uint64_t prod = (uint64_t)0xD256D193u * (uint64_t)c0;
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(prod >> 32), key + (uint32_t)r * 0x9E3779B9u, c1, 0x96);
        c1 = (uint32_t)prod;
    }
    o0 = c0; o1 = c1;
}

// The two uniforms of a bounce, taken from the low bits of the two Philox words:
//   u1 = (r0 & 0x1fffff) / 2^21   (theta)      u2 = (r1 & 0x7fffff) / 2^23   (phi)
// Both are exactly representable in fp32 and fp64, so the device and the
// oracle's Philox mode see the same numbers.  The widths are chosen so that the
// diffuse bounce can build its angles with bit operations instead of an
// int->float conversion and a multiply (scatter_diffuse_bits).
constexpr uint32_t U1_MASK = 0x1fffffu, U2_MASK = 0x7fffffu;
__device__ __forceinline__ float u_theta(uint32_t r0) { return (float)(r0 & U1_MASK) * (1.0f / 2097152.0f); }
__device__ __forceinline__ float u_phi(uint32_t r1)   { return (float)(r1 & U2_MASK) * (1.0f / 8388608.0f); }

// sin/cos of an angle given in REVOLUTIONS (v_sin_f32 / v_cos_f32 take the
// argument pre-divided by 2*pi; |x| <= 256).
__device__ __forceinline__ float sin_rev(float x) { return __builtin_amdgcn_sinf(x); }
__device__ __forceinline__ float cos_rev(float x) { return __builtin_amdgcn_cosf(x); }

// fromSphericalCoords (src/Util.hs:124-129) with theta, phi in revolutions.
__device__ __forceinline__ f3 local_dir(float theta_rev, float phi_rev)
{
    const float st = sin_rev(theta_rev), ct = cos_rev(theta_rev);
    const float sp = sin_rev(phi_rev),   cp = cos_rev(phi_rev);
    return mk3(cp * st, sp * st, ct);
}

constexpr float PHI_REV   = 380.0f / 360.0f;   // phi ~ U(0, 380) degrees (sic), src/Lib.hs:127,132
constexpr float THETA_DIFFUSE_REV = 0.25f;      // theta ~ U(0, 90) degrees, src/Lib.hs:132
constexpr float NOISE_REV = 40.0f / 360.0f;    // specular noise 40 degrees, src/Lib.hs:110

// diffuse (src/Lib.hs:129-132): rotateRel theta phi normal, with the rotation
// about the (per-object constant) normal folded into M at upload.
__device__ __forceinline__ f3 scatter_diffuse(float4 m1, float4 m2, float m22, float u1, float u2)
{
    const f3 l = local_dir(THETA_DIFFUSE_REV * u1, PHI_REV * u2);
    return mk3(fmaf(m1.z, l.z, fmaf(m1.y, l.y, m1.x * l.x)),
               fmaf(m2.y, l.z, fmaf(m2.x, l.y, m1.w * l.x)),
               fmaf(m22,  l.z, fmaf(m2.w, l.y, m2.z * l.x)));
}

// The trace kernel's form: the two Philox words go straight to angles.
//   theta: 0x3f800000 | (r0 & 0x1fffff) is the float 1 + u1/4, and v_sin/v_cos are
//          periodic in whole revolutions, so that IS the angle 90 deg * u1;
//   phi:   0x3f800000 | (r1 & 0x7fffff) is f = 1 + u2, and fma(f, P, -P) rounds
//          P*u2 once, bit for bit the product PHI_REV * u2.
// One v_and_or_b32 (+ one fma) per angle instead of shift, convert, multiply.
__device__ __forceinline__ f3 scatter_diffuse_bits(float4 m1, float4 m2, float m22, uint32_t r0, uint32_t r1)
{
    const float theta_rev = __uint_as_float((r0 & U1_MASK) | 0x3f800000u);
    const float f = __uint_as_float((r1 & U2_MASK) | 0x3f800000u);
    const f3 l = local_dir(theta_rev, fmaf(f, PHI_REV, -PHI_REV));
    // the last row of rotateAbs has no y term (r.z = -sin theta' x + cos theta' z, src/Util.hs:137): M[2][1] = m2.w is 0
    return mk3(fmaf(m1.z, l.z, fmaf(m1.y, l.y, m1.x * l.x)),
               fmaf(m2.y, l.z, fmaf(m2.x, l.y, m1.w * l.x)),
               fmaf(m22,  l.z, m2.z * l.x));
}

// asin(x)/(2*pi) for x in [0,1], in revolutions.  Abramowitz-Stegun 4.4.46
// form acos(x) = sqrt(1-x) * P7(x), |err| < 2e-8 rad before fp32 rounding.
__device__ __forceinline__ float asin_rev01(float x)
{
    float p = -0.0012624911f;
    p = fmaf(p, x,  0.0066700901f);
    p = fmaf(p, x, -0.0170881256f);
    p = fmaf(p, x,  0.0308918810f);
    p = fmaf(p, x, -0.0501743046f);
    p = fmaf(p, x,  0.0889789874f);
    p = fmaf(p, x, -0.2145988016f);
    p = fmaf(p, x,  1.5707963050f);
    const float acos_x = __builtin_amdgcn_sqrtf(fmaxf(1.0f - x, 0.0f)) * p;
    return (1.5707963268f - acos_x) * 0.15915494309f;
}

// rotateRel (src/Util.hs:140-145) applied to the local direction l =
// fromSphericalCoords theta phi: rotate l by the spherical angles of `axis` and
// rescale by |axis|.  toSphericalCoords axis (src/Util.hs:114-122) in algebraic form:
//   cos th' = a.z, sin th' = rho, cos ph' = a.x/rho, sin ph' = a.y/rho,
// a = normalize axis, rho = sqrt(a.x^2 + a.y^2).
__device__ __forceinline__ f3 rotate_rel_local(f3 l, f3 axis)
{
    const float lenp = __builtin_amdgcn_sqrtf(dot3(axis, axis));
    const float invp = __builtin_amdgcn_rcpf(fmaxf(lenp, 1e-6f));   // normalize floors the norm at 1e-6
    const f3 a = mk3(axis.x * invp, axis.y * invp, axis.z * invp);
    const float rho2 = fmaf(a.y, a.y, a.x * a.x);
    float ct, st, cp, sp;
    if (rho2 > 0.0f) {
        const float irho = __builtin_amdgcn_rsqf(rho2);
        ct = a.z; st = rho2 * irho;
        cp = a.x * irho; sp = a.y * irho;
        // arctan2 takes x = -0.0 through its x >= 0 branch with y/x = -+Inf
        // (src/Util.hs:106-112): phi' lands on the opposite side.
        if (a.x == 0.0f && __builtin_signbit(a.x)) sp = -sp;
    } else {
        // x = y = 0 guards (src/Util.hs:116-117): (0,0) or (180,0)
        ct = (a.z >= 0.0f) ? 1.0f : -1.0f; st = 0.0f; cp = 1.0f; sp = 0.0f;
    }
    // rotateAbs local theta' phi' (src/Util.hs:131-138)
    const float x = fmaf(st * cp, l.z, fmaf(-sp, l.y, (ct * cp) * l.x));
    const float y = fmaf(sp * st, l.z, fmaf( cp, l.y, (sp * ct) * l.x));
    const float z = fmaf(ct, l.z, -st * l.x);
    return mk3(x * lenp, y * lenp, z * lenp);
}

// specular (src/Lib.hs:115-127) + rotateRel about the mirror direction
// (src/Util.hs:140-145).  n_hat = normalize normal.  Direction length is
// preserved (the reference rescales by norm2 vector').
__device__ __forceinline__ f3 scatter_specular(f3 v, f3 nh, float u1, float u2, float noise_rev)
{
    const float dvn = dot3(v, nh);
    const f3 vp = fma3(-2.0f * dvn, nh, v);                 // vector + (-2) * projection
    const float len2 = dot3(v, v);
    const float len  = __builtin_amdgcn_sqrtf(len2);
    const float inv  = __builtin_amdgcn_rcpf(fmaxf(len, 1e-6f));   // normalize floors the norm at 1e-6
    // angleWithSurface = 90 - acos |n_hat . v_hat|  ==  asin |n_hat . v_hat|
    const float c = fminf(fabsf(dvn) * inv, 1.0f);
    const float max_theta = fminf(asin_rev01(c), noise_rev);
    const f3 l = local_dir(max_theta * u1, PHI_REV * u2);
    return rotate_rel_local(l, vp);
}

// diffuse (src/Lib.hs:129-132) about a normal that is not a per-object constant
// (the Sphere extension): rotateRel theta phi normal, built per hit.
__device__ __forceinline__ f3 scatter_diffuse_about(f3 normal, uint32_t r0, uint32_t r1)
{
    const float theta_rev = __uint_as_float((r0 & U1_MASK) | 0x3f800000u);
    const float f = __uint_as_float((r1 & U2_MASK) | 0x3f800000u);
    return rotate_rel_local(local_dir(theta_rev, fmaf(f, PHI_REV, -PHI_REV)), normal);
}

// Sphere (extension; the definition is in include/pt.h at PT_SPHERE): centre c,
// r2 = r^2.  m = (oc.v)/(v.v); l = oc - m v; no hit unless l.l <= r2;
// h = sqrt((r2 - l.l)/(v.v)); t = -m - h if positive, else -m + h.  NaN = no hit.
__device__ __forceinline__ float sphere_t(f3 c, float r2, f3 o, f3 v)
{
    const f3 oc = mk3(o.x - c.x, o.y - c.y, o.z - c.z);
    const float ia = __builtin_amdgcn_rcpf(dot3(v, v));
    const float m = dot3(oc, v) * ia;
    const f3 l = mk3(fmaf(-m, v.x, oc.x), fmaf(-m, v.y, oc.y), fmaf(-m, v.z, oc.z));
    const float q = r2 - dot3(l, l);
    const float h = __builtin_amdgcn_sqrtf(q * ia);           // q < 0 -> NaN -> no hit
    const float t1 = -m - h;
    return (t1 > 0.0f) ? t1 : (h - m);
}

// "The hit point is on the form", given d = point - centre: a Disk's
// distanceFromCenter < radius (src/Object.hs:169-170) as |d|^2 < r2, or -- r2 = NaN
// marks the Rectangle extension (definition in include/pt.h at PT_RECTANGLE) --
// |d.u| < width/2 and |d.w| < height/2, with the two axes pre-divided by the half
// sides in rect[2k], rect[2k+1].  EXT = false compiles the Disk test alone.
template <bool EXT>
__device__ __forceinline__ bool inside_form(const float4 *__restrict__ rect, int k, float r2, float dx, float dy, float dz)
{
    if constexpr (EXT) {
        if (r2 != r2) {
            const float4 U = rect[2 * k], W = rect[2 * k + 1];
            const float u = fmaf(dz, U.z, fmaf(dy, U.y, dx * U.x));
            const float w = fmaf(dz, W.z, fmaf(dy, W.y, dx * W.x));
            return fabsf(u) < 1.0f && fabsf(w) < 1.0f;
        }
    }
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r2;
}

// distanceFrom (src/Object.hs:156-177) for one object against one ray.
// Returns t, or NaN where the reference returns Nothing.
__device__ __forceinline__ float distance_from(float4 g0, float4 g1, float4 g2, const float4 *rect, f3 o, f3 v)
{
    if (g1.w < 0.0f) {                                         // Sphere (extension)
        const float ts = sphere_t(mk3(g2.x, g2.y, g2.z), -g1.w, o, v);
        return (__float_as_uint(ts) - 1u < 0x7f7fffffu) ? ts : __builtin_nanf("");
    }
    const float num = fmaf(-o.z, g0.z, fmaf(-o.y, g0.y, fmaf(-o.x, g0.x, g0.w)));
    const float den = fmaf(v.z, g1.z, fmaf(v.y, g1.y, v.x * g1.x));
    // the trace kernel's form (closest_static): the -denorm_min addend turns t = +0
    // negative and leaves every other value as it rounds, so that one unsigned
    // compare of the bits is the guard 0 < t < Inf (src/Object.hs:159)
    const float t = fmaf(num, __builtin_amdgcn_rcpf(den), -__builtin_bit_cast(float, 1u));
    bool ok = __float_as_uint(t) < 0x7f800000u;
    if (g1.w != __builtin_inff()) {                           // Disk, src/Object.hs:165-171
        const f3 d = mk3(fmaf(t, v.x, o.x) - g2.x, fmaf(t, v.y, o.y) - g2.y, fmaf(t, v.z, o.z) - g2.z);
        ok = ok && inside_form<true>(rect, 0, g1.w, d.x, d.y, d.z);
    }
    return ok ? t : __builtin_nanf("");
}

// ---- fp64 primary segment --------------------------------------------------
// rayFromCamToPixel + closestObjectTo for the un-jittered primary ray
// (src/Lib.hs:41-51, :81-98), in the reference's operation order with
// contraction off.  prim = PRIM_DOUBLES doubles per object.
struct PrimaryHit { int idx; double t; double dx, dy, dz; };

// distanceFrom of object k for the primary ray (origin 0) + closestObjectTo's
// update, in the reference's operation order (src/Object.hs:156-177; the Sphere
// extension in the order include/pt.h states), contraction off.
// An equal t keeps the lower index: Data.Vector.minimumBy's first-wins in any
// visiting order (src/Lib.hs:84).
__device__ __forceinline__ void primary_candidate(const double *__restrict__ prim, int k, PrimaryHit &h)
{
#pragma clang fp contract(off)
    const double *p = prim + (size_t)k * PRIM_DOUBLES;
    double t;
    bool ok;
    if (p[10] == 2.0) {                                       // Sphere (extension): o_distance_from_prime, O_SPHERE
        const double ocx = 0.0 - p[0], ocy = 0.0 - p[1], ocz = 0.0 - p[2];
        double a = 0.0;  a = a + h.dx * h.dx;  a = a + h.dy * h.dy;  a = a + h.dz * h.dz;
        double b = 0.0;  b = b + ocx * h.dx;   b = b + ocy * h.dy;   b = b + ocz * h.dz;
        const double m = b / a;
        const double lx = ocx - m * h.dx, ly = ocy - m * h.dy, lz = ocz - m * h.dz;
        const double r2 = p[9] * p[9];
        double d2 = 0.0; d2 = d2 + lx * lx; d2 = d2 + ly * ly; d2 = d2 + lz * lz;
        ok = (p[9] > 0.0) && (d2 <= r2);
        const double hh = sqrt((r2 - d2) / a);
        const double t1 = -m - hh;
        t = (0.0 < t1) ? t1 : -m + hh;
        ok = ok && (0.0 < t) && (t < (double)__builtin_inff());
    } else {
        // (point - origin) with origin = 0, then `dot` as a left fold from 0
        double num = 0.0;
        num = num + (p[0] - 0.0) * p[3];
        num = num + (p[1] - 0.0) * p[4];
        num = num + (p[2] - 0.0) * p[5];
        double den = 0.0;
        den = den + h.dx * p[6];
        den = den + h.dy * p[7];
        den = den + h.dz * p[8];
        t = num / den;
        ok = (0.0 < t) && (t < (double)__builtin_inff());
        if (ok && p[10] != 0.0) {                             // Disk, or the Rectangle extension (kind 3)
            const double px = 0.0 + t * h.dx - p[0];
            const double py = 0.0 + t * h.dy - p[1];
            const double pz = 0.0 + t * h.dz - p[2];
            if (p[10] == 3.0) {
                double du = 0.0; du = du + px * p[11]; du = du + py * p[12]; du = du + pz * p[13];
                double dw = 0.0; dw = dw + px * p[14]; dw = dw + py * p[15]; dw = dw + pz * p[16];
                ok = (fabs(du) < p[17]) && (fabs(dw) < p[18]);
            } else {
                ok = sqrt(px * px + py * py + pz * pz) < p[9];
            }
        }
    }
    if (ok && (h.idx < 0 || t < h.t || (t == h.t && k < h.idx))) { h.idx = k; h.t = t; }
}

__device__ inline PrimaryHit primary_hit_f64(const double *__restrict__ prim, int n,
                                             int H, int W, double camera_depth, int i, int j)
{
#pragma clang fp contract(off)
    PrimaryHit h;
    const double ip = (double)H / 2.0 - (double)i;
    const double jp = (double)j - (double)W / 2.0;
    const double nrm = sqrt(ip * ip + jp * jp + camera_depth * camera_depth);
    const double floor_ = 1e-6;
    const double norm = (floor_ <= nrm) ? nrm : floor_;
    h.dx = ip / norm; h.dy = jp / norm; h.dz = camera_depth / norm;
    h.idx = -1; h.t = 0.0;
    for (int k = 0; k < n; ++k) primary_candidate(prim, k, h);
    return h;
}

// ---- the same search through the spatial index ------------------------------
// Binary BVH over the finite forms (pt_bvh.hpp).  Pruning only: every candidate
// runs the arithmetic of the linear scan, and an equal t keeps the lower object
// index, so the result is the scan's.  The slab test is fp32 on boxes the host
// rounded outward and padded.
// node = (c.xy | c.z e.x | e.yz as binary16 pairs, code): centre and half-extent, pt_bvh.hpp
// The ray's side of the test, computed once per search: 1/d, clamped to +-2^100 so that a direction component of
// exactly 0 gives finite products (the slab is then open or shut by the sign of (c - o) -+ e alone).
struct SlabRay { float ix, iy, iz; };
__device__ __forceinline__ SlabRay slab_ray(float dx, float dy, float dz)
{
    SlabRay r;
    r.ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dx), -0x1p100f, 0x1p100f);
    r.iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dy), -0x1p100f, 0x1p100f);
    r.iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dz), -0x1p100f, 0x1p100f);
    return r;
}
// t at the slab's two faces = ((c -+ e) - o) / d = fma(-+e, |1/d|, (c - o) * (1/d)): the near face is the one with -e
// whatever the sign of d, so no min/max per axis.  Per axis: c - o as one v_fma_mix_f32 (the binary16 source converted
// inside the fp32 FMA), a multiplication, two more v_fma_mix_f32; then max3 + max, min3 + min, one compare: 17
// instructions where the lo / hi form took 30 (six conversions, six subtractions, six multiplications, six min/max).
// (fma(c, 1/d, -o/d) would save the three multiplications and cost three more registers per lane for o/d: the kernel
// is bound by its gathers and by registers, not by the VALU -- profiles/r03_bvh_walk.txt.)
// Rounding: three fp32 roundings per face, relative to |c - o| and |t|: in box space < 2e-7 x (|c - o| + e), which the
// host's padding (2e-5 of the scene's extent) covers while the ray's origin is within ~50 extents of the boxes.
__device__ __forceinline__ bool slab_hit(float4 nd, float ox, float oy, float oz, const SlabRay &r, float t_best)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, nd.x), b = __builtin_bit_cast(h2, nd.y), c = __builtin_bit_cast(h2, nd.z);
    // c - o as ONE v_fma_mix_f32 each (c * 1.0 - o, the binary16 centre converted inside the FMA: the same bits as a
    // conversion and a subtraction, which is what the compiler makes of the C expression -- it folds the multiplication
    // by one away and with it the instruction that converts for free)
    float sx, sy, sz;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,0]" : "=v"(sx) : "v"(nd.x), "v"(ox));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(sy) : "v"(nd.x), "v"(oy));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,0]" : "=v"(sz) : "v"(nd.y), "v"(oz));
    const float mx = sx * r.ix, my = sy * r.iy, mz = sz * r.iz;
    (void)a;
    const float ax = __builtin_fabsf(r.ix), ay = __builtin_fabsf(r.iy), az = __builtin_fabsf(r.iz);
    const float nx = __builtin_fmaf(-(float)b.y, ax, mx), fx = __builtin_fmaf((float)b.y, ax, mx);
    const float ny = __builtin_fmaf(-(float)c.x, ay, my), fy = __builtin_fmaf((float)c.x, ay, my);
    const float nz = __builtin_fmaf(-(float)c.y, az, mz), fz = __builtin_fmaf((float)c.y, az, mz);
    const float tn = fmaxf(fmaxf(fmaxf(nx, ny), 0.0f), nz);
    const float tf = fminf(fminf(fminf(fx, fy), t_best), fz);
    return tn <= tf;
}

#ifndef PT_BVH_LEAF_MAX
#define PT_BVH_LEAF_MAX 2                 // objects per leaf at most (pt_bvh.hpp builds to it; a leaf's count has 3 bits: <= 8)
#endif
constexpr int BVH_LEAF_SLOTS = (PT_BVH_LEAF_MAX + 1) & ~1;     // a leaf's records are fetched two slots at a time
constexpr int BVH_ROOT_NODE = 1;          // = BVH_ROOT of pt_bvh.hpp: node 0 is the "search over" mark

// Near-child-first walk.  `stack` is this lane's column of an LDS array laid out
// [entry][lane] (stride = lanes per workgroup): conflict-free, no scratch memory.
// `t_best()` is read at every node, so subtrees pushed earlier are pruned against
// the best hit found since; `leaf(first, cnt)` tests the objects in slots
// [first, first + cnt) of the leaf-ordered lists.
template <typename STK, typename LeafFn, typename BestFn>
__device__ __forceinline__ void bvh_walk(const float4 *__restrict__ nodes,
                                         float ox, float oy, float oz, float dx, float dy, float dz,
                                         STK *stack, int stride, LeafFn &&leaf, BestFn &&t_best)
{
    const SlabRay sr = slab_ray(dx, dy, dz);
    const uint32_t neg = (dx < 0.f ? 1u : 0u) | (dy < 0.f ? 2u : 0u) | (dz < 0.f ? 4u : 0u);
    // "while-while" (Aila & Laine, HPG'09): every lane first walks inner nodes until
    // it holds a leaf or runs out of nodes; only then do the lanes test their
    // leaves together.  A wave pays for the (long) leaf block once per round
    // instead of on every step in which any one of its 64 lanes is at a leaf.
    int sp = 0, n = BVH_ROOT_NODE;
    auto pop = [&]() -> int {
        if (sp == 0) return 0;
        sp -= 1;
        return (int)stack[sp * stride];
    };
    for (;;) {
        int code = -1;
        while (n != 0) {
            const float4 nd = nodes[n];
            if (!slab_hit(nd, ox, oy, oz, sr, t_best())) { n = pop(); continue; }
            const int c = __float_as_int(nd.w);
            if (c < 0) { code = ~c; break; }
            const int near = (c >> 5) + (int)((neg >> (c & 31)) & 1u);    // siblings are neighbours, the left one even (pt_bvh.hpp)
            stack[sp * stride] = (STK)(near ^ 1);
            sp += 1;
            n = near;
        }
        if (code < 0) break;
        leaf(code >> 3, (code & 7) + 1);                      // first slot, count (<= 4)
        n = pop();
    }
}

template <typename STK>
__device__ inline PrimaryHit primary_hit_f64_bvh(const double *__restrict__ prim, const float4 *__restrict__ nodes,
                                                 const int32_t *__restrict__ prims, int n_unbounded,
                                                 int H, int W, double camera_depth, int i, int j, STK *stack, int stride)
{
#pragma clang fp contract(off)
    PrimaryHit h;
    const double ip = (double)H / 2.0 - (double)i;
    const double jp = (double)j - (double)W / 2.0;
    const double nrm = sqrt(ip * ip + jp * jp + camera_depth * camera_depth);
    const double floor_ = 1e-6;
    const double norm = (floor_ <= nrm) ? nrm : floor_;
    h.dx = ip / norm; h.dy = jp / norm; h.dz = camera_depth / norm;
    h.idx = -1; h.t = 0.0;
    auto test = [&](int k) { primary_candidate(prim, k, h); };
    for (int u = 0; u < n_unbounded; ++u) test(prims[u]);
    // the fp64 t is compared in fp32 by the slab test: widen it by one part in 2^20
    bvh_walk(nodes, 0.f, 0.f, 0.f, (float)h.dx, (float)h.dy, (float)h.dz, stack, stride,
             [&](int first, int cnt) { for (int q = 0; q < cnt; ++q) test(prims[first + q]); },
             [&]() { return h.idx < 0 ? __builtin_inff() : (float)h.t * 1.000001f; });
    return h;
}

} // namespace pt
