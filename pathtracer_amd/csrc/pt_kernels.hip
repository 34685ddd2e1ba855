// pt_kernels.hip -- gfx950 kernels of the radiance loop.
//
//   trace_static_kernel : Lib.traceCanvas / terminalColor / closestObjectTo /
//                         bounceRay (src/Lib.hs:58-132) for n samples per pixel,
//                         scenes of <= 8 objects (object list in the kernel
//                         arguments): THE hot kernel, the one bench.py times.
//                         Large launches run as two sample slices of one grid;
//                         slice_sum_kernel adds the second slice's plane.
//                         Defined in pt_static_kernel.hpp, which pt_jit.hip
//                         compiles a second time, at run time, for one scene
//                         (pt_scene_specialise).
//   trace_bvh_kernel    : larger scenes searched through the spatial index:
//                         resumable per-lane searches, the wave alternates walk
//                         and shade steps.
//   trace_steal_kernel  : the same loop as trace_static_kernel over the scalar-
//                         streamed object list (NS = 0: PT_FLAG_NO_BVH, or too
//                         few finite forms for an index), or one whole BVH
//                         search per trip (NS = -2, kept for A/B).
//   trace_kernel        : the strict variant (PT_FLAG_NO_STEAL /
//                         NO_PRIMARY_REUSE / LDS_TILES): a pixel's samples are
//                         summed strictly in order by its own lane.
//   primary_kernel      : fp64 primary segment, the geometry parity hook.
//   rgb8_kernel         : Conversion.repa3ToImage + listToPixelRGB8
//                         (src/Conversion.hs:17-28), transposing through LDS.
//   dev_* kernels       : device-function parity hooks.
//
// Execution model of the trace kernels (one lane = one pixel, a wave = 64
// adjacent pixels of a canvas row):
//   every lane runs a SEGMENT state machine -- one closest-hit search plus one
//   scatter per loop trip -- and a lane whose path ends (miss, light, depth
//   cap) starts its pixel's next sample in the same trip.  All 64 lanes
//   therefore execute the same instruction stream on every trip no matter how
//   long their individual paths are.  Path depth divergence, which would
//   otherwise leave a wave running at the pace of its longest path, never
//   reaches the exec mask; lanes that run out of samples help the other pixels
//   of their wave (work sharing, see trace_steal_kernel).
//   The object list is indexed wave-uniformly, so for small scenes it rides
//   in SGPRs / wave-uniform VGPRs; materials are per-lane gathers from an LDS
//   copy staged once per workgroup.
#include "pt_internal.hpp"
#include "pt_static_kernel.hpp"

#include <cstdlib>
#include <type_traits>

namespace pt {


// ---------------------------------------------------------------------------
// closest-hit search (closestObjectTo, src/Lib.hs:81-98)
// ---------------------------------------------------------------------------
template <bool SPH>
__device__ __forceinline__ void test_object(float4 g0, float4 g1, float4 g2, const float4 *__restrict__ rect, int k, f3 o, f3 v,
                                            int last, int &best, float &best_t)
{
    if constexpr (SPH) {
        if (g1.w < 0.0f) {                      // wave-uniform: Sphere (extension); NaN = no hit fails every compare
            const float ts = sphere_t(mk3(g2.x, g2.y, g2.z), -g1.w, o, v);
            const bool oks = (ts > 0.0f) & (ts < best_t) & (k != last);
            best = oks ? k : best;
            best_t = oks ? ts : best_t;
            return;
        }
    }
    // t = (A - o.n_raw) / (v.n_hat); every object operand is wave-uniform (SGPR)
    const float num = fmaf(-o.z, g0.z, fmaf(-o.y, g0.y, fmaf(-o.x, g0.x, g0.w)));   // A - o.n_raw, the dot folded into A
    const float den = fmaf(v.z, g1.z, fmaf(v.y, g1.y, v.x * g1.x));
    const float t = num * __builtin_amdgcn_rcpf(den);
    // 0 < t < Inf (src/Object.hs:159), not the last-struck object
    // (src/Lib.hs:86-90), strictly closer than the best so far: an equal t
    // keeps the earlier object, as Data.Vector.minimumBy does.  Plain '&' so
    // the lanes stay predicated instead of branching.
    bool ok = (t > 0.0f) & (t < best_t) & (k != last);
    if (g1.w != __builtin_inff()) {          // wave-uniform: Disk (src/Object.hs:165-171)
        const float dx = fmaf(t, v.x, o.x) - g2.x;
        const float dy = fmaf(t, v.y, o.y) - g2.y;
        const float dz = fmaf(t, v.z, o.z) - g2.z;
        ok = ok & inside_form<SPH>(rect, k, g1.w, dx, dy, dz);
    }
    best = ok ? k : best;
    best_t = ok ? t : best_t;
}

// Streamed object lists (more than 8 objects) run thousands of tests per
// segment, and there the lane-mask bookkeeping of test_object (s_and/s_or on
// SGPR pairs, ~20 SALU per object) saturates the CU's single scalar unit before
// the four SIMDs are busy.  test_object_key keeps every decision in VGPRs:
// the candidate is the ordered integer key bits(t) - 1, so that
//   t <= -0, NaN, +Inf  ->  key >= 0x7f7fffff  (never accepted),
//   t == +0             ->  key == 0xffffffff  (0 < t is strict, src/Object.hs:159),
// and "closer" is one unsigned compare; a strict '<' keeps the earlier object
// on ties as minimumBy does.
constexpr uint32_t KEY_INVALID = 0xffffffffu;
constexpr uint32_t KEY_LIMIT   = 0x7f7fffffu;   // bits(+Inf) - 1: keys below are finite positive t

template <bool SPH>
__device__ __forceinline__ void test_object_key(float4 g0, float4 g1, float4 g2, const float4 *__restrict__ rect, int k, f3 o, f3 v,
                                                int last, int &best, uint32_t &best_key)
{
    if constexpr (SPH) {
        if (g1.w < 0.0f) {                      // wave-uniform: Sphere (extension)
            uint32_t keys = __float_as_uint(sphere_t(mk3(g2.x, g2.y, g2.z), -g1.w, o, v)) - 1u;
            keys = (k == last) ? KEY_INVALID : keys;
            const bool b = keys < best_key;
            best = b ? k : best;
            best_key = b ? keys : best_key;
            return;
        }
    }
    const float num = fmaf(-o.z, g0.z, fmaf(-o.y, g0.y, fmaf(-o.x, g0.x, g0.w)));   // A - o.n_raw, the dot folded into A
    const float den = fmaf(v.z, g1.z, fmaf(v.y, g1.y, v.x * g1.x));
    const float t = num * __builtin_amdgcn_rcpf(den);
    uint32_t key = __float_as_uint(t) - 1u;
    key = (k == last) ? KEY_INVALID : key;                // _lastStruck (src/Lib.hs:86-90)
    if (g1.w != __builtin_inff()) {                       // wave-uniform: Disk (src/Object.hs:165-171)
        const float dx = fmaf(t, v.x, o.x) - g2.x;
        const float dy = fmaf(t, v.y, o.y) - g2.y;
        const float dz = fmaf(t, v.z, o.z) - g2.z;
        key = inside_form<SPH>(rect, k, g1.w, dx, dy, dz) ? key : KEY_INVALID;
    }
    const bool better = key < best_key;
    best = better ? k : best;
    best_key = better ? key : best_key;
}

__device__ __forceinline__ void key_result(int &best, uint32_t best_key, float &best_t)
{
    const bool hit = best_key < KEY_LIMIT;
    best = hit ? best : -1;
    best_t = __uint_as_float(best_key + 1u);
}


// LDS tile of geometry records for the tiled kernel (NS == -1), dynamic size.
extern __shared__ float4 s_tile[];

template <int NS, bool SPH>
__device__ __forceinline__ void closest(const TraceArgs &a, const StaticGeom<NS> &sg, f3 o, f3 v,
                                        int last, int &best, float &best_t, int *stack = nullptr, int stride = 0)
{
    best = -1; best_t = __builtin_inff();
    uint32_t best_key = KEY_LIMIT;
    if constexpr (NS == -2) {
        // Spatial index (pt_bvh.hpp): the unbounded forms first, then a
        // near-child-first walk of the BVH.  Each lane follows its own chain of
        // node indices (32-byte gathers from L2, far children on an LDS stack);
        // candidates run test_object_key's arithmetic, and an equal key keeps the
        // lower object index, so the winner is the linear scan's.
        // a.bvh_geom: the geometry records again, in leaf order (slot s of
        // bvh_prims), with the object index in g2.w -- a leaf's records are
        // contiguous, so all their loads are issued at once (one L2 round trip
        // per leaf instead of two dependent ones per object).
        const float4 *__restrict__ g = a.bvh_geom;
        float bt = __builtin_inff();
        auto test = [&](float4 g0, float4 g1, float4 g2, bool on) {
            const int k = __float_as_int(g2.w);
            const float num = fmaf(-o.z, g0.z, fmaf(-o.y, g0.y, fmaf(-o.x, g0.x, g0.w)));
            const float den = fmaf(v.z, g1.z, fmaf(v.y, g1.y, v.x * g1.x));
            float t = num * __builtin_amdgcn_rcpf(den);
            const float dx = fmaf(t, v.x, o.x) - g2.x;         // Disk (src/Object.hs:165-171); a plane has r2 = +Inf
            const float dy = fmaf(t, v.y, o.y) - g2.y;
            const float dz = fmaf(t, v.z, o.z) - g2.z;
            bool inside = inside_form<SPH>(a.rect, k, g1.w, dx, dy, dz) || (g1.w == __builtin_inff());
            if constexpr (SPH) {                               // per lane here: both forms computed, one selected
                const float ts = sphere_t(mk3(g2.x, g2.y, g2.z), -g1.w, o, v);
                t = (g1.w < 0.0f) ? ts : t;
                inside = inside || (g1.w < 0.0f);
            }
            uint32_t key = __float_as_uint(t) - 1u;
            key = (k == last || !on || !inside) ? KEY_INVALID : key;     // _lastStruck (src/Lib.hs:86-90)
            const bool better = (key < best_key) | ((key == best_key) & (key < KEY_LIMIT) & (k < best));
            best = better ? k : best;
            best_key = better ? key : best_key;
            bt = better ? t : bt;
        };
        for (int u = 0; u < a.n_unbounded; ++u) test(g[3 * u], g[3 * u + 1], g[3 * u + 2], true);
        bvh_walk(a.bvh_nodes, o.x, o.y, o.z, v.x, v.y, v.z, stack, stride,
                 [&](int first, int cnt) {
#pragma unroll 1
                     for (int h = 0; h < BVH_LEAF_SLOTS; h += 2) {
                         float4 r[2][3];
#pragma unroll
                         for (int q = 0; q < 2; ++q) {
                             r[q][0] = g[3 * (first + h + q)]; r[q][1] = g[3 * (first + h + q) + 1]; r[q][2] = g[3 * (first + h + q) + 2];
                         }
#pragma unroll
                         for (int q = 0; q < 2; ++q) test(r[q][0], r[q][1], r[q][2], h + q < cnt);
                         if (h + 2 >= cnt) break;
                     }
                 },
                 [&]() { return bt; });
        key_result(best, best_key, best_t);
        return;
    }
    if constexpr (NS == -1) {
        // Object list staged in LDS per workgroup.  A scene that fits one tile
        // was staged once at kernel start and the waves run free; a larger one
        // streams tile after tile from HBM/L2 with the whole workgroup in
        // lockstep (the caller's loop is block-uniform in that case).  Every
        // lane reads the SAME record: an LDS broadcast, conflict-free.
        const int n = a.n_obj, T = a.tile_objs;
        for (int base = 0; base < n; base += T) {
            const int cnt = min(T, n - base);
            if (n > T) {
                __syncthreads();                            // everyone is done with the previous tile
                for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) s_tile[i] = a.geom[(size_t)base * 3 + i];
                __syncthreads();
            }
#pragma unroll 2
            for (int k = 0; k < cnt; ++k)
                test_object_key<SPH>(s_tile[3 * k + 0], s_tile[3 * k + 1], s_tile[3 * k + 2], a.rect, base + k, o, v, last, best,
                                best_key);
        }
        key_result(best, best_key, best_t);
        return;
    }
    if constexpr (NS > 0) {
        // the host pads the list to NS with objects that can never be hit
#pragma unroll
        for (int k = 0; k < NS; ++k)
            test_object<SPH>(sg.g[k][0], sg.g[k][1], sg.g[k][2], a.rect, k, o, v, last, best, best_t);
    } else {
        // Large scenes: the records stream through the scalar cache from L2/HBM.
        // Scalar loads return out of order, so the only safe wait is
        // lgkmcnt(0); the loop is therefore double-buffered by hand -- the
        // loads of objects k+2, k+3 are issued right after the wait that
        // retires k, k+1 and land while the VALU works on k, k+1.
        // (the host pads the list with 4 never-hit records, so the prefetch
        // needs no bounds logic: one pointer bump per pair)
        const float4 *__restrict__ g = a.geom;
        const int n = a.n_obj;
        float4 A0[3], A1[3], B0[3], B1[3];
        auto load = [&](float4 (&r)[3], const float4 *q) { r[0] = q[0]; r[1] = q[1]; r[2] = q[2]; };
        load(A0, g); load(A1, g + 3);
        for (int k = 0; k < n; k += 4) {
            load(B0, g + 6); load(B1, g + 9);
            __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ahead of the arithmetic
            test_object_key<SPH>(A0[0], A0[1], A0[2], a.rect, k, o, v, last, best, best_key);
            test_object_key<SPH>(A1[0], A1[1], A1[2], a.rect, k + 1, o, v, last, best, best_key);
            g += 12;
            load(A0, g); load(A1, g + 3);
            __builtin_amdgcn_sched_barrier(0);
            test_object_key<SPH>(B0[0], B0[1], B0[2], a.rect, k + 2, o, v, last, best, best_key);
            test_object_key<SPH>(B1[0], B1[1], B1[2], a.rect, k + 3, o, v, last, best, best_key);
        }
        key_result(best, best_key, best_t);
    }
}

// ---------------------------------------------------------------------------
// trace kernel
// ---------------------------------------------------------------------------
constexpr int TRACE_BLOCK = 256;

constexpr int TILED_BLOCK = 512;
constexpr int TILE_OBJECTS = 1024;     // 48 KB of LDS per workgroup: 3 workgroups (24 waves) per CU

template <int NS, bool REUSE, bool SPEC>
__global__ __launch_bounds__(NS == -1 ? TILED_BLOCK : TRACE_BLOCK) void trace_kernel(TraceArgs a, StaticGeom<NS> sg)
{
    constexpr int BLOCK = NS == -1 ? TILED_BLOCK : TRACE_BLOCK;
    // BVH walk: per-lane stack [entry][lane] in dynamic LDS, sized by the launcher from the tree's depth
    int *const stack = NS == -2 ? reinterpret_cast<int *>(s_tile) + threadIdx.x : nullptr;
    bool lockstep = false;
    if constexpr (NS == -1) {
        lockstep = a.n_obj > a.tile_objs;
        if (!lockstep) {                                    // the whole scene fits one tile: stage it once
            for (int i = threadIdx.x; i < a.n_obj * 3; i += BLOCK) s_tile[i] = a.geom[i];
            __syncthreads();
        }
    }
    // Materials: LDS copy for small scenes, global/L2 gather otherwise.  The
    // gather index differs per lane; records are spaced MAT_LDS_STRIDE = 5
    // float4 (80 B) apart so that the same field of up to 8 different objects
    // falls on 8 different 4-bank groups -- a 64-B spacing puts objects k and
    // k+2 on the same banks (measured: 41 % of LDS cycles were conflicts).
    constexpr int MAT_LDS_STRIDE = 5;
    __shared__ float4 s_mat[(NS > 0 ? NS : 1) * MAT_LDS_STRIDE];
    if constexpr (NS > 0) {
        if ((int)threadIdx.x < min(a.n_obj, NS) * 4)
            s_mat[(threadIdx.x >> 2) * MAT_LDS_STRIDE + (threadIdx.x & 3)] = a.mat[threadIdx.x];
        __syncthreads();
    }
    auto mat = [&](int k, int f) -> float4 {
        if constexpr (NS > 0) return s_mat[__mul24(k, MAT_LDS_STRIDE) + f];
        else                  return a.mat[k * 4 + f];
    };
    // The 9 matrix floats of a diffuse bounce.  (An inline-asm 2 x ds_read_b128 +
    // ds_read_b32 form halves the LDS cycles but needs its own lgkmcnt wait,
    // which measured 4 % slower than letting hipcc schedule these reads.)
    auto mat_rot = [&](int k, float4 &m1, float4 &m2, float &m22) {
        m1 = mat(k, 1); m2 = mat(k, 2); m22 = mat(k, 3).x;
    };

    const int npix = a.height * a.width;
    signal_gate(a);
    const int pixel = blockIdx.x * BLOCK + threadIdx.x;
    const bool in_canvas = pixel < npix;
    const int pc = in_canvas ? pixel : 0;
    const int pi = pc / a.width, pj = pc - pi * a.width;

    unsigned long long segs = 0;
    int n_my = in_canvas ? a.n_samples : 0;

    // traceCanvas adds onto the running sum (src/Lib.hs:65); continuing the
    // fp32 sum sample by sample makes k launches of 1 equal 1 launch of k.
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (in_canvas && !a.fresh) acc = mk3(a.canvas[3 * (size_t)pc + 0], a.canvas[3 * (size_t)pc + 1], a.canvas[3 * (size_t)pc + 2]);

    // Primary segment: no jitter, no randomness (src/Lib.hs:41-51), so it is
    // computed once per pixel, in fp64, reference operation order.
    const PrimaryHit ph = primary_hit(a, pi, pj, stack, BLOCK);
    const f3 d0 = mk3((float)ph.dx, (float)ph.dy, (float)ph.dz);
    const int B = a.max_bounces;

    f3 P0 = mk3(0.f, 0.f, 0.f), color0 = mk3(1.f, 1.f, 1.f);
    int idx0 = -1; float e0 = 0.f;
    if constexpr (REUSE) {
        if (in_canvas) segs += 1;
        idx0 = ph.idx;
        if (idx0 < 0) {
            n_my = 0;                                    // every sample is black (src/Lib.hs:72)
        } else {
            const float4 m0 = mat(idx0, 0);
            e0 = m0.w; color0 = mk3(m0.x, m0.y, m0.z);
            P0 = mk3((float)(ph.t * ph.dx), (float)(ph.t * ph.dy), (float)(ph.t * ph.dz));
            if (e0 > 0.f) {                              // primary ray sees a light (src/Lib.hs:74)
                for (int s = 0; s < n_my; ++s) { acc.x += e0; acc.y += e0; acc.z += e0; }
                n_my = 0;
            } else if (B < 2) {
                n_my = 0;                                // terminalColor 0 = black (src/Lib.hs:68)
            }
        }
    }

    // per-lane path state
    f3 o = mk3(0.f, 0.f, 0.f), d = d0, thr = mk3(1.f, 1.f, 1.f);
    int last = -1, b = 0, s = 0;
    uint32_t gs = a.sample_begin;

    auto scatter = [&](int idx, float e, f3 din, int bounce, f3 at) -> f3 {
        uint32_t r0, r1;
        philox2x32_10((uint32_t)pc, (gs << 6) | (uint32_t)bounce, a.seed, r0, r1);
        if constexpr (SPEC) {
            const float4 m3 = mat(idx, 3);
            f3 nrm = mk3(m3.y, m3.z, m3.w);
            const bool sphere = m3.x != m3.x;              // NaN marks the Sphere extension: normal at the hit point
            if (sphere) {
                const float4 m1 = mat(idx, 1);
                nrm = mk3((at.x - m1.x) * m1.w, (at.y - m1.y) * m1.w, (at.z - m1.z) * m1.w);
            }
            if (e < 0.f)                                   // _reflective (src/Lib.hs:110)
                return scatter_specular(din, nrm, u_theta(r0), u_phi(r1), NOISE_REV);
            if (sphere) return scatter_diffuse_about(nrm, r0, r1);
        }
        float4 m1, m2; float m22;
        mat_rot(idx, m1, m2, m22);
        return scatter_diffuse_bits(m1, m2, m22, r0, r1);                          // src/Lib.hs:111
    };

    if constexpr (REUSE) {
        if (n_my > 0) {
            d = scatter(idx0, e0, d0, 0, P0);
            o = P0; last = idx0; thr = color0; b = 1;
        }
    }

    for (;;) {
        const bool alive = s < n_my;
        if constexpr (NS == -1) {
            // streamed tiles need every wave of the workgroup at the barriers
            if (lockstep ? !__syncthreads_or(alive) : !alive) break;
        } else {
            if (!alive) break;
        }
        int k; float t;
        closest<NS, SPEC>(a, sg, o, d, last, k, t, stack, BLOCK);
        segs += alive ? 1 : 0;
        const bool hit = k >= 0;
        const float4 m0 = mat(hit ? k : 0, 0);
        const bool emissive = hit && (m0.w > 0.f) && alive;
        // after this search b+1 have run; terminalColor 0 ends the path black
        const bool terminal = !hit || emissive || (b + 1 >= B);
        if (emissive) {                                    // emittance * pixel (src/Lib.hs:74)
            acc.x = fmaf(m0.w, thr.x, acc.x);
            acc.y = fmaf(m0.w, thr.y, acc.y);
            acc.z = fmaf(m0.w, thr.z, acc.z);
        }
        if constexpr (REUSE) {
            // a finished path restarts from the pixel's primary hit
            if (terminal) { s += 1; gs += a.sample_stride; }
            const int   idx = terminal ? idx0 : k;
            const float e   = terminal ? e0 : m0.w;
            const f3    din = terminal ? d0 : d;
            const int   bn  = terminal ? 0 : b;
            const f3 hp = mk3(terminal ? P0.x : fmaf(t, d.x, o.x),     // march (src/Object.hs:151-153)
                              terminal ? P0.y : fmaf(t, d.y, o.y),
                              terminal ? P0.z : fmaf(t, d.z, o.z));
            thr = mk3(terminal ? color0.x : thr.x * m0.x,               // pixel * color (src/Lib.hs:79)
                      terminal ? color0.y : thr.y * m0.y,
                      terminal ? color0.z : thr.z * m0.z);
            d = scatter(idx, e, din, bn, hp);
            o = hp; last = idx; b = bn + 1;
        } else {
            if (terminal) {
                s += 1; gs += a.sample_stride;
                o = mk3(0.f, 0.f, 0.f); d = d0; thr = mk3(1.f, 1.f, 1.f); last = -1; b = 0;
            } else {
                const f3 hp = fma3(t, d, o);
                thr = mk3(thr.x * m0.x, thr.y * m0.y, thr.z * m0.z);
                d = scatter(k, m0.w, d, b, hp);
                o = hp; last = k; b += 1;
            }
        }
    }

    if (in_canvas) {
        a.canvas[3 * (size_t)pc + 0] = acc.x; a.canvas[3 * (size_t)pc + 1] = acc.y; a.canvas[3 * (size_t)pc + 2] = acc.z;
    }

    // one 64-bit atomic per wave for the segment counter
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) segs += __shfl_down(segs, off, 64);
    if ((threadIdx.x & 63) == 0 && segs) atomicAdd(seg_lane(a.segments), segs);
}

// ---------------------------------------------------------------------------
// trace kernel with wave-level work sharing (the default)
// ---------------------------------------------------------------------------
// Phase 1 is trace_kernel's loop unchanged: every lane works through its own
// pixel's samples in order, no bookkeeping.  It ends, wave-uniformly, the moment
// the first lane of the wave runs out of samples (or at once if some lanes never
// had any: primary ray missed or saw a light).  Phase 2 shares what is left:
// every pixel of the wave gets a counter of samples not yet started (LDS).  A
// lane whose path ends draws the next sample of the pixel it is working on with
// one LDS atomic; lanes without a path ballot, pick (wave-uniformly, s_ff1 on the
// ballot of pixels that still have samples) a pixel to help, draw from ITS
// counter and load its primary-hit constants from an LDS table.  The wave thus
// keeps all 64 lanes on live paths until its pool of samples is dry instead of
// until each lane's own quota is.  Radiance gathered for a foreign pixel goes to
// a per-pixel LDS accumulator (ds_add_f32) that is added to the canvas at the
// end.  Philox is keyed by (pixel, global sample, bounce), so WHO traces a
// sample does not change it; only the fp32 summation order of a pixel's samples
// can change (PT_FLAG_NO_STEAL keeps the strictly sequential sum).
template <int NS, bool SPEC>
__global__ __launch_bounds__(NS == -2 ? 64 : TRACE_BLOCK) void trace_steal_kernel(TraceArgs a, StaticGeom<NS> sg,
                                                                  const float4 *__restrict__ geom_ro)
{
    // geom_ro is the object list again, as a noalias read-only parameter: with LDS
    // atomics in the loop hipcc otherwise stops proving the list unclobbered and
    // turns the wave-uniform s_load stream into per-lane vector loads.
    a.geom = geom_ro;
    static_assert(NS == 0 || NS == -2, "scalar-stream and BVH object lists; small scenes use trace_static_kernel");
    constexpr int BLOCK = NS == -2 ? 64 : TRACE_BLOCK;      // BVH walks diverge: one wave per workgroup
    constexpr int MAT_LDS_STRIDE = 5;
    // BVH walk: per-lane stack [entry][lane] in dynamic LDS, sized by the launcher from the tree's depth
    int *const stack = NS == -2 ? reinterpret_cast<int *>(s_tile) + threadIdx.x : nullptr;
    __shared__ float4   s_mat[(NS > 0 ? NS : 1) * MAT_LDS_STRIDE];
    __shared__ uint32_t s_next[BLOCK];          // per pixel: next sample (launch-local index) not started
    __shared__ float4   s_pix[BLOCK * 3];       // per pixel: P0, d0, colour0, idx0, e0
    __shared__ float    s_acc[BLOCK * 3];       // per pixel: radiance gathered in phase 2
    if constexpr (NS > 0) {
        if ((int)threadIdx.x < min(a.n_obj, NS) * 4)
            s_mat[(threadIdx.x >> 2) * MAT_LDS_STRIDE + (threadIdx.x & 3)] = a.mat[threadIdx.x];
    }
    auto mat = [&](int k, int f) -> float4 {
        if constexpr (NS > 0) return s_mat[__mul24(k, MAT_LDS_STRIDE) + f];
        else                  return a.mat[k * 4 + f];
    };

    const int tid = threadIdx.x;
    const int wbase = tid & ~63;
    const int npix = a.height * a.width;
    signal_gate(a);
    const int pixel0 = blockIdx.x * BLOCK;
    const bool in_canvas = pixel0 + tid < npix;
    const int pc = in_canvas ? pixel0 + tid : 0;
    const int pi = pc / a.width, pj = pc - pi * a.width;
    const uint32_t n = (uint32_t)a.n_samples;
    const int B = a.max_bounces;

    unsigned long long segs = in_canvas ? 1 : 0;          // the fp64 primary search below
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (in_canvas && !a.fresh) acc = mk3(a.canvas[3 * (size_t)pc + 0], a.canvas[3 * (size_t)pc + 1], a.canvas[3 * (size_t)pc + 2]);

    // primary segment, once per pixel, fp64 (src/Lib.hs:41-51, :81-98)
    const PrimaryHit ph = primary_hit(a, pi, pj, stack, BLOCK);
    f3 cd0 = mk3((float)ph.dx, (float)ph.dy, (float)ph.dz);
    f3 cP0 = mk3((float)(ph.t * ph.dx), (float)(ph.t * ph.dy), (float)(ph.t * ph.dz));
    f3 ccol = mk3(1.f, 1.f, 1.f);
    int cidx = ph.idx; float ce = 0.f;
    bool traced = in_canvas && cidx >= 0 && B >= 2;       // else every sample is black (src/Lib.hs:68,72)
    __syncthreads();                                       // s_mat is staged
    if (in_canvas && cidx >= 0) {
        const float4 m0 = mat(cidx, 0);
        ce = m0.w; ccol = mk3(m0.x, m0.y, m0.z);
        if (ce > 0.f) {                                    // the primary ray sees a light (src/Lib.hs:74)
            for (uint32_t s = 0; s < n; ++s) { acc.x += ce; acc.y += ce; acc.z += ce; }
            traced = false;
        }
    }
    s_pix[tid * 3 + 0] = make_float4(cP0.x, cP0.y, cP0.z, cd0.x);
    s_pix[tid * 3 + 1] = make_float4(cd0.y, cd0.z, ccol.x, ccol.y);
    s_pix[tid * 3 + 2] = make_float4(ccol.z, __int_as_float(cidx), ce, 0.f);

    uint64_t pre = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)pc;
    f3 o = cP0, d = cd0, thr = ccol;
    int last = cidx, b = 1;
    uint32_t sloc = 0;                                     // launch-local index of the sample in flight
    uint32_t gs = a.sample_begin;
    bool alive = traced;                                   // has a path in flight

    const PhiloxRegs pregs = philox_regs(a.seed);
    auto scatter = [&](int idx, float e, f3 din, int bounce, f3 at) -> f3 {
        uint32_t r0, r1;
        philox2x32_10_pre_v((uint32_t)(pre >> 32), (uint32_t)pre, (gs << 6) | (uint32_t)bounce, pregs, r0, r1);
        if constexpr (SPEC) {
            const float4 m3 = mat(idx, 3);
            f3 nrm = mk3(m3.y, m3.z, m3.w);
            const bool sphere = m3.x != m3.x;              // NaN marks the Sphere extension: normal at the hit point
            if (sphere) {
                const float4 m1 = mat(idx, 1);
                nrm = mk3((at.x - m1.x) * m1.w, (at.y - m1.y) * m1.w, (at.z - m1.z) * m1.w);
            }
            if (e < 0.f)                                   // _reflective (src/Lib.hs:110)
                return scatter_specular(din, nrm, u_theta(r0), u_phi(r1), NOISE_REV);
            if (sphere) return scatter_diffuse_about(nrm, r0, r1);
        }
        return scatter_diffuse_bits(mat(idx, 1), mat(idx, 2), mat(idx, 3).x, r0, r1);   // src/Lib.hs:111
    };
    // One segment: closest-hit search, radiance of a finished path, then `draw`
    // decides what a finished path does next, then restart or continue.
    // `all_alive` (compile-time): phase 1 only runs while every lane has a path.
    auto trip = [&](auto all_alive, auto &&draw) {
        int k; float t;
        closest<NS, SPEC>(a, sg, o, d, last, k, t, stack, BLOCK);
        const bool live = decltype(all_alive)::value ? true : alive;
        if (!decltype(all_alive)::value) segs += live ? 1 : 0;      // phase 1 counts its trips wave-uniformly
        const bool hit = k >= 0;
        const float4 m0 = mat(hit ? k : 0, 0);
        bool terminal = live && (!hit || (m0.w > 0.f) || (b + 1 >= B));
        {                                                  // emittance * pixel (src/Lib.hs:74)
            const float E = (live && hit) ? fmaxf(m0.w, 0.f) : 0.f;   // e <= 0: not a light
            acc.x = fmaf(E, thr.x, acc.x);
            acc.y = fmaf(E, thr.y, acc.y);
            acc.z = fmaf(E, thr.z, acc.z);
        }
        draw(terminal);                                    // may clear or set `terminal` (= restart from cP0)
        int idx = terminal ? cidx : k;
        if (!decltype(all_alive)::value) idx = max(idx, 0);   // a pathless lane's garbage index stays in range
        const float e   = terminal ? ce : m0.w;
        const f3    din = terminal ? cd0 : d;
        const int   bn  = terminal ? 0 : b;
        const f3 hp = mk3(terminal ? cP0.x : fmaf(t, d.x, o.x),         // march (src/Object.hs:151-153)
                          terminal ? cP0.y : fmaf(t, d.y, o.y),
                          terminal ? cP0.z : fmaf(t, d.z, o.z));
        thr = mk3(terminal ? ccol.x : thr.x * m0.x,                       // pixel * color (src/Lib.hs:79)
                  terminal ? ccol.y : thr.y * m0.y,
                  terminal ? ccol.z : thr.z * m0.z);
        d = scatter(idx, e, din, bn, hp);
        o = hp; last = idx; b = bn + 1;
    };

    if (alive) d = scatter(cidx, ce, cd0, 0, cP0);

    // ---- phase 1: every lane on its own pixel --------------------------------
    const unsigned long long everyone = __ballot(true);
    // (bottom-tested on purpose: a top-tested form makes hipcc copy ~25 state
    // registers per trip to keep the pre-trip values alive for the exit path)
    if (__ballot(alive) == everyone) {
        uint32_t trips = 0;                                // wave-uniform (SALU)
        do {
            trip(std::true_type{}, [&](bool &terminal) {
                if (terminal) { sloc += 1; gs += a.sample_stride; }
            });
            alive = sloc < n;
            trips += 1;
        } while (__ballot(alive) == everyone);
        segs += trips;
    }
    // A lane that has just left phase 1 restarted a path that does not exist
    // (sloc == n); it enters phase 2 pathless.

    // ---- phase 2: the wave shares what is left -------------------------------
    s_next[tid] = alive ? sloc + 1 : n;
    s_acc[tid * 3 + 0] = 0.f; s_acc[tid * 3 + 1] = 0.f; s_acc[tid * 3 + 2] = 0.f;
    __syncthreads();
    int cur = tid;                                         // the pixel (workgroup-relative) this lane works on
    bool dry = false;                                      // wave-uniform: no pixel has an unstarted sample
    if (__ballot(alive) != 0ull || !dry) do {
        trip(std::false_type{}, [&](bool &terminal) {
            uint32_t sn = 0;
            if (terminal) sn = atomicAdd(&s_next[cur], 1u);          // next sample of my current pixel
            const bool want = terminal || !alive;          // lanes that need a (new) path
            bool needy = want && !(terminal && sn < n);
            while (!dry && __ballot(needy) != 0ull) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                const bool has = __hip_atomic_load(&s_next[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n;
                const unsigned long long avail = __ballot(has);      // pixels of this wave with unstarted samples
                if (avail == 0ull) { dry = true; break; }
                const int v = wbase + (__ffsll((long long)avail) - 1);   // wave-uniform choice of pixel to help
                if (needy) {
                    sn = atomicAdd(&s_next[v], 1u);
                    if (sn < n) {
                        if (v != cur) {                    // hand over what was gathered, adopt pixel v
                            atomicAdd(&s_acc[cur * 3 + 0], acc.x);
                            atomicAdd(&s_acc[cur * 3 + 1], acc.y);
                            atomicAdd(&s_acc[cur * 3 + 2], acc.z);
                            acc = mk3(0.f, 0.f, 0.f);
                            cur = v;
                            const float4 p0 = s_pix[v * 3 + 0], p1 = s_pix[v * 3 + 1], p2 = s_pix[v * 3 + 2];
                            cP0 = mk3(p0.x, p0.y, p0.z); cd0 = mk3(p0.w, p1.x, p1.y); ccol = mk3(p1.z, p1.w, p2.x);
                            cidx = __float_as_int(p2.y); ce = p2.z;
                            pre = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)(pixel0 + v);
                        }
                        needy = false;
                    }
                }
            }
            alive = !want || !needy;                       // path continues, or a sample was drawn
            terminal = want && !needy;                     // restart from the (current) pixel's primary hit
            if (terminal) gs = a.sample_begin + sn * a.sample_stride;
        });
    } while (__ballot(alive) != 0ull || !dry);

    atomicAdd(&s_acc[cur * 3 + 0], acc.x);
    atomicAdd(&s_acc[cur * 3 + 1], acc.y);
    atomicAdd(&s_acc[cur * 3 + 2], acc.z);
    __syncthreads();
    if (in_canvas) {
        a.canvas[3 * (size_t)pc + 0] = s_acc[tid * 3 + 0];
        a.canvas[3 * (size_t)pc + 1] = s_acc[tid * 3 + 1];
        a.canvas[3 * (size_t)pc + 2] = s_acc[tid * 3 + 2];
    }

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) segs += __shfl_down(segs, off, 64);
    if ((tid & 63) == 0 && segs) atomicAdd(seg_lane(a.segments), segs);
}

// ---------------------------------------------------------------------------
// trace kernel for scenes searched through the spatial index (pt_bvh.hpp).
// Walks of different rays differ widely in length (10,000 disks: 51 node visits
// on average, 116 for the slowest lane of a wave), and a loop that runs one
// whole closest-hit search per trip (trace_steal_kernel<-2>) keeps every lane
// waiting for the slowest: 22 % of the lanes were active.  Here a lane's search
// is a resumable state (node, stack depth, best hit so far) and the wave
// alternates between two steps:
//   walk  : every lane that is searching visits up to PT_BVH_ROUND_VISITS nodes and
//           at most one leaf;
//   shade : the lanes whose search has finished (as soon as PT_BVH_SHADE_LANES of
//           them wait, or nobody is searching) take the radiance, bounce or
//           restart -- drawing the next sample from the wave's pool exactly as
//           trace_steal_kernel's sharing phase does -- and start their next
//           search at once.
// The arithmetic of a search and of a bounce is that of the other kernels, so the
// paths, the hits and pt_segments are the same; only WHEN a lane does what
// changes (and with it the fp32 summation order of a pixel's samples).
// ---------------------------------------------------------------------------
// One wave per workgroup.  (Round 2 could keep the top of the tree in LDS, shared by the waves of a larger workgroup:
// measured 12-50 % slower with 448-960 nodes, +-3 % with 64-128, profiles/r02_bvh_roofline.txt; removed in round 3 --
// the choice between an LDS and a global address made every node fetch a flat load.)
// Round 3 rewrote the walk step around its VALU issue cost (the kernel issues 0.27 wave-instructions per cycle per
// SIMD against the default scene's 0.34: the VALU, not the gathers, is what its waves queue for): the box test as
// nine v_fma_mix_f32 on centre / half-extent nodes, one 16-byte global load per node through a scalar base and a
// 32-bit offset, a stack pointer instead of a depth, index 0 as the "search over" mark at the bottom of the stack.
constexpr int BVH_BLOCK = 64;
// Registers are allocated for SIX waves per SIMD, stated: the stacks and per-pixel records in LDS allow 25 waves per
// CU, so 80 VGPRs is the budget that matters.  Left to itself the compiler's choice moved with unrelated edits -- 72
// VGPRs (a seventh wave that LDS does not admit, and 7 % slower) for one variant and 84-86 (five waves) for another
// after the sample slices went in; with the target stated: 18.3 / 23.4 / 69.8 ms against 19.8 / 25.2 / 76.5 without and
// 19.2 / 24.4 / 71.3 with (7,7) (profiles/r03_bvh_walk.txt 9).
#ifndef PT_BVH_WAVES_ATTR
#define PT_BVH_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#endif
#ifndef PT_BVH_ROUND_VISITS
#define PT_BVH_ROUND_VISITS 8
#endif
#ifndef PT_BVH_SHADE_LANES
#define PT_BVH_SHADE_LANES 16
#endif

// REC32: the compact object records (TraceArgs::bvh_geom32) -- two 16-byte gathers per record instead of three; an
// object is then named by its position in bvh_prims, and materials / rectangle sides come in that order.
template <bool SPEC, typename STK, bool REC32>
__global__ __launch_bounds__(BVH_BLOCK) PT_BVH_WAVES_ATTR void trace_bvh_kernel(TraceArgs a)
{
    constexpr int BLOCK = BVH_BLOCK;
    // dynamic LDS: the stacks, (bvh_depth + 2) entries x BLOCK lanes x STK, laid out [entry][lane]; entry 0 holds
    // node index 0, the "search over" mark, for good
    // (STK = 16 bits while the tree has fewer than 65,536 nodes: the stacks are what limits the waves per CU)
    STK *const stack = reinterpret_cast<STK *>(s_tile) + threadIdx.x;
    __shared__ uint32_t s_next[BLOCK];          // per pixel: next sample (launch-local index) not started
    __shared__ float4   s_pix[BLOCK * 3];       // per pixel: P0, d0, colour0, idx0, e0
    __shared__ float    s_acc[BLOCK * 3];       // per pixel: radiance gathered for it by other lanes
    const float4 *__restrict__ const mats = REC32 ? a.bvh_mat : a.mat;
    const float4 *__restrict__ const rects = REC32 ? a.bvh_rect : a.rect;
    auto mat = [&](int k, int f) -> float4 { return mats[k * 4 + f]; };

    const int tid = threadIdx.x;
    const int wbase = tid & ~63;                           // this wave's first slot in the per-pixel arrays
    const int npix = a.height * a.width;
    signal_gate(a);
    const int pixel0 = blockIdx.x * BLOCK;
    const bool in_canvas = pixel0 + tid < npix;
    const int pc = in_canvas ? pixel0 + tid : 0;
    const int pi = pc / a.width, pj = pc - pi * a.width;
    // this workgroup's slice of the launch's samples (TraceArgs::slice_*, as in trace_static_kernel: a second, short
    // slice dispatched behind the first leaves only short waves for the end of a long launch); slices beyond the first
    // start their sums at 0 in a plane of their own, which slice_sum_kernel adds to the canvas
    const int slice = blockIdx.y;
    const uint32_t n = (uint32_t)(a.n_slices > 1 ? a.slice_n[slice] : a.n_samples);
    const uint32_t sample_begin = a.sample_begin + (a.n_slices > 1 ? (uint32_t)a.slice_j0[slice] * a.sample_stride : 0u);
    float *const canvas = slice == 0 ? a.canvas : a.partial + (size_t)(slice - 1) * (size_t)(a.height * a.width) * 3;
    const bool fresh = slice == 0 ? a.fresh != 0 : true;
    const int B = a.max_bounces;

    uint32_t segs = in_canvas && slice == 0 ? 1u : 0u;     // the fp64 primary search below, counted once per launch (a lane runs < 2^32 searches per launch)
    // what bounds this kernel: 16-byte per-lane gathers (one per node visited, three per object record).
    // Counted per lane, summed per wave at the end, next to the segment counter (pt_bvh_counters, include/pt.h).
    uint32_t n_nodes_visited = 0, n_records = 0, n_padding = 0;
#ifdef PT_BVH_STATS
    uint32_t st_iv = 0, st_il = 0, st_r = 0, st_s = 0;   // wave-iterations (x 4096) of the visit loop, the leaf loop, walk steps, shade steps
#define PT_STAT(x) x += 4096u / (uint32_t)__popcll(__ballot(true))
#else
#define PT_STAT(x) (void)0
#endif   // (n_padding: the part of n_records spent on the empty slot of odd leaves)
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (in_canvas && !fresh) acc = mk3(canvas[3 * (size_t)pc + 0], canvas[3 * (size_t)pc + 1], canvas[3 * (size_t)pc + 2]);

    // primary segment, once per pixel, fp64 (src/Lib.hs:41-51, :81-98)
    const PrimaryHit ph = primary_hit(a, pi, pj, stack, BLOCK);
    f3 cd0 = mk3((float)ph.dx, (float)ph.dy, (float)ph.dz);
    f3 cP0 = mk3((float)(ph.t * ph.dx), (float)(ph.t * ph.dy), (float)(ph.t * ph.dz));
    f3 ccol = mk3(1.f, 1.f, 1.f);
    int cidx = ph.idx; float ce = 0.f;
    if (REC32 && cidx >= 0) cidx = a.bvh_pos[cidx];        // from here on the object's position in bvh_prims
    bool traced = in_canvas && cidx >= 0 && B >= 2;       // else every sample is black (src/Lib.hs:68,72)
    if (in_canvas && cidx >= 0) {
        const float4 m0 = mat(cidx, 0);
        ce = m0.w; ccol = mk3(m0.x, m0.y, m0.z);
        if (ce > 0.f) {                                    // the primary ray sees a light (src/Lib.hs:74)
            for (uint32_t s = 0; s < n; ++s) { acc.x += ce; acc.y += ce; acc.z += ce; }
            traced = false;
        }
    }
    s_pix[tid * 3 + 0] = make_float4(cP0.x, cP0.y, cP0.z, cd0.x);
    s_pix[tid * 3 + 1] = make_float4(cd0.y, cd0.z, ccol.x, ccol.y);
    s_pix[tid * 3 + 2] = make_float4(ccol.z, __int_as_float(cidx), ce, 0.f);
    s_next[tid] = traced && n > 0 ? 1u : n;                // a traced pixel's own lane starts on its sample 0
    // the per-pixel sums live in LDS for the whole launch (registers are what limits the occupancy of the walk)
    s_acc[tid * 3 + 0] = acc.x; s_acc[tid * 3 + 1] = acc.y; s_acc[tid * 3 + 2] = acc.z;
    __syncthreads();

    f3 o = cP0, d = cd0, thr = ccol;
    int last = cidx, b = 1;
    uint32_t gs = sample_begin;
    int cur = tid;                                         // the pixel (workgroup-relative) this lane works on
    auto scatter = [&](int idx, float e, f3 din, int bounce, f3 at) -> f3 {
        uint32_t r0, r1;
        const uint64_t pre = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)(pixel0 + cur);   // Philox's first product
        // (round keys as scalar operands: this kernel is bound by its gathers and by registers, not by the SGPR port)
        philox2x32_10_pre((uint32_t)(pre >> 32), (uint32_t)pre, (gs << 6) | (uint32_t)bounce, a.seed, r0, r1);
        if constexpr (SPEC) {
            const float4 m3 = mat(idx, 3);
            f3 nrm = mk3(m3.y, m3.z, m3.w);
            const bool sphere = m3.x != m3.x;              // NaN marks the Sphere extension: normal at the hit point
            if (sphere) {
                const float4 m1 = mat(idx, 1);
                nrm = mk3((at.x - m1.x) * m1.w, (at.y - m1.y) * m1.w, (at.z - m1.z) * m1.w);
            }
            if (e < 0.f)                                   // _reflective (src/Lib.hs:110)
                return scatter_specular(din, nrm, u_theta(r0), u_phi(r1), NOISE_REV);
            if (sphere) return scatter_diffuse_about(nrm, r0, r1);
        }
        return scatter_diffuse_bits(mat(idx, 1), mat(idx, 2), mat(idx, 3).x, r0, r1);   // src/Lib.hs:111
    };

    // ---- the resumable closest-hit search (closest<-2>'s arithmetic) ----------
    const float4 *__restrict__ g = a.bvh_geom;
    const float4 *__restrict__ g32 = a.bvh_geom32;
    const float4 *__restrict__ nodes = a.bvh_nodes;
    int wn = 0, wbest = -1;                                // wn: the node to visit next, 0 = the search is over
    STK *wsp = stack;                                      // this lane's top stack entry
    uint32_t wkey = KEY_LIMIT, neg = 0;
    float wbt = __builtin_inff();
    SlabRay sr = {0.f, 0.f, 0.f};
    auto test = [&](float4 g0, float4 g1, float4 g2, int k, bool on, bool finite = true) {    // finite: a leaf's record (no plane)
        const float num = fmaf(-o.z, g0.z, fmaf(-o.y, g0.y, fmaf(-o.x, g0.x, g0.w)));
        const float den = fmaf(d.z, g1.z, fmaf(d.y, g1.y, d.x * g1.x));
        float t = num * __builtin_amdgcn_rcpf(den);
        const float dx = fmaf(t, d.x, o.x) - g2.x;             // Disk (src/Object.hs:165-171); a plane has r2 = +Inf
        const float dy = fmaf(t, d.y, o.y) - g2.y;
        const float dz = fmaf(t, d.z, o.z) - g2.z;
        bool inside = inside_form<SPEC>(rects, k, g1.w, dx, dy, dz) || (!finite && g1.w == __builtin_inff());
        if constexpr (SPEC) {
            const float ts = sphere_t(mk3(g2.x, g2.y, g2.z), -g1.w, o, d);
            t = (g1.w < 0.0f) ? ts : t;
            inside = inside || (g1.w < 0.0f);
        }
        uint32_t key = __float_as_uint(t) - 1u;
        key = (k == last || !on || !inside) ? KEY_INVALID : key;     // _lastStruck (src/Lib.hs:86-90)
        bool better = key < wkey;
        const bool tie = (key == wkey) & (key < KEY_LIMIT);          // an equal t keeps the lower OBJECT index (minimumBy)
        if constexpr (REC32) { if (tie) better = a.bvh_prims[k] < a.bvh_prims[wbest]; }
        else better = better | (tie & (k < wbest));
        wbest = better ? k : wbest;
        wkey = better ? key : wkey;
        wbt = better ? t : wbt;
    };
    auto begin_search = [&]() {
        wbest = -1; wkey = KEY_LIMIT; wbt = __builtin_inff();
        for (int u = 0; u < a.n_unbounded; ++u) test(g[3 * u], g[3 * u + 1], g[3 * u + 2], REC32 ? u : __float_as_int(g[3 * u + 2].w), true, false);
        sr = slab_ray(d.x, d.y, d.z);
        neg = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
        // (starting below the root -- its code word as a kernel argument, one gather less per search -- was measured:
        // rays that leave the scene are then turned away by two boxes instead of one, 7.0 instead of 6.4 node visits
        // per segment on the 1,000-disk scene, 31.6 instead of 32.1 on the 10,000-disk scene, times equal)
        wn = BVH_ROOT_NODE; wsp = stack;
    };
    auto pop = [&]() {                                     // (the bottom entry is node 0: popping it ends the search)
        wn = (int)*wsp;
        wsp -= BLOCK;
    };
    stack[0] = (STK)0;                                     // (after primary_hit, which walks with the same stack)

    bool live = traced && n > 0;                           // has a path in flight
    bool walking = false, ready = !live;                   // pathless lanes ask the pool for work at the first shade step
    if (live) { d = scatter(cidx, ce, cd0, 0, cP0); begin_search(); walking = true; }
    bool dry = false;                                      // wave-uniform: no pixel has an unstarted sample

#ifdef PT_BVH_CLOCKS
    long long clk_shade = 0, clk_walk = 0, n_shade = 0, n_walk = 0;
#endif
    for (;;) {
        const unsigned long long wm = __ballot(walking), rm = __ballot(ready);
        if (wm == 0ull && rm == 0ull) break;
#ifdef PT_BVH_CLOCKS
        const long long c_in = clock64();
        const bool shading = rm != 0ull && (wm == 0ull || __popcll(rm) >= PT_BVH_SHADE_LANES);
#endif
        if (rm != 0ull && (wm == 0ull || __popcll(rm) >= PT_BVH_SHADE_LANES)) {
            // ---- shade: the lanes whose search has finished ----------------------
            int k = -1; float t = 0.f;
            float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f);
            bool terminal = false;
            if (ready) { PT_STAT(st_s); }
            if (ready && live) {
                k = wbest;
                key_result(k, wkey, t);
                segs += 1;
                const bool hit = k >= 0;
                m0 = mat(hit ? k : 0, 0);
                // miss (src/Lib.hs:72), light (:74) or the depth cap (:68) end the path
                terminal = !hit || (m0.w > 0.f) || (b + 1 >= B);
                if (hit && m0.w > 0.f) {                                     // emittance * pixel (src/Lib.hs:74)
                    atomicAdd(&s_acc[cur * 3 + 0], m0.w * thr.x);
                    atomicAdd(&s_acc[cur * 3 + 1], m0.w * thr.y);
                    atomicAdd(&s_acc[cur * 3 + 2], m0.w * thr.z);
                }
            }
            // the next sample: of my current pixel, else of any pixel of the wave that has one left
            uint32_t sn = 0;
            if (ready && live && terminal) sn = atomicAdd(&s_next[cur], 1u);
            const bool want = ready && (terminal || !live);
            bool needy = want && !(live && terminal && sn < n);
            while (!dry && __ballot(needy) != 0ull) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                const bool has = __hip_atomic_load(&s_next[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n;
                const unsigned long long avail = __ballot(has);
                if (avail == 0ull) { dry = true; break; }
                const int v = wbase + (__ffsll((long long)avail) - 1);     // wave-uniform choice of pixel to help
                if (needy) {
                    sn = atomicAdd(&s_next[v], 1u);
                    if (sn < n) {
                        cur = v;                           // adopt pixel v
                        needy = false;
                    }
                }
            }
            // (one flat region for the lanes that go on: nested, every level of the nest cost a copy of each state
            // register -- 42 v_mov_b32 per shade step)
            const bool lane_done = ready && want && needy; // the pool is dry: this lane is done
            const bool go_on = ready && !lane_done;
            if (lane_done) { live = false; ready = false; }
            {
                if (go_on) {
                    const bool restart = want;             // a new sample from the (current) pixel's primary hit,
                    if (restart) gs = sample_begin + sn * a.sample_stride;
                    // whose constants live in LDS, not in registers: the walk needs the registers (occupancy)
                    const float4 p0 = s_pix[cur * 3 + 0], p1 = s_pix[cur * 3 + 1], p2 = s_pix[cur * 3 + 2];
                    const int   idx = restart ? __float_as_int(p2.y) : k;
                    const float e   = restart ? p2.z : m0.w;
                    const f3    din = restart ? mk3(p0.w, p1.x, p1.y) : d;
                    const int   bn  = restart ? 0 : b;
                    const f3 hp = mk3(restart ? p0.x : fmaf(t, d.x, o.x),      // march (src/Object.hs:151-153)
                                      restart ? p0.y : fmaf(t, d.y, o.y),
                                      restart ? p0.z : fmaf(t, d.z, o.z));
                    thr = mk3(restart ? p1.z : thr.x * m0.x,                     // pixel * color (src/Lib.hs:79)
                              restart ? p1.w : thr.y * m0.y,
                              restart ? p2.x : thr.z * m0.z);
                    d = scatter(idx, e, din, bn, hp);
                    o = hp; last = idx; b = bn + 1;
                    live = true; ready = false; walking = true;
                    begin_search();
                }
            }
        } else {
            // ---- walk: up to PT_BVH_ROUND_VISITS steps and one leaf per searching lane
            if (walking) {
                int code = -1, steps = 0;
                PT_STAT(st_r);
                while (wn != 0 && steps < PT_BVH_ROUND_VISITS) {
                    PT_STAT(st_iv);
                    // one 16-byte gather: scalar base + 32-bit byte offset (fewer than 2^26 nodes)
                    const float4 nd = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + ((uint32_t)wn << 4));
                    int c = __float_as_int(nd.w);
                    // (the code word is pinned here so that it arrives with the box, in one dwordx4: left alone, the
                    // compiler fetches it in the hit branch -- a second dependent round trip per node; and the counter
                    // is an opaque add so that it does not become "base + steps", two copies per step)
                    asm volatile("v_add_u32 %0, 1, %0" : "+v"(n_nodes_visited), "+v"(c));
                    steps += 1;
                    if (!slab_hit(nd, o.x, o.y, o.z, sr, wbt)) { pop(); continue; }
#ifndef PT_BVH_NO_POSTPONE
                    // a lane that reaches a leaf keeps it for the end of the step and walks on (its best hit is then
                    // one leaf out of date for the rest of the step); a second leaf ends its step, and is visited
                    // again by the next one
                    if (c < 0) {
                        if (code >= 0) break;
                        code = ~c; pop(); continue;
                    }
#else
                    if (c < 0) { code = ~c; break; }
#endif
                    // inner node: c = left child << 5 | split axis (v_bfe_u32 reads the low five bits of its offset),
                    // the left child even, its sibling index ^ 1
                    const int near = (c >> 5) + (int)__builtin_amdgcn_ubfe(neg, (uint32_t)c, 1u);
                    wsp += BLOCK;
                    *wsp = (STK)(near ^ 1);
                    wn = near;
                }
                if (code >= 0) {
                    const int first = code >> 3, cnt = (code & 7) + 1;
                    n_records += (REC32 ? 2u : 3u) * (uint32_t)((cnt + 1) & ~1);     // 16-byte gathers: three (two) per record slot (slots are fetched in pairs)
                    n_padding += (REC32 ? 2u : 3u) * (uint32_t)(cnt & 1);
#pragma unroll 1
                    for (int h = 0; h < BVH_LEAF_SLOTS; h += 2) {
                        PT_STAT(st_il);
                        // (fetching a record's third vector only where its plane distance can still win saves 10 % of
                        // the gathers and costs 4.5 % in time -- a second dependent round trip per pair; measured, dropped.
                        // Not fetching the pair's second slot when the leaf's count is odd saves 5-9 % of the gathers and
                        // changes the time by -1...+3 %, as constants under a lane mask (82 VGPRs) or as a single record
                        // behind the pairs; measured, dropped: profiles/r02_bvh_roofline.txt; again in round 3 with
                        // leaves of two, under the lanes' mask: +-0.5 %)
                        if constexpr (REC32) {
                            float4 r[2][2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                r[q][0] = g32[2 * (first - a.n_unbounded + h + q)]; r[q][1] = g32[2 * (first - a.n_unbounded + h + q) + 1];
                            }
#pragma unroll
                            for (int q = 0; q < 2; ++q)      // (n_hat, A) (centre, r^2): n_raw = n_hat in this form
                                test(r[q][0], make_float4(r[q][0].x, r[q][0].y, r[q][0].z, r[q][1].w), r[q][1], first + h + q, h + q < cnt);
                        } else {
                            float4 r[2][3];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                r[q][0] = g[3 * (first + h + q)]; r[q][1] = g[3 * (first + h + q) + 1]; r[q][2] = g[3 * (first + h + q) + 2];
                            }
#pragma unroll
                            for (int q = 0; q < 2; ++q) test(r[q][0], r[q][1], r[q][2], __float_as_int(r[q][2].w), h + q < cnt);
                        }
                        if (h + 2 >= cnt) break;
                    }
#ifdef PT_BVH_NO_POSTPONE
                    pop();
#endif
                }
                if (wn == 0) { walking = false; ready = true; }
            }
        }
#ifdef PT_BVH_CLOCKS
        { const long long dc = clock64() - c_in; if (shading) { clk_shade += dc; n_shade += 1; } else { clk_walk += dc; n_walk += 1; } }
#endif
    }

    // (no workgroup barrier: a pixel's sum is touched by lanes of its own wave only, and a wave's LDS operations
    // complete in order -- the other three waves may still be tracing)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (in_canvas) {
        canvas[3 * (size_t)pc + 0] = s_acc[tid * 3 + 0];
        canvas[3 * (size_t)pc + 1] = s_acc[tid * 3 + 1];
        canvas[3 * (size_t)pc + 2] = s_acc[tid * 3 + 2];
    }
    unsigned long long wsegs = segs, wnodes = n_nodes_visited, wrecs = n_records, wpad = n_padding;
#ifdef PT_BVH_STATS
    wnodes = st_iv; wrecs = st_il; wpad = PT_BVH_STATS == 1 ? st_r : st_s;
#endif
#ifdef PT_BVH_CLOCKS
    wnodes = (tid & 63) == 0 ? (unsigned long long)clk_shade : 0ull;      // experiment: clocks in shade / walk steps
    wrecs = (tid & 63) == 0 ? (unsigned long long)clk_walk : 0ull;
    (void)n_shade; (void)n_walk;
#endif
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        wsegs += __shfl_down(wsegs, off, 64);
        wnodes += __shfl_down(wnodes, off, 64);
        wrecs += __shfl_down(wrecs, off, 64);
        wpad += __shfl_down(wpad, off, 64);
    }
    if ((tid & 63) == 0 && wsegs) {
        unsigned long long *line = seg_lane(a.segments);              // one 128-byte line: [0] segments, [1] nodes, [2] records, [3] of which padding
        atomicAdd(line, wsegs);
        atomicAdd(line + 1, wnodes);
        atomicAdd(line + 2, wrecs);
        atomicAdd(line + 3, wpad);
    }
}


// The primary segment of every pixel for trace_static_kernel: fp64, the reference's operation order (primary_hit_f64,
// the function the parity hook pt_primary_hits runs), once per camera and scene -- the first launch after
// pt_scene_upload -- kept in a.pix as two planes of one float4 per pixel: (P0, record offset 16*(idx+1)), then (d0, -).  The searches are counted.
__global__ __launch_bounds__(256) void primary_fill_kernel(TraceArgs a)
{
    const int npix = a.height * a.width;
    const int pc = blockIdx.x * 256 + threadIdx.x;
    if (pc < npix) {
        const int pi = pc / a.width, pj = pc - pi * a.width;
        const PrimaryHit ph = primary_hit(a, pi, pj);
        a.pix[(size_t)pc] = make_float4((float)(ph.t * ph.dx), (float)(ph.t * ph.dy), (float)(ph.t * ph.dz),
                                            __int_as_float(16 * (ph.idx + 1)));
        a.pix[(size_t)npix + (size_t)pc] = make_float4((float)ph.dx, (float)ph.dy, (float)ph.dz, 0.f);
    }
    const unsigned long long n = __builtin_popcountll(__ballot(pc < npix));
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(seg_lane(a.segments), n);
}

// (the same float by float, for canvases whose length is not a multiple of four)
__global__ __launch_bounds__(256) void slice_sum_scalar_kernel(float *__restrict__ canvas, const float *__restrict__ partial,
                                                               size_t n, int planes)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = canvas[i];
    for (int p = 0; p < planes; ++p) v += partial[(size_t)p * n + i];
    canvas[i] = v;
}

// canvas += plane 0 + plane 1 + ... (in this order): the sample slices of one launch
__global__ __launch_bounds__(256) void slice_sum_kernel(float4 *__restrict__ canvas, const float4 *__restrict__ partial,
                                                        size_t n4, int planes)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = canvas[i];
    for (int p = 0; p < planes; ++p) {
        const float4 q = partial[(size_t)p * n4 + i];
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    canvas[i] = s;
}

template <int NS>
static hipError_t launch_trace_ns(TraceArgs a, const float4 *geom_host, bool reuse, bool spec, bool steal,
                                  hipStream_t s, hipFunction_t scene_kernel = nullptr)
{
    // pad the by-value object list to NS with records that never pass the
    // 0 < t < Inf guard (n = 0 gives t = NaN)
    StaticGeom<NS> sg;
    for (int k = 0; k < (NS > 0 ? NS : 1); ++k) {
        const bool real = NS > 0 && k < a.n_obj;
        sg.g[k][0] = real ? geom_host[3 * k + 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        sg.g[k][1] = real ? geom_host[3 * k + 1] : make_float4(0.f, 0.f, 0.f, __builtin_inff());
        sg.g[k][2] = real ? geom_host[3 * k + 2] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int BLOCK = NS == -1 ? TILED_BLOCK : TRACE_BLOCK;
    size_t smem = 0;
    if (NS == -1) {
        a.tile_objs = a.n_obj < TILE_OBJECTS ? a.n_obj : TILE_OBJECTS;
        smem = (size_t)a.tile_objs * 3 * sizeof(float4);
    }
    if (NS == -2) smem = (size_t)(a.bvh_depth + 1) * TRACE_BLOCK * sizeof(int);
    const int npix = a.height * a.width;
    const dim3 grid((npix + BLOCK - 1) / BLOCK), block(BLOCK);
    if constexpr (NS > 0) {
        if (steal && reuse) {
            // Sample slices: the launch ends when its last waves do, and a wave's life is its 64 pixels' whole
            // sample count (2.8 ms at 1024 spp), so the machine drains for about that long at the end.  A second
            // slice with the last 1/16 of the samples, dispatched after the first, leaves only short waves for
            // the end (measured at 1080p/1024 spp: 14.8 -> 13.8 ms; four slices of 1/2, 1/4, 1/8, 1/8: 14.0 ms, and
            // 2 % slower than one slice when consecutive frames overlap, which two slices are not).
            const int n = a.n_samples;
            a.n_slices = 1; a.slice_j0[0] = 0; a.slice_n[0] = n;
            if (a.partial && npix % 4 == 0 && n >= SLICE_MIN_SAMPLES) {
                a.n_slices = 2;
                a.slice_n[1] = n / 16; a.slice_n[0] = n - a.slice_n[1]; a.slice_j0[1] = a.slice_n[0];
            }
            const dim3 sgrid((npix + STATIC_BLOCK - 1) / STATIC_BLOCK, a.n_slices), sblock(STATIC_BLOCK);
            hipError_t e;
            if (!a.pix) return hipErrorInvalidValue;           // (trace_block allocates it or fails)
            if (!a.pix_valid) {                                // first launch of this camera and scene: the fp64 primary pass
                hipLaunchKernelGGL(primary_fill_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, a);
                if ((e = hipGetLastError()) != hipSuccess) return e;
                a.pix_valid = 1;
            }
            // PT_EXTRA_LDS=<bytes>: unused dynamic LDS per workgroup, to lower the waves per SIMD for an occupancy
            // experiment (160 KB per CU: 10240 -> 4 waves per SIMD, 20480 -> 2)
            static const unsigned extra_lds = getenv("PT_EXTRA_LDS") ? (unsigned)atoi(getenv("PT_EXTRA_LDS")) : 0u;
            if (scene_kernel) {                                // the same kernel compiled for this scene (pt_jit.hip)
                void *params[] = {&a, &sg};
                e = hipModuleLaunchKernel(scene_kernel, sgrid.x, sgrid.y, 1, STATIC_BLOCK, 1, 1, extra_lds, s, params, nullptr);
            } else {
                if constexpr (NS <= MAX_STATIC_OBJECTS) {
                    if (spec) hipLaunchKernelGGL((trace_static_kernel<NS, true>), sgrid, sblock, extra_lds, s, a, sg);
                    else      hipLaunchKernelGGL((trace_static_kernel<NS, false>), sgrid, sblock, extra_lds, s, a, sg);
                    e = hipGetLastError();
                } else {
                    e = hipErrorInvalidValue;                  // 9-16 objects by value: as a scene kernel only
                }
            }
            if (e == hipSuccess && a.n_slices > 1) {
                const size_t n4 = (size_t)npix * 3 / 4;
                hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                                   reinterpret_cast<float4 *>(a.canvas), reinterpret_cast<const float4 *>(a.partial), n4,
                                   a.n_slices - 1);
                e = hipGetLastError();
            }
            return e;
        }
    }
    if constexpr (NS == 0 || NS == -2) {
        if (steal && reuse) {
            constexpr int SB = NS == -2 ? 64 : TRACE_BLOCK;
            const dim3 tgrid((npix + SB - 1) / SB), tblock(SB);
            const size_t ssm = NS == -2 ? (size_t)(a.bvh_depth + 1) * SB * sizeof(int) : 0;
            if (NS == -2 && !getenv("PT_BVH_LOCKSTEP")) {      // (PT_BVH_LOCKSTEP: the one-search-per-trip loop, for A/B)
                // LDS per one-wave workgroup: 4 KB of per-pixel pools (static) + the stacks (2.2 KB at depth 16 with
                // 16-bit entries; 32-bit entries held the kernel to 19 waves per CU, now its 78 VGPRs do: 24) + the
                // "search over" entry at the bottom.
                const bool s16 = a.n_nodes < 65536 && !getenv("PT_BVH_STACK32");   // (PT_BVH_STACK32: the 32-bit form for any tree, A/B and tests)
                // sample slices, as for trace_static_kernel above: the last 1/PT_BVH_SLICE_DIV of a long launch's samples
                // as a second slice behind the first (a one-wave workgroup of a 4K / 512-spp launch lives ~25 ms of the
                // launch's 130: without it the launch ends on a long ragged tail)
                const int ns = a.n_samples;
                a.n_slices = 1; a.slice_j0[0] = 0; a.slice_n[0] = ns;
                static const int slice_div = getenv("PT_BVH_SLICE_DIV") ? atoi(getenv("PT_BVH_SLICE_DIV")) : 16;
                if (a.partial && ns >= BVH_SLICE_MIN_SAMPLES && slice_div > 1) {
                    a.n_slices = 2;
                    a.slice_n[1] = ns / slice_div; a.slice_n[0] = ns - a.slice_n[1]; a.slice_j0[1] = a.slice_n[0];
                }
                const dim3 bgrid((npix + BVH_BLOCK - 1) / BVH_BLOCK, a.n_slices), bblock(BVH_BLOCK);
                // PT_BVH_EXTRA_LDS=<bytes>: unused LDS per workgroup, to lower the waves per CU (occupancy experiment)
                static const size_t bvh_extra_lds = getenv("PT_BVH_EXTRA_LDS") ? (size_t)atoi(getenv("PT_BVH_EXTRA_LDS")) : 0;
                const size_t bsm = bvh_extra_lds + (size_t)(a.bvh_depth + 2) * BVH_BLOCK * (s16 ? sizeof(uint16_t) : sizeof(int));
                const bool r32 = a.bvh_geom32 != nullptr;          // the compact records (pt_scene_upload decides)
#define PT_BVH_LAUNCH(SP, ST) do { if (r32) hipLaunchKernelGGL((trace_bvh_kernel<SP, ST, true>), bgrid, bblock, bsm, s, a); \
                                   else     hipLaunchKernelGGL((trace_bvh_kernel<SP, ST, false>), bgrid, bblock, bsm, s, a); } while (0)
                if (s16) { if (spec) PT_BVH_LAUNCH(true, uint16_t); else PT_BVH_LAUNCH(false, uint16_t); }
                else     { if (spec) PT_BVH_LAUNCH(true, int);      else PT_BVH_LAUNCH(false, int); }
#undef PT_BVH_LAUNCH
                hipError_t e = hipGetLastError();
                if (e == hipSuccess && a.n_slices > 1) {
                    const size_t nf = (size_t)npix * 3;                // (float4-wise where the plane's length allows, else float-wise)
                    if (nf % 4 == 0)
                        hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((nf / 4 + 255) / 256)), dim3(256), 0, s,
                                           reinterpret_cast<float4 *>(a.canvas), reinterpret_cast<const float4 *>(a.partial), nf / 4,
                                           a.n_slices - 1);
                    else
                        hipLaunchKernelGGL(slice_sum_scalar_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s,
                                           a.canvas, a.partial, nf, a.n_slices - 1);
                    e = hipGetLastError();
                }
                return e;
            }
            if (spec) hipLaunchKernelGGL((trace_steal_kernel<NS, true>), tgrid, tblock, ssm, s, a, sg, a.geom);
            else      hipLaunchKernelGGL((trace_steal_kernel<NS, false>), tgrid, tblock, ssm, s, a, sg, a.geom);
            return hipGetLastError();
        }
    }
    if (reuse) {
        if (spec) hipLaunchKernelGGL((trace_kernel<NS, true, true>), grid, block, smem, s, a, sg);
        else      hipLaunchKernelGGL((trace_kernel<NS, true, false>), grid, block, smem, s, a, sg);
    } else {
        if (spec) hipLaunchKernelGGL((trace_kernel<NS, false, true>), grid, block, smem, s, a, sg);
        else      hipLaunchKernelGGL((trace_kernel<NS, false, false>), grid, block, smem, s, a, sg);
    }
    return hipGetLastError();
}

hipError_t launch_trace(const TraceArgs &a, const float4 *geom_host, bool reuse, bool spec, bool lds_tiles,
                        bool steal, hipStream_t s, hipFunction_t scene_kernel)
{
    if (a.n_obj <= 4) return launch_trace_ns<4>(a, geom_host, reuse, spec, steal, s, scene_kernel);
    if (a.n_obj <= MAX_STATIC_OBJECTS)
        return launch_trace_ns<MAX_STATIC_OBJECTS>(a, geom_host, reuse, spec, steal, s, scene_kernel);
    if (scene_kernel && a.n_obj <= MAX_JIT_OBJECTS && reuse && steal && !lds_tiles)
        return launch_trace_ns<MAX_JIT_OBJECTS>(a, geom_host, reuse, spec, steal, s, scene_kernel);
    if (lds_tiles) return launch_trace_ns<-1>(a, geom_host, reuse, spec, false, s);
    if (a.n_nodes > 0) return launch_trace_ns<-2>(a, geom_host, reuse, spec, steal, s);
    return launch_trace_ns<0>(a, geom_host, reuse, spec, steal, s);
}

// ---------------------------------------------------------------------------
// primary hits (fp64), parity hook
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void primary_kernel(TraceArgs a, int32_t *idx, double *t, double *normal)
{
    int *const s_stack = reinterpret_cast<int *>(s_tile);     // (bvh_depth + 1) * 256 entries, dynamic
    const int H = a.height, W = a.width;
    const int pixel = blockIdx.x * 256 + threadIdx.x;
    if (pixel >= H * W) return;
    const int i = pixel / W, j = pixel - i * W;
    const PrimaryHit h = primary_hit(a, i, j, &s_stack[threadIdx.x], 256);
    if (idx) idx[pixel] = h.idx;
    if (t) t[pixel] = h.idx >= 0 ? h.t : 0.0;
    if (normal) {
        const double *p = h.idx >= 0 ? a.prim + (size_t)h.idx * PRIM_DOUBLES : nullptr;
        const double dir[3] = {h.dx, h.dy, h.dz};
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            if (p) {
                // getNormal (src/Object.hs:180-183); Sphere extension: (march - centre) / radius, as o_normal_at
                if (p[10] == 2.0) { const double hit = 0.0 + h.t * dir[c]; v = (1.0 / p[9]) * (hit - p[c]); }
                else v = p[3 + c];
            }
            normal[3 * (size_t)pixel + c] = v;
        }
    }
}

hipError_t launch_primary(const TraceArgs &a, int32_t *idx, double *t, double *normal, hipStream_t s)
{
    const int npix = a.height * a.width;
    hipLaunchKernelGGL(primary_kernel, dim3((npix + 255) / 256), dim3(256),
                       a.n_nodes > 0 ? (size_t)(a.bvh_depth + 1) * 256 * sizeof(int) : 0, s, a, idx, t, normal);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// RGB8 output (src/Conversion.hs:17-28)
// ---------------------------------------------------------------------------
// canvas (i*W + j)*3 + k  ->  rgb (j*H + i)*3 + k : a transpose of the pixel
// grid.  A 64(i) x 64(j) tile is read along j (coalesced fp32), quantised, and
// written along i (coalesced bytes) after a trip through LDS.
constexpr int RGB_TILE = 64;

__device__ __forceinline__ uint8_t quantise(float sum, int mode, double inv_samples)
{
    double x = (double)sum;
    if (mode != 0) {
        x = x * inv_samples;
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        if (mode == 2) x = pow(x, 1.0 / 2.2);
    }
    // round :: Double -> Word8 == half-even to Int, then wrap mod 256
    const double r = rint(255.0 * x);
    return (uint8_t)(unsigned long long)(long long)r;
}

__global__ __launch_bounds__(256) void rgb8_kernel(const float *__restrict__ canvas, uint8_t *__restrict__ rgb,
                                                   int H, int W, int mode, double inv_samples)
{
    __shared__ uint8_t tile[RGB_TILE][RGB_TILE * 3 + 4];     // [jj][ii*3+k], padded
    const int i0 = blockIdx.y * RGB_TILE, j0 = blockIdx.x * RGB_TILE;
    const int tj = min(RGB_TILE, W - j0), ti = min(RGB_TILE, H - i0);
    // read: row ii of the tile is tj*3 contiguous floats
    for (int ii = threadIdx.x / 64; ii < ti; ii += 4) {
        const float *row = canvas + ((size_t)(i0 + ii) * W + j0) * 3;
        for (int e = threadIdx.x % 64; e < tj * 3; e += 64) {
            const int jj = e / 3, k = e - jj * 3;
            tile[jj][ii * 3 + k] = quantise(row[e], mode, inv_samples);
        }
    }
    __syncthreads();
    // write: row jj of the output tile is ti*3 contiguous bytes
    for (int jj = threadIdx.x / 64; jj < tj; jj += 4) {
        uint8_t *row = rgb + ((size_t)(j0 + jj) * H + i0) * 3;
        for (int e = threadIdx.x % 64; e < ti * 3; e += 64) row[e] = tile[jj][e];
    }
}

hipError_t launch_rgb8(const float *canvas, uint8_t *rgb, int H, int W, int mode, double inv_samples,
                       hipStream_t s)
{
    const dim3 grid((W + RGB_TILE - 1) / RGB_TILE, (H + RGB_TILE - 1) / RGB_TILE);
    hipLaunchKernelGGL(rgb8_kernel, grid, dim3(256), 0, s, canvas, rgb, H, W, mode, inv_samples);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// device-function parity hooks
// ---------------------------------------------------------------------------
__global__ void dev_diffuse_kernel(int n, const float4 *__restrict__ m, const float *__restrict__ u,
                                   float *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const f3 r = scatter_diffuse(m[4 * k + 1], m[4 * k + 2], m[4 * k + 3].x, u[2 * k], u[2 * k + 1]);
    out[3 * k] = r.x; out[3 * k + 1] = r.y; out[3 * k + 2] = r.z;
}

__global__ void dev_specular_kernel(int n, const float *__restrict__ v, const float *__restrict__ nh,
                                    const float *__restrict__ u, float noise_rev, float *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const f3 r = scatter_specular(mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]),
                                  mk3(nh[3 * k], nh[3 * k + 1], nh[3 * k + 2]), u[2 * k], u[2 * k + 1],
                                  noise_rev);
    out[3 * k] = r.x; out[3 * k + 1] = r.y; out[3 * k + 2] = r.z;
}

__global__ void dev_distance_kernel(int n, const float4 *__restrict__ g, const float *__restrict__ o,
                                    const float *__restrict__ v, float *__restrict__ t)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    t[k] = distance_from(g[0], g[1], g[2], g + 3, mk3(o[3 * k], o[3 * k + 1], o[3 * k + 2]),
                         mk3(v[3 * k], v[3 * k + 1], v[3 * k + 2]));
}

__global__ void dev_philox_kernel(int n, const uint32_t *__restrict__ c0, const uint32_t *__restrict__ c1,
                                  uint32_t key, uint32_t *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t a, b;
    philox2x32_10(c0[k], c1[k], key, a, b);
    out[2 * k] = a; out[2 * k + 1] = b;
}

// the spatial index's box test (slab_hit) on caller-supplied nodes and rays: out[k] = 1 where the walk would enter the box
__global__ void dev_box_kernel(int n, const float4 *__restrict__ nodes, const float *__restrict__ o, const float *__restrict__ v,
                               const float *__restrict__ t_best, uint8_t *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const SlabRay sr = slab_ray(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
    out[k] = slab_hit(nodes[k], o[3 * k], o[3 * k + 1], o[3 * k + 2], sr, t_best[k]) ? 1 : 0;
}

hipError_t launch_dev_box(int n, const float4 *nodes, const float *o, const float *v, const float *t_best, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(dev_box_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, nodes, o, v, t_best, out);
    return hipGetLastError();
}

hipError_t launch_dev_diffuse(int n, const float4 *m, const float *u, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(dev_diffuse_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, m, u, out);
    return hipGetLastError();
}
hipError_t launch_dev_specular(int n, const float *v, const float *nh, const float *u, float noise_rev,
                               float *out, hipStream_t s)
{
    hipLaunchKernelGGL(dev_specular_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, v, nh, u, noise_rev, out);
    return hipGetLastError();
}
hipError_t launch_dev_distance(int n, const float4 *g, const float *o, const float *v, float *t, hipStream_t s)
{
    hipLaunchKernelGGL(dev_distance_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, g, o, v, t);
    return hipGetLastError();
}
hipError_t launch_dev_philox(int n, const uint32_t *c0, const uint32_t *c1, uint32_t key, uint32_t *out,
                             hipStream_t s)
{
    hipLaunchKernelGGL(dev_philox_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, c0, c1, key, out);
    return hipGetLastError();
}

} // namespace pt
