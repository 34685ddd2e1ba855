// pt_api.hip -- the C ABI of libpt (include/pt.h): context, scene flattening,
// kernel launches, read-back.  No CPU fallback exists: every compute entry
// point runs HIP kernels on the context's device or fails.
//
// File:line citations are relative to the reference repository root.
#include "pt_ctx.hpp"
#include "pt_bvh.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

using namespace pt;

namespace pt {

static thread_local std::string g_err;   // for failures before a context exists

int fail(pt_ctx *c, int status, const std::string &msg)
{
    if (c) c->err = msg; else g_err = msg;
    return status;
}

int check_ctx(pt_ctx *c)
{
    if (!c) return fail(nullptr, PT_ERR_INVALID, "null context");
    if (c->group) return PT_OK;                       // a group head has no device of its own
    hipError_t e = hipSetDevice(c->p.device);
    if (e != hipSuccess) return fail(c, PT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return PT_OK;
}

} // namespace pt

namespace {

// ---- host-side scene flattening (fp64, then rounded to fp32) ---------------
// The per-object constants below are the wave-invariant part of
// distanceFrom' (src/Object.hs:176-177) and of rotateRel about the object's
// normal (src/Util.hs:140-145), evaluated once at upload in the reference's
// own operation order.
struct d3 { double x, y, z; };

double norm2_(d3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }   // src/Triple.hs:61-64

d3 normalize_(d3 v)                                                            // src/Triple.hs:66-69
{
    const double fl = std::pow(10.0, -6.0);
    const double n = norm2_(v);
    const double nn = (fl <= n) ? n : fl;
    return d3{v.x / nn, v.y / nn, v.z / nn};
}

const double kPi = 3.141592653589793;
double sine_(double deg) { return std::sin(deg / 180.0 * kPi); }               // AC-Angle Degrees
double cosine_(double deg) { return std::cos(deg / 180.0 * kPi); }

// toSphericalCoords (src/Util.hs:114-122) with arctan2 (:106-112), degrees.
void to_spherical_(d3 vec, double &theta, double &phi)
{
    const d3 n = normalize_(vec);
    if (n.x == 0.0 && n.y == 0.0 && n.z >= 0.0) { theta = 0.0; phi = 0.0; return; }
    if (n.x == 0.0 && n.y == 0.0 && n.z < 0.0) { theta = 180.0; phi = 0.0; return; }
    theta = std::acos(n.z) / kPi * 180.0;
    double a = std::atan(n.y / n.x) / kPi * 180.0;
    if (n.x < 0.0) a += 180.0;
    if (!(a >= 0.0 && a < 360.0)) a = a - std::floor(a / 360.0) * 360.0;       // `mod'` 360
    phi = a;
}

void flatten_object(const pt_object &ob, float4 *geom3, float4 *mat4, double *prim, float4 *rect2 = nullptr)
{
    for (int q = 11; q < PRIM_DOUBLES; ++q) prim[q] = 0.0;
    if (rect2) rect2[0] = rect2[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    const d3 p{ob.point[0], ob.point[1], ob.point[2]};
    const d3 n{ob.normal[0], ob.normal[1], ob.normal[2]};
    const d3 nh = normalize_(n);
    const double A = (0.0 + p.x * n.x + p.y * n.y) + p.z * n.z;
    const bool disk = ob.kind == PT_DISK;
    const float inf = std::numeric_limits<float>::infinity();
    // a disk with a non-positive radius can never satisfy distance < radius (r2 = 0: nothing is below it;
    // a NEGATIVE g1.w marks a sphere)
    const float r2 = !disk ? inf : (ob.radius > 0.0 ? (float)(ob.radius * ob.radius) : 0.0f);
    geom3[0] = make_float4((float)n.x, (float)n.y, (float)n.z, (float)A);
    geom3[1] = make_float4((float)nh.x, (float)nh.y, (float)nh.z, r2);
    geom3[2] = make_float4((float)p.x, (float)p.y, (float)p.z, 0.0f);
    if (ob.kind == PT_SPHERE) {
        // Extension (the reference's TODO, src/Object.hs:191-194).  A sphere that can never be hit
        // (r <= 0 or NaN) becomes a plane record with a zero normal: t = NaN.
        const bool live = ob.radius > 0.0;
        geom3[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        geom3[1] = make_float4(0.f, 0.f, 0.f, live ? -(float)(ob.radius * ob.radius) : inf);
        const float e = ob.emittance > 0.0 ? (float)ob.emittance : (ob.reflective ? -1.0f : 0.0f);
        mat4[0] = make_float4((float)ob.color[0], (float)ob.color[1], (float)ob.color[2], e);
        mat4[1] = make_float4((float)p.x, (float)p.y, (float)p.z, live ? (float)(1.0 / ob.radius) : 0.f);
        mat4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        mat4[3] = make_float4(std::numeric_limits<float>::quiet_NaN(), 0.f, 0.f, 0.f);
        prim[0] = p.x; prim[1] = p.y; prim[2] = p.z;
        for (int q = 3; q < 9; ++q) prim[q] = 0.0;
        prim[9] = ob.radius;
        prim[10] = 2.0;
        return;
    }

    // M = |n| * rotateAbs(., theta', phi')  (src/Util.hs:131-145)
    double th, ph;
    to_spherical_(n, th, ph);
    const double ct = cosine_(th), st = sine_(th), cp = cosine_(ph), sp = sine_(ph);
    const double len = norm2_(n);
    const double M[3][3] = {{ct * cp * len, -sp * len, st * cp * len},
                            {sp * ct * len, cp * len, sp * st * len},
                            {-st * len, 0.0, ct * len}};
    const float e = ob.emittance > 0.0 ? (float)ob.emittance : (ob.reflective ? -1.0f : 0.0f);
    mat4[0] = make_float4((float)ob.color[0], (float)ob.color[1], (float)ob.color[2], e);
    mat4[1] = make_float4((float)M[0][0], (float)M[0][1], (float)M[0][2], (float)M[1][0]);
    mat4[2] = make_float4((float)M[1][1], (float)M[1][2], (float)M[2][0], (float)M[2][1]);
    mat4[3] = make_float4((float)M[2][2], (float)nh.x, (float)nh.y, (float)nh.z);

    prim[0] = p.x; prim[1] = p.y; prim[2] = p.z;
    prim[3] = n.x; prim[4] = n.y; prim[5] = n.z;
    prim[6] = nh.x; prim[7] = nh.y; prim[8] = nh.z;
    prim[9] = ob.radius;
    prim[10] = disk ? 1.0 : 0.0;

    if (ob.kind == PT_RECTANGLE) {
        // Extension (commented out in the reference, src/Object.hs:57-61, :172-175; include/pt.h defines it):
        // the sides run along rotateAbs (1,0,0) and rotateAbs (0,1,0) (src/Util.hs:131-138), evaluated as
        // o_rectangle_axes does.  The device tests |d.(u/hw)| < 1; the fp64 primary pass |d.u| < hw.
        const double hh = ob.extent[0] / 2.0, hw = ob.extent[1] / 2.0;
        const bool live = hw > 0.0 && hh > 0.0;
        const double u[3] = {ct * cp, sp * ct, -st}, w[3] = {-sp, cp, 0.0};
        // r2 = NaN marks the form for inside_form; a side that is not positive is a disk of radius 0: never hit
        geom3[1].w = live ? std::numeric_limits<float>::quiet_NaN() : 0.0f;
        if (rect2 && live) {
            rect2[0] = make_float4((float)(u[0] / hw), (float)(u[1] / hw), (float)(u[2] / hw), 0.f);
            rect2[1] = make_float4((float)(w[0] / hh), (float)(w[1] / hh), (float)(w[2] / hh), 0.f);
        }
        prim[10] = 3.0;
        for (int q = 0; q < 3; ++q) { prim[11 + q] = u[q]; prim[14 + q] = w[q]; }
        prim[17] = hw; prim[18] = hh;
    }
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, n * sizeof(T)); }
    hipError_t up(const T *h, size_t n) { return hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice); }
    hipError_t down(T *h, size_t n) { return hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost); }
};


} // namespace

extern "C" {

int pt_abi_version(void) { return PT_ABI_VERSION; }

const char *pt_status_string(int s)
{
    switch (s) {
    case PT_OK: return "ok";
    case PT_ERR_INVALID: return "invalid argument";
    case PT_ERR_NO_DEVICE: return "no usable HIP device";
    case PT_ERR_HIP: return "HIP runtime error";
    case PT_ERR_NO_SCENE: return "no scene uploaded";
    case PT_ERR_LIMIT: return "limit exceeded";
    case PT_ERR_UNSUPPORTED: return "optional step not available";
    default: return "unknown status";
    }
}

const char *pt_last_error(pt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int pt_create(pt_ctx **out, const pt_params *params)
{
    if (!out || !params) return fail(nullptr, PT_ERR_INVALID, "pt_create: null argument");
    *out = nullptr;
    const pt_params &p = *params;
    if (p.height < 1 || p.width < 1 || (int64_t)p.height * p.width > ((int64_t)1 << 28))
        return fail(nullptr, PT_ERR_INVALID, "pt_create: height/width out of range");
    if (p.max_bounces < 1 || p.max_bounces > MAX_BOUNCES)
        return fail(nullptr, PT_ERR_LIMIT, "pt_create: max_bounces must be in 1..64");
    if (p.sample_stride < 1 || p.sample_offset < 0 || p.sample_offset >= p.sample_stride)
        return fail(nullptr, PT_ERR_INVALID, "pt_create: need 0 <= sample_offset < sample_stride");
    if (!(p.camera_depth == p.camera_depth))
        return fail(nullptr, PT_ERR_INVALID, "pt_create: camera_depth is NaN");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(nullptr, PT_ERR_NO_DEVICE, "pt_create: no HIP device (libpt has no CPU path)");
    }
    if (p.device < 0 || p.device >= ndev)
        return fail(nullptr, PT_ERR_NO_DEVICE, "pt_create: device ordinal out of range");

    pt_ctx *c = new (std::nothrow) pt_ctx;
    if (!c) return fail(nullptr, PT_ERR_INVALID, "pt_create: out of host memory");
    c->p = p;
    auto bail = [&](const char *what, hipError_t he) {
        std::string m = std::string(what) + ": " + hipGetErrorString(he);
        pt_destroy(c);
        return fail(nullptr, PT_ERR_HIP, m);
    };
    if ((e = hipSetDevice(p.device)) != hipSuccess) return bail("hipSetDevice", e);
    if (scratch_init(nullptr, p.device) != PT_OK) { pt_destroy(c); return PT_ERR_HIP; }   // pt_noalloc.hip: the static scratch's address on this device
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    c->stream = c->own_stream;
    const size_t cbytes = npix(c) * 3 * sizeof(float);
    if ((e = hipMalloc((void **)&c->d_canvas, cbytes)) != hipSuccess) return bail("hipMalloc(canvas)", e);
    if ((e = hipMalloc((void **)&c->d_segments, (size_t)SEG_SLOTS * SEG_EPOCH * sizeof(unsigned long long))) != hipSuccess)
        return bail("hipMalloc(segments)", e);
    if ((e = hipMemsetAsync(c->d_canvas, 0, cbytes, c->stream)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMemsetAsync(c->d_segments, 0, (size_t)SEG_SLOTS * SEG_EPOCH * sizeof(unsigned long long), c->stream)) != hipSuccess)
        return bail("hipMemset", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_ctx *c)
{
    if (!c) return;
    if (c->group) { group_destroy(c); delete c; return; }
    (void)hipSetDevice(c->p.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    jit_release(c->scene_kernel);
    if (c->d_canvas) (void)hipFree(c->d_canvas);
    if (c->d_segments) (void)hipFree(c->d_segments);
    if (c->d_geom) (void)hipFree(c->d_geom);
    if (c->d_mat) (void)hipFree(c->d_mat);
    if (c->d_prim) (void)hipFree(c->d_prim);
    if (c->d_rect) (void)hipFree(c->d_rect);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_pix) (void)hipFree(c->d_pix);
    if (c->d_gate) (void)hipFree(c->d_gate);
    if (c->d_bvh_nodes) (void)hipFree(c->d_bvh_nodes);
    if (c->d_bvh_prims) (void)hipFree(c->d_bvh_prims);
    if (c->d_bvh_geom) (void)hipFree(c->d_bvh_geom);
    if (c->d_bvh_geom32) (void)hipFree(c->d_bvh_geom32);
    if (c->d_bvh_mat) (void)hipFree(c->d_bvh_mat);
    if (c->d_bvh_rect) (void)hipFree(c->d_bvh_rect);
    if (c->d_bvh_pos) (void)hipFree(c->d_bvh_pos);
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    if (c->d_hits) (void)hipFree(c->d_hits);
    comm_destroy(c);
    if (c->ev_sync) (void)hipEventDestroy(c->ev_sync);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pt_scene_upload(pt_ctx *c, const pt_object *objs, int32_t n)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return group_scene_upload(c, objs, n);
    // n = 0 is the reference's empty object list: closestObjectTo's guard (src/Lib.hs:83) answers Nothing, every sample is black
    if (n < 0 || n > MAX_OBJECTS) return fail(c, PT_ERR_LIMIT, "pt_scene_upload: need 0 <= n <= 65536");
    if (!objs && n > 0) return fail(c, PT_ERR_INVALID, "pt_scene_upload: null objects");
    for (int k = 0; k < n; ++k)
        if (objs[k].kind != PT_INFINITE_PLANE && objs[k].kind != PT_DISK && objs[k].kind != PT_SPHERE && objs[k].kind != PT_RECTANGLE)
            return fail(c, PT_ERR_INVALID, "pt_scene_upload: unknown Form kind");

    // 8 trailing records that can never be hit (n = 0 gives t = NaN): the
    // streamed closest-hit loop works in groups of 4 and prefetches one group ahead
    constexpr int GEOM_PAD = 8;
    std::vector<float4> geom((size_t)(n + GEOM_PAD) * 3), mat((size_t)(n > 0 ? n : 1) * 4);   // (never empty: hipMalloc(0))
    for (int k = n; k < n + GEOM_PAD; ++k) {
        geom[(size_t)k * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        geom[(size_t)k * 3 + 1] = make_float4(0.f, 0.f, 0.f, std::numeric_limits<float>::infinity());
        geom[(size_t)k * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    std::vector<double> prim((size_t)(n > 0 ? n : 1) * PRIM_DOUBLES);
    bool spec = false, has_rect = false;
    for (int k = 0; k < n; ++k) has_rect = has_rect || objs[k].kind == PT_RECTANGLE;
    std::vector<float4> rect(has_rect ? (size_t)n * 2 : 0);
    for (int k = 0; k < n; ++k) {
        flatten_object(objs[k], &geom[(size_t)k * 3], &mat[(size_t)k * 4], &prim[(size_t)k * PRIM_DOUBLES],
                       has_rect ? &rect[(size_t)k * 2] : nullptr);
        // only a non-emissive reflective object ever scatters specularly (src/Lib.hs:74-78)
        spec = spec || (objs[k].reflective && !(objs[k].emittance > 0.0));
        // a sphere's bounce builds its rotation per hit: the same general-scatter kernels
        spec = spec || objs[k].kind == PT_SPHERE;
        // a rectangle's inside test lives in the same general kernels (inside_form<true>)
        spec = spec || objs[k].kind == PT_RECTANGLE;
    }
    PT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d_geom) { (void)hipFree(c->d_geom); c->d_geom = nullptr; }
    if (c->d_mat) { (void)hipFree(c->d_mat); c->d_mat = nullptr; }
    if (c->d_prim) { (void)hipFree(c->d_prim); c->d_prim = nullptr; }
    if (c->d_rect) { (void)hipFree(c->d_rect); c->d_rect = nullptr; }
    if (c->d_bvh_nodes) { (void)hipFree(c->d_bvh_nodes); c->d_bvh_nodes = nullptr; }
    if (c->d_bvh_prims) { (void)hipFree(c->d_bvh_prims); c->d_bvh_prims = nullptr; }
    if (c->d_bvh_geom) { (void)hipFree(c->d_bvh_geom); c->d_bvh_geom = nullptr; }
    if (c->d_bvh_geom32) { (void)hipFree(c->d_bvh_geom32); c->d_bvh_geom32 = nullptr; }
    if (c->d_bvh_mat) { (void)hipFree(c->d_bvh_mat); c->d_bvh_mat = nullptr; }
    if (c->d_bvh_rect) { (void)hipFree(c->d_bvh_rect); c->d_bvh_rect = nullptr; }
    if (c->d_bvh_pos) { (void)hipFree(c->d_bvh_pos); c->d_bvh_pos = nullptr; }
    c->n_nodes = c->n_unbounded = 0;
    c->n_obj = 0;
    c->have_scene = false;
    c->pix_valid = false;
    PT_HIP(c, hipMalloc((void **)&c->d_geom, geom.size() * sizeof(float4)));
    PT_HIP(c, hipMalloc((void **)&c->d_mat, mat.size() * sizeof(float4)));
    PT_HIP(c, hipMalloc((void **)&c->d_prim, prim.size() * sizeof(double)));
    PT_HIP(c, hipMemcpy(c->d_geom, geom.data(), geom.size() * sizeof(float4), hipMemcpyHostToDevice));
    PT_HIP(c, hipMemcpy(c->d_mat, mat.data(), mat.size() * sizeof(float4), hipMemcpyHostToDevice));
    PT_HIP(c, hipMemcpy(c->d_prim, prim.data(), prim.size() * sizeof(double), hipMemcpyHostToDevice));
    if (has_rect) {
        PT_HIP(c, hipMalloc((void **)&c->d_rect, rect.size() * sizeof(float4)));
        PT_HIP(c, hipMemcpy(c->d_rect, rect.data(), rect.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    c->h_geom = geom;
    jit_release(c->scene_kernel);                // belongs to the previous scene (the stream was drained above)
    c->scene_kernel = nullptr;
    c->n_obj = n;
    c->have_scene = true;
    c->has_specular = spec;

    // Spatial index over the finite forms, for scenes beyond the by-value list.
    if (n > MAX_STATIC_OBJECTS && !(c->p.flags & PT_FLAG_NO_BVH)) {
        std::vector<BvhBox> boxes((size_t)n);
        std::vector<int32_t> finite;
        Bvh bvh;
        double extent = 0.0;
        for (int k = 0; k < n; ++k) {
            if (objs[k].kind == PT_INFINITE_PLANE) { bvh.prims.push_back(k); continue; }
            const double *pr = &prim[(size_t)k * PRIM_DOUBLES];
            const double r = objs[k].radius > 0.0 ? objs[k].radius : 0.0;      // r <= 0 or NaN: never hit
            const double zero3[3] = {0.0, 0.0, 0.0};                           // a sphere: the full +-r box
            boxes[(size_t)k] = disk_bounds(pr, objs[k].kind == PT_SPHERE ? zero3 : pr + 6, r);
            if (objs[k].kind == PT_RECTANGLE) {                                // corners c +- hw u +- hh w
                const double hw = pr[17] > 0.0 ? pr[17] : 0.0, hh = pr[18] > 0.0 ? pr[18] : 0.0;
                for (int a = 0; a < 3; ++a) {
                    const double h = hw * std::fabs(pr[11 + a]) + hh * std::fabs(pr[14 + a]);
                    boxes[(size_t)k].lo[a] = pr[a] - h; boxes[(size_t)k].hi[a] = pr[a] + h;
                }
            }
            for (int a = 0; a < 3; ++a)
                extent = std::max(extent, std::max(std::fabs(boxes[(size_t)k].lo[a]), std::fabs(boxes[(size_t)k].hi[a])));
            finite.push_back(k);
        }
        bvh.n_unbounded = (int32_t)bvh.prims.size();
        if ((int)finite.size() >= BVH_MIN_FINITE && std::isfinite(extent)) {
            // the device's slab test and ray arithmetic are fp32: a few ulp of the largest
            // coordinate (1.2e-7 * extent each); the pad is ~100 times that
            BvhBuilder builder(boxes, finite, 2e-5 * extent + 1e-30);
            builder.build(bvh, bvh.n_unbounded);
            c->bvh_depth = builder.max_depth;
            PT_HIP(c, hipMalloc((void **)&c->d_bvh_nodes, bvh.nodes.size() * sizeof(BvhNode)));
            PT_HIP(c, hipMalloc((void **)&c->d_bvh_prims, bvh.prims.size() * sizeof(int32_t)));
            PT_HIP(c, hipMemcpy(c->d_bvh_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
            PT_HIP(c, hipMemcpy(c->d_bvh_prims, bvh.prims.data(), bvh.prims.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            // the geometry again in leaf order, object index in [2].w, + BVH_LEAF_MAX never-hit records
            std::vector<float4> lg((bvh.prims.size() + BVH_LEAF_MAX) * 3);
            for (size_t sl = 0; sl < bvh.prims.size() + BVH_LEAF_MAX; ++sl) {
                const int k = sl < bvh.prims.size() ? bvh.prims[sl] : n;        // n: the first pad record of `geom`
                for (int f = 0; f < 3; ++f) lg[sl * 3 + f] = geom[(size_t)k * 3 + f];
                lg[sl * 3 + 2].w = __builtin_bit_cast(float, (int32_t)(sl < bvh.prims.size() ? k : 0x7fffffff));
            }
            PT_HIP(c, hipMalloc((void **)&c->d_bvh_geom, lg.size() * sizeof(float4)));
            PT_HIP(c, hipMemcpy(c->d_bvh_geom, lg.data(), lg.size() * sizeof(float4), hipMemcpyHostToDevice));
            c->n_nodes = (int)bvh.nodes.size();
            c->n_unbounded = bvh.n_unbounded;
            // The compact form for trace_bvh_kernel (TraceArgs::bvh_geom32): possible when the raw normal of every
            // finite plane-like form is its unit normal bit for bit in fp32 -- the object test then reads the same
            // numbers from two vectors that it read from three -- (PT_BVH_REC48 in the environment keeps the
            // 48-byte records, for A/B).
            bool compact = !getenv("PT_BVH_REC48");
            for (size_t sl = (size_t)bvh.n_unbounded; sl < bvh.prims.size() && compact; ++sl) {
                const int k = bvh.prims[sl];
                if (objs[k].kind == PT_SPHERE) continue;
                const float4 g0 = geom[(size_t)k * 3], g1 = geom[(size_t)k * 3 + 1];
                compact = std::memcmp(&g0.x, &g1.x, 4) == 0 && std::memcmp(&g0.y, &g1.y, 4) == 0 && std::memcmp(&g0.z, &g1.z, 4) == 0;
            }
            if (compact) {
                const size_t nf = bvh.prims.size() - (size_t)bvh.n_unbounded;
                std::vector<float4> g32((nf + BVH_LEAF_MAX) * 2), pm(bvh.prims.size() * 4), pr2(has_rect ? bvh.prims.size() * 2 : 0);
                std::vector<int32_t> pos((size_t)n, 0);
                for (size_t q = 0; q < nf + BVH_LEAF_MAX; ++q) {
                    const int k = q < nf ? bvh.prims[(size_t)bvh.n_unbounded + q] : n;     // n: a never-hit pad record
                    const float4 g0 = geom[(size_t)k * 3], g1 = geom[(size_t)k * 3 + 1], g2 = geom[(size_t)k * 3 + 2];
                    const bool sphere = k < n && objs[k].kind == PT_SPHERE;
                    g32[q * 2] = sphere ? make_float4(g1.x, g1.y, g1.z, g0.w) : make_float4(g0.x, g0.y, g0.z, g0.w);
                    g32[q * 2 + 1] = make_float4(g2.x, g2.y, g2.z, g1.w);
                }
                for (size_t sl = 0; sl < bvh.prims.size(); ++sl) {
                    const int k = bvh.prims[sl];
                    pos[(size_t)k] = (int32_t)sl;
                    for (int f = 0; f < 4; ++f) pm[sl * 4 + f] = mat[(size_t)k * 4 + f];
                    if (has_rect) { pr2[sl * 2] = rect[(size_t)k * 2]; pr2[sl * 2 + 1] = rect[(size_t)k * 2 + 1]; }
                }
                PT_HIP(c, hipMalloc((void **)&c->d_bvh_geom32, g32.size() * sizeof(float4)));
                PT_HIP(c, hipMemcpy(c->d_bvh_geom32, g32.data(), g32.size() * sizeof(float4), hipMemcpyHostToDevice));
                PT_HIP(c, hipMalloc((void **)&c->d_bvh_mat, pm.size() * sizeof(float4)));
                PT_HIP(c, hipMemcpy(c->d_bvh_mat, pm.data(), pm.size() * sizeof(float4), hipMemcpyHostToDevice));
                PT_HIP(c, hipMalloc((void **)&c->d_bvh_pos, pos.size() * sizeof(int32_t)));
                PT_HIP(c, hipMemcpy(c->d_bvh_pos, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                if (has_rect) {
                    PT_HIP(c, hipMalloc((void **)&c->d_bvh_rect, pr2.size() * sizeof(float4)));
                    PT_HIP(c, hipMemcpy(c->d_bvh_rect, pr2.data(), pr2.size() * sizeof(float4), hipMemcpyHostToDevice));
                }
            }
        }
    }
    // PT_SPECIALISE=1 in the environment: pt_scene_specialise for every scene it applies to (how the whole GPU
    // test suite is run against the scene kernels); where it does not apply the general kernel stays, silently
    if (const char *e = getenv("PT_SPECIALISE")) {
        if (e[0] == '1' && c->n_obj <= MAX_JIT_OBJECTS &&
            !(c->p.flags & (PT_FLAG_NO_STEAL | PT_FLAG_NO_PRIMARY_REUSE | PT_FLAG_LDS_TILES))) {
            std::string why;
            c->scene_kernel = jit_static_kernel(c->p.device, c->n_obj, c->has_specular, c->h_geom.data(), why);
        }
    }
    return PT_OK;
}

// the fields of TraceArgs that describe the scene
static void scene_args(const pt_ctx *c, TraceArgs &a)
{
    a.geom = c->d_geom; a.mat = c->d_mat; a.prim = c->d_prim; a.rect = c->d_rect;
    a.camera_depth = c->p.camera_depth;
    a.n_obj = c->n_obj; a.height = c->p.height; a.width = c->p.width;
    a.bvh_nodes = c->d_bvh_nodes; a.bvh_prims = c->d_bvh_prims; a.bvh_geom = c->d_bvh_geom;
    a.bvh_geom32 = c->d_bvh_geom32; a.bvh_mat = c->d_bvh_mat; a.bvh_rect = c->d_bvh_rect; a.bvh_pos = c->d_bvh_pos;
    a.n_nodes = c->n_nodes; a.n_unbounded = c->n_unbounded; a.bvh_depth = c->bvh_depth;
}

int pt_trace(pt_ctx *c, int32_t iters)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return group_trace(c, iters);
    return trace_block(c, c->samples_done, iters);
}

} // extern "C"

// `iters` more samples per pixel, the global samples offset + k*stride for k in [k_begin, k_begin + iters).
// pt_trace continues the context's own count (k_begin = samples_done); a device group (pt_multi.hip) hands each
// member a block of the call's range.
int pt::trace_block(pt_ctx *c, int64_t k_begin, int32_t iters)
{
    if (int s = check_ctx(c)) return s;
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "pt_trace: no scene uploaded");
    if (iters < 0 || k_begin < 0) return fail(c, PT_ERR_INVALID, "pt_trace: iters < 0");
    c->total_valid = false;
    const int64_t last_global =
        (int64_t)c->p.sample_offset + (k_begin + iters) * (int64_t)c->p.sample_stride;
    if (last_global >= MAX_GLOBAL_SAMPLES)       // the kernels shift (first index beyond the launch) << 6 in 32 bits
        return fail(c, PT_ERR_LIMIT, "pt_trace: global sample index would exceed 2^26");

    const bool reuse = !(c->p.flags & PT_FLAG_NO_PRIMARY_REUSE);
    // planes for the sample slices of trace_static_kernel (small scenes, default flags, large launches)
    // the by-value kernels: up to 8 objects, and up to 16 where a kernel has been compiled for the scene
    const bool static_path = (c->n_obj <= MAX_STATIC_OBJECTS || (c->scene_kernel && c->n_obj <= MAX_JIT_OBJECTS)) && reuse &&
                             !(c->p.flags & PT_FLAG_NO_STEAL);
    // (trace_bvh_kernel slices too: launch_trace's routing to it, restated)
    const bool bvh_path = !static_path && c->n_obj > MAX_STATIC_OBJECTS && c->n_nodes > 0 && reuse &&
                          !(c->p.flags & (PT_FLAG_NO_STEAL | PT_FLAG_LDS_TILES)) && !getenv("PT_BVH_LOCKSTEP");
    if (!c->d_partial && ((iters >= SLICE_MIN_SAMPLES && static_path) || (iters >= BVH_SLICE_MIN_SAMPLES && bvh_path)))
        if (hipMalloc((void **)&c->d_partial, (size_t)(MAX_SLICES - 1) * npix(c) * 3 * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();                 // no memory for the planes: launches stay unsliced
            c->d_partial = nullptr;
        }
    if (static_path && !c->d_pix) {                  // the kept primary hits, 32 B per pixel (primary_fill_kernel)
        const hipError_t e = hipMalloc((void **)&c->d_pix, npix(c) * 2 * sizeof(float4));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->d_pix = nullptr;
            return fail(c, PT_ERR_HIP, std::string("pt_trace: hipMalloc(primary hits): ") + hipGetErrorString(e));
        }
    }
    if (!c->d_gate) {                                // the drain gate's word; without it launches simply do not signal
        if (hipExtMallocWithFlags((void **)&c->d_gate, 8, hipMallocSignalMemory) != hipSuccess &&
            hipMalloc((void **)&c->d_gate, 8) != hipSuccess) c->d_gate = nullptr;
        (void)hipGetLastError();
        if (c->d_gate && hipStreamWriteValue64(c->stream, c->d_gate, 0, 0) != hipSuccess) {
            (void)hipGetLastError(); (void)hipFree(c->d_gate); c->d_gate = nullptr;
        }
    }
    if (c->timing) PT_HIP(c, hipEventRecord(c->ev0, c->stream));
    int launches = 0;
    int32_t left = iters;
    while (left > 0) {
        const int32_t n = left < MAX_SAMPLES_PER_LAUNCH ? left : MAX_SAMPLES_PER_LAUNCH;
        TraceArgs a{};
        a.canvas = c->d_canvas; a.segments = c->d_segments + (size_t)c->seg_slot * SEG_EPOCH;
        a.fresh = c->canvas_stale ? 1 : 0;           // the canvas is logically zero: do not read it
        scene_args(c, a);
        a.max_bounces = c->p.max_bounces; a.seed = c->p.seed;
        a.sample_begin = (uint32_t)(c->p.sample_offset + k_begin * c->p.sample_stride);
        a.sample_stride = (uint32_t)c->p.sample_stride;
        a.n_samples = n;
        a.partial = c->d_partial;
        a.pix = static_path ? c->d_pix : nullptr; a.pix_valid = a.pix && c->pix_valid ? 1 : 0;
        a.gate = c->d_gate; a.gate_value = c->gate_seq + 1;
        PT_HIP(c, launch_trace(a, c->h_geom.data(), reuse, c->has_specular,
                               (c->p.flags & PT_FLAG_LDS_TILES) != 0,
                               (c->p.flags & PT_FLAG_NO_STEAL) == 0, c->stream,
                               static_path ? c->scene_kernel : nullptr));
        if (a.pix) c->pix_valid = true;              // later launches on this stream read what this one stored
        if (c->d_gate) c->gate_seq += 1;
        c->canvas_stale = false;
        c->samples_done += n;
        k_begin += n;
        left -= n;
        ++launches;
    }
    if (c->timing) {
        PT_HIP(c, hipEventRecord(c->ev1, c->stream));
        c->last_launches = launches;
        c->have_timing = true;
    }
    return PT_OK;
}

extern "C" {

int pt_scene_specialise(pt_ctx *c)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return group_scene_specialise(c);
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "pt_scene_specialise: no scene uploaded");
    if (c->n_obj > MAX_JIT_OBJECTS)
        return fail(c, PT_ERR_UNSUPPORTED, "pt_scene_specialise: only scenes of <= 16 objects have a kernel to specialise");
    if (c->p.flags & (PT_FLAG_NO_STEAL | PT_FLAG_NO_PRIMARY_REUSE | PT_FLAG_LDS_TILES))
        return fail(c, PT_ERR_UNSUPPORTED, "pt_scene_specialise: the strict and tiled kernels are not specialised");
    std::string why;
    hipFunction_t fn = jit_static_kernel(c->p.device, c->n_obj, c->has_specular, c->h_geom.data(), why);
    if (!fn) return fail(c, PT_ERR_UNSUPPORTED, "pt_scene_specialise: " + why + " (the general kernel stays in use)");
    if (c->scene_kernel) {                       // called again (normally the same kernel): give the earlier hold back
        PT_HIP(c, hipStreamSynchronize(c->stream));
        jit_release(c->scene_kernel);
    }
    c->scene_kernel = fn;
    return PT_OK;
}

int pt_scene_kernel_ship(const pt_object *objs, int32_t n, const char *arch, const char *source_out, const char *code_in)
{
    if (!objs || n < 1 || n > MAX_JIT_OBJECTS || !arch) return PT_ERR_INVALID;
    std::vector<float4> geom((size_t)n * 3), mat((size_t)n * 4), rect((size_t)n * 2);
    std::vector<double> prim((size_t)n * PRIM_DOUBLES);
    bool spec = false;
    for (int k = 0; k < n; ++k) {
        if (objs[k].kind != PT_INFINITE_PLANE && objs[k].kind != PT_DISK && objs[k].kind != PT_SPHERE && objs[k].kind != PT_RECTANGLE)
            return PT_ERR_INVALID;
        flatten_object(objs[k], &geom[(size_t)k * 3], &mat[(size_t)k * 4], &prim[(size_t)k * PRIM_DOUBLES], &rect[(size_t)k * 2]);
        spec = spec || (objs[k].reflective && !(objs[k].emittance > 0.0)) || objs[k].kind == PT_SPHERE ||
               objs[k].kind == PT_RECTANGLE;                     // pt_scene_upload's rule for the general-scatter kernels
    }
    std::string why;
    const int rc = jit_ship(n, spec, geom.data(), arch, source_out, code_in, why);
    if (rc < 0) {
        std::fprintf(stderr, "pt_scene_kernel_ship: %s\n", why.c_str());
        return PT_ERR_INVALID;
    }
    return rc == 0 ? PT_OK : PT_ERR_UNSUPPORTED;             // (the query form: no such kernel shipped)
}

int pt_jit_cache_size(void) { return jit_cache_size(); }

int pt_jit_info(pt_ctx *c, char *buf, size_t size)
{
    if (!c || !buf || size == 0) return fail(c, PT_ERR_INVALID, "pt_jit_info: bad argument");
    pt_ctx *m = c->group ? group_first(c) : c;
    const std::string info = m->scene_kernel ? jit_kernel_info(m->scene_kernel) : std::string();
    std::snprintf(buf, size, "%s", info.c_str());
    return PT_OK;
}

} // extern "C"

// The canvas is zeroed lazily: the next pt_trace starts its sums from zero
// without reading the buffer (saving a 25-MB fill and a 25-MB read per frame at
// 1080p), and any read before that zeroes it first.  The segment counter moves
// to the next slot of a pre-zeroed ring instead of being cleared.
int pt::materialise_canvas(pt_ctx *c)
{
    if (c->canvas_stale) {
        PT_HIP(c, hipMemsetAsync(c->d_canvas, 0, npix(c) * 3 * sizeof(float), c->stream));
        c->canvas_stale = false;
    }
    return PT_OK;
}

// Conversion.repa3ToImage of `src` (a canvas of c's shape on c's device) into d_out, on `stream`
int pt::rgb8_from(pt_ctx *c, const float *src, void *d_out, int32_t mode, int64_t samples, hipStream_t stream)
{
    if (!d_out) return fail(c, PT_ERR_INVALID, "pt_rgb8_device: null output");
    if (mode < PT_RGB8_REFERENCE || mode > PT_RGB8_MEAN_GAMMA)
        return fail(c, PT_ERR_INVALID, "pt_rgb8_device: unknown mode");
    const double inv = samples > 0 ? 1.0 / (double)samples : 0.0;
    PT_HIP(c, launch_rgb8(src, (uint8_t *)d_out, c->p.height, c->p.width, mode, inv, stream));
    return PT_OK;
}

// What a read returns: the context's canvas, or -- for the head of a device group and for a rank context after
// pt_allreduce_sum -- the sum over all GPUs (src/Lib.hs:53-56: there is ONE `traces`, however many GPUs render it).
// `on` is the context whose stream orders the read.
int pt::read_source(pt_ctx *c, pt_ctx **on, const float **src)
{
    if (c->group) {
        if (int s = group_total(c, on)) return s;
        *src = (*on)->d_total;
        return check_ctx(*on);
    }
    if (int s = materialise_canvas(c)) return s;
    *on = c;
    *src = c->total_valid ? c->d_total : c->d_canvas;
    return PT_OK;
}

extern "C" {

int pt_reset(pt_ctx *c)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return group_reset(c);
    c->canvas_stale = true;
    c->total_valid = false;
    if (++c->seg_slot == SEG_SLOTS) {
        PT_HIP(c, hipMemsetAsync(c->d_segments, 0, (size_t)SEG_SLOTS * SEG_EPOCH * sizeof(unsigned long long), c->stream));
        c->seg_slot = 0;
    }
    c->samples_done = 0;
    return PT_OK;
}

int pt_read_sum_f32(pt_ctx *c, float *out)
{
    if (int s = check_ctx(c)) return s;
    if (!out) return fail(c, PT_ERR_INVALID, "pt_read_sum_f32: null output");
    pt_ctx *on; const float *src;
    if (int s = read_source(c, &on, &src)) return s;
    PT_HIP(c, hipMemcpyAsync(out, src, npix(c) * 3 * sizeof(float), hipMemcpyDeviceToHost, on->stream));
    PT_HIP(c, hipStreamSynchronize(on->stream));
    return PT_OK;
}

int pt_read_sum(pt_ctx *c, double *out)
{
    if (int s = check_ctx(c)) return s;
    if (!out) return fail(c, PT_ERR_INVALID, "pt_read_sum: null output");
    std::vector<float> tmp(npix(c) * 3);
    if (int s = pt_read_sum_f32(c, tmp.data())) return s;
    for (size_t k = 0; k < tmp.size(); ++k) out[k] = (double)tmp[k];
    return PT_OK;
}

// samples to divide by in the MEAN modes: the caller's, else the context's (a group's: the whole group's) count
} // extern "C"

int64_t pt::mean_samples(pt_ctx *c, int64_t samples_override)
{
    if (samples_override > 0) return samples_override;
    int64_t n = c->samples_done;
    if (c->group) (void)group_samples_done(c, &n);
    return n;
}

extern "C" {

int pt_rgb8_device(pt_ctx *c, void *d_out, int32_t mode, int64_t samples_override)
{
    if (int s = check_ctx(c)) return s;
    pt_ctx *on; const float *src;
    if (int s = read_source(c, &on, &src)) return s;
    return rgb8_from(c, src, d_out, mode, mean_samples(c, samples_override), on->stream);
}

int pt_rgb8_device_on(pt_ctx *c, void *d_out, int32_t mode, int64_t samples_override, void *hip_stream)
{
    if (int s = check_ctx(c)) return s;
    if (!hip_stream) return fail(c, PT_ERR_INVALID, "pt_rgb8_device_on: null stream");
    if (c->group)               // a group's sum is formed on its members' own streams, which the caller cannot order against
        return fail(c, PT_ERR_UNSUPPORTED, "pt_rgb8_device_on: not for a pt_create_multi context (use pt_rgb8_device)");
    if (c->canvas_stale)        // the lazy zero-fill would run on the context's stream, unordered with hip_stream
        return fail(c, PT_ERR_INVALID, "pt_rgb8_device_on: nothing traced since pt_reset");
    return rgb8_from(c, c->total_valid ? c->d_total : c->d_canvas, d_out, mode, mean_samples(c, samples_override),
                     (hipStream_t)hip_stream);
}

int pt_read_rgb8(pt_ctx *c, uint8_t *out, int32_t mode)
{
    if (int s = check_ctx(c)) return s;
    if (!out) return fail(c, PT_ERR_INVALID, "pt_read_rgb8: null output");
    const size_t bytes = npix(c) * 3;
    pt_ctx *on; const float *src;
    if (int s = read_source(c, &on, &src)) return s;
    if (!on->d_rgb) PT_HIP(c, hipMalloc((void **)&on->d_rgb, bytes));
    if (int s = rgb8_from(c, src, on->d_rgb, mode, mean_samples(c, 0), on->stream)) return s;
    PT_HIP(c, hipMemcpyAsync(out, on->d_rgb, bytes, hipMemcpyDeviceToHost, on->stream));
    PT_HIP(c, hipStreamSynchronize(on->stream));
    return PT_OK;
}

int pt_primary_hits(pt_ctx *c, int32_t *idx, double *t, double *normal)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) {
        pt_ctx *m = group_first(c);
        const int s = pt_primary_hits(m, idx, t, normal);
        if (s) c->err = m->err;
        return s;
    }
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "pt_primary_hits: no scene uploaded");
    // device staging for the three outputs (4 + 8 + 24 bytes per pixel), allocated once per context
    const size_t n = npix(c);
    if (!c->d_hits) PT_HIP(c, hipMalloc(&c->d_hits, n * 36));
    double *d_t = (double *)c->d_hits, *d_n = d_t + n;
    int32_t *d_idx = (int32_t *)(d_n + 3 * n);
    TraceArgs a{};
    scene_args(c, a);
    PT_HIP(c, launch_primary(a, idx ? d_idx : nullptr, t ? d_t : nullptr, normal ? d_n : nullptr, c->stream));
    if (idx) PT_HIP(c, hipMemcpyAsync(idx, d_idx, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (t) PT_HIP(c, hipMemcpyAsync(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (normal) PT_HIP(c, hipMemcpyAsync(normal, d_n, n * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_segments(pt_ctx *c, uint64_t *out)
{
    if (int s = check_ctx(c)) return s;
    if (!out) return fail(c, PT_ERR_INVALID, "pt_segments: null output");
    if (c->group) return group_segments(c, out);
    unsigned long long lanes[SEG_EPOCH];
    PT_HIP(c, hipMemcpyAsync(lanes, c->d_segments + (size_t)c->seg_slot * SEG_EPOCH, sizeof lanes, hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    for (int l = 0; l < SEG_LANES; ++l) v += lanes[l * SEG_LANE_STRIDE];
    *out = (uint64_t)v;
    return PT_OK;
}

int pt_bvh_counters(pt_ctx *c, uint64_t *node_visits, uint64_t *records)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) c = group_first(c);
    unsigned long long lanes[SEG_EPOCH];
    PT_HIP(c, hipMemcpyAsync(lanes, c->d_segments + (size_t)c->seg_slot * SEG_EPOCH, sizeof lanes, hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long v = 0, r = 0;
    for (int l = 0; l < SEG_LANES; ++l) { v += lanes[l * SEG_LANE_STRIDE + 1]; r += lanes[l * SEG_LANE_STRIDE + 2]; }
    if (node_visits) *node_visits = (uint64_t)v;
    if (records) *records = (uint64_t)r;
    return PT_OK;
}

int pt_bvh_padding_gathers(pt_ctx *c, uint64_t *padding)
{
    if (int s = check_ctx(c)) return s;
    if (!padding) return fail(c, PT_ERR_INVALID, "pt_bvh_padding_gathers: null argument");
    if (c->group) c = group_first(c);
    unsigned long long lanes[SEG_EPOCH];
    PT_HIP(c, hipMemcpyAsync(lanes, c->d_segments + (size_t)c->seg_slot * SEG_EPOCH, sizeof lanes, hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    for (int l = 0; l < SEG_LANES; ++l) v += lanes[l * SEG_LANE_STRIDE + 3];
    *padding = (uint64_t)v;
    return PT_OK;
}

int pt_samples_done(pt_ctx *c, int64_t *out)
{
    if (!c || !out) return fail(c, PT_ERR_INVALID, "pt_samples_done: null argument");
    if (c->group) return group_samples_done(c, out);
    *out = c->samples_done;
    return PT_OK;
}

int pt_canvas_device_ptr(pt_ctx *c, void **dptr, size_t *bytes)
{
    if (!c || !dptr) return fail(c, PT_ERR_INVALID, "pt_canvas_device_ptr: null argument");
    if (int s = check_ctx(c)) return s;
    if (c->group) {             // the group's sum, on the first device; formed and complete when this returns
        pt_ctx *on; const float *src;
        if (int s = read_source(c, &on, &src)) return s;
        PT_HIP(c, hipStreamSynchronize(on->stream));
        *dptr = (void *)src;
    } else {
        if (int s = materialise_canvas(c)) return s;
        *dptr = c->d_canvas;
    }
    if (bytes) *bytes = npix(c) * 3 * sizeof(float);
    return PT_OK;
}

int pt_set_stream(pt_ctx *c, void *hip_stream)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return fail(c, PT_ERR_UNSUPPORTED, "pt_set_stream: a pt_create_multi context launches on its members' own streams");
    PT_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PT_OK;
}

int pt_trace_gate(pt_ctx *c, pt_ctx *after)
{
    if (int s = check_ctx(c)) return s;
    if (!after || after == c) return fail(c, PT_ERR_INVALID, "pt_trace_gate: need another context");
    if (c->group || after->group) return fail(c, PT_ERR_UNSUPPORTED, "pt_trace_gate: not for pt_create_multi contexts");
    if (c->p.device != after->p.device) return fail(c, PT_ERR_INVALID, "pt_trace_gate: the contexts are on different devices");
    if (!after->d_gate || after->gate_seq == 0) return PT_OK;         // nothing traced there yet: nothing to wait for
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->p.device) != hipSuccess || !can) {
        (void)hipGetLastError();
        return fail(c, PT_ERR_UNSUPPORTED, "pt_trace_gate: this device cannot make a stream wait for a value (hipStreamWaitValue64)");
    }
    PT_HIP(c, hipStreamWaitValue64(c->stream, after->d_gate, after->gate_seq, hipStreamWaitValueGte, ~0ull));
    return PT_OK;
}

int pt_synchronize(pt_ctx *c)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return group_synchronize(c);
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_write_sum_f32(pt_ctx *c, const float *in, int64_t samples_done)
{
    if (int s = check_ctx(c)) return s;
    if (!in || samples_done < 0) return fail(c, PT_ERR_INVALID, "pt_write_sum_f32: bad argument");
    if (c->group) return group_write_sum_f32(c, in, samples_done);
    PT_HIP(c, hipMemcpyAsync(c->d_canvas, in, npix(c) * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    c->canvas_stale = false;
    c->total_valid = false;
    c->samples_done = samples_done;
    return PT_OK;
}

// ---- device-function parity hooks -------------------------------------------
int pt_dev_diffuse(pt_ctx *c, int32_t n, const float *normal, const float *u, float *out)
{
    if (c && c->group) c = group_first(c);
    if (int s = check_ctx(c)) return s;
    if (n < 1 || !normal || !u || !out) return fail(c, PT_ERR_INVALID, "pt_dev_diffuse: bad argument");
    std::vector<float4> mat((size_t)n * 4);
    float4 g[3]; double prim[PRIM_DOUBLES];
    for (int k = 0; k < n; ++k) {
        pt_object ob{};
        ob.kind = PT_INFINITE_PLANE;
        for (int q = 0; q < 3; ++q) ob.normal[q] = (double)normal[3 * k + q];
        flatten_object(ob, g, &mat[(size_t)k * 4], prim);
    }
    DevBuf<float4> dm; DevBuf<float> du, dout;
    PT_HIP(c, dm.alloc(mat.size())); PT_HIP(c, du.alloc((size_t)n * 2)); PT_HIP(c, dout.alloc((size_t)n * 3));
    PT_HIP(c, dm.up(mat.data(), mat.size())); PT_HIP(c, du.up(u, (size_t)n * 2));
    PT_HIP(c, launch_dev_diffuse(n, dm.p, du.p, dout.p, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, dout.down(out, (size_t)n * 3));
    return PT_OK;
}

int pt_dev_specular(pt_ctx *c, int32_t n, const float *vector, const float *normal, const float *u,
                    float noise, float *out)
{
    if (c && c->group) c = group_first(c);
    if (int s = check_ctx(c)) return s;
    if (n < 1 || !vector || !normal || !u || !out) return fail(c, PT_ERR_INVALID, "pt_dev_specular: bad argument");
    std::vector<float> nh((size_t)n * 3);
    for (int k = 0; k < n; ++k) {
        const d3 h = normalize_(d3{(double)normal[3 * k], (double)normal[3 * k + 1], (double)normal[3 * k + 2]});
        nh[3 * k] = (float)h.x; nh[3 * k + 1] = (float)h.y; nh[3 * k + 2] = (float)h.z;
    }
    const float noise_rev = noise >= 0.0f ? noise / 360.0f : NOISE_REV;
    DevBuf<float> dv, dn, du, dout;
    PT_HIP(c, dv.alloc((size_t)n * 3)); PT_HIP(c, dn.alloc((size_t)n * 3));
    PT_HIP(c, du.alloc((size_t)n * 2)); PT_HIP(c, dout.alloc((size_t)n * 3));
    PT_HIP(c, dv.up(vector, (size_t)n * 3)); PT_HIP(c, dn.up(nh.data(), (size_t)n * 3));
    PT_HIP(c, du.up(u, (size_t)n * 2));
    PT_HIP(c, launch_dev_specular(n, dv.p, dn.p, du.p, noise_rev, dout.p, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, dout.down(out, (size_t)n * 3));
    return PT_OK;
}

int pt_dev_distance(pt_ctx *c, int32_t n, const pt_object *obj, const float *origin, const float *vector,
                    float *t_out)
{
    if (c && c->group) c = group_first(c);
    if (int s = check_ctx(c)) return s;
    if (n < 1 || !obj || !origin || !vector || !t_out) return fail(c, PT_ERR_INVALID, "pt_dev_distance: bad argument");
    if (obj->kind != PT_INFINITE_PLANE && obj->kind != PT_DISK && obj->kind != PT_SPHERE && obj->kind != PT_RECTANGLE)
        return fail(c, PT_ERR_INVALID, "pt_dev_distance: unknown Form kind");
    float4 g[5], m[4]; double prim[PRIM_DOUBLES];                          // geometry record + the Rectangle sides
    flatten_object(*obj, g, m, prim, g + 3);
    DevBuf<float4> dg; DevBuf<float> dorg, dv, dt;
    PT_HIP(c, dg.alloc(5)); PT_HIP(c, dorg.alloc((size_t)n * 3)); PT_HIP(c, dv.alloc((size_t)n * 3));
    PT_HIP(c, dt.alloc((size_t)n));
    PT_HIP(c, dg.up(g, 5)); PT_HIP(c, dorg.up(origin, (size_t)n * 3)); PT_HIP(c, dv.up(vector, (size_t)n * 3));
    PT_HIP(c, launch_dev_distance(n, dg.p, dorg.p, dv.p, dt.p, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, dt.down(t_out, (size_t)n));
    return PT_OK;
}

int pt_dev_box_test(pt_ctx *c, int32_t n, const double *boxes, double pad, const float *origin, const float *vector,
                    const float *t_best, uint8_t *out)
{
    if (c && c->group) c = group_first(c);
    if (int s = check_ctx(c)) return s;
    if (n < 1 || !boxes || !origin || !vector || !t_best || !out || !(pad >= 0.0))
        return fail(c, PT_ERR_INVALID, "pt_dev_box_test: bad argument");
    std::vector<BvhNode> nodes((size_t)n);
    for (int k = 0; k < n; ++k) {                            // encoded as pt_scene_upload's builder encodes a node's box
        BvhBox b;
        for (int a = 0; a < 3; ++a) { b.lo[a] = boxes[6 * (size_t)k + a]; b.hi[a] = boxes[6 * (size_t)k + 3 + a]; }
        bvh_encode_box(b, pad, nodes[(size_t)k].h);
        nodes[(size_t)k].code = 0;
    }
    DevBuf<float4> dn; DevBuf<float> dorg, dv, dt; DevBuf<uint8_t> dout;
    PT_HIP(c, dn.alloc((size_t)n)); PT_HIP(c, dorg.alloc((size_t)n * 3)); PT_HIP(c, dv.alloc((size_t)n * 3));
    PT_HIP(c, dt.alloc((size_t)n)); PT_HIP(c, dout.alloc((size_t)n));
    PT_HIP(c, dn.up(reinterpret_cast<const float4 *>(nodes.data()), (size_t)n));
    PT_HIP(c, dorg.up(origin, (size_t)n * 3)); PT_HIP(c, dv.up(vector, (size_t)n * 3)); PT_HIP(c, dt.up(t_best, (size_t)n));
    PT_HIP(c, launch_dev_box(n, dn.p, dorg.p, dv.p, dt.p, dout.p, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, dout.down(out, (size_t)n));
    return PT_OK;
}

int pt_dev_philox(pt_ctx *c, int32_t n, const uint32_t *c0, const uint32_t *c1, uint32_t key, uint32_t *out)
{
    if (c && c->group) c = group_first(c);
    if (int s = check_ctx(c)) return s;
    if (n < 1 || !c0 || !c1 || !out) return fail(c, PT_ERR_INVALID, "pt_dev_philox: bad argument");
    DevBuf<uint32_t> d0, d1, dout;
    PT_HIP(c, d0.alloc((size_t)n)); PT_HIP(c, d1.alloc((size_t)n)); PT_HIP(c, dout.alloc((size_t)n * 2));
    PT_HIP(c, d0.up(c0, (size_t)n)); PT_HIP(c, d1.up(c1, (size_t)n));
    PT_HIP(c, launch_dev_philox(n, d0.p, d1.p, key, dout.p, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, dout.down(out, (size_t)n * 2));
    return PT_OK;
}

int pt_set_timing(pt_ctx *c, int32_t enable)
{
    if (!c) return fail(nullptr, PT_ERR_INVALID, "pt_set_timing: null context");
    if (c->group) return group_set_timing(c, enable);
    c->timing = enable != 0;
    c->have_timing = false;
    return PT_OK;
}

int pt_last_trace_ms(pt_ctx *c, float *ms, int32_t *launches)
{
    if (int s = check_ctx(c)) return s;
    if (!ms) return fail(c, PT_ERR_INVALID, "pt_last_trace_ms: null output");
    if (c->group) return group_last_trace_ms(c, ms, launches);
    if (!c->have_timing) return fail(c, PT_ERR_INVALID, "pt_last_trace_ms: no timed pt_trace yet");
    PT_HIP(c, hipEventSynchronize(c->ev1));
    PT_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    if (launches) *launches = c->last_launches;
    return PT_OK;
}

} // extern "C"
