// pt_internal.hpp -- launch-parameter structs shared by the kernels and the
// C ABI implementation.  Not part of the public interface (include/pt.h is).
#pragma once

#ifndef __HIPCC_RTC__          // hiprtc brings its own runtime header and the fixed-width integer types
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "pt_device.hpp"

namespace pt {

// Geometry of a small scene passed BY VALUE in the kernel-argument segment:
// the closest-hit loop indexes objects with a wave-uniform index, so hipcc
// fetches these with s_load into SGPRs and the VALU reads them as scalar
// operands -- no VGPRs and no LDS bandwidth spent on the object list.
template <int NS> struct StaticGeom {
    float4 g[NS > 0 ? NS : 1][3];
};

// Segment counter layout: SEG_LANES 64-bit counters per epoch, one per 128-byte line.
constexpr int MAX_SLICES = 2;
#ifndef PT_SLICE_MIN
#define PT_SLICE_MIN 512
#endif
constexpr int SLICE_MIN_SAMPLES = PT_SLICE_MIN;        // smaller launches are not sliced: the second set-up pass costs what the
                                              // drain saves (and more when consecutive frames overlap: 256 spp +2 %, 128 spp +8 %)
constexpr int BVH_SLICE_MIN_SAMPLES = 256;    // trace_bvh_kernel: its one-wave workgroups live longer per sample, so slicing pays earlier
                                              // (4K, serial launches: 256 spp -5...7 %, 128 spp -3 %; pipelined frames 256 spp -0.4 %, 128 spp +2 %)
constexpr int SEG_LANES = 64;
constexpr int SEG_LANE_STRIDE = 16;                 // in counters (128 bytes)
constexpr int SEG_EPOCH = SEG_LANES * SEG_LANE_STRIDE;

struct TraceArgs {
    float              *canvas;        // H*W*3 fp32 radiance sums, (i*W+j)*3+k
    unsigned long long *segments;      // closest-hit searches executed: SEG_LANES counters, SEG_LANE_STRIDE apart
    const float4       *geom;          // n_obj * 3 float4   (dynamic path)
    const float4       *mat;           // n_obj * 4 float4
    const double       *prim;          // n_obj * PRIM_DOUBLES
    const float4       *rect;          // n_obj * 2 float4, the sides of Rectangle forms (extension); null without one
    double   camera_depth;
    int      n_obj;
    int      height, width;
    int      max_bounces;
    uint32_t seed;
    uint32_t sample_begin;             // global index of this launch's first sample
    uint32_t sample_stride;
    int      n_samples;                // samples per pixel in this launch
    int      fresh;                    // 1: the canvas is logically zero, start the sums at 0 without reading it
    int      tile_objs;                // objects per LDS tile (tiled kernel; set by the launcher)
    // spatial index over the finite forms (pt_bvh.hpp); n_nodes == 0: none, linear scan
    const float4  *bvh_nodes;          // 1 float4 per node: centre and half-extent as six binary16 + code (pt_bvh.hpp); node 0 unused, root = node 1
    const int32_t *bvh_prims;          // [unbounded objects..., finite objects in leaf order...]
    const float4  *bvh_geom;           // geometry records in bvh_prims order (3 float4 each, object index in [2].w), padded
    // trace_bvh_kernel's compact form, present when every finite form's raw normal is already unit (n_raw = n_hat
    // bit for bit in fp32; spheres have none): TWO float4 per finite object in leaf order -- (n_hat, A) (centre, r^2 |
    // -r^2 | NaN) -- so that a record costs two 16-byte gathers instead of three.  A record no longer carries its
    // object's index: inside that kernel an object is named by its POSITION in bvh_prims, and materials, rectangle
    // sides and the object -> position map are given in that order.
    const float4  *bvh_geom32;         // (n_prims - n_unbounded + pad) * 2 float4, or null: the 48-byte records
    const float4  *bvh_mat;            // 4 float4 per position
    const float4  *bvh_rect;           // 2 float4 per position, or null
    const int32_t *bvh_pos;            // position of object k
    // Sample slices of one launch (trace_static_kernel): slice y = blockIdx.y traces samples
    // [slice_j0[y], slice_j0[y] + slice_n[y]) of the launch's n_samples; slice 0 continues the canvas, slice
    // y > 0 writes plane y - 1 of `partial`, and slice_sum_kernel adds the planes to the canvas in order.
    // The primary hits (fp64 pass, per pixel: point, direction, record offset) of this camera and scene, kept
    // from the first launch for all later ones (trace_static_kernel): two planes of one float4 per pixel
    float4  *pix;
    int      pix_valid;                // 1: read them, 0: compute (and store if pix != null)
    float   *partial;                  // (MAX_SLICES - 1) * H*W*3 floats, or null: no slicing
    int      n_slices;
    int      slice_j0[MAX_SLICES], slice_n[MAX_SLICES];
    // The drain gate (pt_trace_gate, include/pt.h): the workgroup of a trace launch that is dispatched LAST writes
    // gate_value to *gate as it starts -- from then on the launch has no work left to hand out and its waves only
    // finish, so a launch waiting for that value on another stream fills the slots they leave.  Null: no signal.
    unsigned long long *gate;
    unsigned long long gate_value;
    int      n_nodes;
    int      bvh_depth;                // deepest leaf: the walk's stack holds at most this many far children
    int      n_unbounded;
};

#ifndef __HIPCC_RTC__
// Launchers (pt_kernels.hip).  All return the hipError_t of the launch.
// `scene_kernel`: trace_static_kernel compiled for this scene (jit_static_kernel), launched in place of the
// general one where that would run; null: none.
hipError_t launch_trace(const TraceArgs &a, const float4 *geom_host, bool reuse_primary,
                        bool has_specular, bool lds_tiles, bool steal, hipStream_t s,
                        hipFunction_t scene_kernel = nullptr);
hipError_t launch_primary(const TraceArgs &a, int32_t *idx, double *t, double *normal, hipStream_t s);
hipError_t launch_rgb8(const float *canvas, uint8_t *rgb, int H, int W, int mode,
                       double inv_samples, hipStream_t s);
hipError_t launch_dev_diffuse(int n, const float4 *mat1, const float *u, float *out, hipStream_t s);
hipError_t launch_dev_specular(int n, const float *v, const float *nrm, const float *u,
                               float noise_rev, float *out, hipStream_t s);
hipError_t launch_dev_distance(int n, const float4 *geom1_rect, const float *o, const float *v,
                               float *t, hipStream_t s);
hipError_t launch_dev_box(int n, const float4 *nodes, const float *o, const float *v, const float *t_best, uint8_t *out, hipStream_t s);
hipError_t launch_dev_philox(int n, const uint32_t *c0, const uint32_t *c1, uint32_t key,
                             uint32_t *out, hipStream_t s);
#endif

#ifndef __HIPCC_RTC__
} // namespace pt
#include <string>
namespace pt {
// pt_jit.hip: trace_static_kernel<NS, spec> compiled by hiprtc with this geometry (n_obj * 3 float4, as uploaded)
// as constants, loaded on `device`; cached per process.  Null and a reason in `err` on failure.
hipFunction_t jit_static_kernel(int device, int n_obj, bool spec, const float4 *geom_host, std::string &err);
// what compiled it: "hiprtc <major>.<minor> (<library file>), --offload-arch=..., __builtin_amdgcn_inverse_ballot_w64 available|missing"
std::string jit_kernel_info(hipFunction_t fn);
// A context that stops using a kernel from jit_static_kernel (scene change, pt_destroy) gives it back -- after its
// stream has drained.  Unused kernels beyond a small number are unloaded.
int jit_ship(int n_obj, bool spec, const float4 *geom_host, const char *arch, const char *src_out, const char *code_in,
             std::string &err);
void jit_release(hipFunction_t fn);
int jit_cache_size();
#endif

} // namespace pt
