// pt_ctx.hpp -- the context behind include/pt.h's opaque pt_ctx, shared by the
// C ABI implementation (pt_api.hip) and the multi-GPU layer (pt_multi.hip).
// Not part of the public interface.
#pragma once

#include "../../include/pt.h"
#include "pt_internal.hpp"

#include <memory>
#include <string>
#include <vector>

struct pt_group;                       // pt_multi.hip: the members of a pt_create_multi context

struct pt_ctx {
    pt_params p{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    float *d_canvas = nullptr;
    unsigned long long *d_segments = nullptr;   // ring of SEG_SLOTS counters, one per pt_reset epoch
    int seg_slot = 0;
    bool canvas_stale = false;                   // pt_reset defers the zeroing to the next pt_trace / read
    float4 *d_geom = nullptr, *d_mat = nullptr;
    double *d_prim = nullptr;
    float4 *d_pix = nullptr;           // primary hits kept across launches (TraceArgs::pix); valid until the next scene upload
    bool pix_valid = false;
    unsigned long long *d_gate = nullptr;   // the drain gate's word (TraceArgs::gate), signal memory where the runtime has it
    unsigned long long gate_seq = 0;        // value the most recent trace launch of this context writes there
    float *d_partial = nullptr;        // sample-slice planes of a launch (pt_internal.hpp), allocated at the first large pt_trace
    float4 *d_rect = nullptr;          // sides of Rectangle forms, 2 float4 per object; null without one
    float4 *d_bvh_nodes = nullptr;               // spatial index over the finite forms (pt_bvh.hpp), or null
    int32_t *d_bvh_prims = nullptr;
    float4 *d_bvh_geom = nullptr;
    float4 *d_bvh_geom32 = nullptr, *d_bvh_mat = nullptr, *d_bvh_rect = nullptr;   // the indexed kernel's compact form (TraceArgs::bvh_geom32), or null
    int32_t *d_bvh_pos = nullptr;
    int n_nodes = 0, n_unbounded = 0, bvh_depth = 0;
    uint8_t *d_rgb = nullptr;
    void *d_hits = nullptr;            // device staging of pt_primary_hits (idx, t, normal: 36 B per pixel), kept between calls
    int n_obj = 0;
    bool have_scene = false;           // pt_scene_upload has run (an EMPTY scene is a scene: src/Lib.hs:83, every ray misses)
    bool has_specular = false;
    std::vector<float4> h_geom;         // host copy: small scenes ride in the kernel arguments
    hipFunction_t scene_kernel = nullptr;   // trace_static_kernel compiled for this scene (pt_scene_specialise), or null
    int64_t samples_done = 0;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int last_launches = 0;
    bool have_timing = false;
    std::string err;

    // ---- multi-GPU (pt_multi.hip) ----
    pt_group *group = nullptr;         // non-null: the head of a single-process device group (pt_create_multi);
                                       // a head owns no device memory of its own
    void *comm = nullptr;              // ncclComm_t of a process-per-GPU context (pt_comm_init_rank, pt_comm_share)
    std::shared_ptr<void> comm_ref;    // owns it: the communicator is destroyed when the last context that uses it lets go
    int comm_rank = 0, comm_size = 1;
    float *d_total = nullptr;          // the all-reduced canvas (the context's own canvas stays its partial sum)
    bool total_valid = false;          // d_total holds the sum over all ranks of the current canvases
    hipEvent_t ev_sync = nullptr;      // stream-to-stream ordering inside a group
};

namespace pt {

constexpr int64_t MAX_GLOBAL_SAMPLES = (int64_t)1 << 26;   // sample index shares a Philox word with a 6-bit bounce
constexpr int     MAX_BOUNCES = 64;
constexpr int     MAX_OBJECTS = 65536;
constexpr int     MAX_SAMPLES_PER_LAUNCH = 4096;
constexpr int     SEG_SLOTS = 64;                 // pt_reset epochs before the ring is cleared

int fail(pt_ctx *c, int status, const std::string &msg);
int check_ctx(pt_ctx *c);
inline size_t npix(const pt_ctx *c) { return (size_t)c->p.height * (size_t)c->p.width; }

#define PT_HIP(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return pt::fail((c), PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// pt_api.hip, used by the group layer
int trace_block(pt_ctx *c, int64_t k_begin, int32_t iters);    // samples offset + k*stride, k in [k_begin, k_begin + iters)
int materialise_canvas(pt_ctx *c);
int rgb8_from(pt_ctx *c, const float *src, void *d_out, int32_t mode, int64_t samples, hipStream_t stream);
int read_source(pt_ctx *c, pt_ctx **on, const float **src);   // what a read returns (canvas or all-reduced total) and on whose stream
int64_t mean_samples(pt_ctx *c, int64_t samples_override);
int scratch_init(pt_ctx *c, int device);                       // pt_noalloc.hip

// pt_multi.hip: what the C ABI entry points do for the head of a device group
int  group_scene_upload(pt_ctx *head, const pt_object *objs, int32_t n);
int  group_scene_specialise(pt_ctx *head);
int  group_trace(pt_ctx *head, int32_t iters);
int  group_reset(pt_ctx *head);
int  group_total(pt_ctx *head, pt_ctx **leader);             // all-reduce if needed; leader = the member whose d_total to read
int  group_segments(pt_ctx *head, uint64_t *out);
int  group_samples_done(pt_ctx *head, int64_t *out);
int  group_synchronize(pt_ctx *head);
int  group_write_sum_f32(pt_ctx *head, const float *in, int64_t samples_done);
int  group_set_timing(pt_ctx *head, int32_t enable);
int  group_last_trace_ms(pt_ctx *head, float *ms, int32_t *launches);
pt_ctx *group_first(pt_ctx *head);                            // member 0 (parity hooks run there)
void group_destroy(pt_ctx *head);
void comm_destroy(pt_ctx *c);                                 // a rank context's communicator and d_total

} // namespace pt
