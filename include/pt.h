/*
 * pt.h -- C ABI of libpt: the MI355X (gfx950) radiance loop of
 * ethanabrooks/pathtracer behind plain C linkage.
 *
 * The reference is pure Haskell and has NO FFI or plugin interface of its own
 * (SURVEY.md F2).  The seam this ABI replaces is the lazy list
 *     Lib.traces :: [Array D DIM3 Double]            (src/Lib.hs:53-56)
 * with its implicit inputs Params.* (src/Params.hs:3-11) and Object.objects
 * (src/Object.hs:147-148), and its consumer Conversion.repa3ToImage
 * (src/Conversion.hs:22-28).  A Haskell host binds these entry points with
 * `foreign import ccall safe` (INTEGRATION.md shows the stub); every citation
 * below names the reference code the entry point stands in for.
 * All file:line citations are relative to the reference repository root.
 *
 * Conventions: every function returns PT_OK (0) or a negative pt_status and
 * never throws or aborts across the boundary; all buffers are caller-allocated
 * and the library never frees or keeps caller memory; a pt_ctx is bound to one
 * GPU (a pt_create_multi context: to a list of GPUs) and is not thread-safe,
 * distinct contexts are independent.  There is no
 * CPU fallback: without a usable HIP device pt_create fails with
 * PT_ERR_NO_DEVICE.
 */
#ifndef PT_H
#define PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 3   /* 2: pt_object gained extent[2] (PT_RECTANGLE); 3: multi-GPU entry points */

typedef enum {
    PT_OK             =  0,
    PT_ERR_INVALID    = -1,   /* bad argument (null pointer, size out of range) */
    PT_ERR_NO_DEVICE  = -2,   /* no HIP device / device index out of range      */
    PT_ERR_HIP        = -3,   /* a HIP runtime call failed; see pt_last_error   */
    PT_ERR_NO_SCENE   = -4,   /* pt_trace / pt_primary_hits before pt_scene_upload */
    PT_ERR_LIMIT      = -5,   /* a documented limit exceeded (objects, samples, bounces) */
    PT_ERR_UNSUPPORTED = -6   /* an optional step does not apply here or cannot run; nothing changed (pt_scene_specialise) */
} pt_status;

/* Form constructors (src/Object.hs:51-56).  PT_SPHERE is an EXTENSION: the
 * reference only lists it as a TODO (src/Object.hs:191-194), so it has no
 * reference parity; this comment is its definition.  point = centre c,
 * radius = r > 0 (else never hit), normal unused.  For a ray o + t v (v not
 * unit, t a parameter), with oc = o - c:
 *     m = (oc.v)/(v.v);   l = oc - m v;   no hit unless l.l <= r^2;
 *     h = sqrt((r^2 - l.l)/(v.v));   t = -m - h if that is positive, else -m + h;
 * then distanceFrom's guard 0 < t < Inf (src/Object.hs:159).  The normal at a
 * hit point p is (p - c)/r, outward, never flipped toward the ray (as for the
 * reference's forms); diffuse and specular scatter use it as they use _normal.
 *
 * PT_RECTANGLE is an EXTENSION too: the reference carries it as a commented-out
 * constructor { _center, _normal, _height, _width } with an unfinished guard
 * (src/Object.hs:57-61, :172-175) and never says how the sides are oriented, so
 * this comment is its definition.  point = centre c, normal = _normal (raw, any
 * length), extent = { _height, _width }.  The distance is the guarded plane
 * distance through c, exactly as for a Disk (src/Object.hs:166-167); with
 * (theta', phi') = toSphericalCoords _normal (src/Util.hs:114-122), the sides run
 * along the images of the local x and y axes under the rotation rotateRel uses
 * for this normal (src/Util.hs:131-145):
 *     u = rotateAbs (1,0,0) theta' phi',   w = rotateAbs (0,1,0) theta' phi',
 * and the hit point p = march ray t (src/Object.hs:151-153) is inside when
 *     |(p - c).u| < _width / 2   and   |(p - c).w| < _height / 2
 * (strict, like the Disk's distanceFromCenter < radius).  A non-positive or NaN
 * side is never hit.  Scatter uses _normal as for a Disk. */
enum { PT_INFINITE_PLANE = 0, PT_DISK = 1, PT_SPHERE = 2, PT_RECTANGLE = 3 };

/* One Object with its Form flattened (src/Object.hs:30-36, :51-56).  Doubles,
 * as the Haskell records hold; the library converts to its fp32 SoA layout at
 * upload.  _name is not carried: the reference uses it only inside Eq Object
 * (src/Object.hs:38-40), which the device replaces by index identity. */
typedef struct {
    int32_t kind;          /* PT_INFINITE_PLANE | PT_DISK | PT_SPHERE | PT_RECTANGLE (the last two: extensions, see above) */
    int32_t reflective;    /* _reflective  (src/Object.hs:33)                  */
    double  point[3];      /* _point / _center (src/Object.hs:52,55)           */
    double  normal[3];     /* _normal, raw, not normalised (src/Object.hs:53,56) */
    double  radius;        /* _radius, Disk only (src/Object.hs:54)            */
    double  color[3];      /* _color       (src/Object.hs:31)                  */
    double  emittance;     /* _emittance   (src/Object.hs:32)                  */
    double  extent[2];     /* PT_RECTANGLE only: { _height, _width } (src/Object.hs:60-61); else ignored */
} pt_object;

/* Params.hs made a run-time value (src/Params.hs:3-11; numIters is the
 * argument of pt_trace), plus what a counter-based RNG and a multi-GPU sample
 * split need. */
typedef struct {
    int32_t  height;         /* Params.height (src/Params.hs:3)                 */
    int32_t  width;          /* Params.width  (src/Params.hs:5)                 */
    double   camera_depth;   /* Params.cameraDepth (src/Params.hs:7)            */
    int32_t  max_bounces;    /* Params.maxBounces (src/Params.hs:11); 1..64     */
    uint32_t seed;           /* Philox key (the reference seeds StdGen from 0, src/Lib.hs:34-39) */
    int32_t  device;         /* HIP device ordinal                              */
    int32_t  sample_offset;  /* this context renders global samples             */
    int32_t  sample_stride;  /*   offset + k*stride, k = 0,1,...  (1-GPU: 0, 1) */
    int32_t  flags;          /* PT_FLAG_*                                       */
} pt_params;

/* Trace every sample's primary segment in fp32 instead of reusing the
 * per-pixel fp64 primary hit (the primary ray has no jitter, src/Lib.hs:41-51,
 * so the hit is the same for every sample of a pixel). */
#define PT_FLAG_NO_PRIMARY_REUSE 1
/* Scenes of more than 8 objects: stage the object list in LDS tiles per
 * workgroup (streamed in lockstep beyond 1,024 objects) instead of streaming
 * it through scalar loads from L2.  Measured 5-12 % slower on MI355X
 * (DESIGN.md 3.1), so it is the option, not the default. */
#define PT_FLAG_LDS_TILES 2
/* Keep every lane on its own pixel for the whole launch (no wave-level work
 * sharing): a pixel's samples are then summed strictly in sample order, which
 * makes k calls of pt_trace(1) equal one call of pt_trace(k) bit for bit.  By
 * default idle lanes help other pixels of their wave and the two differ by
 * fp32 rounding of the per-pixel sum. */
#define PT_FLAG_NO_STEAL 4
/* Scenes of more than 8 objects: test every object against every ray, as
 * closestObjectTo does (src/Lib.hs:81-98), instead of pruning the finite forms
 * with the bounding-volume hierarchy built at pt_scene_upload.  The two find
 * the same object (same arithmetic per candidate, ties to the lower index);
 * this flag is the A/B and parity switch. */
#define PT_FLAG_NO_BVH 8

typedef struct pt_ctx pt_ctx;

/* Library / ABI identification. */
int         pt_abi_version(void);
const char *pt_status_string(int status);

/* Replaces the compile-time constants of src/Params.hs:3-11 and the canvas
 * set-up of src/Lib.hs:31-39 (blackCanvas, startingGens): allocates the fp32
 * radiance-sum canvas (H*W*3, zeroed) on params->device. */
int pt_create(pt_ctx **out, const pt_params *params);
void pt_destroy(pt_ctx *ctx);

/* Replaces the top-level value Object.objects (src/Object.hs:147-148).  Copies
 * and flattens `n` objects to the device SoA; index order is preserved because
 * closestObjectTo breaks ties by position (src/Lib.hs:84).  0 <= n <= 65536;
 * n = 0 (objs may be NULL) is the empty object list: closestObjectTo's guard
 * (src/Lib.hs:83) answers Nothing for every ray, so every sample is black and
 * pt_segments counts the one search per pixel that finds nothing. */
int pt_scene_upload(pt_ctx *ctx, const pt_object *objs, int32_t n);

/* The synthetic n-object scenes of BASELINE.json configs[3]/[4], generated as
 * SURVEY.md 8(d) specifies so that every host builds the same scene from (n,
 * seed): a SplitMix64 stream seeded with `seed`; object 0 the back plane z = 1500
 * (normal (0,0,-1), white); objects 1..n-1 of form `finite_kind` (PT_DISK: the
 * variant the reference's Form can express, src/Object.hs:51-56; PT_SPHERE,
 * PT_RECTANGLE: the extensions) with, drawn in this order per object: centre
 * uniform in x,y in [-400,400], z in [200,1200]; a uniformly random unit normal
 * (rejection from the unit ball; zeroed for spheres); radius U(5,40) (a
 * rectangle: height 2r, width 1.2r); colour U(0.2,1)^3; reflective with
 * probability 1/2; emissive with probability 0.02, emittance U(1,5).  configs[3]
 * is (1000, seed 1), configs[4] (10000, seed 2).  Host code: needs no GPU and no
 * context.  No counterpart in the reference (its scene is the constant `objects`,
 * src/Object.hs:147-148).  `out` holds n objects. */
int pt_synthetic_scene(pt_object *out, int32_t n, uint64_t seed, int32_t finite_kind);

/* `iters` applications of traceCanvas (src/Lib.hs:58-65): adds `iters` more
 * samples per pixel to the device canvas.  Sample indices continue from the
 * running count, so k calls of 1 render the same paths as one call of k (equal
 * up to fp32 summation order; bit for bit under PT_FLAG_NO_STEAL).  Returns
 * after the work is queued on the context's stream. */
int pt_trace(pt_ctx *ctx, int32_t iters);

/* Zero the canvas and the sample/segment counters (back to `traces !! 0`).
 * The device buffer is zeroed lazily (by the next pt_trace, which then starts
 * from zero without reading it, or by the next read): a host holding the raw
 * pointer of pt_canvas_device_ptr must not read it between pt_reset and the
 * next pt_trace. */
int pt_reset(pt_ctx *ctx);

/* Element `samples_done` of Lib.traces (src/Lib.hs:53-56): the SUM (not mean)
 * of all samples so far, out[(i*W + j)*3 + k], H*W*3 values
 * (layout of fromTripleArray, src/Util.hs:63-67).  Synchronises. */
int pt_read_sum(pt_ctx *ctx, double *out);
int pt_read_sum_f32(pt_ctx *ctx, float *out);

/* Conversion.repa3ToImage + listToPixelRGB8 (src/Conversion.hs:17-28):
 * out[(j*H + i)*3 + k], an image H pixels wide and W pixels tall.
 * PT_RGB8_REFERENCE: round-half-even(255 * sum), wrapped mod 256, no clamp, no
 * division by the sample count -- what the reference does.
 * PT_RGB8_MEAN_CLAMP: sum / samples, clamped to [0,1], then as above.
 * PT_RGB8_MEAN_GAMMA: as MEAN_CLAMP with x^(1/2.2) before quantising. */
enum { PT_RGB8_REFERENCE = 0, PT_RGB8_MEAN_CLAMP = 1, PT_RGB8_MEAN_GAMMA = 2 };
int pt_read_rgb8(pt_ctx *ctx, uint8_t *out, int32_t mode);

/* pt_read_rgb8 for callers that cannot allocate.  No counterpart in the
 * reference; it is for hosts that may not call hipMalloc where they read a frame:
 * a tight serving loop (one pt_trace(1) + one read per client message, the
 * pull-next-frame contract of app/Web.hs:42-65), harnesses that forbid
 * allocation.  (Not for HIP graph capture: like pt_read_rgb8 it copies to the
 * caller's host buffer and synchronises the stream.)  pt_read_rgb8 stages the quantised image in a
 * device buffer it allocates on first use; this variant stages it in a static
 * device array of 8 MiB that every GPU the library runs on holds until the
 * process ends, and calls no hipMalloc / hipFree or allocating library routine.
 * Same output, byte for byte.  Images above 8,388,608 bytes (H*W*3; 1920x1080
 * fits) are refused with PT_ERR_LIMIT.  All contexts on one GPU share that GPU's
 * array; the call holds a per-GPU lock from the kernel that fills it to the end
 * of the copy out of it, so calls from several contexts or threads on one GPU
 * take turns (each context still from one thread at a time, as everywhere). */
int pt_read_rgb8_noalloc(pt_ctx *ctx, uint8_t *out, int32_t mode);

/* Parity hook for rayFromCamToPixel + closestObjectTo on the primary ray
 * (src/Lib.hs:41-51, :81-98), evaluated in fp64 on the device: idx[i*W+j] is
 * the struck object or -1; t its ray parameter; normal[3] the struck form's
 * raw _normal (getNormal, src/Object.hs:180-183).  Any pointer may be NULL. */
int pt_primary_hits(pt_ctx *ctx, int32_t *idx, double *t, double *normal);

/* Counters: closestObjectTo executions (src/Lib.hs:81) so far, and samples
 * per pixel added so far by this context.  The primary ray has no jitter
 * (src/Lib.hs:41-51), so its search runs once per pixel for a camera and scene
 * -- by the first pt_trace after pt_scene_upload on scenes of <= 8 objects with
 * default flags, once per pt_trace otherwise -- and is counted when it runs. */
int pt_segments(pt_ctx *ctx, uint64_t *out);
int pt_samples_done(pt_ctx *ctx, int64_t *out);

/* Measurement hook for scenes searched through the spatial index (more than 8
 * objects, default flags): since the last pt_reset, how many index nodes the
 * per-lane walks visited (one 16-byte gather each) and how many 16-byte gathers
 * they spent on object records (three per record) -- the traffic that bounds
 * that kernel.  Zero for the other kernels; a pt_create_multi context reports its
 * first device.  Any pointer may be NULL.  No counterpart in the reference,
 * whose closestObjectTo is a linear scan (src/Lib.hs:81-98). */
int pt_bvh_counters(pt_ctx *ctx, uint64_t *node_gathers, uint64_t *record_gathers);
/* The part of record_gathers that fetched nothing useful: leaves' records are
 * fetched two slots at a time, and a leaf with an odd number of objects fetches
 * one empty slot (three gathers).  Algorithmic gathers = node_gathers +
 * record_gathers - this. */
int pt_bvh_padding_gathers(pt_ctx *ctx, uint64_t *padding_gathers);

/* Device-side access for hosts that own GPU memory (torch, RCCL): the canvas
 * device pointer (H*W*3 floats) for an all-reduce, the stream the context
 * launches on, and a device-resident RGB8 image.  pt_set_stream takes a
 * hipStream_t (NULL = the context's own stream). */
int pt_canvas_device_ptr(pt_ctx *ctx, void **dptr, size_t *bytes);
int pt_set_stream(pt_ctx *ctx, void *hip_stream);
int pt_synchronize(pt_ctx *ctx);
int pt_rgb8_device(pt_ctx *ctx, void *d_out, int32_t mode, int64_t samples_override);
/* The same output kernel (Conversion.repa3ToImage, src/Conversion.hs:22-28)
 * launched on a caller-named hipStream_t instead of the context's stream, for
 * hosts that overlap frame k's output (and its all-reduce) with the tracing of
 * frame k+1 on another context -- the pull-next-frame loop of app/Web.hs:42-65.
 * The caller orders `hip_stream` after the trace and the all-reduce (events);
 * fails with PT_ERR_INVALID if nothing was traced since pt_reset. */
int pt_rgb8_device_on(pt_ctx *ctx, void *d_out, int32_t mode, int64_t samples_override, void *hip_stream);

/* Host -> device canvas write, for the quantiser's parity test and for
 * checkpoint/resume of (sum, samples_done). */
int pt_write_sum_f32(pt_ctx *ctx, const float *in, int64_t samples_done);

/* Device-function parity hooks: run the device's fp32 scatter / intersection
 * code on `n` caller-supplied inputs (host arrays, 3 floats per vector).
 *  pt_dev_diffuse : diffuse  (src/Lib.hs:129-132) with uniforms u1,u2.
 *  pt_dev_specular: specular (src/Lib.hs:115-127), noise 40 unless noise>=0 given.
 *  pt_dev_distance: distanceFrom (src/Object.hs:156-177) of rays against ONE
 *                   object; t_out = NaN where the reference returns Nothing. */
int pt_dev_diffuse(pt_ctx *ctx, int32_t n, const float *normal, const float *u, float *out);
int pt_dev_specular(pt_ctx *ctx, int32_t n, const float *vector, const float *normal,
                    const float *u, float noise, float *out);
int pt_dev_distance(pt_ctx *ctx, int32_t n, const pt_object *obj, const float *origin,
                    const float *vector, float *t_out);
/* The spatial index's box test (an extension: the reference's closestObjectTo is a
 * linear scan, src/Lib.hs:81-98) on `n` caller-supplied boxes and rays: box k =
 * boxes[6k..6k+2] (lo) and [6k+3..6k+5] (hi) is encoded as pt_scene_upload encodes
 * a node's box (padded by `pad`, centre and half-extent in binary16) and tested
 * on the device against the ray (origin, vector) up to t_best[k]; out[k] = 1
 * where the walk would enter the box.  The index only prunes, so the property
 * to hold is one-sided: never 0 where the exact segment touches the box. */
int pt_dev_box_test(pt_ctx *ctx, int32_t n, const double *boxes, double pad, const float *origin,
                    const float *vector, const float *t_best, uint8_t *out);
/* Philox2x32-10 on the device for ctr=(c0[k], c1[k]), key: out[2k], out[2k+1]. */
int pt_dev_philox(pt_ctx *ctx, int32_t n, const uint32_t *c0, const uint32_t *c1,
                  uint32_t key, uint32_t *out);

/* Optional: compile the trace kernel for the uploaded scene.  No counterpart in
 * the reference, where `objects` is a compile-time constant of the program
 * anyway (src/Object.hs:147-148) -- this gives the device code the same
 * knowledge.  The general kernel takes the object list as arguments and
 * evaluates `(p - o).n / (v . n^)` (src/Object.hs:176-177) in full for every
 * object; this call compiles the same kernel source at run time (hiprtc) with
 * the scene's geometry as constants, so that terms with a zero coefficient
 * (walls along the coordinate axes, as all four objects of the default scene
 * are) are not computed.  The arithmetic that remains is the general kernel's
 * rounding for rounding: images agree bit for bit, except that a ray direction
 * with an exactly zero component may see -0 where the general form sees +0.
 * Costs a compilation (about a second) the first time a scene is seen: compiled
 * kernels are shared by the contexts of a process (pt_jit_cache_size) and kept
 * as files for later processes -- in $PT_KERNEL_CACHE, else
 * $XDG_CACHE_HOME/pathtracer_amd, else $HOME/.cache/pathtracer_amd; each file
 * spells out the kernel source, geometry, target and compiler it was built
 * from and is used only if all of it matches; PT_KERNEL_CACHE=off disables
 * the files.  Applies to scenes of <= 16 objects on the default flags (for 9-16
 * objects, whose list the general path streams through the scalar cache, the
 * scene kernel is the only by-value kernel: measured 2.3x the streamed rate);
 * the kernel is dropped by the next pt_scene_upload.  A kernel shipped with the
 * library for this scene (pt_scene_kernel_ship) is used before any compiler is
 * looked for.  On PT_ERR_UNSUPPORTED (no libhiprtc on the machine, a scene
 * or flags it does not cover) the context keeps using the general kernel.
 * PT_SPECIALISE=1 in the environment makes pt_scene_upload do this for every
 * scene it applies to (failures ignored). */
int pt_scene_specialise(pt_ctx *ctx);
/* Which compiler built the context's scene kernel, as text: hiprtc's version
 * and file, the target, and whether it had __builtin_amdgcn_inverse_ballot_w64
 * (a process that loaded PyTorch's bundled ROCm first compiles with that older
 * hiprtc, which lacks it; the kernel then uses a v_cndmask_b32 form of the
 * same decision).  Empty when the general kernel is in use. */
int pt_jit_info(pt_ctx *ctx, char *buf, size_t size);
/* How many scene kernels the process holds loaded right now.  A kernel is shared
 * by every context with that scene on that device, stays loaded after its last
 * user has gone (the next one skips the compile), and is unloaded once more
 * than 16 unused ones have piled up, least recently used first. */
int pt_jit_cache_size(void);
/* Build support, no GPU needed: scene kernels SHIPPED with the library.  The
 * reference's scene is a constant of the program (src/Object.hs:147-148), so
 * its kernel can be compiled when the library is built and found by
 * pt_scene_specialise without libhiprtc and without the compile: files in
 * <directory of libpt.so>/scene_kernels/, in the format of the kernel cache
 * (each spells out source, geometry and target and is used only where all of
 * it matches; the compiler is not part of the key -- the file is the build's
 * own artefact).  With source_out, writes the kernel source of the scene
 * `objs` to that file (compile it with the Makefile's flags and -I csrc); with
 * code_in, stores the code object compiled from it for target `arch` (the
 * --offload-arch value, e.g. "gfx950:sramecc+:xnack-") in that directory.
 * `make scene_kernels` / tools/build_scene_kernels.py do both for the default
 * scene.  With neither, a query: PT_OK if a kernel for this scene and target is
 * shipped, PT_ERR_UNSUPPORTED if not.  PT_SHIPPED_KERNELS=off in the environment
 * ignores the directory; PT_NO_HIPRTC=1 keeps pt_scene_specialise from loading a
 * compiler (shipped kernels only). */
int pt_scene_kernel_ship(const pt_object *objs, int32_t n, const char *arch,
                         const char *source_out, const char *code_in);

/* Last kernel's launch duration in milliseconds, measured with HIP events on
 * the context's stream around the most recent pt_trace (all its launches).
 * Enabled by pt_set_timing(ctx, 1); synchronises when read. */
int pt_set_timing(pt_ctx *ctx, int32_t enable);
int pt_last_trace_ms(pt_ctx *ctx, float *ms, int32_t *launches);

/* Frames in flight on one GPU (the pull-next-frame loop of app/Web.hs:42-65 made
 * asynchronous: frame k+1 is traced into one context while frame k's all-reduce
 * and output run on another).  Two trace launches that run at the same time on
 * two streams share the GPU evenly, finish together and leave it idle together
 * while their last waves end; what a frame loop wants is that the next launch
 * starts exactly when the current one has no work left to hand out.
 * pt_trace_gate(ctx, after) queues on ctx's stream a wait for that moment of
 * `after`'s most recent pt_trace (its last launch: the workgroup dispatched last
 * writes a word the wait polls; hipStreamWaitValue64): what is queued on ctx's
 * stream afterwards -- the next frame's pt_trace -- then fills the wave slots
 * that `after`'s launch leaves as it drains.  Call it after `after`'s pt_trace
 * has been called (the wait names work that is already queued, so no cycle can
 * form); `after` must outlive the wait.  Same device only.  PT_OK and no wait if
 * `after` has traced nothing yet; PT_ERR_UNSUPPORTED where the device cannot
 * make a stream wait for a value.  No counterpart in the reference. */
int pt_trace_gate(pt_ctx *ctx, pt_ctx *after);

/* ---- Multi-GPU ----------------------------------------------
 * The reference has ONE `traces` (src/Lib.hs:53-56), read by app/Write.hs:24-28
 * and app/Web.hs:42-65; these entry points keep it one however many GPUs render
 * it.  The work shards by sample: the random stream is keyed by (pixel, global
 * sample index, bounce), so any partition of a frame's sample indices over GPUs
 * renders the same set of paths (equal up to fp32 summation order), and ONE
 * RCCL all-reduce (sum) of the fp32 canvases over xGMI combines them.  No
 * counterpart in the reference (it is single-threaded).  librccl is
 * opened on first use; without it these calls return PT_ERR_UNSUPPORTED.
 *
 * One process, several devices: pt_create_multi is pt_create for a list of HIP
 * device ordinals (params->device is ignored).  The context it returns is used
 * exactly like any other: pt_scene_upload and pt_scene_specialise apply to every
 * device; pt_trace(ctx, k) splits the call's k sample indices into contiguous
 * blocks, one per device (k calls of 1 thus take turns on the first device --
 * shard frames of many samples, not single ones); the first read after it
 * (pt_read_sum*, pt_read_rgb8, pt_rgb8_device, pt_canvas_device_ptr) performs
 * the all-reduce (ncclAllReduce per device, fp32 sum of H*W*3 values, out of
 * place, so each device's canvas stays its partial sum and the render can go
 * on) and returns the total from the first device; pt_segments sums the
 * devices' counters.  A device may be listed up to 8 times (its members are
 * then added by a kernel before the collective); pt_set_stream and
 * pt_rgb8_device_on do not apply (PT_ERR_UNSUPPORTED). */
int pt_create_multi(pt_ctx **out, const pt_params *params, const int32_t *devices, int32_t n_devices);

/* One process per GPU: every rank creates an ordinary context with
 * sample_offset = rank, sample_stride = n_ranks (or any other disjoint split).
 * Rank 0 calls pt_comm_unique_id and hands the PT_COMM_ID_BYTES bytes to the
 * others by its own means (MPI, a file, torch.distributed); every rank then
 * calls pt_comm_init_rank (collective: it returns when all n_ranks have
 * joined).  pt_allreduce_sum queues ncclAllReduce(sum) of the canvas on the
 * context's stream into a second buffer; from then until the next pt_trace,
 * pt_reset or pt_write_sum_f32 the reads of this context return the total of
 * all ranks (in the MEAN modes pass the frame's whole sample count as
 * samples_override).  pt_canvas_device_ptr keeps naming the rank's own partial
 * sum.  Every rank must call pt_allreduce_sum the same number of times. */
#define PT_COMM_ID_BYTES 128
int pt_comm_unique_id(void *id);
int pt_comm_init_rank(pt_ctx *ctx, const void *id, int32_t rank, int32_t n_ranks);
int pt_allreduce_sum(pt_ctx *ctx);
/* A second context of the same rank, on the same device and of the same canvas
 * size (frames in flight: frame k+1 is traced into one canvas while frame k's
 * all-reduce runs on another), uses `owner`'s communicator instead of creating
 * one of its own: one set of RCCL channels and proxy threads per rank, one
 * collective initialisation.  RCCL runs the all-reduces of one communicator in
 * the order they were queued; every rank must queue them in the same order over
 * all the contexts that share it.  The communicator lives until the last
 * context using it is destroyed.  No counterpart in the reference. */
int pt_comm_share(pt_ctx *ctx, pt_ctx *owner);
/* What a context's collective runs over: RCCL ranks, this context's rank, the
 * devices driven from this process, and RCCL's version code (0: no
 * communicator -- an ordinary single-GPU context).  Any pointer may be NULL. */
int pt_comm_info(pt_ctx *ctx, int32_t *n_ranks, int32_t *rank, int32_t *n_local, int32_t *rccl_version);

const char *pt_last_error(pt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PT_H */
