/*
 * pt_oracle.h -- CPU fp64 restatement of the reference's radiance loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (libpt.so, the
 * pathtracer_amd package, the C++ host) may include, link or call this.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / the reported CPU baseline.
 *
 * It is a RESTATEMENT of ethanabrooks/pathtracer (Haskell), not the reference
 * itself: no GHC toolchain exists in the build container, so the reference
 * cannot be compiled or run.  What pins it:
 *   - the known answers and properties of the reference's own test/Spec.hs
 *     (tests/test_oracle_spec.py restates all 18),
 *   - analytic known answers derived from src/Object.hs:104-148 and
 *     src/Lib.hs:41-51 (tests/golden/),
 *   - the widely published first output of System.Random's StdGen
 *     (next (mkStdGen 0) = 2147482884, state "40014 40692").
 * End-to-end images are PARITY UNPINNED by the reference (it holds no golden
 * image and no test of traces/terminalColor/closestObjectTo).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Triple.hs ---------------------------------------------------------- */
typedef struct { double x, y, z; } o_vec3;

double o_dot(o_vec3 a, o_vec3 b);            /* src/Triple.hs:47-50  */
o_vec3 o_cross(o_vec3 a, o_vec3 b);          /* src/Triple.hs:52-59  */
double o_norm2(o_vec3 v);                    /* src/Triple.hs:61-64  */
o_vec3 o_normalize(o_vec3 v);                /* src/Triple.hs:66-69  */

/* ---- Util.hs (angles are in DEGREES, as AC-Angle's Degrees newtype) ------ */
double o_arctan2(double x, double y);                         /* src/Util.hs:106-112 */
void   o_to_spherical(o_vec3 v, double *theta, double *phi);  /* src/Util.hs:114-122 */
o_vec3 o_from_spherical(double theta, double phi);            /* src/Util.hs:124-129 */
o_vec3 o_rotate_abs(o_vec3 v, double theta, double phi);      /* src/Util.hs:131-138 */
o_vec3 o_rotate_rel(double theta, double phi, o_vec3 v);      /* src/Util.hs:140-145 */

/* ---- Object.hs ---------------------------------------------------------- */
enum { O_INFINITE_PLANE = 0, O_DISK = 1, O_SPHERE = 2 /* extension, no reference */,
       O_RECTANGLE = 3 /* extension: commented out in the reference, src/Object.hs:57-61, :172-175 */ };

/* Same field order and meaning as the C ABI's pt_object (include/pt.h) so a
 * test can hand one byte image to both. */
typedef struct {
    int32_t kind;
    int32_t reflective;
    double  point[3];    /* _point / _center   src/Object.hs:52,55 */
    double  normal[3];   /* _normal (raw)      src/Object.hs:53,56 */
    double  radius;      /* _radius            src/Object.hs:54    */
    double  color[3];    /* _color             src/Object.hs:31    */
    double  emittance;   /* _emittance         src/Object.hs:32    */
    double  extent[2];   /* O_RECTANGLE: { _height, _width }  src/Object.hs:60-61 (commented out there) */
} o_object;

/* The reference's hard-coded scene: objects = [infPlane, infPlane2, infPlane3,
 * light]  (src/Object.hs:104-148). Writes 4 objects. */
int o_default_scene(o_object *out4);

/* distanceFrom' / distanceFrom (src/Object.hs:156-177).  Return 1 and *t when
 * the Maybe is Just, 0 when Nothing. */
int o_distance_from_prime(o_vec3 origin, o_vec3 vector, const o_object *form, double *t);
/* getNormal (src/Object.hs:180-183); for the O_SPHERE extension the outward normal at `point`. */
o_vec3 o_normal_at(const o_object *form, o_vec3 point);
int o_distance_from(o_vec3 origin, o_vec3 vector, const o_object *form, double *t);
/* O_RECTANGLE (extension): the unit vectors its _width (u) and _height (w) sides run along */
void o_rectangle_axes(const o_object *form, o_vec3 *u, o_vec3 *w);
o_vec3 o_march(o_vec3 origin, o_vec3 vector, double t);       /* src/Object.hs:151-153 */

/* ---- random number streams --------------------------------------------- */
/* System.Random 1.1 StdGen (L'Ecuyer 1988), restated from the published
 * package source (not vendored under /root/reference; stack.yaml:5-9 pins the
 * snapshot).  */
typedef struct { int32_t s1, s2; } o_stdgen;
o_stdgen o_mk_stdgen(int64_t seed);
int32_t  o_stdgen_next(o_stdgen *g);
int64_t  o_stdgen_random_int(o_stdgen *g);                    /* random :: Int          */
double   o_stdgen_random_double(o_stdgen *g);                 /* random :: Double [0,1) */
double   o_stdgen_randomR(o_stdgen *g, double l, double h);   /* randomR (l,h) :: Double */

/* Philox2x32-10 (Salmon et al., SC'11), the device's generator.  ctr/key in,
 * two 32-bit words out. */
void o_philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t out[2]);

enum { O_RNG_STDGEN = 0, O_RNG_PHILOX = 1 };

/* ---- Lib.hs ------------------------------------------------------------- */
typedef struct {
    int32_t height, width;    /* src/Params.hs:3-5  */
    double  camera_depth;     /* src/Params.hs:7    */
    int32_t max_bounces;      /* src/Params.hs:11   */
    int32_t rng;              /* O_RNG_STDGEN (reference stream) or O_RNG_PHILOX (device stream) */
    uint32_t seed;            /* Philox key; ignored for StdGen (reference seeds from mkStdGen 0) */
} o_params;

/* specular (src/Lib.hs:115-127) and diffuse (:129-132) with the two uniforms
 * u1,u2 in [0,1) supplied by the caller: theta = randomR(0,maxTheta),
 * phi = randomR(0,380) evaluated as System.Random does,
 * 2*(0.5*l + u*(0.5*h - 0.5*l)). */
o_vec3 o_specular_u(double u1, double u2, double noise, o_vec3 vector, o_vec3 normal);
o_vec3 o_diffuse_u(double u1, double u2, o_vec3 normal);

/* rayFromCamToPixel (src/Lib.hs:41-51): direction only; origin is 0. */
o_vec3 o_primary_direction(const o_params *p, int i, int j);

/* closestObjectTo (src/Lib.hs:81-98).  last = index of _lastStruck or -1.
 * Returns the winning index or -1; *t its distance. */
int o_closest_object_to(const o_object *objs, int n, o_vec3 origin, o_vec3 vector,
                        int last, double *t);

/* Primary hits for the whole canvas: idx[i*W+j] (-1 = miss), t, normal[3]
 * (the RAW _normal of the struck form, getNormal src/Object.hs:180-183). */
void o_primary_hits(const o_params *p, const o_object *objs, int n,
                    int32_t *idx, double *t, double *normal);

/* traces !! iters restricted to samples [s_begin, s_begin+iters) of every
 * pixel (src/Lib.hs:53-65): sum[(i*W+j)*3+k] += radiance.  For StdGen the
 * per-pixel generator state is carried in `gens` (H*W entries; initialised by
 * o_starting_gens) exactly as traceCanvas threads it.  For Philox the draw for
 * (pixel, sample s, bounce b) is philox(ctr=(pixel, s<<6|b), key=seed).
 * sumsq (may be NULL) accumulates per-channel squares for variance estimates.
 * segments (may be NULL) += number of closestObjectTo calls.
 * threads <= 1 runs serially; >1 uses OpenMP over rows.
 * sample_stride lets a caller render the subset {s_begin + k*stride}
 * (Philox only; multi-GPU sample split). */
void o_starting_gens(const o_params *p, o_stdgen *gens);      /* src/Lib.hs:34-39 */
void o_trace(const o_params *p, const o_object *objs, int n,
             o_stdgen *gens, int64_t s_begin, int64_t iters, int64_t sample_stride,
             double *sum, double *sumsq, uint64_t *segments, int threads);

/* Same as o_trace for a chosen subset of pixels (spot checks at full canvas
 * sizes): pixels[k] = i*W + j; sum/sumsq hold 3 values per listed pixel. */
void o_trace_pixels(const o_params *p, const o_object *objs, int n, const int32_t *pixels, int64_t count,
                    o_stdgen *gens, int64_t s_begin, int64_t iters, int64_t sample_stride,
                    double *sum, double *sumsq, uint64_t *segments, int threads);

/* TEST AID (no counterpart in the reference): o_trace_pixels on the Philox stream with each listed pixel's primary
 * segment supplied by the caller -- idx[q] (struck object or -1), point[3q..] (hit point), dir[3q..] (primary
 * direction) -- so that a path-wise comparison with the device can start from the device's own fp32-rounded hit
 * point (the analytic tie lines).  The supplied segment is not counted in *segments. */
void o_trace_pixels_from_primary(const o_params *p, const o_object *objs, int n, const int32_t *pixels, int64_t count,
                                 const int32_t *idx, const double *point, const double *dir,
                                 int64_t s_begin, int64_t iters, int64_t sample_stride,
                                 double *sum, uint64_t *segments, int threads);

/* ---- Conversion.hs ------------------------------------------------------ */
/* listToPixelRGB8 (src/Conversion.hs:17-20): round-half-even of 255*x, then
 * Int -> Word8 wrap. */
uint8_t o_to_word8(double x);
/* repa3ToImage (src/Conversion.hs:22-28): out[(j*H + i)*3 + k] from
 * canvas[(i*W + j)*3 + k]; the image is H pixels wide and W pixels tall. */
void o_repa3_to_image(const o_params *p, const double *canvas, uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
