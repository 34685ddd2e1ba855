/* shim_replay.c -- the Haskell shim's call sequence (haskell/Lib.hs), replayed in
 * plain C.  The shim cannot be compiled in this image (no GHC);
 * this program does what it does, the way it does it:
 *   * pt_params and the pt_object array are filled BYTE-WISE at the offsets
 *     written by hand in the shim's Storable instances (shim_offsets.h is
 *     generated from haskell/Lib.hs by tools/shim_offsets.py), never through
 *     the C structs -- a wrong offset there shows up here;
 *   * `context`: pt_create -> pt_scene_upload (Object.objects, src/Object.hs:
 *     104-148) -> pt_scene_specialise (result ignored, as in the shim);
 *   * `fetch k` for every k on the command line, in order: pt_samples_done;
 *     pt_reset if k lies behind; pt_trace (k - from) if k lies ahead;
 *     pt_read_sum into the ONE buffer of the process.
 * app/Web.hs:42-65 is `shim_replay ... 0 1 2 3`, app/Write.hs:24-28 is
 * `shim_replay ... <numIters>`.  Each fetched element (H*W*3 doubles) is
 * appended to the output file; tests/test_gpu_host.py compares them with the
 * oracle.
 *
 * usage: shim_replay HEIGHT WIDTH CAMERA_DEPTH MAX_BOUNCES OUT.bin K [K ...]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pt.h"
#include "shim_offsets.h"

static void poke_i32(unsigned char *p, int off, int32_t v) { memcpy(p + off, &v, 4); }
static void poke_u32(unsigned char *p, int off, uint32_t v) { memcpy(p + off, &v, 4); }
static void poke_f64(unsigned char *p, int off, double v) { memcpy(p + off, &v, 8); }

/* poke of `PtObject` (haskell/Lib.hs): kind, reflective, point/centre, normal, radius, colour, emittance, extent = 0 */
static void poke_object(unsigned char *p, int32_t kind, int refl, const double *pt, const double *n, double radius,
                        const double *rgb, double emit)
{
    poke_i32(p, SHIM_OBJECT_KIND, kind);
    poke_i32(p, SHIM_OBJECT_REFLECTIVE, refl ? 1 : 0);
    for (int k = 0; k < 3; ++k) poke_f64(p, SHIM_OBJECT_POINT + 8 * k, pt[k]);
    for (int k = 0; k < 3; ++k) poke_f64(p, SHIM_OBJECT_NORMAL + 8 * k, n[k]);
    poke_f64(p, SHIM_OBJECT_RADIUS, radius);
    for (int k = 0; k < 3; ++k) poke_f64(p, SHIM_OBJECT_COLOR + 8 * k, rgb[k]);
    poke_f64(p, SHIM_OBJECT_EMITTANCE, emit);
    poke_f64(p, SHIM_OBJECT_EXTENT0, 0.0);
    poke_f64(p, SHIM_OBJECT_EXTENT1, 0.0);
}

static int die(pt_ctx *ctx, const char *what, int st)
{
    fprintf(stderr, "shim_replay: %s: libpt status %d: %s\n", what, st, pt_last_error(ctx));
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: shim_replay HEIGHT WIDTH CAMERA_DEPTH MAX_BOUNCES OUT.bin K [K ...]\n");
        return 2;
    }
    const int H = atoi(argv[1]), W = atoi(argv[2]);
    const double depth = atof(argv[3]);
    const int bounces = atoi(argv[4]);
    const char *out_path = argv[5];

    /* `params` of the shim: PtParams height width cameraDepth maxBounces 0 0 0 1 0 */
    unsigned char par[SHIM_PARAMS_SIZE];
    memset(par, 0xA5, sizeof par);                       /* bytes the shim does not write stay visibly wrong */
    poke_i32(par, SHIM_PARAMS_HEIGHT, H);
    poke_i32(par, SHIM_PARAMS_WIDTH, W);
    poke_f64(par, SHIM_PARAMS_CAMERA_DEPTH, depth);
    poke_i32(par, SHIM_PARAMS_MAX_BOUNCES, bounces);
    poke_u32(par, SHIM_PARAMS_SEED, 0);
    poke_i32(par, SHIM_PARAMS_DEVICE, 0);
    poke_i32(par, SHIM_PARAMS_SAMPLE_OFFSET, 0);
    poke_i32(par, SHIM_PARAMS_SAMPLE_STRIDE, 1);
    poke_i32(par, SHIM_PARAMS_FLAGS, 0);

    /* Object.objects = [infPlane, infPlane2, infPlane3, light]  (src/Object.hs:104-148) */
    unsigned char objs[4 * SHIM_OBJECT_SIZE];
    memset(objs, 0xA5, sizeof objs);
    const double p[3] = {20, -20, 100}, white[3] = {1, 1, 1};
    const double n0[3] = {0, 0, -1}, n1[3] = {0, 1, 0}, n2[3] = {-1, 0, 0};
    const double c1[3] = {0.1, 1, 1}, c2[3] = {1, 1, 0.1};
    const double lc[3] = {0, 0, -1}, ln[3] = {0, 0, 1};
    poke_object(objs + 0 * SHIM_OBJECT_SIZE, 0, 0, p, n0, 0.0, white, 0.0);
    poke_object(objs + 1 * SHIM_OBJECT_SIZE, 0, 0, p, n1, 0.0, c1, 0.0);
    poke_object(objs + 2 * SHIM_OBJECT_SIZE, 0, 0, p, n2, 0.0, c2, 0.0);
    poke_object(objs + 3 * SHIM_OBJECT_SIZE, 1, 1, lc, ln, 20.0, white, 2.0);

    /* `context` */
    pt_ctx *ctx = NULL;
    int st = pt_create(&ctx, (const pt_params *)par);
    if (st) return die(NULL, "pt_create", st);
    st = pt_scene_upload(ctx, (const pt_object *)objs, 4);
    if (st) return die(ctx, "pt_scene_upload", st);
    (void)pt_scene_specialise(ctx);                      /* optional; the shim ignores the status too */

    /* `slot`: ONE buffer, and the element it holds */
    const size_t n = (size_t)H * W * 3;
    double *buf = (double *)malloc(n * sizeof(double));
    long have = -1;
    FILE *f = fopen(out_path, "wb");
    if (!buf || !f) { fprintf(stderr, "shim_replay: cannot allocate or open %s\n", out_path); return 1; }

    for (int a = 6; a < argc; ++a) {
        const long k = atol(argv[a]);
        if (have != k) {                                  /* `fetch k` */
            int64_t done = 0;
            if ((st = pt_samples_done(ctx, &done))) return die(ctx, "pt_samples_done", st);
            long from = (long)done;
            if (k < done) {
                if ((st = pt_reset(ctx))) return die(ctx, "pt_reset", st);
                from = 0;
            }
            if (k > from && (st = pt_trace(ctx, (int32_t)(k - from)))) return die(ctx, "pt_trace", st);
            if ((st = pt_read_sum(ctx, buf))) return die(ctx, "pt_read_sum", st);
            have = k;
        }
        if (fwrite(buf, sizeof(double), n, f) != n) { fprintf(stderr, "shim_replay: short write\n"); return 1; }
    }
    fclose(f);
    uint64_t segs = 0;
    (void)pt_segments(ctx, &segs);
    printf("{\"elements\": %d, \"last\": %ld, \"segments\": %llu}\n", argc - 6, have, (unsigned long long)segs);
    free(buf);
    pt_destroy(ctx);
    return 0;
}
