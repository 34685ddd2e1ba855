// kbench.cpp -- times pt_trace of the reference default scene through the C ABI
// (include/pt.h), without Python: the quick A/B harness for kernel experiments.
//   kbench [spp=1024] [reps=5] [flags=0] [H=1920] [W=1080] [bounces=8]
// Prints the per-launch kernel time (HIP events, pt_last_trace_ms), segments
// and a checksum of the canvas, so that two builds can be compared for speed and
// for equality of output.
// Build: g++ -O2 -std=c++17 -I../include -o kbench kbench.cpp -L../pathtracer_amd -Wl,-rpath,'$ORIGIN/../pathtracer_amd' -lpt
#include "pt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static pt_object plane(double px, double py, double pz, double nx, double ny, double nz, double r, double g, double b)
{
    pt_object o; memset(&o, 0, sizeof o);
    o.kind = PT_INFINITE_PLANE; o.point[0] = px; o.point[1] = py; o.point[2] = pz;
    o.normal[0] = nx; o.normal[1] = ny; o.normal[2] = nz; o.color[0] = r; o.color[1] = g; o.color[2] = b;
    return o;
}

#define CHECK(x) do { int rc_ = (x); if (rc_ != PT_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, ctx ? pt_last_error(ctx) : ""); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int spp = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 5;
    const int flags = argc > 3 ? atoi(argv[3]) : 0;
    const int H = argc > 4 ? atoi(argv[4]) : 1920, W = argc > 5 ? atoi(argv[5]) : 1080;
    const int bounces = argc > 6 ? atoi(argv[6]) : 8;
    pt_ctx *ctx = nullptr;
    pt_params p; memset(&p, 0, sizeof p);
    p.height = H; p.width = W; p.camera_depth = 500.0; p.max_bounces = bounces; p.seed = 0; p.device = 0;
    p.sample_offset = 0; p.sample_stride = 1; p.flags = flags;
    CHECK(pt_create(&ctx, &p));
    // Object.objects (src/Object.hs:104-148): infPlane, infPlane2, infPlane3, light
    pt_object objs[4];
    objs[0] = plane(20, -20, 100, 0, 0, -1, 1, 1, 1);
    objs[1] = plane(20, -20, 100, 0, 1, 0, 0.1, 1, 1);
    objs[2] = plane(20, -20, 100, -1, 0, 0, 1, 1, 0.1);
    objs[3] = plane(0, 0, -1, 0, 0, 1, 1, 1, 1);
    objs[3].kind = PT_DISK; objs[3].radius = 20; objs[3].emittance = 2; objs[3].reflective = 1;
    CHECK(pt_scene_upload(ctx, objs, 4));
    CHECK(pt_set_timing(ctx, 1));
    double best = 1e30, sum = 0;
    uint64_t segs = 0;
    for (int r = 0; r < reps + 1; ++r) {
        CHECK(pt_reset(ctx));
        CHECK(pt_trace(ctx, spp));
        float ms; int32_t launches;
        CHECK(pt_last_trace_ms(ctx, &ms, &launches));
        CHECK(pt_segments(ctx, &segs));
        if (r == 0) continue;                         // warm-up
        sum += ms; if (ms < best) best = ms;
    }
    std::vector<float> c((size_t)H * W * 3);
    CHECK(pt_read_sum_f32(ctx, c.data()));
    double cs = 0; for (float v : c) cs += v;
    const double mean = sum / reps;
    printf("spp=%d flags=%d  kernel mean %.4f ms  best %.4f ms  segments %llu  -> %.1f Grays/s (mean)  frac=%.4f  checksum %.6e\n",
           spp, flags, mean, best, (unsigned long long)segs, segs / mean / 1e6, 97.0 * segs / (mean * 1e-3) / 157.3e12, cs);
    pt_destroy(ctx);
    return 0;
}
