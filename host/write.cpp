// write.cpp -- C++ twin of app/Write.hs:24-28:
//     main = savePngImage "image.png" . ImageRGB8 . repa3ToImage $ traces !! Params.numIters
// with Params (src/Params.hs:3-11) settable at run time.  All rendering runs
// on the GPU through libpt.so; without a GPU this program fails with libpt's
// error, it has no CPU path.
//
// usage: write [--height H] [--width W] [--camera-depth D] [--iters N] [--bounces B]
//              [--seed S] [--mode reference|mean|gamma] [--out image.png] [--device K]
//              [--scene default|disksN|spheresN|rectanglesN]   N-object synthetic scene of BASELINE configs[3]/[4]
//                                             (pt_synthetic_scene: seed 1 up to 1000 objects, else 2)
//              [--gpus N | --devices a,b,c]   several GPUs of this process: the iterations are sharded by
//                                             sample index and all-reduced over RCCL inside libpt (pt_create_multi)
#include "pathtracer.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

using namespace pathtracer;

int main(int argc, char **argv)
{
    Params p;                        // the reference's defaults: 500x500, depth 500, 10 iters, 30 bounces
    std::string out = "image.png";
    uint32_t seed = 0;
    int mode = PT_RGB8_REFERENCE, device = 0;
    std::vector<int32_t> devices;
    std::string scene = "default";
    for (int i = 1; i < argc; ++i) {
        auto arg = [&](const char *name) { return !std::strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--height")) p.height = std::atoi(argv[++i]);
        else if (arg("--width")) p.width = std::atoi(argv[++i]);
        else if (arg("--camera-depth")) p.cameraDepth = std::atof(argv[++i]);
        else if (arg("--iters")) p.numIters = std::atoi(argv[++i]);
        else if (arg("--bounces")) p.maxBounces = std::atoi(argv[++i]);
        else if (arg("--seed")) seed = (uint32_t)std::strtoul(argv[++i], nullptr, 10);
        else if (arg("--device")) device = std::atoi(argv[++i]);
        else if (arg("--gpus")) { const int n = std::atoi(argv[++i]); devices.clear(); for (int d = 0; d < n; ++d) devices.push_back(d); }
        else if (arg("--devices")) {
            devices.clear();
            for (char *tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ",")) devices.push_back(std::atoi(tok));
        }
        else if (arg("--scene")) scene = argv[++i];
        else if (arg("--out")) out = argv[++i];
        else if (arg("--mode")) {
            const std::string m = argv[++i];
            mode = m == "mean" ? PT_RGB8_MEAN_CLAMP : m == "gamma" ? PT_RGB8_MEAN_GAMMA : PT_RGB8_REFERENCE;
        } else {
            std::fprintf(stderr, "write: unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try {
        std::vector<Object> objs = objects();
        if (scene != "default") {
            const size_t digits = scene.find_first_of("0123456789");
            const std::string kind = scene.substr(0, digits);
            const int n = digits == std::string::npos ? 0 : std::atoi(scene.c_str() + digits);
            const int code = kind == "disks" ? PT_DISK : kind == "spheres" ? PT_SPHERE : kind == "rectangles" ? PT_RECTANGLE : -1;
            if (code < 0 || n < 1) { std::fprintf(stderr, "write: unknown --scene %s\n", scene.c_str()); return 2; }
            objs = syntheticScene(n, n <= 1000 ? 1 : 2, code);
        }
        std::unique_ptr<Lib::Traces> owner(devices.empty() ? new Lib::Traces(p, objs, seed, device)
                                                           : new Lib::Traces(p, objs, devices, seed));
        Lib::Traces &traces = *owner;
        const auto t0 = std::chrono::steady_clock::now();
        if ((int64_t)p.height * p.width * p.numIters >= (int64_t)1 << 26)
            traces.specialise();                                      // a frame of seconds: worth the second of compilation
        traces.advance(p.numIters);                                   // traces !! Params.numIters
        const Conversion::Image img = Conversion::repa3ToImage(traces, mode);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        Conversion::savePngImage(out, img);
        const uint64_t segs = traces.segments();
        std::printf("{\"image\": \"%s\", \"width_px\": %d, \"height_px\": %d, \"iters\": %d, \"max_bounces\": %d, "
                    "\"segments\": %llu, \"render_ms\": %.3f, \"mrays_per_s\": %.1f, \"gpus\": %d}\n",
                    out.c_str(), img.width, img.height, p.numIters, p.maxBounces, (unsigned long long)segs, ms,
                    segs / (ms * 1e3), devices.empty() ? 1 : (int)devices.size());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "write: %s\n", e.what());
        return 1;
    }
    return 0;
}
