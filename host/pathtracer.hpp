// pathtracer.hpp -- C++ host side of the radiance loop, mirroring the
// reference's Haskell interface for this path (same names, same meaning):
//
//   Params::*                 src/Params.hs:3-11   (now a run-time struct)
//   Object / Form / objects   src/Object.hs:30-56, :104-148
//   Lib::Traces               src/Lib.hs:53-56     (traces :: [Array D DIM3 Double])
//   Conversion::repa3ToImage  src/Conversion.hs:22-28
//   Conversion::imageToText   src/Conversion.hs:37-42  (PNG -> base64)
//
// The reference is compiled Haskell; no GHC exists in the build image, so the
// host above the C ABI (include/pt.h) is C++ here, and haskell/ holds the
// `foreign import ccall` shim a maintainer with GHC would use instead
// (INTEGRATION.md).  Everything compute-heavy happens in libpt.so on the GPU;
// this header only marshals values and owns the context.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/pt.h"

namespace pathtracer {

struct Triple { double x, y, z; };                      // src/Triple.hs:19-24

namespace Params_defaults {                             // src/Params.hs:3-11
constexpr int height = 500, width = 500;
constexpr double cameraDepth = 500;
constexpr int numIters = 10, maxBounces = 30;
}

struct Params {
    int height = Params_defaults::height;
    int width = Params_defaults::width;
    double cameraDepth = Params_defaults::cameraDepth;
    int numIters = Params_defaults::numIters;
    int maxBounces = Params_defaults::maxBounces;
};

struct Form {                                           // src/Object.hs:51-61
    enum Kind { InfinitePlane = PT_INFINITE_PLANE, Disk = PT_DISK,
                Sphere = PT_SPHERE,        // extension, src/Object.hs:191-194 is a TODO
                Rectangle = PT_RECTANGLE   // extension, src/Object.hs:57-61 is commented out (include/pt.h defines it)
    } kind;
    Triple point;      // _point | _center
    Triple normal;     // _normal
    double radius;     // _radius (Disk)
    double height = 0.0, width = 0.0;   // _height, _width (Rectangle)
    static Form infinitePlane(Triple p, Triple n) { return Form{InfinitePlane, p, n, 0.0}; }
    static Form disk(Triple c, Triple n, double r) { return Form{Disk, c, n, r}; }
    static Form sphere(Triple c, double r) { return Form{Sphere, c, Triple{0, 0, 0}, r}; }
    static Form rectangle(Triple c, Triple n, double height, double width) { return Form{Rectangle, c, n, 0.0, height, width}; }
};

struct Object {                                         // src/Object.hs:30-36
    Triple color;
    double emittance;
    bool reflective;
    Form form;
    std::string name;

    pt_object to_c() const
    {
        pt_object o{};
        o.kind = form.kind;
        o.reflective = reflective ? 1 : 0;
        o.point[0] = form.point.x; o.point[1] = form.point.y; o.point[2] = form.point.z;
        o.normal[0] = form.normal.x; o.normal[1] = form.normal.y; o.normal[2] = form.normal.z;
        o.radius = form.radius;
        o.color[0] = color.x; o.color[1] = color.y; o.color[2] = color.z;
        o.emittance = emittance;
        o.extent[0] = form.height; o.extent[1] = form.width;
        return o;
    }
};

// objects = V.fromList [infPlane, infPlane2, infPlane3, light]  (src/Object.hs:104-148)
inline std::vector<Object> objects()
{
    const Triple p{20, -20, 100};
    return {
        Object{{1, 1, 1}, 0, false, Form::infinitePlane(p, {0, 0, -1}), "plane 1"},
        Object{{0.1, 1, 1}, 0, false, Form::infinitePlane(p, {0, 1, 0}), "plane 2"},
        Object{{1, 1, 0.1}, 0, false, Form::infinitePlane(p, {-1, 0, 0}), "plane 3"},
        Object{{1, 1, 1}, 2, true, Form::disk({0, 0, -1}, {0, 0, 1}, 20), "light"},
    };
}

// The synthetic scenes of BASELINE.json configs[3]/[4], from libpt's generator so that every
// host builds the same one: kind = PT_DISK (reference-expressible), PT_SPHERE or PT_RECTANGLE (extensions).
inline std::vector<Object> syntheticScene(int n, uint64_t seed, int kind = PT_DISK)
{
    std::vector<pt_object> c((size_t)n);
    if (pt_synthetic_scene(c.data(), n, seed, kind) != PT_OK) throw std::runtime_error("pt_synthetic_scene: bad argument");
    std::vector<Object> out;
    for (const pt_object &o : c) {
        Form f{(Form::Kind)o.kind, {o.point[0], o.point[1], o.point[2]}, {o.normal[0], o.normal[1], o.normal[2]}, o.radius,
               o.extent[0], o.extent[1]};
        out.push_back(Object{{o.color[0], o.color[1], o.color[2]}, o.emittance, o.reflective != 0, f, ""});
    }
    return out;
}

struct PtError : std::runtime_error {
    int status;
    PtError(int s, const std::string &m) : std::runtime_error("libpt status " + std::to_string(s) + ": " + m), status(s) {}
};

namespace Lib {

// Lib.traces (src/Lib.hs:53-56) as a lazy sequence: element k is the
// (height, width, 3) SUM of k samples per pixel, element 0 the black canvas.
// Asking for element k runs only the traceCanvas steps not done yet -- the
// reference re-runs all k (and three times, once per channel).
class Traces {
public:
    Traces(const Params &p, const std::vector<Object> &objs, uint32_t seed = 0, int device = 0,
           int sample_offset = 0, int sample_stride = 1)
        : params_(p)
    {
        pt_params cp{};
        cp.height = p.height; cp.width = p.width; cp.camera_depth = p.cameraDepth;
        cp.max_bounces = p.maxBounces; cp.seed = seed; cp.device = device;
        cp.sample_offset = sample_offset; cp.sample_stride = sample_stride; cp.flags = 0;
        check(pt_create(&ctx_, &cp));
        upload(objs);
    }
    // The same `traces` rendered by several GPUs of this process (pt_create_multi, include/pt.h): every
    // advance() is sharded by sample index over `devices`, every read all-reduced over RCCL inside libpt.
    Traces(const Params &p, const std::vector<Object> &objs, const std::vector<int32_t> &devices, uint32_t seed = 0)
        : params_(p)
    {
        pt_params cp{};
        cp.height = p.height; cp.width = p.width; cp.camera_depth = p.cameraDepth;
        cp.max_bounces = p.maxBounces; cp.seed = seed; cp.device = devices.empty() ? 0 : devices[0];
        cp.sample_offset = 0; cp.sample_stride = 1; cp.flags = 0;
        check(pt_create_multi(&ctx_, &cp, devices.data(), (int32_t)devices.size()));
        upload(objs);
    }
    void upload(const std::vector<Object> &objs)
    {
        std::vector<pt_object> c;
        for (const Object &o : objs) c.push_back(o.to_c());
        check(pt_scene_upload(ctx_, c.data(), (int32_t)c.size()));
    }
    ~Traces() { pt_destroy(ctx_); }
    Traces(const Traces &) = delete;
    Traces &operator=(const Traces &) = delete;

    // traces !! k : advance to element k (k >= current) and return it.
    std::vector<double> operator[](int64_t k)
    {
        advance(k);
        std::vector<double> out((size_t)params_.height * params_.width * 3);
        check(pt_read_sum(ctx_, out.data()));
        return out;
    }
    void advance(int64_t k)
    {
        int64_t done = 0;
        check(pt_samples_done(ctx_, &done));
        if (k < done) throw PtError(PT_ERR_INVALID, "Traces: elements are produced in order");
        while (done < k) {
            const int32_t n = (int32_t)std::min<int64_t>(k - done, 1 << 20);
            check(pt_trace(ctx_, n));
            done += n;
        }
    }
    // pt_scene_specialise: the trace kernel compiled for this scene (seconds once, then faster launches; the
    // image is the same).  False where the library cannot (no hiprtc, more than 8 objects): nothing changes.
    bool specialise()
    {
        const int s = pt_scene_specialise(ctx_);
        if (s == PT_ERR_UNSUPPORTED) return false;
        check(s);
        return true;
    }
    uint64_t segments()
    {
        uint64_t s = 0;
        check(pt_segments(ctx_, &s));
        return s;
    }
    pt_ctx *ctx() { return ctx_; }
    const Params &params() const { return params_; }

    void check(int status)
    {
        if (status != PT_OK) throw PtError(status, pt_last_error(ctx_));
    }

private:
    Params params_;
    pt_ctx *ctx_ = nullptr;
};

} // namespace Lib

namespace Conversion {

// A P.Image P.PixelRGB8: `width` x `height` pixels, row-major RGB bytes.
struct Image {
    int width, height;
    std::vector<uint8_t> rgb;
};

// repa3ToImage (src/Conversion.hs:22-28) of the current element of `traces`:
// generateImage fromCoords Params.height Params.width, so the image is
// Params.height pixels WIDE and Params.width pixels TALL.
// mode PT_RGB8_REFERENCE reproduces listToPixelRGB8 (round(255*x) :: Word8,
// no clamp, no division by the sample count).
inline Image repa3ToImage(Lib::Traces &t, int mode = PT_RGB8_REFERENCE)
{
    Image img{t.params().height, t.params().width, {}};
    img.rgb.resize((size_t)img.width * img.height * 3);
    t.check(pt_read_rgb8(t.ctx(), img.rgb.data(), mode));
    return img;
}

std::vector<uint8_t> encodePng(const Image &img);        // P.encodePng
std::string base64(const std::vector<uint8_t> &bytes);  // Data.ByteString.Base64.encode
inline std::string imageToText(const Image &img) { return base64(encodePng(img)); }   // src/Conversion.hs:37-42
void savePngImage(const std::string &path, const Image &img);                         // P.savePngImage

} // namespace Conversion
} // namespace pathtracer
