// pt_synth.hip -- pt_synthetic_scene (include/pt.h): the generator of the
// synthetic n-object scenes of BASELINE.json configs[3]/[4] as SURVEY.md 8(d)
// specifies it, in C behind the ABI so that every host (Haskell, C++, Python)
// builds the SAME scene from (kind, n, seed).  Host code only; no counterpart in
// the reference, whose one scene is the constant `objects` (src/Object.hs:147-148).
#include "../../include/pt.h"

#include <cmath>
#include <cstdint>
#include <cstring>

namespace {

// SplitMix64 (Steele, Lea, Flood: "Fast splittable pseudorandom number generators", OOPSLA 2014)
struct SplitMix64 {
    uint64_t s;
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }     // [0, 1), 53 bits
    double uniform(double lo, double hi) { return lo + (hi - lo) * unit(); }
};

} // namespace

extern "C" int pt_synthetic_scene(pt_object *out, int32_t n, uint64_t seed, int32_t finite_kind)
{
    if (!out || n < 1) return PT_ERR_INVALID;
    if (finite_kind != PT_DISK && finite_kind != PT_SPHERE && finite_kind != PT_RECTANGLE) return PT_ERR_INVALID;
    SplitMix64 g{seed};
    std::memset(out, 0, (size_t)n * sizeof(pt_object));
    // object 0: the back plane z = 1500, so that most paths hit something
    out[0].kind = PT_INFINITE_PLANE;
    out[0].point[2] = 1500.0;
    out[0].normal[2] = -1.0;
    out[0].color[0] = out[0].color[1] = out[0].color[2] = 1.0;
    for (int32_t k = 1; k < n; ++k) {
        pt_object &o = out[k];
        o.kind = finite_kind;
        // centre uniform in the box x, y in [-400, 400], z in [200, 1200]
        o.point[0] = g.uniform(-400.0, 400.0);
        o.point[1] = g.uniform(-400.0, 400.0);
        o.point[2] = g.uniform(200.0, 1200.0);
        // a uniformly random unit normal: a point of the unit ball (rejection), normalised
        double x, y, z, r2;
        do {
            x = g.uniform(-1.0, 1.0); y = g.uniform(-1.0, 1.0); z = g.uniform(-1.0, 1.0);
            r2 = x * x + y * y + z * z;
        } while (r2 > 1.0 || r2 < 1e-4);
        const double inv = 1.0 / std::sqrt(r2);
        o.normal[0] = x * inv; o.normal[1] = y * inv; o.normal[2] = z * inv;
        o.radius = g.uniform(5.0, 40.0);                               // radii U(5, 40)
        for (int c = 0; c < 3; ++c) o.color[c] = g.uniform(0.2, 1.0);  // colour U(0.2, 1)^3
        o.reflective = g.unit() < 0.5 ? 1 : 0;                         // 50 % reflective
        const bool emissive = g.unit() < 0.02;                         // 2 % emissive, emittance U(1, 5)
        const double e = g.uniform(1.0, 5.0);                          // (drawn either way: the stream does not branch)
        o.emittance = emissive ? e : 0.0;
        if (finite_kind == PT_SPHERE) { o.normal[0] = o.normal[1] = o.normal[2] = 0.0; }
        if (finite_kind == PT_RECTANGLE) { o.extent[0] = 2.0 * o.radius; o.extent[1] = 1.2 * o.radius; o.radius = 0.0; }
    }
    return PT_OK;
}
