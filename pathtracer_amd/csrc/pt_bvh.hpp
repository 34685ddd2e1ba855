// pt_bvh.hpp -- host-side builder of the spatial index over the FINITE forms of
// a scene (Disk, src/Object.hs:51-54); InfinitePlane (:55-56) has no bounds and
// stays on a short list that every ray tests.
//
// The reference's closestObjectTo is a linear scan over all objects
// (src/Lib.hs:81-98).  The index only prunes candidates: every object whose
// (inflated) bounding box the ray can reach before the best hit so far is tested
// with the SAME arithmetic as the linear scan, and ties are broken by object
// index as Data.Vector.minimumBy does, so the winner is the linear scan's winner.
//
// Layout: binary BVH in BREADTH-first order, one 16-byte node per box, the two
// children of a node next to each other.  A lane walks it near child first (by
// the sign of the ray direction along the node's split axis) with a short
// per-lane stack of far children in LDS; every node is one 16-byte gather.
// Node 0 is not a box: index 0 is the walk's "search over" mark (the bottom entry
// of every lane's stack), the root is node 1 and the pairs of children follow
// from node 2 on, so a left child's index is even and its sibling is index ^ 1.
// Depth is bounded (BVH_MAX_DEPTH) so the stack cannot overflow whatever the scene.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace pt {

// 16 bytes per node = ONE 16-byte gather per step of the walk: the box as centre and half-extent, six binary16
// values, plus one code word.  The device tests it with nine v_fma_mix_f32 (binary16 sources converted inside the
// fp32 FMA: t = (c -+ e - o) / d as fma(-+e, |1/d|, fma(c, 1/d, -o/d))) instead of six conversions, six subtractions,
// six multiplications and six min/max (round 3, profiles/r03_bvh_walk.txt).
//   h[0..2] = centre.xyz rounded to nearest, h[3..5] = half-extent.xyz rounded UP so that [c - e, c + e] evaluated
//   in fp32 contains the padded box (a coordinate beyond binary16's range gives e = +Inf: still conservative)
//   code >= 0: inner node, left_child << 5 | split_axis (left_child even; right child = left child + 1)
//   code <  0: leaf, ~code = first_prim << 3 | (count - 1)
constexpr int32_t BVH_ROOT = 1;
struct BvhNode {
    uint16_t h[6];
    int32_t code;
};
static_assert(sizeof(BvhNode) == 16, "one float4 per node");

// binary16 bits of the largest half <= x (up = false) or the smallest half >= x (up = true)
inline uint16_t half_round_dir(float x, bool up)
{
    if (x != x) return up ? 0x7c00 : 0xfc00;                 // NaN bound: open the box
    const bool neg = std::signbit(x);
    const float ax = std::fabs(x);
    uint32_t mag;                                             // magnitude rounded toward zero ...
    bool exact;
    if (ax == std::numeric_limits<float>::infinity()) { mag = 0x7c00; exact = true; }
    else if (ax > 65504.0f) { mag = 0x7bff; exact = false; }  // beyond the largest finite half
    else if (ax < 6.103515625e-05f) {                         // subnormal half: units of 2^-24
        const float q = ax * 16777216.0f;
        mag = (uint32_t)q; exact = (float)mag == q;
    } else {
        uint32_t au; std::memcpy(&au, &ax, 4);
        const uint32_t e = (au >> 23) - 112, m = au & 0x7fffffu;
        mag = (e << 10) | (m >> 13); exact = (m & 0x1fffu) == 0;
    }
    if (!exact && up != neg) mag += 1;                        // ... then away from zero where the direction asks (0x7bff + 1 = Inf)
    return (uint16_t)((neg ? 0x8000u : 0u) | mag);
}

inline float half_bits_to_float(uint16_t h)
{
    const uint32_t e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? (m ? std::nanf("") : std::numeric_limits<float>::infinity())
                                                          : std::ldexp((float)(m | 1024), (int)e - 25);
    return (h & 0x8000) ? -v : v;
}

#ifndef PT_BVH_LEAF_MAX
#define PT_BVH_LEAF_MAX 2                // (pt_device.hpp has the same default)
#endif
// Objects per leaf: 8 was best while the walk waited on latency (round 1), 6 once it was bound by gathered bytes
// (round 2); with round 3's walk step (a node visit costs the wave ~24 VALU instructions, a pair of object tests 82)
// 2 is: 1,000 disks 23.5 -> 19.4 ms, 10,000 disks 85.3 -> 74.5 (profiles/r03_bvh_walk.txt).
constexpr int BVH_LEAF_MAX = PT_BVH_LEAF_MAX;
#ifndef PT_BVH_SAH_DEPTH
#define PT_BVH_SAH_DEPTH 14
#endif
constexpr int BVH_SAH_DEPTH = PT_BVH_SAH_DEPTH;       // surface-area splits down to here, balanced median splits below:
constexpr int BVH_MAX_DEPTH = 30;       //   14 + ceil(log2(65536 / 2)) = 29 levels at most
constexpr int BVH_STACK = 32;           // per-lane stack entries on the device (> BVH_MAX_DEPTH)
constexpr int BVH_MIN_FINITE = 16;      // below this the linear scan is used

struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<int32_t> prims;          // [unbounded objects..., finite objects in leaf order...]
    int32_t n_unbounded = 0;
    bool usable() const { return !nodes.empty(); }
};

struct BvhBox { double lo[3], hi[3]; };

// a box as a node holds it: pad, widen by fp32 rounding, then the centre as a binary16 and the half-extent rounded UP to
// cover both sides from that centre -- the device's test is plain fp32 on these values (slab_hit, pt_device.hpp)
inline void bvh_encode_box(const BvhBox &bb, double pad, uint16_t h[6])
{
    for (int a = 0; a < 3; ++a) {
        const float lo = (float)(bb.lo[a] - pad) - std::fabs((float)(bb.lo[a] - pad)) * 1e-7f;
        const float hi = (float)(bb.hi[a] + pad) + std::fabs((float)(bb.hi[a] + pad)) * 1e-7f;
        uint16_t ch = half_round_dir(0.5f * lo + 0.5f * hi, false);
        if ((ch & 0x7fffu) >= 0x7c00u) ch = (uint16_t)((ch & 0x8000u) | 0x7bffu);     // keep the centre finite
        const double c = (double)half_bits_to_float(ch);
        const double ext = std::max((double)hi - c, c - (double)lo);
        h[a]     = ch;
        h[3 + a] = half_round_dir((float)(ext * (1.0 + 2e-7)) + 1e-30f, true);
    }
}

// centre, unit normal, radius of a Disk -> axis-aligned bounds
inline BvhBox disk_bounds(const double c[3], const double nhat[3], double r)
{
    BvhBox b;
    for (int a = 0; a < 3; ++a) {
        const double h = r * std::sqrt(std::max(0.0, 1.0 - nhat[a] * nhat[a]));
        b.lo[a] = c[a] - h; b.hi[a] = c[a] + h;
    }
    return b;
}

class BvhBuilder {
public:
    BvhBuilder(const std::vector<BvhBox> &boxes, const std::vector<int32_t> &ids, double pad)
        : boxes_(boxes), pad_(pad), order_(ids) {}

    void build(Bvh &out, int32_t prim_base)
    {
        prim_base_ = prim_base;
        out_ = &out;
        // level order: a queue of (range of order_, depth, node index); children are allocated in pairs
        struct Item { int b, e, depth, me; };
        std::vector<Item> queue;
        out.nodes.push_back(BvhNode{});                  // node 0: the "search over" mark, never fetched
        out.nodes.push_back(BvhNode{});                  // node BVH_ROOT
        queue.push_back(Item{0, (int)order_.size(), 0, BVH_ROOT});
        for (size_t head = 0; head < queue.size(); ++head) {
            const Item it = queue[head];
            int mid = 0, axis = 0;
            if (!emit(it.b, it.e, it.depth, it.me, mid, axis)) continue;       // a leaf
            const int32_t first = (int32_t)out.nodes.size();
            out.nodes.push_back(BvhNode{});
            out.nodes.push_back(BvhNode{});
            out.nodes[it.me].code = (first << 5) | axis;        // first is even: pairs start at node 2
            queue.push_back(Item{it.b, mid, it.depth + 1, first});               // the low side of `axis` first
            queue.push_back(Item{mid, it.e, it.depth + 1, first + 1});
            if (it.depth + 1 > max_depth) max_depth = it.depth + 1;
        }
        for (int32_t id : order_) out.prims.push_back(id);
    }

private:
    const std::vector<BvhBox> &boxes_;
    double pad_;
    std::vector<int32_t> order_;          // positions into boxes_ == object ids are looked up via ids
    Bvh *out_ = nullptr;
    int32_t prim_base_ = 0;

    BvhBox bounds(int b, int e) const
    {
        BvhBox r;
        for (int a = 0; a < 3; ++a) { r.lo[a] = std::numeric_limits<double>::infinity(); r.hi[a] = -r.lo[a]; }
        for (int i = b; i < e; ++i)
            for (int a = 0; a < 3; ++a) {
                r.lo[a] = std::min(r.lo[a], boxes_[order_[i]].lo[a]);
                r.hi[a] = std::max(r.hi[a], boxes_[order_[i]].hi[a]);
            }
        return r;
    }
    static double area(const BvhBox &b)
    {
        const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
        return 2.0 * (x * y + y * z + z * x);
    }
    double centroid(int i, int a) const { return 0.5 * (boxes_[order_[i]].lo[a] + boxes_[order_[i]].hi[a]); }

    // fills node `me` with the box of order_[b, e); a leaf if it holds few enough objects (-> false), else splits
    // the range at `mid` along `axis` (-> true; the caller allocates the children)
    bool emit(int b, int e, int depth, int me, int &mid, int &axis)
    {
        const BvhBox bb = bounds(b, e);
        BvhNode n;
        bvh_encode_box(bb, pad_, n.h);
        const int cnt = e - b;
        if (cnt <= BVH_LEAF_MAX) {
            n.code = ~(((prim_base_ + b) << 3) | (cnt - 1));
            out_->nodes[me] = n;
            return false;
        }
        n.code = 0;
        out_->nodes[me] = n;
        // binned SAH over the centroids of the widest axes; median split as the fallback
        int best_axis = -1, best_split = -1;
        double best_cost = std::numeric_limits<double>::infinity();
#ifndef PT_BVH_BINS
#define PT_BVH_BINS 16
#endif
        constexpr int BINS = PT_BVH_BINS;
        for (int a = 0; a < 3 && depth < BVH_SAH_DEPTH; ++a) {
            double cmin = std::numeric_limits<double>::infinity(), cmax = -cmin;
            for (int i = b; i < e; ++i) { cmin = std::min(cmin, centroid(i, a)); cmax = std::max(cmax, centroid(i, a)); }
            if (!(cmax > cmin)) continue;
            BvhBox bin_box[BINS]; int bin_cnt[BINS] = {0};
            for (auto &q : bin_box) for (int x = 0; x < 3; ++x) { q.lo[x] = std::numeric_limits<double>::infinity(); q.hi[x] = -q.lo[x]; }
            const double scale = BINS / (cmax - cmin);
            for (int i = b; i < e; ++i) {
                int k = std::min(BINS - 1, (int)((centroid(i, a) - cmin) * scale));
                bin_cnt[k]++;
                for (int x = 0; x < 3; ++x) {
                    bin_box[k].lo[x] = std::min(bin_box[k].lo[x], boxes_[order_[i]].lo[x]);
                    bin_box[k].hi[x] = std::max(bin_box[k].hi[x], boxes_[order_[i]].hi[x]);
                }
            }
            double right_area[BINS]; int right_cnt[BINS];
            BvhBox acc = bin_box[BINS - 1]; int c = 0;
            for (int k = BINS - 1; k >= 1; --k) {
                if (k < BINS - 1)
                    for (int x = 0; x < 3; ++x) { acc.lo[x] = std::min(acc.lo[x], bin_box[k].lo[x]); acc.hi[x] = std::max(acc.hi[x], bin_box[k].hi[x]); }
                c += bin_cnt[k];
                right_area[k] = c ? area(acc) : 0.0; right_cnt[k] = c;
            }
            acc = bin_box[0]; c = 0;
            for (int k = 0; k < BINS - 1; ++k) {
                if (k > 0)
                    for (int x = 0; x < 3; ++x) { acc.lo[x] = std::min(acc.lo[x], bin_box[k].lo[x]); acc.hi[x] = std::max(acc.hi[x], bin_box[k].hi[x]); }
                c += bin_cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const double cost = area(acc) * c + right_area[k + 1] * right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
            }
            if (best_axis == a) { split_min_ = cmin; split_scale_ = scale; }
        }
        axis = best_axis;
        if (best_axis >= 0) {
            const int a = best_axis; const double cmin = split_min_, scale = split_scale_; const int ks = best_split;
            auto it = std::partition(order_.begin() + b, order_.begin() + e, [&](int32_t id) {
                const double cc = 0.5 * (boxes_[id].lo[a] + boxes_[id].hi[a]);
                return std::min(BINS - 1, (int)((cc - cmin) * scale)) <= ks;
            });
            mid = (int)(it - order_.begin());
        } else {
            mid = b;
        }
        if (mid <= b || mid >= e) {                    // degenerate (coincident centroids): median by the widest axis
            int a = 0; double w = -1;
            for (int x = 0; x < 3; ++x) if (bb.hi[x] - bb.lo[x] > w) { w = bb.hi[x] - bb.lo[x]; a = x; }
            mid = (b + e) / 2;
            axis = a;
            std::nth_element(order_.begin() + b, order_.begin() + mid, order_.begin() + e, [&](int32_t p, int32_t q) {
                return boxes_[p].lo[a] + boxes_[p].hi[a] < boxes_[q].lo[a] + boxes_[q].hi[a];
            });
        }
        return true;
    }
    double split_min_ = 0, split_scale_ = 0;
public:
    int max_depth = 0;
};

} // namespace pt
