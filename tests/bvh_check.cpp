// bvh_check.cpp -- host-side check of the spatial index's builder (pathtracer_amd/csrc/pt_bvh.hpp), compiled and run by
// tests/test_bvh_builder.py with g++ (no GPU).  Builds the index over random boxes and checks what the device walk
// relies on: node 0 unused, the root at node 1, children in adjacent pairs with the left one at an even index, codes
// well-formed, every object in exactly one leaf of at most BVH_LEAF_MAX objects, depth within the stack's bound, and --
// the conservative part -- every node's box, DECODED FROM ITS BINARY16 CENTRE AND HALF-EXTENT as the device decodes
// it, contains the padded boxes of all objects below it and its children's decoded boxes' objects.
// usage: bvh_check <objects> <seed> <scale> <offset>     prints "ok nodes=.. depth=.. leaves=.." or a complaint
#include "pt_bvh.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace pt;

static int fail(const char *what, long a = 0, long b = 0)
{
    std::printf("FAIL %s (%ld, %ld)\n", what, a, b);
    return 1;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 1000;
    const unsigned seed = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u;
    const double scale = argc > 3 ? std::atof(argv[3]) : 100.0, offset = argc > 4 ? std::atof(argv[4]) : 0.0;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<BvhBox> boxes((size_t)n);
    std::vector<int32_t> ids;
    double extent = 0.0;
    for (int k = 0; k < n; ++k) {
        for (int a = 0; a < 3; ++a) {
            const double c = offset + scale * (U(rng) - 0.5), h = scale * 0.02 * U(rng) * (k % 7 == 0 ? 0.0 : 1.0);   // some degenerate boxes
            boxes[(size_t)k].lo[a] = c - h; boxes[(size_t)k].hi[a] = c + h;
            extent = std::max(extent, std::max(std::fabs(c - h), std::fabs(c + h)));
        }
        ids.push_back(k);
    }
    if (n > 10) { boxes[3] = boxes[4]; boxes[5] = boxes[4]; boxes[9] = boxes[4]; }      // coincident objects: the median fallback
    const double pad = 2e-5 * extent + 1e-30;
    Bvh bvh;
    const int32_t base = 3;                                   // as if three unbounded objects came first
    BvhBuilder b(boxes, ids, pad);
    b.build(bvh, base);
    const int nn = (int)bvh.nodes.size();
    if (nn < 2) return fail("fewer than two nodes", nn);
    if ((int)bvh.prims.size() != n) return fail("prims", (long)bvh.prims.size(), n);
    if (b.max_depth > BVH_MAX_DEPTH) return fail("depth", b.max_depth);

    std::vector<int> seen((size_t)n, 0), visited((size_t)nn, 0);
    int leaves = 0, deepest = 0;
    // decoded box of a node, as the device computes it: fp32 centre -+ fp32 half-extent
    auto decoded = [&](int i, float lo[3], float hi[3]) {
        for (int a = 0; a < 3; ++a) {
            const float c = half_bits_to_float(bvh.nodes[(size_t)i].h[a]), e = half_bits_to_float(bvh.nodes[(size_t)i].h[3 + a]);
            lo[a] = c - e; hi[a] = c + e;
        }
    };
    struct It { int node, depth; };
    std::vector<It> st{{BVH_ROOT, 0}};
    // objects below every node, bottom-up by recursion on an explicit stack: collect (node -> object range) at leaves,
    // and check containment against every ancestor on the path
    std::vector<int> path;
    struct Frame { int node, depth, state; };
    std::vector<Frame> fs{{BVH_ROOT, 0, 0}};
    while (!fs.empty()) {
        Frame f = fs.back(); fs.pop_back();
        if (f.state == 1) { path.pop_back(); continue; }
        const int i = f.node;
        if (i <= 0 || i >= nn) return fail("child index out of range", i, nn);
        if (visited[(size_t)i]++) return fail("node reached twice", i);
        deepest = std::max(deepest, f.depth);
        path.push_back(i);
        fs.push_back({i, f.depth, 1});
        const int32_t c = bvh.nodes[(size_t)i].code;
        if (c >= 0) {
            const int left = c >> 5, axis = c & 31;
            if (axis > 2) return fail("split axis", i, axis);
            if (left & 1) return fail("left child at an odd index", i, left);
            if (left < 2 || left + 1 >= nn) return fail("children out of range", i, left);
            fs.push_back({left + 1, f.depth + 1, 0});
            fs.push_back({left, f.depth + 1, 0});
        } else {
            const int code = ~c, first = code >> 3, cnt = (code & 7) + 1;
            ++leaves;
            if (cnt > BVH_LEAF_MAX) return fail("leaf too large", i, cnt);
            if (first < base || first - base + cnt > n) return fail("leaf range", first, cnt);
            for (int q = 0; q < cnt; ++q) {
                const int obj = bvh.prims[(size_t)(first - base + q)];
                if (obj < 0 || obj >= n) return fail("object id", obj);
                if (seen[(size_t)obj]++) return fail("object in two leaves", obj);
                for (int anc : path) {
                    float lo[3], hi[3];
                    decoded(anc, lo, hi);
                    for (int a = 0; a < 3; ++a) {
                        // the padded box in fp32, as the device's arithmetic could see it at worst
                        if (!((double)lo[a] <= boxes[(size_t)obj].lo[a] - pad) || !((double)hi[a] >= boxes[(size_t)obj].hi[a] + pad))
                            return fail("a node's decoded box does not contain a padded object below it", anc, obj);
                    }
                }
            }
        }
    }
    if (visited[0]) return fail("node 0 is part of the tree");
    for (int k = 0; k < n; ++k) if (seen[(size_t)k] != 1) return fail("object not in exactly one leaf", k, seen[(size_t)k]);
    for (int i = 1; i < nn; ++i) if (!visited[(size_t)i]) return fail("orphan node", i);
    if (deepest != b.max_depth) return fail("max_depth", deepest, b.max_depth);
    std::printf("ok nodes=%d depth=%d leaves=%d\n", nn, deepest, leaves);
    return 0;
}
