"""bvh_stats.py -- wave-level step counts of trace_bvh_kernel from the -DPT_BVH_STATS=1|2 builds (PT_LIBPT_PATH names the
library): visit-loop iterations, leaf-loop iterations, walk steps (STATS=1) or shade steps (STATS=2), per segment, and the
mean number of lanes each serves.  Counts are summed per lane as 4096 / (lanes active), so they are exact to ~0.1 %."""
import sys, os
sys.path.insert(0, '.')
import pathtracer_amd as pa
which = "rounds" if "stats1" in os.environ.get("PT_LIBPT_PATH", "") else "shade steps"
for kind, n, seed, spp, b in (("disks", 1000, 1, 64, 8), ("spheres", 1000, 1, 64, 8), ("disks", 10000, 2, 32, 16)):
    objs = pa.synthetic_scene(n, seed, kind)
    with pa.Renderer(pa.Params(3840, 2160, 500.0, spp, b), objs, seed=1) as r:
        r.trace(spp); r.synchronize()
        segs = r.segments; iv, il = r.bvh_counters(); x = r.bvh_padding_gathers()
        iv /= 4096; il /= 4096; x /= 4096
        print(f"{kind}{n}: per segment: visit iterations {iv/segs:.3f}, leaf-pair iterations {il/segs:.3f}, {which} {x/segs:.3f}; "
              f"wave totals: visit its {iv:.4g}, leaf its {il:.4g}, {which} {x:.4g}, segments {segs}")
