"""bvh_primary_cost.py -- what the per-launch fp64 primary pass costs the indexed kernel: launch time against samples per
launch (1080p and 4K, 1,000 and 10,000 disks); the intercept of the line is the pass, the slope a sample per pixel."""
import sys
sys.path.insert(0, '.')
import pathtracer_amd as pa
for (H, W) in ((1920, 1080), (3840, 2160)):
    for n, seed in ((1000, 1), (10000, 2)):
        objs = pa.synthetic_scene(n, seed, "disks")
        with pa.Renderer(pa.Params(H, W, 500.0, 64, 8), objs, seed=1) as r:
            r.set_timing(True)
            r.trace(1); r.synchronize()
            row = []
            for spp in (1, 2, 4, 8, 16):
                best = 1e9
                for rep in range(3):
                    r.trace(spp); ms, _ = r.last_trace_ms(); best = min(best, ms)
                row.append((spp, best))
            slope = (row[-1][1] - row[0][1]) / (row[-1][0] - row[0][0])
            print(f"{H}x{W} disks{n}: " + ", ".join(f"{s} spp {m:.3f} ms" for s, m in row) + f"; per sample {slope:.3f} ms, intercept {row[0][1] - slope:.3f} ms")
