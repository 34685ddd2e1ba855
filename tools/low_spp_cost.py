"""low_spp_cost.py -- launch time against samples per launch on the default scene (1080p, scene kernel and general kernel):
the way app/Web.hs pulls frames is one sample per message (src/Lib.hs:53, app/Web.hs:43)."""
import sys
sys.path.insert(0, '.')
import pathtracer_amd as pa
for spec in (True, False):
    with pa.Renderer(pa.Params(1920, 1080, 500.0, 64, 8), pa.objects, seed=1) as r:
        if spec:
            r.specialise()
        r.set_timing(True)
        r.trace(1); r.synchronize()
        row = []
        for spp in (1, 2, 4, 8, 16, 64):
            best = 1e9
            for rep in range(5):
                r.trace(spp); ms, _ = r.last_trace_ms(); best = min(best, ms)
            row.append((spp, best))
        slope = (row[-1][1] - row[-2][1]) / (row[-1][0] - row[-2][0])
        print(("scene kernel: " if spec else "general kernel: ") + ", ".join(f"{s} spp {m*1e3:.0f} us" for s, m in row) + f"; per sample {slope*1e3:.1f} us")
