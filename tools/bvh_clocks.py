"""Experiment (libpt built with -DPT_BVH_CLOCKS): shader clocks the waves of trace_bvh_kernel spend in shade and in
walk steps, reported through pt_bvh_counters' two numbers.  usage: PT_LIBPT_PATH=.../libpt_exp_clk.so python3 tools/bvh_clocks.py"""
import sys
sys.path.insert(0, '.')
import pathtracer_amd as pa
for kind, n, seed, spp, b in (("disks", 1000, 1, 64, 8), ("spheres", 1000, 1, 64, 8), ("disks", 10000, 2, 32, 16)):
    objs = pa.synthetic_scene(n, seed, kind)
    with pa.Renderer(pa.Params(3840, 2160, 500.0, spp, b), objs, seed=1) as r:
        r.trace(spp); r.synchronize()
        r.reset(); r.set_timing(True); r.trace(spp); ms, _ = r.last_trace_ms()
        segs = r.segments; shade, walk = r.bvh_counters()
        waves = 3840 * 2160 // 64
        print(f"{kind}{n}: {ms:.1f} ms; per wave: shade {shade / waves / 1e3:.0f} kclk, walk {walk / waves / 1e3:.0f} kclk "
              f"-> shade share {shade / (shade + walk):.2f}; per segment (64 lanes): shade {shade * 64 / segs:.0f} clk, walk {walk * 64 / segs:.0f} clk")
