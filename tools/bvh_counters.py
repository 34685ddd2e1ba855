import sys, time
sys.path.insert(0,'.')
import pathtracer_amd as pa
for kind,n,seed,spp,b in (("disks",1000,1,64,8),("spheres",1000,1,64,8),("disks",10000,2,32,16)):
    objs=pa.synthetic_scene(n,seed,kind)
    with pa.Renderer(pa.Params(3840,2160,500.0,spp,b),objs,seed=1) as r:
        r.trace(spp); r.synchronize()
        r.reset(); r.set_timing(True); r.trace(spp); ms,_=r.last_trace_ms()
        segs=r.segments; nv,rec=r.bvh_counters(); pad=r.bvh_padding_gathers()
        g=nv+rec
        print(f"{kind}{n}: {ms:.1f} ms, {segs/ms/1e3:.0f} Mrays/s, nodes/seg {nv/segs:.1f}, record gathers/seg {rec/segs:.1f} (of which padding {pad/segs:.1f}), gathers/seg {g/segs:.1f}, {g/ms/1e6:.1f} G gathers/s = {g*16/ms/1e9:.2f} TB/s")
