"""Rates of the by-value kernels on scenes of 4 to 16 objects (1080p, 256 spp, depth 8): the default scene, a
six-rectangle room with a lamp and a mirror card (8 objects, general-scatter kernel), the same without the mirror,
and the room furnished with 4 and 8 more objects (12 and 16: by value only as scene kernels, --specialise; the
general path streams their object list)."""
import sys, time
sys.path.insert(0, ".")
import pathtracer_amd as pa
R = pa.Rectangle
room = [pa.Object((0.9, 0.9, 0.9), 0, False, R((0, 0, 300), (0, 0, -1), 240, 240), "back"),
        pa.Object((0.9, 0.3, 0.3), 0, False, R((0, -120, 180), (0, 1, 0), 240, 240), "left"),
        pa.Object((0.3, 0.9, 0.3), 0, False, R((0, 120, 180), (0, -1, 0), 240, 240), "right"),
        pa.Object((0.9, 0.9, 0.9), 0, False, R((120, 0, 180), (-1, 0, 0), 240, 240), "ceiling"),
        pa.Object((0.9, 0.9, 0.9), 0, False, R((-120, 0, 180), (1, 0, 0), 240, 240), "floor"),
        pa.Object((1, 1, 1), 6.0, True, R((119, 0, 180), (-1, 0, 0), 80, 60), "lamp"),
        pa.Object((0.9, 0.9, 1.0), 0, True, R((-20, 40, 250), (0.3, -0.5, -1.0), 120, 70), "mirror card"),
        pa.Object((0.3, 0.4, 0.9), 0, False, pa.Disk((-119, 0, 180), (1, 0, 0), 50), "rug")]
disks8 = [pa.Object((0.9, 0.9, 0.9), 0, False, pa.InfinitePlane((0, 0, 300), (0, 0, -1)), "back"),
          pa.Object((0.9, 0.3, 0.3), 0, False, pa.InfinitePlane((0, -120, 180), (0, 1, 0)), "left"),
          pa.Object((0.3, 0.9, 0.3), 0, False, pa.InfinitePlane((0, 120, 180), (0, -1, 0)), "right"),
          pa.Object((0.9, 0.9, 0.9), 0, False, pa.InfinitePlane((120, 0, 180), (-1, 0, 0)), "ceiling"),
          pa.Object((0.9, 0.9, 0.9), 0, False, pa.InfinitePlane((-120, 0, 180), (1, 0, 0)), "floor"),
          pa.Object((1, 1, 1), 6.0, True, pa.Disk((119, 0, 180), (-1, 0, 0), 40), "lamp"),
          pa.Object((0.3, 0.4, 0.9), 0, False, pa.Disk((-119, 0, 180), (1, 0, 0), 50), "rug"),
          pa.Object((0.8, 0.8, 0.2), 0, False, pa.Disk((0, 0, 299), (0, 0, -1), 60), "picture")]
extra = [pa.Object((0.7, 0.2, 0.6), 0, False, pa.Disk((-60, -60, 220), (0.1, 0.9, -0.3), 30), "plate"),
         pa.Object((0.2, 0.6, 0.9), 0, False, R((-119, 60, 120), (1, 0, 0), 60, 90), "second rug"),
         pa.Object((0.9, 0.6, 0.2), 0, False, R((60, -119, 200), (0, 1, 0), 90, 60), "picture, left wall"),
         pa.Object((0.9, 0.9, 0.4), 0, False, pa.Disk((0, 0, 299), (0, 0, -1), 45), "clock, back wall"),
         pa.Object((0.6, 0.9, 0.9), 0, False, R((60, 119, 200), (0, -1, 0), 90, 60), "picture, right wall"),
         pa.Object((0.4, 0.4, 0.9), 0, False, pa.Disk((-119, -70, 240), (1, 0, 0), 25), "mat"),
         pa.Object((0.9, 0.3, 0.9), 0, False, R((-40, -30, 150), (0.2, 0.1, -1), 50, 50), "board"),
         pa.Object((0.3, 0.3, 0.3), 0, False, pa.Disk((90, 50, 260), (-1, -0.2, -0.2), 20), "dish")]
disks8_plus4, disks8_plus8 = disks8 + extra[:4], disks8 + extra
P = pa.Params(1920, 1080, 500.0, 256, 8)
for name, objs in (("default (4 objects)", pa.objects), ("room, 8 objects, mirror + rectangles", room),
                   ("8 planes and disks, diffuse", disks8),
                   ("12 objects: the 8 + 4 more, diffuse", disks8_plus4),
                   ("16 objects: the 8 + 8 more, diffuse", disks8_plus8)):
    with pa.Renderer(P, objs, seed=0) as r:
        if "--specialise" in sys.argv:
            assert r.specialise(), r.last_error()
        r.trace(8); r.synchronize()
        ts = []
        for _ in range(3):
            r.reset(); t0 = time.perf_counter(); r.trace(256); r.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(f"{name:42s} {t * 1e3:8.2f} ms  {r.segments / t / 1e9:7.1f} Grays/s  ({r.segments / (1920 * 1080 * 256):.2f} segments per path)")
