#!/usr/bin/env python3
"""Throughput of the trace kernel on the synthetic n-object disk scenes of
SURVEY.md 8(d) (BASELINE configs 4/5 shapes at reduced image size): segments/s
and ray-object tests/s (= segments/s * n)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[8, 100, 1000, 10000])
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=270)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--flags", type=int, default=0)
args = ap.parse_args()
for n in args.n:
    objs = pa.random_disk_scene(n, 1 if n <= 1000 else 2)
    spp = args.spp
    with pa.Renderer(pa.Params(args.height, args.width, 500.0, spp, args.bounces), objs, seed=1, flags=args.flags) as r:
        r.set_timing(True)
        r.trace(spp)
        r.synchronize()
        r.reset()
        r.trace(spp)
        ms, launches = r.last_trace_ms()
        segs = r.segments
        print(json.dumps({"flags": args.flags, "n_objects": n, "pixels": args.height * args.width, "spp": spp, "ms": round(ms, 3),
                          "segments": segs, "segs_per_path": round(segs / (args.height * args.width * spp), 3),
                          "Mrays_per_s": round(segs / ms / 1e3, 1),
                          "G_ray_object_tests_per_s": round(segs * n / ms / 1e6, 1)}), flush=True)
