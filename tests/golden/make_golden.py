#!/usr/bin/env python3
"""Regenerates the fixtures in this directory from the CPU oracle
(oracle/pt_oracle.c, the fp64 restatement of the reference).

The reference itself cannot produce golden vectors: it is Haskell and no GHC
exists in the build image, and its own test suite holds no
fixtures beyond the known answers restated in tests/test_oracle_spec.py.  These
files therefore pin (a) the oracle against accidental change and (b) the GPU
path against the oracle on the GPU box; what ties them to the reference is the
analytic cross-check in tests/test_oracle_golden.py (hit regions derived
independently from src/Object.hs:104-148 and src/Lib.hs:41-51).

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402


def primary(H, W, depth, name):
    idx, t, nrm = O.primary_hits(O.params(H, W, depth, 4), O.default_scene())
    np.savez_compressed(os.path.join(HERE, name), idx=idx.astype(np.int8), t=t, normal=nrm.astype(np.int8),
                        H=H, W=W, camera_depth=depth)


def image(H, W, depth, bounces, spp, rng, seed, name):
    tr = O.Tracer(O.params(H, W, depth, bounces, rng, seed), O.default_scene(), threads=4)
    tr.trace(spp)
    np.savez_compressed(os.path.join(HERE, name), sum=tr.sum, segments=tr.segments.value, H=H, W=W,
                        camera_depth=depth, bounces=bounces, spp=spp, rng=rng, seed=seed)


# Three 128x128 windows of the bench frame (1920x1080, depth 8) on the reference's own StdGen stream at 2048
# samples per pixel, for the per-pixel statistical test at the bench resolution: the centre, a
# corner, and a window over the analytic tie column j = W/2 - 100.
CROPS = {"centre": (1920 // 2 - 64, 1080 // 2 - 64), "corner": (0, 0), "tie": (300, 1080 // 2 - 100 - 64)}


def crops(H, W, depth, bounces, spp, name, threads=8):
    out = {}
    for key, (i0, j0) in CROPS.items():
        px = O.crop_pixels(W, i0, j0)
        s, sq, segs = O.trace_pixels_stdgen(O.params(H, W, depth, bounces, O.STDGEN), O.default_scene(), px, spp, threads)
        out[key + "_sum"] = s.reshape(128, 128, 3)
        out[key + "_sumsq"] = sq.reshape(128, 128, 3)
        out[key + "_origin"] = np.array([i0, j0])
        out[key + "_segments"] = segs
    np.savez_compressed(os.path.join(HERE, name), H=H, W=W, camera_depth=depth, bounces=bounces, spp=spp, **out)


def spot_pixels(H, W, count, seed):
    """The spot set of tests/test_gpu_fullsize.py::_spot_check_scene: `count` random pixels plus corners and wave edges."""
    rng = np.random.default_rng(seed)
    return np.unique(np.concatenate([rng.integers(0, H * W, count),
                                     [0, W - 1, (H - 1) * W, H * W - 1, 63, 64, 255, 256, H * W - 65]])).astype(np.int32)


def config5_spots(name, count=64, seed=46, threads=8):
    """BASELINE.json configs[4] as written -- 10,000 disks (libpt's generator, SplitMix64 seed 2), 3840x2160, 4096 spp,
    depth 16 -- at `count` spot pixels on the device's Philox stream (seed 1): about a minute of CPU per pixel (the
    oracle scans all 10,000 objects per segment, as closestObjectTo does), so the sums are a fixture."""
    import pathtracer_amd as pa
    H, W, spp, bounces = 3840, 2160, 4096, 16
    objs = pa.synthetic_scene(10000, 2, "disks")
    px = spot_pixels(H, W, count, seed)
    s, segs = O.trace_pixels(O.params(H, W, 500.0, bounces, O.PHILOX, 1), O.scene_from(objs), px, spp, threads=threads)
    np.savez_compressed(os.path.join(HERE, name), pixels=px, sum=s, segments=segs, H=H, W=W, camera_depth=500.0,
                        bounces=bounces, spp=spp, philox_seed=1, scene_objects=10000, scene_seed=2, scene_kind="disks")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "config5":      # (ten minutes on 8 cores; the rest takes about as long)
        config5_spots("spots_config5_10000disks_3840x2160_4096spp_d16.npz")
        sys.exit(0)
    crops(1920, 1080, 500.0, 8, 2048, "crops_stdgen_1920x1080_d8_2048spp.npz")
    primary(64, 64, 500.0, "primary_default_64x64.npz")
    primary(200, 200, 500.0, "primary_default_200x200.npz")
    # BASELINE config 1: default scene, 200x200, 16 spp, depth 4, the reference's own StdGen stream
    image(200, 200, 500.0, 4, 16, O.STDGEN, 0, "image_config1_stdgen_200x200_16spp_d4.npz")
    # the device's Philox stream, for path-wise comparison on the GPU box
    image(64, 64, 500.0, 8, 8, O.PHILOX, 11, "image_philox_64x64_8spp_d8_seed11.npz")
    print("golden fixtures written to", HERE)
