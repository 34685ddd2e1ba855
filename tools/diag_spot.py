import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle as O
import pathtracer_amd as pa
import test_gpu_fullsize as T
H, W = T.H, T.W
px = T._spot_pixels(np.random.default_rng(2026), 1500)
for spp in (1024, 256):
    with pa.Renderer(pa.Params(H, W, 500.0, spp, 8), seed=0) as r:
        r.trace(spp)
        dev = r.read_sum().reshape(-1, 3)[px]
    ora, _ = O.trace_pixels(O.params(H, W, 500.0, 8, O.PHILOX, 0), O.default_scene(), px, spp)
    d = np.abs(dev - ora).max(axis=1)
    order = np.argsort(-d)[:25]
    rel = np.abs(dev - ora) / np.maximum(np.abs(ora), 1e-9)
    print("spp", spp, "close frac", (rel <= 2e-4).all(axis=1).mean())
    for k in order:
        i, j = divmod(int(px[k]), W)
        print(f"  px ({i},{j}) tiecol={j == W//2-100} tierow={i == H//2-100} maxdiff={d[k]:.4f} dev={dev[k]} ora={ora[k]}")
