"""The C++ host twins of app/Write.hs and app/Web.hs on the GPU: the PNG that
`write` saves holds exactly repa3ToImage of `traces !! numIters`, and `frames`
answers client messages with consecutive elements of `traces`."""
import base64
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from png_util import decode_png

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_write_twin_matches_python_path_and_oracle(tmp_path):
    import pathtracer_amd as pa
    out = tmp_path / "image.png"
    H, W, N, B = 96, 64, 3, 6
    r = subprocess.run([os.path.join(ROOT, "host", "write"), "--height", str(H), "--width", str(W), "--iters", str(N),
                        "--bounces", str(B), "--seed", "5", "--out", str(out)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    img = decode_png(out.read_bytes())
    assert img.shape == (W, H, 3)                          # H pixels wide, W pixels tall
    with pa.Renderer(pa.Params(H, W, 500.0, N, B), seed=5) as rr:
        rr.trace(N)
        rgb = rr.read_rgb8()
        s32 = rr.read_sum_f32()
    assert np.array_equal(img, rgb)
    assert np.array_equal(img, O.repa3_to_image(O.params(H, W, 500.0, B), s32.astype(np.float64)))
    assert b'"segments"' in r.stdout


def test_frames_twin_streams_consecutive_elements(tmp_path):
    import pathtracer_amd as pa
    H, W, B = 24, 16, 4
    r = subprocess.run([os.path.join(ROOT, "host", "frames"), "--height", str(H), "--width", str(W), "--bounces", str(B)],
                       input=b"m\nm\nm\n", capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.decode().strip().split("\n")
    assert len(lines) == 3 and all(l.startswith("data:image/png;base64,") for l in lines)
    frames = [decode_png(base64.b64decode(l.split(",", 1)[1])) for l in lines]
    assert not frames[0].any()                             # traces !! 0 is the black canvas
    with pa.Renderer(pa.Params(H, W, 500.0, 1, B)) as rr:
        for k in (1, 2):
            rr.trace(1)
            assert np.array_equal(frames[k], rr.read_rgb8())
