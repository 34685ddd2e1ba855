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


def _replay(tmp_path, H, W, B, ks):
    out = tmp_path / "elements.bin"
    r = subprocess.run([os.path.join(ROOT, "host", "shim_replay"), str(H), str(W), "500", str(B), str(out)] + [str(k) for k in ks],
                       capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    return np.fromfile(out, np.float64).reshape(len(ks), H, W, 3)


def test_haskell_shim_call_sequence_web(tmp_path):
    """host/shim_replay.c = haskell/Lib.hs's calls with its structs filled at the shim's own byte offsets.
    app/Web.hs:42-65 pulls traces !! 0, 1, 2, 3; then an old element again (rendered afresh: same numbers)."""
    import pathtracer_amd as pa
    H, W, B = 64, 48, 6
    el = _replay(tmp_path, H, W, B, [0, 1, 2, 3, 3, 1])
    assert not el[0].any()                                 # traces !! 0 is the black canvas
    with pa.Renderer(pa.Params(H, W, 500.0, 1, B), seed=0) as r:
        r.specialise()
        for k in (1, 2, 3):
            r.trace(1)
            assert np.array_equal(el[k], r.read_sum()), k
    assert np.array_equal(el[4], el[3])                    # the buffer already held element 3
    assert np.array_equal(el[5], el[1])                    # pt_reset + one fresh sample: element 1 again
    ora = O.Tracer(O.params(H, W, 500.0, B, O.PHILOX, 0), O.default_scene(), threads=4).trace(3)
    bad = (np.abs(el[3] - ora) > 2e-4 * np.abs(ora) + 1e-5).any(axis=2).mean()
    assert bad <= 0.01, bad


def test_haskell_shim_interleaved_elements(tmp_path):
    """Two elements read in turns (R.zipWith over traces !! 2 and traces !! 5: the one buffer flips at every switch,
    forwards by pt_trace, backwards by pt_reset + fresh samples): every read returns its own element's numbers."""
    H, W, B = 48, 40, 6
    el = _replay(tmp_path, H, W, B, [2, 5, 2, 5, 5, 2])
    assert np.array_equal(el[1], el[3]) and np.array_equal(el[1], el[4])
    assert not np.array_equal(el[0], el[1])
    # element 2 rendered as 2 fresh samples each time: the same numbers bit for bit
    assert np.array_equal(el[0], el[2]) and np.array_equal(el[0], el[5])
    ora = O.Tracer(O.params(H, W, 500.0, B, O.PHILOX, 0), O.default_scene(), threads=4)
    two = ora.trace(2).copy()
    five = ora.trace(3)
    for got, want in ((el[0], two), (el[1], five)):
        assert (np.abs(got - want) > 2e-4 * np.abs(want) + 1e-5).any(axis=2).mean() <= 0.01


def test_haskell_shim_call_sequence_write(tmp_path):
    """app/Write.hs:28 is traces !! numIters: ONE pt_trace numIters and ONE read."""
    import pathtracer_amd as pa
    H, W, B, N = 96, 64, 8, 10
    el = _replay(tmp_path, H, W, B, [N])
    with pa.Renderer(pa.Params(H, W, 500.0, N, B), seed=0) as r:
        r.trace(N)
        assert np.array_equal(el[0], r.read_sum())
    ora = O.Tracer(O.params(H, W, 500.0, B, O.PHILOX, 0), O.default_scene(), threads=4).trace(N)
    bad = (np.abs(el[0] - ora) > 2e-4 * np.abs(ora) + 1e-5).any(axis=2).mean()
    assert bad <= 0.01, bad


def test_write_twin_on_a_device_group(tmp_path):
    """host/write --devices a,b: pt_create_multi behind the C++ host (here: device 0 twice)."""
    import pathtracer_amd as pa
    out = tmp_path / "image.png"
    H, W, N, B = 96, 64, 6, 6
    r = subprocess.run([os.path.join(ROOT, "host", "write"), "--height", str(H), "--width", str(W), "--iters", str(N),
                        "--bounces", str(B), "--seed", "5", "--devices", "0,0", "--mode", "mean", "--out", str(out)],
                       capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert b'"gpus": 2' in r.stdout
    img = decode_png(out.read_bytes())
    with pa.Renderer(pa.Params(H, W, 500.0, N, B), seed=5, devices=[0, 0]) as g:
        g.trace(N)
        assert np.array_equal(img, g.read_rgb8(pa.PT_RGB8_MEAN_CLAMP))
