"""Full-size parity: the bench workload itself (BASELINE.json configs[1]/[2]:
default scene, Params.height=1920, Params.width=1080, maxBounces 8) through the
C ABI, checked where the oracle can follow in seconds and through
size-independent properties elsewhere.

  * primary-hit indices and t at all 2,073,600 pixels: bit-exact against the
    oracle (fp64, reference operation order);
  * 1024 spp: the oracle (fp64 RESTATEMENT of the reference, same Philox
    stream) re-traces a seeded spot-check set of pixels -- tie lines, canvas
    corners, wave and workgroup boundaries included -- and each pixel's sum must
    agree path-wise;
  * linearity of `traces`: pt_trace(a) then pt_trace(b) == pt_trace(a+b) up to
    fp32 summation order, executed-search count equal;
  * the RGB8 image of the full canvas, byte for byte against the oracle's
    quantiser applied to the device's own sum.
"""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

H, W, DEPTH, BOUNCES = 1920, 1080, 500.0, 8


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


_cache = {}


def _oracle_spot(px, spp):
    """The oracle's Philox-stream sums at the spot pixels (once per run: both kernel modes compare with them)."""
    key = ("spot", px.tobytes(), spp)
    if key not in _cache:
        _cache[key] = O.trace_pixels(O.params(H, W, DEPTH, BOUNCES, O.PHILOX, 0), O.default_scene(), px, spp, threads=16)[0]
    return _cache[key]


def _oracle_frame_stdgen(nref):
    """The oracle's full frame on the reference's StdGen stream: (sum, sumsq), once per run."""
    key = ("frame", nref)
    if key not in _cache:
        tr = O.Tracer(O.params(H, W, DEPTH, BOUNCES, O.STDGEN), O.default_scene(), threads=16)
        tr.trace(nref)
        _cache[key] = (tr.sum, tr.sumsq)
    return _cache[key]


def _spot_pixels(rng, count):
    """Seeded spot-check set: random pixels plus the places where indexing bugs live."""
    ii = rng.integers(0, H, count)
    jj = rng.integers(0, W, count)
    px = list(ii * W + jj)
    px += [0, W - 1, (H - 1) * W, H * W - 1]                          # canvas corners
    px += [63, 64, 255, 256, 257, H * W - 256, H * W - 257]           # wave / workgroup edges
    px += [(H // 2) * W + W // 2]                                     # centre pixel (KAT: object 0, t = 100)
    px += [i * W + (W // 2 - 100) for i in range(0, H, 97)]           # tie column
    px += [(H // 2 - 100) * W + j for j in range(0, W, 53)]           # tie row
    return np.unique(np.array(px, np.int64)).astype(np.int32)


def test_primary_hits_full_canvas(pa):
    with pa.Renderer(pa.Params(H, W, DEPTH, 1, BOUNCES)) as r:
        idx, t, nrm = r.primary_hits()
    oidx, ot, onrm = O.primary_hits(O.params(H, W, DEPTH, BOUNCES), O.default_scene())
    assert np.array_equal(idx, oidx)
    assert np.array_equal(t, ot)
    assert np.array_equal(nrm, onrm)
    assert idx[H // 2, W // 2] == 0 and t[H // 2, W // 2] == 100.0
    assert set(np.unique(idx)) == {0, 1, 2}               # the three planes; the light is behind the camera


@pytest.mark.parametrize("spp,flags", [(1024, 0), (256, 0), (256, 4)])
def test_spot_check_pixels_at_bench_size(pa, spp, flags):
    """configs[1] (256 spp) and the metric's own frame (1024 spp)."""
    px = _spot_pixels(np.random.default_rng(2026), 1500)
    with pa.Renderer(pa.Params(H, W, DEPTH, spp, BOUNCES), seed=0, flags=flags) as r:
        r.trace(spp)
        dev = r.read_sum().reshape(-1, 3)[px]
        segs = r.segments
    ora = _oracle_spot(px, spp)
    # Same draws on both sides: a pixel differs only where fp32/fp64 rounding flips a hit.
    # Measured on MI355X: every one of the ~1,500 ordinary pixels agrees to <= 4e-4 absolute
    # at 1024 spp.  The exception is the analytic TIE LINES: there the primary
    # hit point lies on two planes at once, so whether a bounce ray "sees" the second plane
    # is decided by the last bit of the hit point -- in the reference's fp64 as much as in
    # fp32.  Their primary hits are bit-exact (test above); their sums are only sanity-bounded.
    ii, jj = px // W, px % W
    tie = (jj == W // 2 - 100) | (ii == H // 2 - 100)
    assert 20 <= tie.sum() <= 80
    rel = np.abs(dev - ora) / np.maximum(np.abs(ora), 1e-9)
    close = (rel[~tie] <= 2e-4).all(axis=1)
    assert close.mean() >= 0.995, close.mean()
    # one flipped path moves a sum by at most emittance * 1 = 2 per channel
    assert np.max(np.abs(dev[~tie] - ora[~tie])) <= 8.0
    assert np.all(np.isfinite(dev)) and dev.min() >= 0.0 and dev.max() <= 2.0 * spp
    # means over the spot set agree far below the Monte-Carlo noise
    assert np.all(np.abs(dev[~tie].mean(0) - ora[~tie].mean(0)) <= 1e-4 * ora[~tie].mean(0))
    # executed searches: bounded by paths * depth, above paths * 1
    assert H * W * spp <= segs <= H * W * spp * BOUNCES


def test_spot_check_of_exactly_what_the_bench_runs(pa):
    """bench.py's own configuration against the oracle: the kernel compiled for the scene (pt_scene_specialise),
    a 1024-spp launch (so: two sample slices + slice_sum_kernel), frames double-buffered over two contexts and
    two streams (PipelinedFrames) with the RGB8 output on a side stream.  Both frames' canvases must agree
    path-wise with the oracle's fp64 restatement on the same Philox stream at the spot-check pixels, and the
    RGB8 image with the oracle's quantiser of the device's own sums."""
    import torch
    from pathtracer_amd.distributed import PipelinedFrames, canvas_tensor
    spp = 1024
    px = _spot_pixels(np.random.default_rng(2026), 1500)
    P = pa.Params(H, W, DEPTH, spp, BOUNCES)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rs = [pa.Renderer(P, seed=0) for _ in range(2)]
        for q in rs:
            assert q.specialise(), q.last_error()
            assert "hiprtc" in q.jit_info() or "hipcc at build time" in q.jit_info()   # compiled here, or shipped with the library
        cvs = [canvas_tensor(q, 0) for q in rs]
        rgbs = [torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda") for _ in rs]
        pipe = PipelinedFrames(rs, cvs, rgbs, rank=0, spp_total=spp, my_spp=spp, mode=pa.PT_RGB8_MEAN_CLAMP)
        for _ in range(3):                                 # frames 0, 1, 2: canvases 0, 1, 0 (reused after its output)
            pipe.step()
        pipe.drain()
        devs = [q.read_sum().reshape(-1, 3) for q in rs]
        imgs = [g.cpu().numpy().reshape(W, H, 3) for g in rgbs]
        segs = [q.segments for q in rs]
        for q in rs:
            q.close()
    assert np.array_equal(devs[0], devs[1])               # same frame, same bits, from either context
    assert segs[1] == segs[0] + H * W                     # context 0's last frame read the primary hits its first one kept
    ora = _oracle_spot(px, spp)
    ii, jj = px // W, px % W
    tie = (jj == W // 2 - 100) | (ii == H // 2 - 100)
    dev = devs[0][px]
    rel = np.abs(dev - ora) / np.maximum(np.abs(ora), 1e-9)
    assert (rel[~tie] <= 2e-4).all(axis=1).mean() >= 0.995
    assert np.all(np.abs(dev[~tie].mean(0) - ora[~tie].mean(0)) <= 1e-4 * ora[~tie].mean(0))
    want = O.repa3_to_image(O.params(H, W, DEPTH, BOUNCES),
                            np.clip(devs[0].reshape(H, W, 3).astype(np.float32).astype(np.float64) * (1.0 / spp), 0, 1))
    for img in imgs:
        assert np.array_equal(img, want)


def test_per_pixel_statistics_on_crops_of_the_bench_frame(pa, kernel_mode):
    """SURVEY.md 8c's statistical bar, unchanged and PER PIXEL, at the bench's own resolution and depth: the
    device's 1024-spp frame (Philox) against the oracle on the REFERENCE's StdGen stream at 2048 spp
    (tests/golden/crops_stdgen_*.npz, written by make_golden.py) as independent estimates, on three 128x128
    windows: the centre, a corner, and one over the analytic tie column j = W/2 - 100.  Per pixel and channel
    z = (mu_dev - mu_ref) / sqrt(var/N + var/N_ref); fraction of |z| > 4 below 1e-3; per-channel mean bias below
    0.5 %; RMSE within 1.2x of the predicted value.  (2048 reference samples are what makes the per-pixel
    variance usable at this field of view; the 8x8 block-mean test below stays as an addition.)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crops_stdgen_1920x1080_d8_2048spp.npz"))
    N, NREF = 1024, int(g["spp"])
    assert (int(g["H"]), int(g["W"]), int(g["bounces"])) == (H, W, BOUNCES)
    with pa.Renderer(pa.Params(H, W, DEPTH, N, BOUNCES), seed=7) as r:
        r.trace(N)
        frame = r.read_sum() / N
    for key in ("centre", "corner", "tie"):
        i0, j0 = (int(x) for x in g[key + "_origin"])
        mu_dev = frame[i0:i0 + 128, j0:j0 + 128]
        mu_ref = g[key + "_sum"] / NREF
        var = np.maximum(g[key + "_sumsq"] / NREF - mu_ref**2, 0.0) * NREF / (NREF - 1)
        se2 = var / N + var / NREF
        ok = se2 > 0
        assert ok.mean() > 0.99
        z = (mu_dev[ok] - mu_ref[ok]) / np.sqrt(se2[ok])
        frac = (np.abs(z) > 4).mean()
        rmse, pred = np.sqrt(np.mean((mu_dev - mu_ref) ** 2)), np.sqrt(np.mean(se2))
        bias = [abs(mu_dev[..., k].mean() - mu_ref[..., k].mean()) / mu_ref[..., k].mean() for k in range(3)]
        print(f"{kernel_mode} {key}: |z|>4 {frac:.2e}, z mean {z.mean():+.3f} std {z.std():.3f}, rmse/pred {rmse / pred:.3f}, "
              f"bias {max(bias):.2e}")
        assert frac < 1e-3, (key, frac)
        assert max(bias) < 5e-3, (key, bias)
        assert 0.5 * pred <= rmse <= 1.2 * pred, (key, rmse, pred)


def test_trace_is_additive_at_full_size(pa):
    P = pa.Params(H, W, DEPTH, 64, BOUNCES)
    with pa.Renderer(P, seed=3) as one, pa.Renderer(P, seed=3) as two:
        one.trace(64)
        two.trace(24)
        two.trace(40)
        a, b = one.read_sum_f32(), two.read_sum_f32()
        assert two.samples_done == one.samples_done == 64
        assert two.segments == one.segments               # the second launch reads the primary hits the first one kept
    assert np.allclose(a, b, rtol=4e-6, atol=1e-6)
    assert a.sum() > 0


def test_rgb8_of_full_canvas(pa):
    spp = 16
    with pa.Renderer(pa.Params(H, W, DEPTH, spp, BOUNCES), seed=1) as r:
        r.trace(spp)
        s = r.read_sum_f32()
        ref_mode = r.read_rgb8(pa.PT_RGB8_REFERENCE)
        mean_mode = r.read_rgb8(pa.PT_RGB8_MEAN_CLAMP)
    p = O.params(H, W, DEPTH, BOUNCES)
    assert ref_mode.shape == (W, H, 3)                    # Params.height px wide, Params.width px tall (F10)
    assert np.array_equal(ref_mode, O.repa3_to_image(p, s.astype(np.float64)))
    want = O.repa3_to_image(p, np.clip(s.astype(np.float64) * (1.0 / spp), 0, 1))
    assert np.array_equal(mean_mode, want)


# --- BASELINE.json configs[3] / configs[4]: the synthetic 1,000- and 10,000-object scenes --------
# Disk variant (the reference has no sphere Form, so only disks have an oracle).
# configs[3] runs at its full size (3840x2160, 512 spp, depth 8: ~1e13 ray-object tests, seconds on
# one MI355X).  configs[4] at its full 4096 spp / depth 16 is ~1.5e15 brute-force tests (a quarter of
# an hour per GPU), so it runs at its full image size, object count and depth with 256 spp (the spatial index makes that seconds;
# the full 4096 spp would be about a minute and a half on one GPU).

def _spot_check_scene(pa, H, W, objs, spp, bounces, cam, count, seed, chunk=None, golden=None):
    """golden: a fixture of tests/golden/make_golden.py holding the oracle's sums for this very spot set (where the
    oracle is too slow to run inside the test)."""
    rng = np.random.default_rng(seed)
    px = np.unique(np.concatenate([rng.integers(0, H * W, count),
                                   [0, W - 1, (H - 1) * W, H * W - 1, 63, 64, 255, 256, H * W - 65]])).astype(np.int32)
    with pa.Renderer(pa.Params(H, W, cam, spp, bounces), objs, seed=1) as r:
        for _ in range(spp // (chunk or spp)):             # (launches of a second or two each)
            r.trace(chunk or spp)
            r.synchronize()
        assert r.samples_done == spp
        dev = r.read_sum().reshape(-1, 3)
        segs = r.segments
    if golden is not None:
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", golden))
        assert np.array_equal(g["pixels"], px) and (int(g["H"]), int(g["W"]), int(g["spp"]), int(g["bounces"])) == (H, W, spp, bounces)
        assert int(g["philox_seed"]) == 1 and float(g["camera_depth"]) == cam
        ora = g["sum"]
    else:
        ora, _ = O.trace_pixels(O.params(H, W, cam, bounces, O.PHILOX, 1), O.scene_from(objs), px, spp, threads=16)
    d = dev[px]
    rel = np.abs(d - ora) / np.maximum(np.abs(ora), 1e-9)
    close = ((rel <= 1e-3) | (np.abs(d - ora) <= 1e-4)).all(axis=1)
    # specular chains amplify rounding differences more than diffuse ones (test_pathwise_parity_random_disks)
    assert close.mean() >= 0.97, close.mean()
    assert np.all(np.isfinite(dev)) and dev.min() >= 0.0
    assert np.abs(d.mean(0) - ora.mean(0)).max() <= 0.02 * max(ora.mean(), 1e-3)
    assert H * W <= segs <= H * W * (1 + spp * (bounces - 1))
    return segs


def test_config4_thousand_disks_full_size(pa):
    objs = pa.synthetic_scene(1000, 1, "disks")      # BASELINE configs[3]'s scene (disk variant), libpt's generator
    segs = _spot_check_scene(pa, 3840, 2160, objs, 512, 8, 500.0, 160, 44)
    assert segs > 3840 * 2160 * 512                      # most primary rays hit something and bounce on


def test_config5_ten_thousand_disks_full_image_reduced_spp(pa):
    objs = pa.synthetic_scene(10000, 2, "disks")
    _spot_check_scene(pa, 3840, 2160, objs, 256, 16, 500.0, 40, 45)


def test_config5_as_written_4096_spp_depth_16_on_one_gpu(pa):
    """BASELINE.json configs[4] at its stated size -- 10,000 objects, 3840x2160, 4096 spp, depth 16 -- on ONE GPU
    (the configuration names 8; the sample split only divides these launches among ranks): 3.4e10 paths, about a
    quarter of a minute of kernels, spot-checked against the oracle on the same Philox stream at 73 pixels (64 random,
    corners and wave edges).  The oracle scans all 10,000 objects per segment, about a minute of CPU per pixel, so its
    sums are the committed fixture tests/golden/spots_config5_*.npz (make_golden.py config5)."""
    objs = pa.synthetic_scene(10000, 2, "disks")
    segs = _spot_check_scene(pa, 3840, 2160, objs, 4096, 16, 500.0, 64, 46, chunk=256,
                             golden="spots_config5_10000disks_3840x2160_4096spp_d16.npz")
    assert segs > 3840 * 2160 * 4096


def test_statistical_parity_at_bench_resolution(pa):
    """north_star's third level at the bench's own resolution and depth: the
    device (Philox, 256 spp) against the oracle on the REFERENCE's StdGen stream
    (128 spp; the oracle is an fp64 restatement of the reference, which cannot be
    built here) as independent estimates of the same 1920x1080 image.  Same
    thresholds as test_statistical_parity_reference_stream."""
    N, NREF = 256, 128
    with pa.Renderer(pa.Params(H, W, DEPTH, N, BOUNCES), seed=5) as r:
        r.trace(N)
        mu_dev = r.read_sum() / N
    ref_sum, ref_sumsq = _oracle_frame_stdgen(NREF)
    mu_ref = ref_sum / NREF
    var = np.maximum(ref_sumsq / NREF - mu_ref**2, 0.0) * NREF / (NREF - 1)
    se2 = var / N + var / NREF
    # At this field of view most pixels see the light with a small probability per sample, so a
    # pixel's 128 reference samples give neither a Gaussian mean nor a usable variance (measured:
    # 3 % of the per-pixel |z| exceed 4).  The z criterion is therefore applied to 8x8 BLOCK means
    # (64 pixels x 128 samples on the reference side), whose variance is the blocks' mean se2 / 64.
    def blocks(a):
        return a.reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
    bz_var = blocks(se2) / 64.0
    ok = bz_var > 0
    z = (blocks(mu_dev)[ok] - blocks(mu_ref)[ok]) / np.sqrt(bz_var[ok])
    assert ok.mean() > 0.9
    assert (np.abs(z) > 4).mean() < 1e-3, (np.abs(z) > 4).mean()
    assert abs(z.mean()) < 0.05 and 0.9 < z.std() < 1.1, (z.mean(), z.std())
    for k in range(3):
        bias = abs(mu_dev[..., k].mean() - mu_ref[..., k].mean()) / mu_ref[..., k].mean()
        assert bias < 5e-3, (k, bias)
    rmse = np.sqrt(np.mean((mu_dev - mu_ref) ** 2))
    pred = np.sqrt(np.mean(se2))
    assert 0.5 * pred <= rmse <= 1.2 * pred, (rmse, pred)
