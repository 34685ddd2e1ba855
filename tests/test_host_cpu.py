"""Host-side codecs (PNG over zlib, base64) of the C++ twins of app/Write.hs /
app/Web.hs, exercised without a GPU through host/selftest."""
import base64
import os
import subprocess

import numpy as np
import pytest

from png_util import decode_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "host", "selftest")


@pytest.fixture(scope="module")
def selftest():
    if not os.path.exists(SELFTEST):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    return SELFTEST


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (64, 33), (500, 3)])
def test_png_and_base64_roundtrip(selftest, tmp_path, w, h):
    out = tmp_path / "t.png"
    text = subprocess.check_output([selftest, str(out), str(w), str(h)]).decode().strip()
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(3), indexing="ij")
    want = ((x * 7 + y * 13 + k * 101) & 255).astype(np.uint8)
    png = out.read_bytes()
    assert np.array_equal(decode_png(png), want)
    assert base64.b64decode(text) == png                   # imageToText = base64 . encodePng
    assert text == base64.b64encode(png).decode()


def test_write_without_gpu_fails_loudly(selftest):
    if os.path.exists("/dev/kfd"):
        pytest.skip("checks the no-GPU failure mode")
    r = subprocess.run([os.path.join(ROOT, "host", "write"), "--height", "8", "--width", "8", "--out", "/dev/null"],
                       capture_output=True)
    assert r.returncode == 1 and b"no HIP device" in r.stderr
