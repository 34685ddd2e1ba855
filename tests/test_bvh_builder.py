"""Host-side checks of the spatial index's builder (pathtracer_amd/csrc/pt_bvh.hpp; no GPU): tests/bvh_check.cpp is
compiled with g++ against the header and run over random scenes.  What the device walk relies on is checked there --
node numbering, code words, leaf sizes, depth, and that every node's box as the device decodes it (binary16 centre and
half-extent) contains the padded boxes of the objects below it.  The reference has no index (closestObjectTo is a
linear scan, src/Lib.hs:81-98): this is an extension's own invariant, parity with the scan is tested on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bvh_check(tmp_path_factory):
    exe = tmp_path_factory.mktemp("bvh") / "bvh_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "pathtracer_amd", "csrc"),
                           "-o", str(exe), os.path.join(ROOT, "tests", "bvh_check.cpp")])
    return str(exe)


@pytest.mark.parametrize("n,seed,scale,offset", [
    (16, 1, 100.0, 0.0),            # the smallest scene that gets an index
    (17, 2, 1.0, 0.0),
    (1000, 3, 100.0, 0.0),
    (1000, 4, 1e-3, 0.0),           # tiny coordinates: binary16 subnormals
    (1000, 5, 100.0, 5000.0),       # boxes far from the origin: coarse binary16 centres
    (2000, 6, 1e5, 0.0),            # coordinates beyond binary16's range: infinite half-extents
    (10000, 7, 500.0, -300.0),
    (65536, 8, 2000.0, 0.0),        # the largest scene (more than 65,535 nodes: the 32-bit stack)
])
def test_builder_invariants(bvh_check, n, seed, scale, offset):
    out = subprocess.run([bvh_check, str(n), str(seed), str(scale), str(offset)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
    fields = dict(kv.split("=") for kv in out.stdout.split()[1:])
    assert int(fields["depth"]) <= 30                       # BVH_MAX_DEPTH: the device stack is sized from it
    assert int(fields["leaves"]) >= (n + 1) // 2            # leaves of at most two
    assert int(fields["nodes"]) == 2 * int(fields["leaves"])         # node 0 + a full binary tree of 2L - 1 nodes
