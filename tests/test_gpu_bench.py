"""bench.py itself on the GPU box: the one-GPU line, and the N > 1 control flow rehearsed with every rank on device 0
(PT_BENCH_REHEARSE_ONE_GPU=1: self-launched ranks, gloo rendezvous, sample split, fences, max of the time, sum of the
segments; no collective, because RCCL refuses two ranks on one device -- the collectives themselves are tested with
worlds of 1 in tests/test_gpu_multi.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--height", "256", "--width", "192", "--spp", "64", "--bounces", "6", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]


def _bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_one_gpu_line_and_two_rank_rehearsal():
    one = _bench(SMALL, {})
    assert one["n_gpus"] == 1 and one["unit"] == "Mrays/s" and one["value"] > 0
    assert 0 < one["roofline"]["frac"] < 1 and one["roofline"]["kernel_ms"] > 0
    assert one["config"]["collective"] is None
    for key in ("kernel_ms_mode", "traffic_source"):
        assert key in one["roofline"]
    assert "value_mode" in one["config"] and "scene_kernel_compiler" in one["config"]

    two = _bench(["--gpus", "2"] + SMALL, {"PT_BENCH_REHEARSE_ONE_GPU": "1"})
    assert two["n_gpus"] == 2 and "rehearsal" in two["config"]["collective"]
    # two ranks of 32 samples each trace the frame's 64: the same paths as one rank (Philox is keyed by the global
    # sample index), plus one more per-pixel primary search for the second rank's context
    assert two["config"]["paths_per_step"] == one["config"]["paths_per_step"]
    # (the step whose segments bench.py reports is each context's first, which runs the primary search: once per rank)
    assert two["config"]["segments_per_step"] == one["config"]["segments_per_step"] + 256 * 192
