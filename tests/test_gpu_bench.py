"""bench.py itself on the GPU box: the one-GPU line, and the N > 1 control flow rehearsed with every rank on device 0
(PT_BENCH_REHEARSE_ONE_GPU=1: self-launched ranks, gloo rendezvous, sample split, fences, max of the time, sum of the
segments; libpt's ncclAllReduce runs in that loop over a communicator of ONE rank per rank process, because RCCL
refuses two ranks of one communicator on one device)."""
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

    two = _bench(["--gpus", "2", "--collective", "libpt"] + SMALL, {"PT_BENCH_REHEARSE_ONE_GPU": "1"})
    cfg = two["config"]
    assert two["n_gpus"] == 2 and "REHEARSAL" in cfg["collective"] and "libpt ncclAllReduce" in cfg["collective"]
    # libpt's collective sat in the two-rank loop: communicators of one rank (on this one-GPU box), one per pipelined
    # context of each rank process
    assert len(cfg["comm_info_per_context"]) == 2
    assert all(ci["n_ranks"] == 1 and ci["rccl_version"] > 0 for ci in cfg["comm_info_per_context"])
    assert len(cfg["per_rank_ms_per_step"]["all"]) == 2
    assert cfg["per_rank_ms_per_step"]["min"] <= cfg["per_rank_ms_per_step"]["max"]
    assert abs(cfg["per_rank_ms_per_step"]["max"] - two["ms_per_step"]) < 1e-3        # value is timed on the slowest rank
    assert cfg["collective_ms"]["rank0_mean"] > 0 and len(cfg["collective_ms"]["per_rank_mean"]) == 2
    # two ranks of 32 samples each trace the frame's 64: the same paths as one rank (Philox is keyed by the global
    # sample index), plus one more per-pixel primary search for the second rank's context
    assert two["config"]["paths_per_step"] == one["config"]["paths_per_step"]
    # (the step whose segments bench.py reports is each context's first, which runs the primary search: once per rank)
    assert two["config"]["segments_per_step"] == one["config"]["segments_per_step"] + 256 * 192
