"""bench.py --gpus N typed as is (no torch.distributed.run): the parent starts
the N rank processes itself, before it imports torch or touches a GPU, and
relays rank 0's single JSON line.  Checked here without a GPU: the rank
environments, and a dry run in which the ranks only rendezvous over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_environments():
    import bench
    envs = bench.rank_environments(4, 29555, base={"PATH": "/bin", "WORLD_SIZE": "stale"})
    assert len(envs) == 4
    for r, e in enumerate(envs):
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r)
        assert e["WORLD_SIZE"] == "4" and e["LOCAL_WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29555"
        assert e["PATH"] == "/bin"


def _run(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PT_BENCH_DRY_RUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                     # stdout carries exactly one line, the JSON
    return json.loads(lines[0])


def test_self_launch_starts_n_ranks_and_relays_one_json_line():
    out = _run(3)
    assert out == {"dry_run": True, "n_gpus": 3, "rank_sum": 1 + 2 + 3, "local_rank_sum": 1 + 2 + 3}


def test_self_launch_reports_a_dead_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # no dry run and no GPU: every rank fails at "bench.py needs a GPU" (or on the box: this test is CPU-only)
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("needs a machine without a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""


def test_a_rank_that_never_gets_its_peers_gives_up():
    """--watchdog-s: a rank stuck in a rendezvous (here: a world of 2 of which only rank 0 exists) ends itself with
    status 124 and a line on stderr instead of holding its GPU until somebody else's limit."""
    import socket
    import time
    import bench
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = bench.rank_environments(2, port)[0]
    env.update(PT_BENCH_DRY_RUN="1", GLOO_SOCKET_IFNAME="lo")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--watchdog-s", "4"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 124 and "giving up" in p.stderr
    assert p.stdout.strip() == "" and time.time() - t0 < 60


def test_rank_to_visible_device():
    """A driver may give each rank its own HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES: the rank's GPU is then ordinal 0
    of what it can see, not its local rank."""
    import bench
    import pytest
    assert bench.visible_devices_from_env({}) is None
    assert bench.visible_devices_from_env({"HIP_VISIBLE_DEVICES": "3"}) == 1
    assert bench.visible_devices_from_env({"ROCR_VISIBLE_DEVICES": "0,1,2,3", "HIP_VISIBLE_DEVICES": "1,2"}) == 2
    assert bench.visible_devices_from_env({"CUDA_VISIBLE_DEVICES": "4,5,6,7"}) == 4
    assert bench.visible_devices_from_env({"HIP_VISIBLE_DEVICES": ""}) == 0
    assert [bench.device_for_rank(r, 8) for r in range(8)] == list(range(8))      # the whole node visible
    assert [bench.device_for_rank(r, 1) for r in range(8)] == [0] * 8             # one device per rank process
    assert [bench.device_for_rank(r, 4) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    for n in (0, None):
        with pytest.raises(RuntimeError):
            bench.device_for_rank(0, n)


def _run_with(n, fail_at):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PT_BENCH_DRY_RUN="1", PT_BENCH_DRY_FAIL=fail_at)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--watchdog-s", "60"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0]), p.stderr


def test_every_rank_falls_back_together_when_rank_0_cannot_make_an_id():
    """the id used to be made inside the try that held the broadcast, so a failure on rank 0 skipped the
    broadcast the other ranks were blocked in.  Now the broadcast always runs and the outcome is a MIN all-reduce:
    whatever fails, every rank executes the same gloo collectives and they all agree."""
    out, err = _run_with(3, "id")
    assert out["libpt_collective"] is False and out["in_step"] is True
    assert "no RCCL id" in err


def test_every_rank_falls_back_together_when_one_rank_cannot_join():
    out, err = _run_with(3, "init2")
    assert out["libpt_collective"] is False and out["in_step"] is True
    assert "pt_comm_init_rank" in err
    out, _ = _run_with(2, "nothing-fails")
    assert out["libpt_collective"] is True and out["in_step"] is True
