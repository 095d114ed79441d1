"""bench.py's launch contract (no GPU): `python bench.py --gpus N` starts N ranks by itself, never fewer, and exactly one JSON
line reaches stdout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen_by_rccl"] == 2 and j["config"]["self_launched"] is True
    assert j["config"]["rank_id_sum"] == 3.0          # ranks 0 and 1 both took part in the all-reduce


def test_under_a_launcher_the_world_size_must_equal_gpus():
    # WORLD_SIZE=1 with --gpus 2: a launcher that started the wrong number of ranks is an error, not a one-rank bench
    r = _run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)


def test_fewer_devices_than_ranks_is_refused():
    # this container has no HIP device: a real (non-dry) --gpus 2 must refuse, not fall back to fewer ranks
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 devices")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr
    assert r.stdout.strip() == ""
