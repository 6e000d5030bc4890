"""`bench.py --gpus N` starts N ranks itself: the argument / environment matrix, and the launch path
end to end without a device (FHES_BENCH_LAUNCH_ONLY=1: every rank reports and leaves before anything touches HIP)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("gpus,env,want", [
    (1, {}, "run"),
    (2, {}, "spawn"),
    (8, {}, "spawn"),
    (8, {"WORLD_SIZE": ""}, "spawn"),
    (1, {"WORLD_SIZE": "1"}, "run"),
    (8, {"WORLD_SIZE": "8"}, "run"),          # the driver's own command: torch.distributed.run around us
    (8, {"WORLD_SIZE": "1"}, "error"),
    (1, {"WORLD_SIZE": "8"}, "error"),        # a forgotten --gpus under torchrun must not print n_gpus: 8 for "--gpus 1"
    (2, {"WORLD_SIZE": "4"}, "error"),
    (2, {"WORLD_SIZE": "two"}, "error"),
    (0, {}, "error"),
])
def test_launch_plan_matrix(gpus, env, want):
    assert bench.launch_plan(gpus, env)[0] == want


def _run(args, extra_env, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1"], {"FHES_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gpus_arg"] == 2


def test_a_failing_rank_makes_the_whole_command_fail():
    r = _run(["--gpus", "2"], {"FHES_BENCH_LAUNCH_ONLY": "1", "FHES_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "asked to fail" in r.stderr


def test_world_size_mismatch_is_refused_before_anything_runs():
    r = _run(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2
    assert "refusing" in r.stderr and r.stdout.strip() == ""
