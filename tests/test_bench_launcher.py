"""bench.py --gpus N without an outer torchrun: the launcher starts N ranks itself (the reference is
single-GPU, train.py:33).  CPU test: `--dry-launch` runs the same launch over gloo with the iteration's two collectives
on host buffers."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=600, env=e)


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_launch"] is True and out["steps"] == 3
    c = out["config"]
    assert c["rccl_ranks"] == 2 and c["objects_per_rank"] == [50, 50] and c["collectives_per_step"] == 2
    assert c["collectives_ok"] is True


def test_strong_scaling_config_deals_objects_to_the_ranks():
    r = _run("--gpus", "2", "--steps", "1", "--config", "c4", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["config"]["objects_per_rank"] == [60, 60] and out["scaling"] == "strong"


def test_more_ranks_than_gpus_is_refused():
    """No GPU in the build container: --gpus 2 must refuse loudly (count read in a child process) and print no line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs present")
    r = _run("--gpus", "2", "--steps", "1")
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr


def test_a_failing_rank_fails_the_launch():
    """A rank that exits non-zero turns into a non-zero exit of the launcher and NO line (OBJNERF_BENCH_FAIL_RANK is the
    dry launch's test hook: that rank returns 3 after the run)."""
    r = _run("--gpus", "2", "--dry-launch", "--steps", "1", env={"OBJNERF_BENCH_FAIL_RANK": "1"})
    # (rank 0 has normally finished by then: [0, 3]; had it not, the launcher would have stopped it: [-15, 3])
    assert r.returncode != 0 and r.stdout.strip() == "" and ", 3]" in r.stderr and "exit codes [" in r.stderr


def test_a_rank_that_dies_before_the_rendezvous_ends_the_launch_quickly():
    """Rank 1 exits before init_process_group: rank 0 would sit in the rendezvous until torch's timeout (minutes); the
    launcher supervises every child, stops rank 0 (the child it started) and fails within seconds."""
    import time
    t0 = time.time()
    r = _run("--gpus", "2", "--dry-launch", "--steps", "1", env={"OBJNERF_BENCH_DIE_EARLY": "1"})
    assert r.returncode != 0 and r.stdout.strip() == "" and ", 4]" in r.stderr
    assert time.time() - t0 < 90
