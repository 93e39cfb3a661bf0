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


def test_line_is_short():
    """The driver parses ONE stdout line; round 5's had grown to 20.8 KB and was not parsed (BENCH_r05.parsed = null).
    The line is a fixed flat selection: < 4096 bytes on the dry-launch path AND for a full default-run report (round 5's
    recorded one, profiles/r05_bench_default_v2.json, fattened further), with `roofline` and `cpu_baseline` present."""
    sys.path.insert(0, ROOT)
    import bench
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT == 4096
    with open(os.path.join(ROOT, "profiles", "r05_bench_default_v2.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 4 * bench.LINE_LIMIT                  # the report that broke the parser
    full["other_configs"]["more"] = [full["other_configs"]] * 3          # growth of the detail must not reach the line
    full["config"]["workload"] = full["config"]["workload"] * 4
    full["summary"]["a_nested_block"] = {"x": [1] * 1000}
    line = bench.short_line(full, "gpurun_out/bench_detail.json")
    assert len(line) < bench.LINE_LIMIT and "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in out, k
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                                    "algorithmic_bytes_per_launch"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert all(not isinstance(v, (dict, list)) for v in out["summary"].values())
    assert abs(out["value"] / full["value"] - 1) < 1e-5 and out["steps"] == full["steps"]
    # a summary that cannot fit is shed before the contract keys are
    full["summary"] = {f"k{i}": float(i) for i in range(400)}
    out = json.loads(bench.short_line(full, None))
    assert "summary" not in out and "roofline" in out and "cpu_baseline" in out


def test_detail_file_holds_the_full_report(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_bench_default_v2.json")) as f:
        full = json.load(f)
    rd, wr = os.pipe()
    path = str(tmp_path / "sub" / "detail.json")
    bench.emit(full, wr, path)
    os.close(wr)
    line = json.loads(os.read(rd, 1 << 16).decode())
    os.close(rd)
    assert line["detail"] == path
    with open(path) as f:
        assert json.load(f) == full
