"""CPU test of bench.py's own launcher: `python bench.py --gpus N` (no WORLD_SIZE in the environment) must start N rank
processes itself, rendezvous them over 127.0.0.1, print ONE JSON line from rank 0 whose n_gpus is the number of ranks the
process group saw, and hand a failing rank's exit code back.  --dry-run keeps it on the CPU (gloo, no HIP library)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", *extra],
                          env=env, capture_output=True, text=True, timeout=240)


@pytest.mark.timeout(300)
def test_gpus_2_launches_two_ranks_unwrapped():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # one line, from rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line, key
    # the N > 1 line's diagnostics (bench.comm_diagnostics, the same code the RCCL run goes through): payload, a stand-alone
    # all-reduce's algorithm / bus bandwidth, the step with its collectives stubbed, every rank's launch mode
    assert line["ranks_seen"] == 2
    assert line["allreduce_payload_mb"] == round(4 * (3 * 4096 + 2048 + 512) / 1e6, 2) and len(line["allreduce_messages_mb"]) == 5
    assert line["allreduce_standalone_ms"] > 0 and line["allreduce_bus_gbs"] == pytest.approx(line["allreduce_alg_gbs"], abs=0.11)   # 2 (N - 1) / N = 1
    assert line["launch_mode_per_rank"] == ["dry run (rank 0)", "dry run (rank 1)"]
    assert line["ms_per_step_collectives_stubbed"] > 0
    assert line["comm_exposed_ms"] == pytest.approx(line["ms_per_step"] - line["ms_per_step_collectives_stubbed"], abs=2e-3)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", [4, 8])
def test_gpus_4_and_8_report_the_ranks_seen(n):
    """The driver's scaling run is `--gpus 1, 2, 4, 8` back to back: both launch forms at 4 and 8 ranks -- bench.py's own
    launcher, and `python -m torch.distributed.run --nproc-per-node N` with bench.py as one of the ranks (the driver's form)."""
    r = _run("--gpus", str(n))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == n, r.stdout
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + n), os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", str(n), "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == n, r.stdout


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch():
    r = _run("--gpus", "2", "--dry-run-fail-rank", "1")
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    r = _run("--gpus", "2", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_single_rank_needs_no_launcher():
    r = _run("--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
