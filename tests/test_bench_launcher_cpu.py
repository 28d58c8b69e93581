"""bench.py must run BY ITSELF at N > 1 (`python bench.py --gpus N`): the parent, which has made no GPU call, starts
torch.distributed.run as a child process, relays rank 0's JSON line and returns the child's exit code.  Exercised here
with 2 ranks on CPU (gloo) and the `--stub` step (no model): launcher, process group, barrier-bracketed timed region,
max-over-ranks clock, the per-step gather of the expression matrix in query order, one JSON line on stdout."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_self_launch_two_ranks_stub_step():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--genes-per-step", "3", "--tissues", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"exactly one line on stdout, got {lines}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["data"] == "stub"
    assert d["scaling"] == "weak" and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 2 * 3 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2      # whole-job aggregate over both ranks


def test_single_process_stub_and_launcher_mismatch():
    r = _run("--gpus", "1", "--steps", "2", "--warmup", "0", "--stub")
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1
    # started under a launcher whose world size disagrees with --gpus: refuse (a clear message, not an assert trace)
    r = _run("--gpus", "4", "--stub", env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_a_failing_rank_fails_the_parent():
    r = _run("--gpus", "2", "--stub", env_extra={"VF_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]       # and no result line
