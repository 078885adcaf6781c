"""`python bench.py --gpus N` must start its own ranks (the driver runs exactly that command): the parent launches
torch.distributed.run as a child BEFORE touching the GPU and relays rank 0's single JSON line and the exit code.
Exercised here with --dry-run (gloo rendezvous, no GPU, no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus2_self_launches_two_ranks_and_prints_one_json_line():
    p = _run("--gpus", "2", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks"] == 2 and rec["config"]["rank_sum"] == 1.0


def test_a_failing_rank_fails_the_command():
    # WORLD_SIZE disagreeing with --gpus: the rank refuses; the parent must not report success
    p = _run("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
