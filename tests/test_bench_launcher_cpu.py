"""CPU: `python bench.py --gpus N` launches its own rank processes (the driver's invocation form) — rehearsed with
`--dry-run` (gloo, CPU tensors, placeholder step): N ranks come up, the barriers / MAX-over-ranks timing / all-gather run,
rank 0 prints ONE JSON line with the contract's fields, the exit code is the children's.  Weak and strong (--total-bh)
scaling.  Also: the torchrun form (WORLD_SIZE already set) does not spawn again."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=e,
                          timeout=600)


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    return json.loads(lines[0])


def test_self_launch_two_ranks_weak_scaling():
    r = _run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    for k in CONTRACT:
        assert k in d
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["steps"] == 2 and d["warmup"] == 1 and d["dry_run"] is True and d["value"] is None
    assert d["config"]["bh_total"] == 2 * d["config"]["bh_per_gpu"] and d["gather_ms"] is not None and d["ms_per_step"] > 0


def test_self_launch_strong_scaling_ragged_split():
    r = _run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--total-bh", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    assert d["scaling"] == "strong" and d["config"]["bh_total"] == 5 and d["config"]["bh_per_gpu"] == 3 and d["gather_ms"] is not None


def test_children_failures_become_the_exit_code():
    import torch

    if torch.cuda.is_available():
        return  # on a GPU box the real run would start; the failure path needs a machine without one
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0")
    assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_under_a_launcher_no_second_spawn():
    # WORLD_SIZE=1 in the environment = launched by torch.distributed.run with one rank: runs in place
    r = _run("--gpus", "1", "--dry-run", "--steps", "1", "--warmup", "0",
             env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    assert d["n_gpus"] == 1 and d["gather_ms"] is None
