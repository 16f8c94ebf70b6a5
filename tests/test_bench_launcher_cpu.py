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
    # one global problem, sliced by rank: rank 0 recomputed rank 1's block from the same global inputs and found it in the gather
    chk = d["gather_check"]
    assert chk["block_of_rank"] == 1 and chk["units"] == [4, 8] and chk["bitwise_equal"] is True and max(chk["max_abs_diff"].values()) == 0.0


def test_self_launch_strong_scaling_ragged_split():
    r = _run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--total-bh", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    assert d["scaling"] == "strong" and d["config"]["bh_total"] == 5 and d["config"]["bh_per_gpu"] == 3 and d["gather_ms"] is not None
    assert d["gather_check"]["units"] == [3, 5] and d["gather_check"]["bitwise_equal"] is True      # the ragged (padded) gather too


def test_global_tensors_do_not_depend_on_who_draws_them():
    """Unit u of the global problem has the same values whichever rank draws it (per-block seeded draws): slices drawn with
    different bounds agree where they overlap."""
    import importlib.util

    import torch

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.draw_shards(200, 8, 4, torch.float32, torch.device("cpu"), 3, [(0, 200)])[0]
    b = bench.draw_shards(200, 8, 4, torch.float32, torch.device("cpu"), 3, [(50, 130), (190, 200)])
    for t in range(4):
        assert torch.equal(a[t][50:130], b[0][t]) and torch.equal(a[t][190:200], b[1][t])
    c = bench.draw_shards(200, 8, 4, torch.float32, torch.device("cpu"), 4, [(0, 10)])[0]
    assert not torch.equal(c[0], a[0][:10]) and not torch.equal(a[0][:10], a[1][:10])     # seed and tensor index both matter


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
