"""CPU: the benchmark harness keeps the reference's record schema, FLOP convention and flags
(/root/reference/benchmarks/bench_utils.py:161-224,247-263,287-325)."""
import argparse
import csv
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
import bench_utils as bu  # noqa: E402


def test_flop_convention_and_flags():
    assert bu.attention_flops(2, 4, 128, 64, "forward") == 4.0 * 2 * 4 * 128 * 128 * 64
    assert bu.attention_flops(2, 4, 128, 64, "backward") == 8.0 * 2 * 4 * 128 * 128 * 64
    assert abs(bu.compute_tflops(1, 1, 4096, 128, 1.0, "forward") - 4.0 * 4096 * 4096 * 128 / 1e-3 / 1e12) < 1e-9
    assert bu.compute_tflops(1, 1, 8, 8, None, "forward") is None
    ap = argparse.ArgumentParser()
    bu.add_common_args(ap)
    a = ap.parse_args([])
    assert a.seqlen == [512, 1024, 2048, 4096, 8192, 16384] and a.head_dim == [64, 128, 256]
    assert a.batch_size == [1, 2] and a.num_heads == [4] and a.dtypes == ["fp16", "bf16"] and (a.warmup, a.iters) == (5, 20)
    assert bu.iter_causal_flags(a) == [False, True]
    assert bu.iter_causal_flags(ap.parse_args(["--causal"])) == [True]


def test_record_schema_roundtrip(tmp_path):
    r = bu.BenchmarkRecord("FA2", "fa2", "cuda", "backward", "bf16", True, 4096, 128, 8, 32, 9.0, 0.1, 500.0, 1024.0, "ok", None, "x", None)
    assert list(r.to_dict().keys()) == bu.FIELDS
    paths = bu.write_results("unit test", [r], out_dir=str(tmp_path))
    assert json.load(open(paths["json"]))[0]["seqlen"] == 4096
    rows = list(csv.DictReader(open(paths["csv"])))
    assert rows[0]["method"] == "FA2" and list(rows[0].keys()) == bu.FIELDS
    assert bu.load_results([paths["json"]])[0] == r
    assert "B8 H32 N4096 D128" in r.to_row()


def test_make_qkv_is_the_seeded_generator_order():
    import torch

    q, k, v = bu.make_qkv(1, 2, 8, 4, "cpu", torch.float32)
    g = torch.Generator().manual_seed(0)
    assert torch.equal(q, torch.randn((1, 2, 8, 4), generator=g)) and torch.equal(k, torch.randn((1, 2, 8, 4), generator=g))
    mean, std, mem = bu.benchmark_fn(lambda: q @ k.transpose(-1, -2), "cpu", 1, 3)
    assert mean > 0 and std >= 0 and mem is None


def test_per_algorithm_entry_points_exist_and_refuse_to_run_without_a_gpu():
    """benchmarks/bench_fa{1,2,3}.py mirror the reference's per-algorithm scripts; with no GPU they must exit loudly
    (no CPU fallback), not print an empty table."""
    import subprocess

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for a in (1, 2, 3):
        path = os.path.join(root, "benchmarks", f"bench_fa{a}.py")
        assert os.path.exists(path)
        if not torch.cuda.is_available():
            r = subprocess.run([sys.executable, path, "--seqlen", "128", "--no-save"], capture_output=True, text=True,
                               cwd=os.path.join(root, "benchmarks"))
            assert r.returncode != 0 and "no CPU backend" in (r.stderr + r.stdout)
