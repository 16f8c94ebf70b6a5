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


def test_reference_command_lines_parse_unchanged():
    """Every flag of the reference's scripts is accepted (bench_compare_all.py:70-91, bench_fa3.py:50-70, bench_fa{1,2}.py:47-50)
    and --directions defaults to forward as it does there."""
    import bench_compare_all as bca

    a = bca.build_parser().parse_args([])
    assert a.directions == ["forward"] and a.tag == "compare_all" and a.no_plot is False and a.config_label is None and a.plot_dtype is None
    assert a.caption.startswith("Figure 6")
    a = bca.build_parser("fa3").parse_args(
        "--device cuda --seqlen 512 16384 --head-dim 64 128 256 --batch-size 1 2 --num-heads 4 --causal --dtypes fp16 bf16 "
        "--warmup 5 --iters 20 --fp8 --directions forward backward --tag run1 --config-label cfgA --plot-dtype bf16 --no-plot".split()
        + ["--caption", "Figure 6: x"])
    assert a.fp8 and a.directions == ["forward", "backward"] and a.tag == "run1" and a.config_label == "cfgA"
    assert a.plot_dtype == "bf16" and a.no_plot and a.caption == "Figure 6: x" and a.causal and a.seqlen == [512, 16384]
    assert bca.build_parser("fa2").parse_args(["--non-causal-only", "--no-plot"]).tag == "fa2"


def test_cpu_device_is_recorded_not_a_usage_error(tmp_path):
    """`--device cpu` is a legal reference command line; without a CPU backend every sweep point becomes an error record (with
    the --config-label in its config field), the table is printed and the exit status is non-zero with the reason."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "benchmarks", "bench_fa3.py"), "--device", "cpu", "--seqlen", "128", "--head-dim", "64",
                        "--batch-size", "1", "--dtypes", "fp16", "--fp8", "--no-plot", "--config-label", "lbl", "--no-save"],
                       capture_output=True, text=True, cwd=os.path.join(root, "benchmarks"))
    assert r.returncode != 0 and "no CPU backend" in r.stderr
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("FA3")]
    assert len(rows) == 4 and all("error" in ln for ln in rows), r.stdout   # causal x {bf16 path, fp8 path}
