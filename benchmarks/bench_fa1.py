#!/usr/bin/env python3
"""FlashAttention-1 sweep on the HIP backend: the reference's benchmarks/bench_fa1.py entry point (same common flags,
same records and result files, tag "fa1"), served by the one sweep driver in bench_compare_all.py.

    python benchmarks/bench_fa1.py --seqlen 4096 --head-dim 128 --batch-size 8 --num-heads 32 --dtypes bf16
"""
import sys

import bench_compare_all

if __name__ == "__main__":
    bench_compare_all.main(["--algos", "fa1"] + sys.argv[1:], default_tag="fa1")
