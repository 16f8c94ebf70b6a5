"""Tile / variant sweep on one MI355X: runs bench.py per variant (the library reads its override env vars once per
process) and tabulates per-kernel HIP-event times.  Usage: python tools/tile_sweep.py > gpurun_out/sweep.md"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    "C4 shard: B=8 H=32 N=4096 d=128 bf16": ["--batch", "8", "--heads", "32", "--seqlen", "4096", "--head-dim", "128"],
    "C3: B=4 H=32 N=8192 d=128 bf16 causal": ["--batch", "4", "--heads", "32", "--seqlen", "8192", "--head-dim", "128", "--causal"],
}
VARIANTS = [
    ("default (fwd 128-key tiles, lazy rescale; split bwd, 8-wave dK/dV)", {}),
    ("fwd K/V tile 32 keys", {"FA_FWD_KB": "1"}),
    ("fwd K/V tile 64 keys", {"FA_FWD_KB": "2"}),
    ("dK/dV 4 waves x 64 keys (512 regs/wave)", {"FA_DKDV": "w4"}),
    ("single-kernel bwd, dQ by float atomics", {"FA_BWD_VARIANT": "atomic"}),
]


def run(cfg_args, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "0"] + cfg_args,
                         env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return None, out.stderr[-300:]
    return json.loads(lines[-1]), None


def main():
    print("# Tile / variant sweep (HIP-event kernel times, ms per launch; bench.py --steps 5)\n")
    for cname, cargs in CONFIGS.items():
        print(f"## {cname}\n")
        print("| variant | fwd ms | bwd dK/dV (or fused) ms | bwd dQ ms | step ms | fwd+bwd TFLOP/s |")
        print("|---|---|---|---|---|---|")
        for vname, env in VARIANTS:
            j, err = run(cargs, env)
            if j is None:
                print(f"| {vname} | failed: {err} | | | | |")
                continue
            k = j["roofline"]["kernels"]
            g = lambda n: ("%.3f" % k[n]["avg_launch_ms"]) if n in k else "-"
            print(f"| {vname} | {g('fwd_mfma')} | {g('bwd_mfma')} | {g('bwd_dq_mfma')} | {j['ms_per_step']:.3f} | {j['value']:.1f} |", flush=True)
        print()


if __name__ == "__main__":
    main()
