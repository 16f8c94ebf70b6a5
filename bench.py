#!/usr/bin/env python3
"""bench.py — attention fwd+bwd TFLOP/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--total-bh 2048]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work: without a launcher (no WORLD_SIZE in the environment) `--gpus N` starts N rank processes itself,
before this process has touched the GPU, and relays rank 0's line.

One "step" = one forward + one backward of FA2 attention over the rank's shard of (b,h) units, through the
reference-shaped wrapper (`fa2_cuda` autograd Function -> ctypes -> C-ABI -> HIP kernels), with q, k, v and dO
already resident in HBM.  Workload (default): BASELINE.json config 4's per-GPU shard, B=8 H=32 N=4096 d=128
bf16 non-causal = 256 independent (b,h) units per GPU; with N ranks every rank owns 256 units (weak scaling,
N=8 is config 4 exactly: B=64 H=32) and there is no collective on the data path; `--total-bh 2048` is the
strong-scaling form of config 4 (2048 units split over the ranks).  The final RCCL all-gather of
o/dq/dk/dv that would rebuild the full (B,H,N,d) tensors on every rank is timed separately ("gather_ms").

Prints ONE JSON line on rank 0 (fields: see the driver contract) including
  "roofline":     dominant kernel, algorithmic FLOPs per launch / HIP-event kernel time, vs 2.5 PFLOP/s dense bf16
  "cpu_baseline": the CPU oracle's tile loops (oracle/attention_oracle.py, a restatement of the reference's
                  src/fa1/torch/impl.py) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "flashattention-pytorch_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def alg_flops(bh, n, d, causal, direction):
    f = {"fwd": 4.0, "bwd": 10.0, "fwd+bwd": 14.0}[direction] * bh * n * n * d
    return f * ((n + 1) / (2.0 * n) if causal else 1.0)


def cpu_baseline(n, d, dtype, causal, budget_s):
    """Time the oracle's tiled fwd+bwd (the reference's CPU path restated) on a bounded number of (b,h) units."""
    from oracle import attention_oracle as orc
    from fa2.spec import pick_fa2_spec

    spec = pick_fa2_spec(d)
    # the reference's tile loops issue many small matmuls: beyond ~16 threads they get slower, not faster
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(0)
    units, elapsed = 0, 0.0
    t_all = time.perf_counter()
    while True:
        q, k, v, do = (torch.randn((1, n, d), generator=g).to(dtype) for _ in range(4))
        t0 = time.perf_counter()
        o, lse = orc.tiled_forward(q, k, v, causal, d ** -0.5, spec.br, spec.bc)
        orc.tiled_backward(q, k, v, o, do, lse, causal, d ** -0.5, spec.br, spec.bc)
        elapsed += time.perf_counter() - t0
        units += 1
        if time.perf_counter() - t_all > budget_s or units >= 64:
            break
    tflops = alg_flops(units, n, d, causal, "fwd+bwd") / elapsed / 1e12
    return {
        "value": round(tflops, 5), "unit": "TFLOP/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{units} of the (b,h) units of the same workload (N={n}, d={d}, {'causal' if causal else 'non-causal'}), "
                  f"oracle tiled_forward+tiled_backward (br={spec.br}, bc={spec.bc}), {elapsed:.1f} s of CPU time, "
                  f"host has {os.cpu_count()} logical cores",
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="batch per GPU (weak scaling: every rank owns batch x heads units)")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=list(DT))
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--total-bh", type=int, default=0,
                    help="strong scaling: this many (b,h) units in total, split over the ranks (config 4: 2048); "
                         "0 = weak scaling, batch x heads units per rank")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-gather", action="store_true", help="skip the separately timed RCCL all-gather")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing rehearsal without a GPU: gloo, CPU tensors, a placeholder step (no attention is "
                         "computed, value is null); exercises the launcher, the sharding, the barriers and the gather")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, as torch.distributed.run would), relay rank 0's JSON line, exit with the worst
    return code.  Runs BEFORE this process touches the GPU and never re-executes it: the children are new processes."""
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world   # under a launcher the launcher's world size is the truth
    dry = args.dry_run
    if not dry:
        assert torch.cuda.is_available(), "bench.py needs a GPU (use --dry-run to rehearse the launch plumbing on CPU)"
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    from common.shard import shard_bounds
    from fa2.spec import pick_fa2_spec

    B, H, N, D = args.batch, args.heads, args.seqlen, args.head_dim
    dtype = DT[args.dtype]
    strong = args.total_bh > 0
    if strong:   # fixed total, contiguous split of the merged (b,h) axis (common/shard.py)
        lo, hi = shard_bounds(args.total_bh, world, rank)
        bh, bh_total = hi - lo, args.total_bh
    else:        # every rank owns B x H units
        bh, bh_total = B * H, B * H * world
    if dry:
        N, D, bh = min(N, 64), min(D, 16), max(1, min(bh, 4))
        bh_total = bh * world if not strong else args.total_bh
    spec = pick_fa2_spec(D)
    scale = D ** -0.5
    g = torch.Generator(device=dev)
    g.manual_seed(rank)  # benchmarks/bench_utils.py:83-97 order q, k, v (+ dO)
    q, k, v = (torch.randn((bh, N, D), device=dev, dtype=dtype, generator=g).requires_grad_(True) for _ in range(3))
    do = torch.randn((bh, N, D), device=dev, dtype=dtype, generator=g)

    if dry:
        def step():   # placeholder with the step's tensor traffic shape; NOT attention (there is no CPU compute path)
            q.grad = k.grad = v.grad = None
            o = q + k + v
            torch.autograd.backward(o, do)
            return o
        ext = None
    else:
        import flashattention_lab_cuda as ext
        from fa2.cuda.impl import fa2_cuda

        def step():
            q.grad = k.grad = v.grad = None
            o, _ = fa2_cuda(q, k, v, args.causal, scale, spec)
            torch.autograd.backward(o, do)
            return o

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=None if dry else [local_rank])

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        o = step()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- per-kernel durations (HIP events on the launch stream, inside the library), separate untimed pass
    prof = {}
    if ext is not None:
        ext.profile_enable(True)
        for _ in range(args.steps):
            step()
        sync()
        prof = ext.profile_report()
        ext.profile_enable(False)

    # ---- separately timed all-gather of the outputs (rebuilds the full (BH_total, N, d) tensors on every rank)
    gather_ms = None
    if dist is not None and not args.no_gather:
        from common.shard import all_gather_bh

        outs = [t_ for t_ in (o.detach(), q.grad, k.grad, v.grad)]
        for rep in range(3):
            sync(); barrier(); t1 = time.perf_counter()
            full = [all_gather_bh(t_, bh_total) for t_ in outs]  # one fused collective per output (ragged: padded)
            sync(); barrier()
            gather_ms = (time.perf_counter() - t1) * 1e3
        assert all(f.shape[0] == bh_total for f in full)
        del full
        tg = torch.tensor([gather_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = tg.item()

    if rank == 0:
        total_flops = alg_flops(bh_total, N, D, args.causal, "fwd+bwd") * args.steps
        value = None if dry else total_flops / elapsed / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        # algorithmic FLOPs per launch of each kernel (per (b,h): S, dP, dV, dK, dQ = 2*N^2*d each, x (N+1)/2N causal).
        # The split backward prices its dK/dV kernel at the 4 products it alone is responsible for and the dQ kernel at
        # the one product it adds; the S / dP it recomputes are overhead, not algorithmic work (DESIGN.md "Roofline").
        gemm = alg_flops(bh, N, D, args.causal, "fwd") / 2.0
        fused = os.environ.get("FA_BWD_VARIANT") == "atomic"
        alg = {"fwd_mfma": 2 * gemm, "fwd_f32": 2 * gemm, "bwd_mfma": (5 if fused else 4) * gemm, "bwd_dq_mfma": gemm,
               "bwd_dkdv_f32": 4 * gemm, "bwd_dq_f32": gemm}
        traffic = {}
        tpath = next((pth for pth in (os.path.join(ROOT, "profiles", "r02_traffic.json"), os.path.join(ROOT, "profiles", "r01_traffic.json"))
                      if os.path.exists(pth)), None)
        default_cfg = (bh, N, D, args.dtype, args.causal) == (256, 4096, 128, "bf16", False) and not fused
        if default_cfg and tpath:  # PMC counters are collected in separate rocprofv3 --pmc passes
            traffic = json.load(open(tpath))
        kern = max((kname for kname in prof if kname in alg), key=lambda kname: prof[kname][1], default=None)
        roof = None
        if kern is not None:
            cnt, tot_ms = prof[kern]
            ach = alg[kern] / (tot_ms / cnt * 1e-3) / 1e12
            per_kernel = {}
            for kname, (c_, ms_) in prof.items():
                per_kernel[kname] = {"avg_launch_ms": round(ms_ / c_, 4)}
                if kname in alg:
                    per_kernel[kname]["achieved_tflops"] = round(alg[kname] / (ms_ / c_ * 1e-3) / 1e12, 1)
                if kname in traffic:
                    per_kernel[kname]["hbm_bytes"] = traffic[kname]["hbm_bytes"]
            if "bwd_delta" in prof and "bwd_dq_mfma" in per_kernel and D == 128:
                # dS hand-over: the dQ kernel is one product over the stored dS tiles, bound by reading them once.
                # Algorithmic bytes per launch: dS (N^2 * 2 per (b,h), the visible half + diagonal under the mask) + K + dQ.
                vis = (N + 1) / (2.0 * N) if args.causal else 1.0
                c_, ms_ = prof["bwd_dq_mfma"]
                launches_per_step = max(1.0, c_ / max(1, prof.get("fwd_mfma", (c_, 0))[0]))   # > 1: chunks of (b,h) units
                dq_bytes = bh * (N * N * 2.0 * vis + 2 * N * D * 2.0) / launches_per_step
                gbps = dq_bytes / (ms_ / c_ * 1e-3) / 1e9
                per_kernel["bwd_dq_mfma"].update({"bound": "hbm", "algorithmic_bytes_per_launch": dq_bytes, "achieved_GBps": round(gbps, 1),
                                                  "peak_GBps": 8000.0, "frac": round(gbps / 8000.0, 4)})
                # the dK/dV kernel's algorithmic bytes with the hand-over: q, k, v, dO in, dK, dV out, row constants, and the dS tiles
                # it writes for the dQ kernel (compare with hbm_bytes: measured traffic per launch)
                per_kernel["bwd_mfma"]["algorithmic_bytes_per_launch"] = bh * (6 * N * D * 2.0 + 8.0 * N + N * N * 2.0 * vis) / launches_per_step
            bwd_ms = sum(ms_ / c_ for kname, (c_, ms_) in prof.items() if kname.startswith("bwd"))
            all_ms = sum(ms_ / c_ for kname, (c_, ms_) in prof.items())
            roof = {"bound": "mfma", "kernel": kern, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4),
                    "traffic": traffic.get(kern, {}).get("hbm_bytes"),
                    "avg_launch_ms": round(tot_ms / cnt, 4), "algorithmic_flop_per_launch": alg[kern],
                    "timing": "HIP events recorded on the launch stream around each kernel (fa_profile_enable)",
                    "kernels": per_kernel,
                    "backward_all_kernels_tflops": round(5 * gemm / (bwd_ms * 1e-3) / 1e12, 1) if bwd_ms > 0 else None,
                    # the whole path (forward + both backward kernels) against the same peak: the honest headline fraction
                    "whole_step_frac": round(7 * gemm / (all_ms * 1e-3) / 1e12 / peak, 4) if all_ms > 0 else None}
        cpu = None
        if args.cpu_seconds > 0 and not dry:
            cpu = cpu_baseline(N, D, dtype, args.causal, args.cpu_seconds)
        per_rank = f"{bh} of {bh_total}" if strong else f"{bh}"
        line = {
            "metric": "attention fwd+bwd TFLOP/s (algorithmic 14*N^2*d per (b,h)), N=%d d=%d" % (N, D),
            "value": None if value is None else round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"FA2 fwd+bwd, {bh_total} (b,h) units in total ({per_rank} per GPU), H={H} N={N} d={D} {args.dtype} "
                                   f"{'causal' if args.causal else 'non-causal'} (BASELINE config 4: 2048 units over 8 GPUs = 256 per GPU)",
                       "bh_total": bh_total, "bh_per_gpu": bh, "heads": H, "seq_len": N, "head_dim": D, "causal": args.causal,
                       "parallelism": f"(b,h)-shard x{world}, no data-path collective"},
            "world_size": world, "backend": None if dist is None else dist.get_backend(),
            "per_gpu_tflops": None if value is None else round(value / world, 2),
            "frac_of_peak_per_gpu": None if value is None else round(value / world / peak, 4),
            "reference_convention_tflops": None if value is None else round(8.0 * bh_total * N * N * D * args.steps / elapsed / 1e12, 2),
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if dry:
            line["dry_run"] = True
        print(json.dumps(line), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
