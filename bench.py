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
strong-scaling form of config 4 (2048 units split over the ranks).  The global (BH_total, N, d) tensors are drawn
from ONE seed and every rank takes its `shard_bounds` slice of them; the final RCCL all-gather of o/dq/dk/dv that
rebuilds the full tensors on every rank is timed separately ("gather_ms") and rank 0 checks one other rank's block of
the gathered result against its own recomputation of that block ("gather_check").

Protocol: `--settle-seconds` of untimed steps (the chip's power governor settles over a window longer than 20 steps),
W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs; warm-up and timed loops are the
same code (same allocation pattern).  Every timed step is also bracketed by a pair of events on the launch stream.

Prints ONE JSON line on rank 0 (fields: see the driver contract) including
  "step_ms":      {min, median, max} of the per-step event times
  "kernel_ms_sum", "host_overhead_frac", "inconsistent": the library's per-kernel HIP-event times (a separate, untimed
                  pass of the same steps) summed, against the wall time per step — `inconsistent` is true when the wall
                  clock is more than 1.1 x the kernels
  "allocator":    device allocations / retries / OOMs of torch's caching allocator INSIDE the timed region (must be 0)
  "roofline":     dominant kernel, algorithmic FLOPs per launch / HIP-event kernel time, vs 2.5 PFLOP/s dense bf16;
                  algorithmic bytes per kernel (12 tensor passes + row constants) next to the PMC bytes of
                  the last profiled run (named by file: they are NOT measured in this process, so `traffic` is null)
  "cpu_baseline": the CPU oracle's tile loops (oracle/attention_oracle.py, a restatement of the reference's
                  src/fa1/torch/impl.py) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "flashattention-pytorch_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
TRAFFIC_FILES = ("r03_traffic.json", "r02_traffic.json")   # newest first; PMC passes of this command (tools/collect_traffic.py)


def alg_flops(bh, n, d, causal, direction):
    f = {"fwd": 4.0, "bwd": 10.0, "fwd+bwd": 14.0}[direction] * bh * n * n * d
    return f * ((n + 1) / (2.0 * n) if causal else 1.0)


def alg_bytes(bh, n, d, esize):
    """the minimum HBM traffic of the step as three kernels = 12 tensor passes + the row vectors.
    forward: read q, k, v, write o (+ lse);  dK/dV: read q, k, v, dO, write dK, dV (+ 2 row constants);
    dQ: read k, write dQ (q, v, dO again when it recomputes) — priced at its hand-over form, k + dQ + o/dO for the row constants."""
    t = float(bh) * n * d * esize
    row = float(bh) * n * 4
    return {"fwd_mfma": 4 * t + row, "fwd_f32": 4 * t + row, "bwd_mfma": 6 * t + 2 * row, "bwd_dkdv_f32": 6 * t + 2 * row,
            "bwd_dq_mfma": 2 * t, "bwd_dq_f32": 2 * t, "bwd_delta": 2 * t + 3 * row}


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
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: a >= 2 s timed region at the default workload)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed steps for this long before the warm-up: allocator and power governor in steady state")
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
    ap.add_argument("--no-gather", action="store_true", help="skip the separately timed RCCL all-gather and its check")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the gather + its check even with ONE rank: a rehearsal of the "
                         "multi-GPU code path on a one-GPU box")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing rehearsal without a GPU: gloo, CPU tensors, a placeholder step (no attention is "
                         "computed, value is null); exercises the launcher, the sharding, the barriers and the gather")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, as torch.distributed.run would), relay rank 0's JSON line, exit with the worst
    return code.  Runs BEFORE this process touches the GPU and never re-executes it: the children are new processes.
    The children are polled: when one dies the others (fresh children of this process) are terminated, so a bad rank
    ends the run instead of leaving rank 0 in a rendezvous until its timeout."""
    import socket
    import subprocess
    import tempfile

    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if rank == 0 else subprocess.DEVNULL, text=True))
    sock.close()   # held until the children exist: nobody else was handed this port in between
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    pr.terminate()
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = pr.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        rcs[i] = pr.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def draw_shards(bh_total, n, d, dtype, dev, seed, bounds):
    """q, k, v, dO of the GLOBAL (bh_total, n, d) problem from one seeded generator (order q, k, v, dO as
    benchmarks/bench_utils.py:83-97 + the tests' dO), returning for every (lo, hi) in `bounds` the four slices.
    Drawn in blocks of (b,h) units from per-block seeds, so no rank ever holds a whole global tensor: the value of unit u
    does not depend on who draws it."""
    out = [[] for _ in bounds]
    for t in range(4):
        parts = [[] for _ in bounds]
        for u0 in range(0, bh_total, 64):
            u1 = min(bh_total, u0 + 64)
            need = [(i, max(lo, u0), min(hi, u1)) for i, (lo, hi) in enumerate(bounds) if max(lo, u0) < min(hi, u1)]
            if not need:
                continue
            g = torch.Generator(device=dev)
            g.manual_seed((seed * 4 + t) * 1_000_003 + u0)
            blk = torch.randn((u1 - u0, n, d), device=dev, dtype=dtype, generator=g)
            for i, a, b in need:
                parts[i].append(blk[a - u0:b - u0])
        for i in range(len(bounds)):
            out[i].append(torch.cat(parts[i], dim=0) if len(parts[i]) != 1 else parts[i][0].clone())
    return out


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
    if world > 1 or args.force_dist:
        import datetime

        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        tmo = datetime.timedelta(seconds=120)
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)  # nccl == RCCL on ROCm

    from common.shard import shard_bounds
    from fa2.spec import pick_fa2_spec

    B, H, N, D = args.batch, args.heads, args.seqlen, args.head_dim
    dtype = DT[args.dtype]
    strong = args.total_bh > 0
    bh_total = args.total_bh if strong else B * H * world
    if dry:
        N, D = min(N, 64), min(D, 16)
        if not strong:
            bh_total = max(1, min(B * H, 4)) * world
    # contiguous split of the merged (b,h) axis (common/shard.py); weak scaling: equal blocks of B x H units
    lo, hi = shard_bounds(bh_total, world, rank)
    bh = hi - lo
    chk_rank = world - 1                        # rank 0 recomputes this rank's block after the gather
    chk_lo, chk_hi = shard_bounds(bh_total, world, chk_rank)
    spec = pick_fa2_spec(D)
    scale = D ** -0.5
    want = [(lo, hi)] + ([(chk_lo, chk_hi)] if rank == 0 and (world > 1 or args.force_dist) and not args.no_gather else [])
    shards = draw_shards(bh_total, N, D, dtype, dev, args.seed, want)
    q, k, v, do = shards[0]
    q, k, v = (t.requires_grad_(True) for t in (q, k, v))

    if dry:
        def run(q_, k_, v_, do_):   # placeholder with the step's tensor traffic shape; NOT attention (there is no CPU compute path)
            q_.grad = k_.grad = v_.grad = None
            o = q_ + k_ + v_
            torch.autograd.backward(o, do_)
            return o
        ext = None
    else:
        import flashattention_lab_cuda as ext
        from fa2.cuda.impl import fa2_cuda

        def run(q_, k_, v_, do_):
            q_.grad = k_.grad = v_.grad = None
            o, _ = fa2_cuda(q_, k_, v_, args.causal, scale, spec)
            torch.autograd.backward(o, do_)
            return o

    def step():
        return run(q, k, v, do)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=None if dry else [local_rank])

    # ---- settle + warm-up: the SAME loop as the timed one (the previous step's `o` stays alive across the call, as a
    # training loop's activations would), so the caching allocator is in its periodic state before the clock starts
    o = None
    settle_steps = 0
    t_settle = time.perf_counter()
    while not dry and time.perf_counter() - t_settle < args.settle_seconds:
        o = step()
        settle_steps += 1
        if settle_steps % 8 == 0:
            sync()
    for _ in range(args.warmup):
        o = step()
    sync()
    ev = None
    if not dry:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.reset_peak_memory_stats(dev)
        ms0 = torch.cuda.memory_stats(dev)
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev is not None:
            ev[i][0].record()
        o = step()
        if ev is not None:
            ev[i][1].record()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    step_ms, allocator = None, None
    if ev is not None:
        ms1 = torch.cuda.memory_stats(dev)
        per = sorted(a.elapsed_time(b) for a, b in ev)
        step_ms = {"min": round(per[0], 4), "median": round(statistics.median(per), 4), "max": round(per[-1], 4)}
        allocator = {key: int(ms1.get(key, 0) - ms0.get(key, 0)) for key in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")}
        allocator["peak_allocated_GiB"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 3)
        allocator["reserved_GiB"] = round(torch.cuda.memory_reserved(dev) / 2 ** 30, 3)
        allocator["shim_workspace_GiB"] = round(ext.workspace_stats()["bytes"] / 2 ** 30, 3)

    # ---- per-kernel durations (HIP events on the launch stream, inside the library), separate untimed pass
    prof, prof_steps = {}, min(args.steps, 50)
    if ext is not None:
        ext.profile_enable(True)
        for _ in range(prof_steps):
            o = step()
        sync()
        prof = ext.profile_report()
        ext.profile_enable(False)

    # ---- separately timed all-gather of the outputs (rebuilds the full (BH_total, N, d) tensors on every rank), and the
    # drop-in check: rank 0 recomputes rank `chk_rank`'s block from the same global inputs and compares it with the gathered one
    gather_ms, gather_check = None, None
    if dist is not None and not args.no_gather:
        from common.shard import all_gather_bh

        outs = [t_ for t_ in (o.detach(), q.grad, k.grad, v.grad)]
        for rep in range(3):
            sync(); barrier(); t1 = time.perf_counter()
            full = [all_gather_bh(t_, bh_total) for t_ in outs]  # one fused collective per output (ragged: padded)
            sync(); barrier()
            gather_ms = (time.perf_counter() - t1) * 1e3
        assert all(f.shape[0] == bh_total for f in full)
        tg = torch.tensor([gather_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = tg.item()
        if rank == 0:
            q2, k2, v2, do2 = shards[1]
            q2, k2, v2 = (t.requires_grad_(True) for t in (q2, k2, v2))
            o2 = run(q2, k2, v2, do2)
            sync()
            diffs = [(f[chk_lo:chk_hi].float() - t_.detach().float()).abs().max().item()
                     for f, t_ in zip(full, (o2, q2.grad, k2.grad, v2.grad))]
            gather_check = {"block_of_rank": chk_rank, "units": [chk_lo, chk_hi], "recomputed_on_rank": 0,
                            "max_abs_diff": {"o": diffs[0], "dq": diffs[1], "dk": diffs[2], "dv": diffs[3]},
                            "bitwise_equal": all(torch.equal(f[chk_lo:chk_hi], t_.detach()) for f, t_ in
                                                 zip(full, (o2, q2.grad, k2.grad, v2.grad)))}
            del q2, k2, v2, do2, o2
        del full

    if rank == 0:
        total_flops = alg_flops(bh_total, N, D, args.causal, "fwd+bwd") * args.steps
        value = None if dry else total_flops / elapsed / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        ms_per_step = elapsed / args.steps * 1e3
        # algorithmic FLOPs per STEP of each kernel (per (b,h): S, dP, dV, dK, dQ = 2*N^2*d each, x (N+1)/2N causal).
        # The split backward prices its dK/dV kernel at the 4 products it alone is responsible for and the dQ kernel at
        # the one product it adds; the S / dP it recomputes are overhead, not algorithmic work (DESIGN.md "Roofline").
        gemm = alg_flops(bh, N, D, args.causal, "fwd") / 2.0
        fused = os.environ.get("FA_BWD_VARIANT") == "atomic"
        alg = {"fwd_mfma": 2 * gemm, "fwd_f32": 2 * gemm, "bwd_mfma": (5 if fused else 4) * gemm, "bwd_dq_mfma": gemm,
               "bwd_dkdv_f32": 4 * gemm, "bwd_dq_f32": gemm}
        abytes = alg_bytes(bh, N, D, 4 if args.dtype == "fp32" else 2)
        # PMC traffic is collected by rocprofv3 --pmc in separate passes of this command and is NOT measured here
        tpath = next((os.path.join("profiles", f) for f in TRAFFIC_FILES if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
        default_cfg = (bh, N, D, args.dtype, args.causal) == (256, 4096, 128, "bf16", False) and not fused
        profiled = json.load(open(os.path.join(ROOT, tpath))) if (default_cfg and tpath) else {}
        kern = max((kname for kname in prof if kname in alg), key=lambda kname: prof[kname][1], default=None)
        roof, kernel_ms_sum = None, None
        if kern is not None:
            per_kernel = {}
            kernel_ms_sum = sum(ms_ for (_c, ms_) in prof.values()) / prof_steps
            handover = "bwd_delta" in prof and D == 128 and prof.get("bwd_dq_mfma", (0, 0))[0] >= prof.get("bwd_delta", (1, 0))[0]
            vis = (N + 1) / (2.0 * N) if args.causal else 1.0
            for kname, (c_, ms_) in prof.items():
                lps = c_ / prof_steps                         # launches per step (> 1: the dS hand-over's chunks of (b,h) units)
                e = {"launches_per_step": round(lps, 3), "avg_launch_ms": round(ms_ / c_, 4), "ms_per_step": round(ms_ / prof_steps, 4)}
                if kname in alg:
                    e["algorithmic_flop_per_launch"] = alg[kname] / lps
                    e["achieved_tflops"] = round(alg[kname] / (ms_ / prof_steps * 1e-3) / 1e12, 1)
                    e["frac_of_mfma_peak"] = round(e["achieved_tflops"] / peak, 4)
                if kname in abytes:
                    e["algorithmic_bytes_per_step"] = abytes[kname]
                    if handover and kname in ("bwd_mfma", "bwd_dq_mfma"):
                        # the dS tiles handed from the dK/dV kernel to the dQ kernel are extra traffic BY DESIGN:
                        # named separately, not folded into the algorithmic bytes
                        e["handover_bytes_per_step"] = bh * N * N * 2.0 * vis
                    e["algorithmic_GBps"] = round(abytes[kname] / (ms_ / prof_steps * 1e-3) / 1e9, 1)
                if kname in profiled:
                    e["profiled_hbm_bytes_per_step"] = profiled[kname]["hbm_bytes"] * profiled[kname].get("launches_per_step", 1)
                    if kname in abytes:
                        e["profiled_over_algorithmic"] = round(e["profiled_hbm_bytes_per_step"] / abytes[kname], 2)
                per_kernel[kname] = e
            if handover:
                e = per_kernel["bwd_dq_mfma"]   # one product over the stored dS tiles: bound by reading them once
                moved = e["algorithmic_bytes_per_step"] + e["handover_bytes_per_step"]
                e.update({"bound": "hbm", "moved_GBps": round(moved / (e["ms_per_step"] * 1e-3) / 1e9, 1), "peak_GBps": PEAK_HBM_GBPS})
                e["frac_of_hbm_peak"] = round(e["moved_GBps"] / PEAK_HBM_GBPS, 4)
            cnt, tot_ms = prof[kern]
            ach = alg[kern] / (tot_ms / prof_steps * 1e-3) / 1e12
            bwd_ms = sum(ms_ for kname, (c_, ms_) in prof.items() if kname.startswith("bwd")) / prof_steps
            alg_step_bytes = sum(abytes[kname] for kname in prof if kname in abytes)
            roof = {"bound": "mfma", "kernel": kern, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4),
                    "traffic": None,
                    "traffic_source": (f"{tpath}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on an earlier run "
                                       "(tools/collect_traffic.py), per kernel under kernels.*.profiled_hbm_bytes_per_step; not measured in this process")
                    if profiled else None,
                    "avg_launch_ms": round(tot_ms / cnt, 4), "launches_per_step": round(cnt / prof_steps, 3),
                    "algorithmic_flop_per_launch": alg[kern] / (cnt / prof_steps),
                    "timing": "HIP events recorded on the launch stream around each kernel (fa_profile_enable), "
                              f"{prof_steps} untimed steps after the timed region",
                    "kernels": per_kernel,
                    "algorithmic_bytes_per_step": alg_step_bytes,
                    "backward_all_kernels_tflops": round(5 * gemm / (bwd_ms * 1e-3) / 1e12, 1) if bwd_ms > 0 else None,
                    # the whole path (forward + all backward kernels) against the same peak: the honest headline fraction
                    "whole_step_frac": round(7 * gemm / (kernel_ms_sum * 1e-3) / 1e12 / peak, 4) if kernel_ms_sum > 0 else None}
        cpu = None
        if args.cpu_seconds > 0 and not dry:
            cpu = cpu_baseline(N, D, dtype, args.causal, args.cpu_seconds)
        per_rank = f"{bh} of {bh_total}" if strong else f"{bh}"
        line = {
            "metric": "attention fwd+bwd TFLOP/s (algorithmic 14*N^2*d per (b,h)), N=%d d=%d" % (N, D),
            "value": None if value is None else round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"FA2 fwd+bwd, {bh_total} (b,h) units in total ({per_rank} per GPU), H={H} N={N} d={D} {args.dtype} "
                                   f"{'causal' if args.causal else 'non-causal'} (BASELINE config 4: 2048 units over 8 GPUs = 256 per GPU)",
                       "bh_total": bh_total, "bh_per_gpu": bh, "heads": H, "seq_len": N, "head_dim": D, "causal": args.causal,
                       "parallelism": f"(b,h)-shard x{world}, no data-path collective", "seed": args.seed},
            "world_size": world if dist is None else dist.get_world_size(), "backend": None if dist is None else dist.get_backend(),
            "per_gpu_tflops": None if value is None else round(value / world, 2),
            "frac_of_peak_per_gpu": None if value is None else round(value / world / peak, 4),
            "reference_convention_tflops": None if value is None else round(8.0 * bh_total * N * N * D * args.steps / elapsed / 1e12, 2),
            "settle": {"seconds": args.settle_seconds, "steps": settle_steps},
            "timed_region_s": round(elapsed, 4),
            "step_ms": step_ms,
            "kernel_ms_sum": None if kernel_ms_sum is None else round(kernel_ms_sum, 4),
            "host_overhead_frac": None if kernel_ms_sum is None else round(ms_per_step / kernel_ms_sum - 1.0, 4),
            "inconsistent": None if kernel_ms_sum is None else bool(ms_per_step > 1.1 * kernel_ms_sum),
            "allocator": allocator,
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "gather_check": gather_check,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if dry:
            line["dry_run"] = True
        if line["inconsistent"]:
            print(f"bench.py: INCONSISTENT: {ms_per_step:.3f} ms per step on the wall clock against {kernel_ms_sum:.3f} ms of kernels "
                  f"(allocator in the timed region: {allocator})", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
