"""CPU, world_size 2 over gloo: the (b,h) sharding + gather plumbing (`common/shard.py`) with the oracle injected
as the compute function — the N>1 path of bench.py without GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import shard


def test_shard_bounds_cover_and_balance():
    for bh in (1, 5, 8, 255, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(bh, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == bh
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard.shard_sizes(2048, 8) == [256] * 8  # BASELINE config 4: 256 (b,h) units per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bh, causal, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import attention_oracle as orc

        def fwd(q_, k_, v_, causal_, scale, br, bc):
            return orc.exact_attention(q_, k_, v_, causal_, scale)

        def bwd(q_, k_, v_, o_, do_, lse_, causal_, scale, br, bc):
            return orc.exact_attention_backward(q_, k_, v_, do_, causal_, scale)[:3]

        g = torch.Generator().manual_seed(3)
        full = [torch.randn(bh, 24, 16, generator=g) for _ in range(4)]
        sa = shard.ShardedAttention(forward_fn=fwd, backward_fn=bwd)
        assert sa.world == world and sa.rank == rank
        outs = sa.forward_backward(*full, causal=causal)
        ro, rlse = orc.exact_attention(full[0], full[1], full[2], causal, 0.25)
        rq, rk, rv, _, _ = orc.exact_attention_backward(*full, causal, 0.25)
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(outs, (ro, rlse, rq, rk, rv)))
        ok = ok and outs[0].shape == full[0].shape and outs[1].shape == (bh, 24)
        local = sa.forward_backward(*full, causal=causal, gather=False)
        lo, hi = shard.shard_bounds(bh, world, rank)
        ok = ok and local[0].shape[0] == hi - lo and torch.allclose(local[0], ro[lo:hi], atol=1e-6)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bh,causal", [(8, False), (5, True)])  # even split (one fused all-gather) and ragged split
def test_sharded_forward_backward_matches_unsharded(bh, causal):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bh, causal, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
