"""(batch x head) sharding of the attention hot path across the GPUs of one node.

The reference has no distributed code; what it does have is an outermost loop over independent
(b,h) problems (`csrc/fa2/fa2_fwd.cu:56`, `fa2_bwd.cu:59`).  On an 8 x MI355X node that loop is the natural
partition: one process per GPU, each rank owns a contiguous block of the merged BH axis, computes forward and
backward entirely locally (no collective on the data path), and — only when a caller needs the full tensors on
every rank — one RCCL all-gather per output over xGMI rebuilds them (`torch.distributed` backend "nccl" is RCCL
on ROCm; the node is fully connected, so a direct all-gather rides all seven links at once).

Everything here is plumbing around `forward_fn` / `backward_fn`, which default to the HIP extension's FA2 entry
points; the CPU tests inject the oracle instead and run the same code under `gloo` with world_size 2.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(bh: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of range(bh): the first bh % world ranks get one extra unit."""
    q, r = divmod(bh, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(bh: int, world: int) -> Sequence[int]:
    return [shard_bounds(bh, world, r)[1] - shard_bounds(bh, world, r)[0] for r in range(world)]


def take_shard(x: torch.Tensor, world: int, rank: int) -> torch.Tensor:
    """This rank's block of a full (BH, ...) tensor (a view)."""
    lo, hi = shard_bounds(x.shape[0], world, rank)
    return x[lo:hi]


def all_gather_bh(x: torch.Tensor, bh_total: int, group=None) -> torch.Tensor:
    """Gather every rank's (bh_r, ...) block into the full (bh_total, ...) tensor, on every rank.

    Equal blocks use one `all_gather_into_tensor` (a single fused collective); ragged splits pad to the largest
    block first.
    """
    world = dist.get_world_size(group)
    sizes = shard_sizes(bh_total, world)
    x = x.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((bh_total,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out
    big = max(sizes)
    pad = torch.zeros((big,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def _default_fns():
    from fa2.cuda.impl import _load_ext

    ext = _load_ext()
    return ext.forward, ext.backward


class ShardedAttention:
    """Forward / backward of attention over this rank's block of (b,h) units, with an optional final gather.

    forward_fn(q, k, v, causal, scale, br, bc) -> (o, lse)                 (extension signature, §8b)
    backward_fn(q, k, v, o, do, lse, causal, scale, br, bc) -> (dq, dk, dv)
    """

    def __init__(self, group=None, forward_fn: Optional[Callable] = None, backward_fn: Optional[Callable] = None,
                 br: int = 64, bc: int = 128):
        if forward_fn is None or backward_fn is None:
            f, b = _default_fns()
            forward_fn, backward_fn = forward_fn or f, backward_fn or b
        self.group, self.forward_fn, self.backward_fn, self.br, self.bc = group, forward_fn, backward_fn, br, bc
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def local(self, x_full: torch.Tensor) -> torch.Tensor:
        return take_shard(x_full, self.world, self.rank)

    def forward_backward_local(self, q, k, v, do, causal=False, softmax_scale=None):
        """q, k, v, do: this rank's (bh_r, N, d) blocks.  Returns local (o, lse, dq, dk, dv); no communication."""
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** -0.5
        o, lse = self.forward_fn(q, k, v, bool(causal), float(softmax_scale), self.br, self.bc)
        dq, dk, dv = self.backward_fn(q, k, v, o, do, lse, bool(causal), float(softmax_scale), self.br, self.bc)
        return o, lse, dq, dk, dv

    def gather(self, tensors: Sequence[torch.Tensor], bh_total: int):
        if self.world == 1:
            return list(tensors)
        return [all_gather_bh(t, bh_total, self.group) for t in tensors]

    def forward_backward(self, q_full, k_full, v_full, do_full, causal=False, softmax_scale=None, gather=True):
        """Full (BH, N, d) tensors on every rank in, full tensors out (the drop-in equivalence check of §8e)."""
        bh = q_full.shape[0]
        outs = self.forward_backward_local(self.local(q_full), self.local(k_full), self.local(v_full),
                                           self.local(do_full), causal, softmax_scale)
        return self.gather(outs, bh) if gather else outs
