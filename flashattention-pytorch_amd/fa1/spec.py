"""Tile hints for FA1 — counterpart of /root/reference/src/fa1/spec.py.

The reference's table (d <= 64 -> br=128, bc=128; else br=64, bc=128; num_warps=8) is kept
field for field because callers construct and pass these objects; on MI355X they are HINTS ONLY: the
HIP library picks its own wave64 tiling ("Tile table") and results are tile independent.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class FA1Spec:
    br: int
    bc: int
    num_warps: int


def pick_fa1_spec(head_dim: int) -> FA1Spec:
    if head_dim <= 64:
        return FA1Spec(br=128, bc=128, num_warps=8)
    return FA1Spec(br=64, bc=128, num_warps=8)
