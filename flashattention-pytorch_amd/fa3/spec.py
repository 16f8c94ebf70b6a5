"""Tile hints for FA3 — counterpart of /root/reference/src/fa3/spec.py.

The reference's table (d <= 64 -> br=128, bc=128; else br=64, bc=128; num_warps=8; stages=2) is kept
field for field because callers construct and pass these objects; on MI355X they are HINTS ONLY: the
HIP library picks its own wave64 tiling ("Tile table") and results are tile independent.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class FA3Spec:
    br: int
    bc: int
    num_warps: int
    stages: int


def pick_fa3_spec(head_dim: int) -> FA3Spec:
    if head_dim <= 64:
        return FA3Spec(br=128, bc=128, num_warps=8, stages=2)
    return FA3Spec(br=64, bc=128, num_warps=8, stages=2)
