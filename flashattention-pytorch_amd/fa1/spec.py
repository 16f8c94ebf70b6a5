"""Tile hints for FA1 (`FA1Spec`, `pick_fa1_spec`) — see common/tile_hints.py for the table and why they are hints."""
from common.tile_hints import make_spec_class, pick

FA1Spec = make_spec_class("FA1Spec", with_stages=False)


def pick_fa1_spec(head_dim: int):
    return pick(FA1Spec, head_dim)
