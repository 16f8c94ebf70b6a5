"""Tile hints for FA2 (`FA2Spec`, `pick_fa2_spec`) — see common/tile_hints.py for the table and why they are hints."""
from common.tile_hints import make_spec_class, pick

FA2Spec = make_spec_class("FA2Spec", with_stages=False)


def pick_fa2_spec(head_dim: int):
    return pick(FA2Spec, head_dim)
