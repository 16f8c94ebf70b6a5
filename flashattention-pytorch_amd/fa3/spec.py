"""Tile hints for FA3 (`FA3Spec`, `pick_fa3_spec`) — see common/tile_hints.py for the table and why they are hints."""
from common.tile_hints import make_spec_class, pick

FA3Spec = make_spec_class("FA3Spec", with_stages=True)


def pick_fa3_spec(head_dim: int):
    return pick(FA3Spec, head_dim)
