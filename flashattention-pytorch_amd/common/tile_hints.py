"""Tile-hint records behind `fa{1,2,3}/spec.py`.

The reference's table (/root/reference/src/fa2/spec.py:9-12, src/fa3/spec.py:10-13): head_dim <= 64 -> br=128, bc=128,
otherwise br=64, bc=128; num_warps=8; FA3 adds stages=2.  The objects exist because callers construct and pass them;
on MI355X they are HINTS ONLY — the HIP library picks its own wave64 tiling and results do not depend
on them.
"""
from dataclasses import make_dataclass

_SMALL_HEAD = dict(br=128, bc=128)
_LARGE_HEAD = dict(br=64, bc=128)


def make_spec_class(name: str, with_stages: bool):
    fields = [("br", int), ("bc", int), ("num_warps", int)] + ([("stages", int)] if with_stages else [])
    return make_dataclass(name, fields, frozen=True)


def pick(spec_cls, head_dim: int):
    kw = dict(_SMALL_HEAD if head_dim <= 64 else _LARGE_HEAD, num_warps=8)
    if "stages" in spec_cls.__dataclass_fields__:
        kw["stages"] = 2
    return spec_cls(**kw)
