#!/usr/bin/env python3
"""Fillers per MFMA gap of a kernel's main loop, priced as MI355X_MICROARCH.md prices vector issue (v_exp & co 8 cycles,
everything else 4): an MFMA leaves about 24 of its 32 cycles to the wave's other instructions, a gap whose fillers cost
more than that stretches.  Reads a hipcc -S file (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only X.hip -o X.s
    python tools/asm_gaps.py X.s <substring of the mangled kernel name> [--list]
"""
import collections
import re
import sys


def cost(op):
    return 8 if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")) else 4


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and key in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], []
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\S+:", l):
            blocks.append(cur)
            cur = []
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur.append(t.split()[0])
    blocks.append(cur)
    body = max(blocks, key=lambda b: sum(1 for t in b if t.startswith("v_mfma")))
    gaps, g = [], []
    for op in body:
        if op.startswith("v_mfma"):
            gaps.append(g)
            g = []
        else:
            g.append(op)
    gaps[0] = g + gaps[0]   # the loop's tail runs into its head
    n = len(gaps)
    costs = [sum(cost(o) for o in g) for g in gaps]
    over = sum(max(0, c - 24) for c in costs)
    print(f"{n} MFMAs, {sum(len(g) for g in gaps)} other instructions ({sum(costs) / n:.1f} cycles of issue per gap on average)")
    print(f"gaps over 24 cycles: {sum(1 for c in costs if c > 24)}; cycles over: {over} = {over / n:.2f} per MFMA; empty gaps: {sum(1 for c in costs if c == 0)}")
    print("instruction mix:", dict(collections.Counter(o for g in gaps for o in g).most_common(12)))
    if "--list" in sys.argv:
        for i, (g, c) in enumerate(zip(gaps, costs)):
            print(f"{i:3d} {c:3d} {' '.join(g)}")


if __name__ == "__main__":
    main()
