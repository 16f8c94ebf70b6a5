#!/usr/bin/env python3
"""Placement table of the dK/dV stream kernel's vector work (csrc/fa_bwd_dkdv_w4.hip, kSched).

One (32-query x 64-key) block of the kernel is 64 MFMAs; between two MFMAs the wave has about 24 cycles of instruction
issue of its own (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost': v_exp 8, everything else 4, the MFMA's own 8
already taken out of its 32).  The operand requests (LDS reads + their counted wait) sit in front of every even MFMA and
take their share of that gap; what is left is the budget the block's other work has to fit into:

    MUL / EXP   one element of S' -> P = exp2(c S')             (kb, element): after the S' chain, before the dV products
    PC          one packed dword of P                           (kb, pair)
    SU          one packed dword of dS = P dP'                  (kb, pair): after the dP' chain, before the dK products
    ACC         row constants -> initial accumulator (4 LDS reads from asm: the stream's counted waits cover them), once the
                accumulator's last reader is behind
    DMA         one LDS-DMA piece of the tile three blocks ahead
    QADDR / TADDR / LADDR   operand addresses moved to the next tile's buffer, after their last use in this block

A gap that is over its budget stretches by the excess (tools/asm_gaps.py measures a built kernel the same way).  This
script packs the work greedily, earliest deadline first, and prints the table as C++.

    python tools/gen_dkdv_schedule.py [--prescaled] [--ds]    # --prescaled: no MUL (K tile already multiplied by c)
"""
import argparse


def reads(g):
    g %= 32
    return 3 if g < 8 else (1 if g < 16 else 2)


def budget(S, nop=True):
    m = S + 1                      # the MFMA this gap leads to
    if m % 2:
        return 24
    g = (m // 2) % 32
    # (the kernel waits once per two groups, but pricing a wait into EVERY request gap places better: filling the slots the
    # missing waits leave measured 39.4 cycles per MFMA against 38.7)
    return 24 - (4 * reads(g + 3) + 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prescaled", action="store_true")
    ap.add_argument("--ds", action="store_true")
    args = ap.parse_args()
    ops = []   # name, cost, earliest slice, deadline slice, deps (names that must sit in an EARLIER slice), same-slice deps

    def add(name, cost, lo, hi, before=(), same=()):
        ops.append(dict(name=name, cost=cost, lo=lo, hi=hi, before=tuple(before), same=tuple(same)))

    # S' of key block kb is complete after MFMA 14 + kb, a reader sits two MFMAs behind: slices >= 16 + kb.
    # P[kb][pairs 0..3] feeds MFMA 32 + kb, pairs 4..7 MFMA 40 + kb; one whole gap between the pack and the MFMA.
    for kb in (0, 1):
        for m in range(8):
            dl = (30 if m < 4 else 38) + kb
            for e in (2 * m, 2 * m + 1):
                if not args.prescaled:
                    add(f"MUL({kb},{e})", 4, 16 + kb, dl - 1)
                add(f"EXP({kb},{e})", 8, 16 + kb, dl - 1, same=() if args.prescaled else (f"MUL({kb},{e})",))
            add(f"PC({kb},{m})", 4, 17 + kb, dl, before=(f"EXP({kb},{2 * m})", f"EXP({kb},{2 * m + 1})"))
    # dP' of key block kb is complete after MFMA 30 + kb: slices >= 32 + kb.  dS[kb][pairs 0..3] feeds MFMA 48 + kb, 4..7 MFMA 56 + kb.
    for kb in (0, 1):
        for m in range(8):
            dl = (46 if m < 4 else 54) + kb
            add(f"SU({kb},{m})", 12, 32 + kb, dl, before=(f"EXP({kb},{2 * m})", f"EXP({kb},{2 * m + 1})"))
    # initial accumulators of the next block: S'[kb] / dP'[kb] are free once every SU of kb is behind
    for kb in (0, 1):
        sus = tuple(f"SU({kb},{m})" for m in range(8))
        add(f"ACC({kb})", 16, 40, 61, before=sus + ("LADDR",))        # S'[kb]
        add(f"ACC({2 + kb})", 16, 40, 61, before=sus + ("LADDR",))    # dP'[kb]
    add("LADDR", 4, 1, 60)                       # the row constants of this block were read before its first MFMA
    for j in range(5):
        add(f"DMA({j})", 16, 0, 40)              # the target buffer was last read in the previous block
    for i in range(8):
        add(f"QADDR({i})", 4, 10 + 2 * i, 56)    # Q rows / dO rows: last requested in front of MFMA 10 + 2 i; group 0 of the next block is requested in front of MFMA 58
    for j in range(4):
        add(f"TADDR({j})", 8, 56, 63)            # two addresses; last transposed request in front of MFMA 56

    if args.ds:
        for kb in (0, 1):
            for half in (0, 1):
                add(f"DSST({kb},{half})", 8, 40, 63, before=tuple(f"SU({kb},{4 * half + m})" for m in range(4)))

    placed, where = {S: [] for S in range(64)}, {}
    left = {S: budget(S) for S in range(64)}
    pending = list(ops)
    for S in range(64):
        progress = True
        while progress:
            progress = False
            ready = [o for o in pending if o["lo"] <= S and all(where.get(d, 99) < S for d in o["before"])
                     and all(where.get(d, 99) <= S for d in o["same"])]
            ready.sort(key=lambda o: (o["hi"], o["lo"]))
            for o in ready:
                # take it if it fits, or if it cannot wait any longer
                if o["cost"] <= left[S] or o["hi"] <= S:
                    placed[S].append(o["name"])
                    where[o["name"]] = S
                    left[S] -= o["cost"]
                    pending.remove(o)
                    progress = True
                    break
    assert not pending, pending
    over = sum(-v for v in left.values() if v < 0)
    print(f"// generated by tools/gen_dkdv_schedule.py{' --prescaled' if args.prescaled else ''}{' --ds' if args.ds else ''}: {over} cycles over budget in "
          f"{sum(1 for v in left.values() if v < 0)} gaps")
    width = max(len(v) for v in placed.values())
    print(f"// slice S (after MFMA S): up to {width} operations")
    for S in range(64):
        print(f"    /* {S:2d} ({budget(S):2d}) */ {{{', '.join(placed[S])}}},")


if __name__ == "__main__":
    main()
