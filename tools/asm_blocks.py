#!/usr/bin/env python3
"""Per basic block of one kernel in a hipcc .s file: MFMA / scratch / v_mov / LDS / VALU counts (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only X.hip -o X.s
    python tools/asm_blocks.py X.s <substring of the mangled kernel name>
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l) and key in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    lab, order, stats = "entry", ["entry"], {"entry": {}}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            lab = m.group(1)
            order.append(lab)
            stats[lab] = {}
            continue
        t = l.strip().split(" ")[0].split("\t")[0]
        if not t or t.startswith(";") or t.startswith("."):
            continue
        d = stats[lab]
        kind = ("mfma" if t.startswith("v_mfma") else "scratch_st" if t.startswith("scratch_store") else
                "scratch_ld" if t.startswith("scratch_load") else "accmov" if t.startswith("v_accvgpr") else
                "vmov" if t.startswith("v_mov") else "ds" if t.startswith("ds_") else "vmem" if t.startswith(("buffer_", "global_")) else
                "valu" if t.startswith("v_") else "salu" if t.startswith("s_") else "other")
        d[kind] = d.get(kind, 0) + 1
    for lab in order:
        if stats[lab]:
            print(lab, " ".join(f"{k}={v}" for k, v in sorted(stats[lab].items())))


if __name__ == "__main__":
    main()
