#!/usr/bin/env python3
"""Static MFMA hazard audit of the built code object (no GPU needed).

hipcc pads the data hazards of the instructions IT emits, but nothing around inline asm: an `asm` MFMA whose result a
later instruction reads too early returns stale data on some waves of some launches, and a green test run cannot prove
that it does not happen (one such bug was met in round 1).  This tool disassembles every gfx950 code object
embedded in libfa_mi355x.so and checks, for EVERY v_mfma (asm or builtin):

  R1  result -> first consumer: from an MFMA to the first later instruction that reads or writes any register of its
      destination tile — other than an MFMA that takes the whole tile as its C operand and writes it back (an
      accumulation chain: no wait states needed) — at least passes + 3 wait states (passes + 2 for the f32-input forms) must lie
      (LLVM's GFX940 rule "XDL write VGPR -> VALU / memory / export read or write", the number hipcc itself pads to).
      One relaxation, LLVM's "SMFMA write VGPR -> overlapped SrcC of an SMFMA": the f32-input forms (v_mfma_f32_*_f32) are not
      XDL operations, and another of them may take such a result as its C operand alone (not as A / B, not overwriting it) after
      passes - 2 states — hipcc's own code does (fa_generic.hip carries one tile as the initial value of several accumulators).
  R2  VALU-written operand -> MFMA: an MFMA may not read, as A or B operand, a register that a VALU instruction wrote
      in the previous instruction slot pair (2 wait states: fa_common.h's mfma32_v carries an `s_nop 1` for it).

Wait states are counted the way LLVM's hazard recognizer counts them — every instruction issued is one state, `s_nop N`
is N + 1 — with one sound refinement: the SIMD has ONE in-order matrix pipe, so an MFMA cannot issue before the previous
MFMA of the wave has left it (`passes` states after that one issued).  Two independent MFMAs between a chain and its
first reader therefore cover the distance; one does not.  Control flow: the scan follows both the fall-through and the target of every branch (labels come from the
disassembler's `<func+0xOFF>` annotations) until the required distance is covered on each path.

    python tools/mfma_hazard_audit.py [path/to/libfa_mi355x.so] [--verbose] [--skip REGEX]

Kernels whose mangled name matches --skip are not audited; the default skips the profiling-ablation instantiations
(wrong results on purpose, never dispatched without an explicit fwd_abl / dkdv_abl option).

Exit code 1 when a rule is broken.  tests/test_hazard_audit_cpu.py runs it on every build.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flashattention-pytorch_amd",
                           "flashattention_lab_cuda", "libfa_mi355x.so")

# profiling-ablation instantiations: bwd_{dkdv,dq}_w4_kernel<Tag, CAUSAL, ABL != 0, TPW>, fwd_mfma_kernel<..., ABL != 0, W4>
DEFAULT_SKIP = r"bwd_(dkdv|dq)_w4_kernelINS_\w+ELb[01]ELi[1-9]|fwd_mfma_kernelINS_\w+?ELi\d+ELb[01]ELi\d+ELb[01]ELb[01]ELb[01]ELi\d+ELb[01]ELi[1-9]"

# passes (4 cycles each) of the MFMA shapes this library uses
PASSES = [
    (re.compile(r"v_mfma_scale_f32_32x32x64"), 16),
    (re.compile(r"v_mfma_scale_f32_16x16x128"), 8),
    (re.compile(r"v_mfma_f32_32x32x16_(bf16|f16|fp8|bf8)"), 8),
    (re.compile(r"v_mfma_f32_16x16x32_(bf16|f16|fp8|bf8)"), 4),
    (re.compile(r"v_mfma_f32_32x32x2_f32"), 16),
    (re.compile(r"v_mfma_f32_16x16x4_f32"), 8),
    (re.compile(r"v_mfma_f32_32x32x"), 16),
    (re.compile(r"v_mfma_"), 16),   # anything else: assume the longest
]
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")
LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")
TARGET = re.compile(r"<(\S+?)\+0x([0-9a-fA-F]+)>\s*$")


def passes_of(mn):
    for rx, p in PASSES:
        if rx.match(mn):
            return p
    return 16


def regs_of(text):
    """set of (file, index) named in an operand string"""
    out = set()
    for m in REG.finditer(text):
        f = m.group(1)
        if m.group(4) is not None:
            out.add((f, int(m.group(4))))
        else:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Ins:
    __slots__ = ("mn", "ops", "addr", "regs", "states", "target", "is_mfma", "dst", "srcs", "chain_c")

    def __init__(self, mn, ops, addr, target):
        self.mn, self.ops, self.addr, self.target = mn, ops, addr, target
        self.states = 1
        if mn == "s_nop":
            try:
                self.states = int(ops.split()[0], 0) + 1
            except (ValueError, IndexError):
                pass
        self.is_mfma = mn.startswith("v_mfma")
        parts = [p.strip() for p in ops.split(",")]
        self.regs = regs_of(ops)
        self.dst, self.srcs, self.chain_c = set(), [], False
        if self.is_mfma and len(parts) >= 4:
            self.dst = regs_of(parts[0])
            self.srcs = [regs_of(parts[1]), regs_of(parts[2]), regs_of(parts[3])]
            self.chain_c = parts[0] == parts[3]


def extract_code_objects(lib, tmp):
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f)


def disassemble(co):
    out = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True,
                         text=True).stdout
    return parse_disassembly(out)


def parse_disassembly(out):
    """{function: (start address, [Ins])} from llvm-objdump -d text"""
    funcs, cur, start = {}, None, 0
    for line in out.splitlines():
        m = FUNC.match(line)
        if m:
            cur, start = m.group(2), int(m.group(1), 16)
            funcs[cur] = (start, [])
            continue
        m = LINE.match(line)
        if m and cur is not None:
            tgt = None
            t = TARGET.search(line)
            if t and m.group(1).startswith(("s_cbranch", "s_branch")):
                tgt = funcs[cur][0] + int(t.group(2), 16) if t.group(1) == cur else None
            funcs[cur][1].append(Ins(m.group(1), m.group(2), int(m.group(3), 16), tgt))
    return funcs


def audit_function(name, ins, verbose):
    index = {x.addr: i for i, x in enumerate(ins)}
    viol, n_mfma, min_slack = [], 0, None

    def scan(i0, need, dst, t, pipe_free, seen, need_c=None):
        """walk forward from instruction i0 at time t (wait states since the producer issued; the matrix pipe is busy
        until pipe_free); return (states elapsed, offender) for the first consumer found too early, or None"""
        i = i0
        while i < len(ins) and t < need:
            x = ins[i]
            if x.is_mfma:
                t = max(t, pipe_free)            # one matrix pipe per SIMD, in order: this MFMA waits for the previous one
                if x.chain_c and x.dst == dst:
                    return None                  # accumulation chain: the next MFMA is audited on its own
                if t >= need:
                    return None
                if x.regs & dst:
                    c_only = (need_c is not None and x.mn.endswith("_f32") and len(x.srcs) == 3 and (x.srcs[2] & dst)
                              and not ((x.srcs[0] | x.srcs[1] | x.dst) & dst))
                    if not (c_only and t >= need_c):
                        return (t, x)
                pipe_free = t + passes_of(x.mn)      # (t + 1) + passes - 1
                t += 1
                i += 1
                continue
            if x.regs & dst:
                return (t, x)
            if x.mn == "s_endpgm":
                return None
            if x.target is not None and x.target in index:
                key = (index[x.target], t)
                if key not in seen:
                    seen.add(key)
                    r = scan(index[x.target], need, dst, t + x.states, pipe_free, seen, need_c)
                    if r is not None:
                        return r
                if x.mn == "s_branch":
                    return None
            t += x.states
            i += 1
        return None

    for i, x in enumerate(ins):
        if not x.is_mfma:
            continue
        n_mfma += 1
        # XDL (16 / 8-bit inputs): passes + 3; the f32-input forms are not XDL ops: passes + 2 (LLVM GFX940 SMFMA rule)
        smfma = x.mn.endswith("x2_f32") or x.mn.endswith("x4_f32")
        need = passes_of(x.mn) + (2 if smfma else 3)
        # t = wait states BETWEEN the producer and the instruction looked at (LLVM's count); the pipe holds the
        # producer for `passes` states from its own issue slot
        r = scan(i + 1, need, x.dst, 0, passes_of(x.mn) - 1, set(), passes_of(x.mn) - 2 if smfma else None)
        if r is not None:
            viol.append(f"R1 {name}: {x.mn} {x.ops.split(',')[0]} @0x{x.addr:x}: consumer `{r[1].mn} {r[1].ops}` "
                        f"@0x{r[1].addr:x} after {r[0]} wait states, {need} needed")
        # R2: VALU write of an A / B operand within the previous 2 wait states
        w, j = 0, i - 1
        ab = x.srcs[0] | x.srcs[1] if x.srcs else set()
        while j >= 0 and w < 2:
            y = ins[j]
            if y.mn.startswith("v_") and not y.is_mfma and not y.mn.startswith("v_cmp") and y.ops:
                wr = regs_of(y.ops.split(",")[0])
                if wr & ab:
                    viol.append(f"R2 {name}: {x.mn} @0x{x.addr:x} reads `{y.mn} {y.ops}` @0x{y.addr:x} after {w} wait states, 2 needed")
                    break
            if y.target is not None or y.mn.startswith("s_cbranch") or y.mn == "s_branch":
                break
            w += y.states
            j -= 1
    if verbose and n_mfma:
        print(f"  {name}: {n_mfma} MFMAs, {len(viol)} violations")
    return n_mfma, viol


def audit_text(disassembly, verbose=False):
    """(number of MFMAs, violations) for llvm-objdump -d text (the unit tests feed hand-written snippets)"""
    total, viol = 0, []
    for name, (_, ins) in parse_disassembly(disassembly).items():
        n, v = audit_function(name, ins, verbose)
        total += n
        viol += v
    return total, viol


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=DEFAULT_LIB)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--skip", default=DEFAULT_SKIP)
    args = ap.parse_args()
    skip = re.compile(args.skip) if args.skip else None
    if not os.path.exists(args.lib):
        sys.exit(f"{args.lib} not found: build it first (make -C flashattention-pytorch_amd/csrc)")
    total, kernels, viol, skipped = 0, 0, [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in extract_code_objects(args.lib, tmp):
            for name, (_, ins) in disassemble(co).items():
                if skip is not None and skip.search(name):
                    skipped += 1
                    continue
                n, v = audit_function(name, ins, args.verbose)
                total += n
                kernels += 1 if n else 0
                viol += v
    print(f"mfma_hazard_audit: {total} MFMA instructions in {kernels} kernels, {len(viol)} violations ({skipped} ablation builds skipped)")
    for v in viol[:50]:
        print("  " + v)
    return 1 if viol else 0


if __name__ == "__main__":
    sys.exit(main())
