"""The stream kernels' placement tables (w4sched::kSched in csrc/fa_bwd_dkdv_w4.hip, dqsched::kSched in csrc/fa_bwd_dq_w4.hip) are
GENERATED (tools/gen_dkdv_schedule.py, tools/gen_dq_schedule.py).  These tests keep source and generator in step and check the
properties the kernels rely on: every operation placed exactly once, inside its window, and no gap over its issue budget."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flashattention-pytorch_amd", "csrc")


def _generated(tool):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], check=True, capture_output=True, text=True).stdout
    return out, [l.strip() for l in out.splitlines() if l.strip().startswith("/*")]


def _embedded(src, namespace):
    text = open(os.path.join(CSRC, src)).read()
    body = text[text.index("namespace " + namespace):]
    body = body[body.index("kSched["):]
    body = body[:body.index("};")]
    return [l.strip() for l in body.splitlines() if l.strip().startswith("/*")]


def _ops(rows):
    return [re.findall(r"[A-Z]+(?:\([0-9, ]*\))?", r.split("*/", 1)[1]) for r in rows]


def test_dkdv_table_is_the_generators_output_and_within_budget():
    out, rows = _generated("gen_dkdv_schedule.py")
    assert rows == _embedded("fa_bwd_dkdv_w4.hip", "w4sched"), "regenerate w4sched::kSched with tools/gen_dkdv_schedule.py"
    assert "0 cycles over budget in 0 gaps" in out
    ops = [o for row in _ops(rows) for o in row]
    assert len(rows) == 64 and len(ops) == len(set(ops))
    # 32 elements of S' -> P (MUL + EXP each), 16 packed dwords of P and of dS, 4 accumulator loads, 5 DMA pieces, 13 address updates
    count = lambda p: sum(1 for o in ops if o.startswith(p))
    assert (count("MUL("), count("EXP("), count("PC("), count("SU("), count("ACC("), count("DMA("), count("QADDR("), count("TADDR(")) == (32, 32, 16, 16, 4, 5, 8, 4)
    where = {o: s for s, row in enumerate(_ops(rows)) for o in row}
    for kb in (0, 1):
        for m in range(8):
            # P[kb][pairs 0..3] feeds MFMA 32 + kb, pairs 4..7 MFMA 40 + kb; dS: 48 + kb / 56 + kb; a whole gap in between
            assert where[f"PC({kb},{m})"] <= (30 if m < 4 else 38) + kb
            assert 32 + kb <= where[f"SU({kb},{m})"] <= (46 if m < 4 else 54) + kb
            for e in (2 * m, 2 * m + 1):
                assert 16 + kb <= where[f"MUL({kb},{e})"] <= where[f"EXP({kb},{e})"] < where[f"PC({kb},{m})"]
                assert where[f"EXP({kb},{e})"] < where[f"SU({kb},{m})"]
        assert max(where[f"SU({kb},{m})"] for m in range(8)) < min(where[f"ACC({kb})"], where[f"ACC({2 + kb})"])
    assert where["LADDR"] < min(where[f"ACC({i})"] for i in range(4))


def test_dkdv_ds_variant_adds_only_its_four_stores():
    """The dK/dV kernel's DS variant (dS handed to fa_bwd_dq_ds.hip) issues its four stores in slices 52 .. 55 by hand: that is
    where the generator puts them, moving nothing else and staying within every gap's budget."""
    plain, rows = _generated("gen_dkdv_schedule.py")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_dkdv_schedule.py"), "--ds"], check=True, capture_output=True, text=True).stdout
    ds_rows = [l.strip() for l in out.splitlines() if l.strip().startswith("/*")]
    assert "0 cycles over budget in 0 gaps" in out
    stores = {52: "DSST(0,0)", 53: "DSST(0,1)", 54: "DSST(1,0)", 55: "DSST(1,1)"}
    for s_, (a, b) in enumerate(zip(_ops(rows), _ops(ds_rows))):
        assert b == a + ([stores[s_]] if s_ in stores else []), (s_, a, b)
    src = open(os.path.join(CSRC, "fa_bwd_dkdv_w4.hip")).read()
    assert "if constexpr (DS && !(ABL & 64) && S >= 52 && S < 56) DSST(integral_constant<int, ((S - 52) / 2)>{}, integral_constant<int, ((S - 52) % 2)>{});" in src


def test_dq_table_is_the_generators_output_and_within_budget():
    out, rows = _generated("gen_dq_schedule.py")
    assert rows == _embedded("fa_bwd_dq_w4.hip", "dqsched"), "regenerate dqsched::kSched with tools/gen_dq_schedule.py"
    assert "0 cycles over budget in 0 gaps, 0 gaps over 5 instructions" in out
    ops = [o for row in _ops(rows) for o in row]
    assert len(rows) == 48 and len(ops) == len(set(ops))
    count = lambda p: sum(1 for o in ops if o.startswith(p))
    assert (count("FMA("), count("EXP("), count("EXPL("), count("SM("), count("SC("), count("DMA("), count("KADDR("), count("TADDR(")) == (32, 28, 4, 32, 16, 4, 8, 4)
    where = {o: s for s, row in enumerate(_ops(rows)) for o in row}
    for qb in (0, 1):
        for m in range(8):
            assert where[f"SC({qb},{m})"] <= (30 if m < 4 else 38) + qb       # dS^T[qb][s] feeds MFMA 32 + 8 s + qb
            for e in (2 * m, 2 * m + 1):
                assert 16 + qb <= where[f"SM({qb},{e})"] <= where[f"SC({qb},{m})"]
    for e in range(16):
        assert where[f"FMA(0,{e})"] <= where[f"EXP(0,{e})"] < where[f"SM(1,{e})"]   # P^T(b), rows 32 .. 63
        assert where[f"FMA(1,{e})"] >= 32                                          # S^T(b+1)[0] is complete after MFMA 30
        late = f"EXPL({e})" in where
        assert late == (e >= 12) and (where[f"EXPL({e})"] < 16 if late else where[f"EXP(1,{e})"] >= where[f"FMA(1,{e})"])
    for i in range(8):
        assert max(0, 2 * (i - 3)) <= where[f"KADDR({i})"] <= 9 + 2 * i
