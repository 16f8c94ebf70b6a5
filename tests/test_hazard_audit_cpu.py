"""CPU: tools/mfma_hazard_audit.py — the static check that every MFMA in the built gfx950 code (the inline-asm ones
above all: hipcc pads nothing around asm) keeps the wait states the hardware needs before its result is touched.
The rule set is exercised on hand-written disassembly first (a checker that cannot fail proves nothing), then the
real library must come out clean.  A future edit that breaks the padding fails here, in the container."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mfma_hazard_audit as audit  # noqa: E402

HEAD = "0000000000001000 <kern>:\n"


def _ins(lines):
    return HEAD + "".join(f"\t{t:<60}// {0x1000 + 8 * i:012X}: 00000000\n" for i, t in enumerate(lines))


MFMA = "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]"


def test_result_read_too_early_is_flagged():
    n, v = audit.audit_text(_ins([MFMA, "v_mul_f32_e32 v40, s0, v3", "s_endpgm"]))
    assert n == 1 and len(v) == 1 and v[0].startswith("R1") and "v_mul_f32" in v[0]
    n, v = audit.audit_text(_ins([MFMA, "s_nop 9", "v_mul_f32_e32 v40, s0, v3", "s_endpgm"]))
    assert len(v) == 1        # 10 states, 11 needed (8 passes + 3)
    n, v = audit.audit_text(_ins([MFMA, "s_nop 10", "v_mul_f32_e32 v40, s0, v3", "s_endpgm"]))
    assert v == []
    # an overwrite (WAW) counts like a read, and a store of the tile too
    assert len(audit.audit_text(_ins([MFMA, "v_mov_b32_e32 v7, 0", "s_endpgm"]))[1]) == 1
    assert len(audit.audit_text(_ins([MFMA, "global_store_dwordx4 v[30:31], v[4:7], off", "s_endpgm"]))[1]) == 1


def test_accumulation_chain_and_unrelated_work_are_fine():
    other = "v_mfma_f32_32x32x16_bf16 v[40:55], v[16:19], v[20:23], v[40:55]"
    n, v = audit.audit_text(_ins([MFMA, MFMA, other, MFMA, "s_nop 10", "v_exp_f32_e32 v0, v0", "s_nop 15", "s_endpgm"]))
    assert n == 4 and v == []
    # eleven independent instructions are eleven wait states
    n, v = audit.audit_text(_ins([MFMA] + ["v_add_f32_e32 v60, v61, v62"] * 11 + ["v_exp_f32_e32 v0, v0", "s_endpgm"]))
    assert v == []
    n, v = audit.audit_text(_ins([MFMA] + ["v_add_f32_e32 v60, v61, v62"] * 10 + ["v_exp_f32_e32 v0, v0", "s_endpgm"]))
    assert len(v) == 1
    # the matrix pipe is in order: ONE independent MFMA between a chain and its reader covers 8 + 1 states (not enough),
    # two cover it
    assert len(audit.audit_text(_ins([MFMA, other, "v_exp_f32_e32 v0, v0", "s_endpgm"]))[1]) == 1
    other2 = "v_mfma_f32_32x32x16_bf16 v[60:75], v[16:19], v[20:23], v[60:75]"
    assert audit.audit_text(_ins([MFMA, other, other2, "v_exp_f32_e32 v0, v0", "s_endpgm"]))[1] == []
    # a partial overlap as C operand is not a chain
    bad = "v_mfma_f32_32x32x16_bf16 v[8:23], v[30:33], v[34:37], v[8:23]"
    assert len(audit.audit_text(_ins([MFMA, bad, "s_endpgm"]))[1]) >= 1


def test_f32_forms_may_take_a_result_as_c_operand_after_passes_minus_two():
    """LLVM's SMFMA -> overlapped SrcC rule (hipcc's own code uses it): 8 passes -> 6 states; as A / B operand or for any
    other reader the full distance (8 + 2) holds; the 16-bit (XDL) forms have no such relaxation here."""
    prod = "v_mfma_f32_16x16x4_f32 a[8:11], v9, v120, a[8:11]"
    as_c = "v_mfma_f32_16x16x4_f32 a[28:31], v9, v120, a[8:11]"
    as_b = "v_mfma_f32_16x16x4_f32 a[28:31], v9, a8, a[28:31]"
    assert audit.audit_text(_ins([prod, "s_nop 5", as_c, "s_nop 15", "s_endpgm"]))[1] == []
    # (closer than that it cannot get: the matrix pipe is in order, the consumer issues 7 states behind the producer at the earliest)
    assert audit.audit_text(_ins([prod, as_c, "s_nop 15", "s_endpgm"]))[1] == []
    assert len(audit.audit_text(_ins([prod, "s_nop 5", as_b, "s_nop 15", "s_endpgm"]))[1]) == 1
    assert len(audit.audit_text(_ins([prod, "s_nop 5", as_c, "v_accvgpr_read_b32 v1, a8", "s_nop 15", "s_endpgm"]))[1]) == 1   # still a reader too early
    xdl_c = "v_mfma_f32_32x32x16_bf16 v[40:55], v[16:19], v[20:23], v[0:15]"
    assert len(audit.audit_text(_ins([MFMA, "s_nop 5", xdl_c, "s_nop 15", "s_endpgm"]))[1]) >= 1


def test_branch_targets_are_followed():
    # the consumer sits at the branch target, 2 states after the MFMA
    text = HEAD
    lines = [MFMA, "s_cbranch_scc1 2", "s_nop 15", "s_endpgm", "v_exp_f32_e32 v1, v1", "s_endpgm"]
    for i, t in enumerate(lines):
        ann = " <kern+0x20>" if t.startswith("s_cbranch") else ""
        text += f"\t{t:<60}// {0x1000 + 8 * i:012X}: 00000000{ann}\n"
    n, v = audit.audit_text(text)
    assert n == 1 and len(v) == 1 and "v_exp_f32" in v[0]


def test_valu_written_operand_needs_two_states():
    m = "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]"
    assert any(x.startswith("R2") for x in audit.audit_text(_ins(["v_cvt_pk_bf16_f32 v16, v40, v41", m, "s_nop 15", "s_endpgm"]))[1])
    assert audit.audit_text(_ins(["v_cvt_pk_bf16_f32 v16, v40, v41", "s_nop 1", m, "s_nop 15", "s_endpgm"]))[1] == []


def test_the_built_library_is_clean():
    lib = audit.DEFAULT_LIB
    assert os.path.exists(lib), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_audit.py"), lib], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    n = int(r.stdout.split("mfma_hazard_audit:")[1].split()[0])
    assert n > 5000   # the check really saw the kernels (the stream kernels alone hold hundreds of asm MFMAs)
