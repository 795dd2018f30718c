"""The built library must not contain the instruction form that gfx950 executes wrongly beside bf16 MFMAs.

Root cause of the round-3 "two-context corruption" (profiles/r04_pk_hazard.txt, tools/ubench/pkf32_vs_bf16mfma.hip): a packed-fp32
VALU instruction whose op_sel bit for src1 is set returns a wrong LOW half in lanes 48-63 while another wave on the CU executes
v_mfma_f32_16x16x32_bf16.  The SLP vectoriser forms such instructions (176 of them in stft.hip without -fno-slp-vectorize); the
flag is the first defence, this disassembly check is the second - it also covers hand-written vector code and compiler upgrades."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402


@pytest.mark.skipif(not os.path.exists(isa_audit.OBJDUMP), reason="llvm-objdump not found")
def test_no_packed_fp32_instruction_selects_the_high_half_of_src1(built_lib):
    objs = isa_audit.code_objects(built_lib)
    assert objs and all("gfx950" in t for t, _ in objs), [t for t, _ in objs]
    n_pk, offenders = isa_audit.audit(built_lib)
    assert n_pk > 1000          # the disassembly really was read (swish2's explicit v_pk_mul / v_pk_add pairs alone are thousands)
    assert not offenders, offenders[:10]


def test_build_keeps_the_slp_vectoriser_off():
    import birdnet_go_amd  # noqa: F401
    from birdnet_go_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS


@pytest.mark.skipif(not os.path.exists(isa_audit.OBJDUMP), reason="llvm-objdump not found")
def test_instruction_class_listing_of_a_fused_kernel(built_lib):
    """tools/isa_audit.py --classes (round 6: the audit of the small-K fused kernels): the static listing finds the
    instantiation b2 runs on the recorded plan, and its classes are what the source says the kernel is made of - f32 MFMAs,
    transcendentals in exp / rcp pairs for the swish, LDS traffic for the expanded footprint."""
    got = isa_audit.classes(built_lib, "k_expand_dw_sk<3, 2, 8, 8, 17, false, 16, true, 4, 0, 1>")
    assert len(got) == 1, list(got)
    c = next(iter(got.values()))
    assert c["mfma"] >= 40 and c["valu:transcendental"] >= 2 * c["mfma"] and c["lds"] >= 50 and c["barrier"] >= 2
    assert c["valu:transcendental"] % 2 == 0 or c["valu:transcendental"] > 100       # exp + rcp per element (plus the cold activation switch)
    assert isa_audit.insn_class("v_pk_mul_f32") == "valu:mul_f32" and isa_audit.insn_class("ds_read_b128") == "lds"
    assert isa_audit.insn_class("v_lshl_add_u64") == "valu:integer" and isa_audit.insn_class("s_cbranch_scc1") == "branch"
