"""The 16x16x4 conv kernel waits for its inline-asm global loads itself; the generated ISA must not touch a register that
such a load still has in flight (unet_amd/isa_check.py).  Compiles conv_igemm.hip to assembly (about a minute)."""
import shutil

import pytest

from unet_amd import isa_check
from unet_amd.build import CSRC, HIPCC


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")
@pytest.mark.parametrize("src,min_kernels", [("conv_igemm.hip", 30), ("conv_bf16.hip", 30)])
def test_no_instruction_touches_an_in_flight_register(src, min_kernels):
    asm = isa_check.compile_to_asm(CSRC / src)
    kernels, bad = isa_check.check_asm(asm)
    assert len(kernels) >= min_kernels, f"expected every direct-operand conv instantiation of {src} to carry asm loads, saw {len(kernels)}"
    assert not bad, "\n".join(bad[:20])


def test_checker_flags_a_copy_of_an_in_flight_register():
    asm = """
    .type k,@function
k:
    ;;#ASMSTART
    global_load_dwordx4 v[4:7], v1, s[2:3]
    ;;#ASMEND
    v_mov_b32_e32 v9, v5
    ;;#ASMSTART
    s_waitcnt vmcnt(0)
    ;;#ASMEND
    v_mov_b32_e32 v10, v5
    s_endpgm
"""
    kernels, bad = isa_check.check_asm(asm)
    assert kernels == {"k": 1}
    assert len(bad) == 1 and "v_mov_b32_e32 v9, v5" in bad[0]


def test_checker_follows_branches():
    # the copy sits on the path that did NOT issue the load: not a violation
    asm = """
    .type k,@function
k:
    s_cbranch_scc1 .LBB0_2
    ;;#ASMSTART
    global_load_dwordx4 v[4:7], v1, s[2:3]
    ;;#ASMEND
    s_branch .LBB0_3
.LBB0_2:
    v_mov_b32_e32 v9, v5
.LBB0_3:
    ;;#ASMSTART
    s_waitcnt vmcnt(0)
    ;;#ASMEND
    s_endpgm
"""
    kernels, bad = isa_check.check_asm(asm)
    assert not bad
