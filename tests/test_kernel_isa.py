"""CPU: static check of the built prefill kernels (tools/check_prefill_isa.py).

The prefill kernel issues LDS fragment reads early and waits for them later (csrc/mfma_attn_common.h, lds_issue_* /
lds_wait_all); inline asm pins the order of those statements, not register allocation, so the property "no instruction
between a read and its wait touches the read's destination registers" is checked on the compiler's output.
"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_prefill_isa as chk  # noqa: E402


def test_checker_sees_a_use_before_the_wait():
    asm = """
_ZN2mi_prefill_mfma_kernel_fake:
	s_barrier
	ds_read_b128 v[10:13], v2 offset:16
	v_add_f32_e32 v1, v3, v4
	s_cbranch_scc1 .LBB0_2
	v_mov_b32_e32 v20, v11
.LBB0_2:
	s_waitcnt lgkmcnt(0)
	v_mov_b32_e32 v21, v12
	s_endpgm
.Lfunc_end0:
"""
    kernels, problems = chk.check(asm)
    assert kernels == 1
    assert [p[2] for p in problems] == ["v_mov_b32_e32 v20, v11"]


def test_checker_follows_branches():
    asm = """
_ZN2mi_prefill_mfma_kernel_fake2:
	ds_read_b64_tr_b16 v[30:31], v2 offset:0
	s_branch .LBB0_9
.LBB0_3:
	v_mov_b32_e32 v5, v30
	s_endpgm
.LBB0_9:
	v_mul_f32_e32 v6, v31, v31
	s_waitcnt lgkmcnt(0)
	s_branch .LBB0_3
.Lfunc_end1:
"""
    kernels, problems = chk.check(asm)
    assert kernels == 1
    assert [p[2] for p in problems] == ["v_mul_f32_e32 v6, v31, v31"]


@pytest.mark.skipif(shutil.which(chk.HIPCC) is None, reason="hipcc not installed")
def test_prefill_kernels_do_not_touch_lds_destinations_before_their_wait():
    assert chk.main() == 0
