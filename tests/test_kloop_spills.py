"""CPU: no register spill inside the K loop of the LDS-DMA GEMM kernels (scripts/kloop_spill_audit.py on the built objects).  A
spill reload there makes hipcc wait vmcnt(0) -- for the whole LDS-DMA ring -- once per K tile: correct results, lost pipelining."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.parametrize("obj", ["gemm3.o", "gemm4.o", "gemm5.o"])
def test_no_scratch_instruction_inside_a_k_loop(obj):
    import kloop_spill_audit as A
    path = os.path.join(ROOT, "flash_diffusion_amd", "csrc", obj)
    if not os.path.exists(path) or not os.path.exists(A.OBJDUMP):
        pytest.skip("needs the built object and llvm-objdump (python __graft_entry__.py)")
    res = A.audit(A.device_disassembly(path))
    assert len(res) >= (2 if obj == "gemm5.o" else 7)              # every instantiation was looked at
    bad = {k: v[0] for k, v in res.items() if v[0]}
    assert not bad, bad


def test_audit_sees_a_spill_between_mfmas():
    import kloop_spill_audit as A
    text = "\n".join(["0000000000001000 <k1>:", "  v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]", "  scratch_load_dword v5, off, off offset:4",
                      "  s_waitcnt vmcnt(0)", "  v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]", "  scratch_store_dword off, v9, off",
                      "0000000000002000 <k2>:", "  scratch_load_dword v5, off, off", "  v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]"])
    r = A.audit(text)
    assert len(r["k1"][0]) == 1 and r["k1"][1] == 1 and r["k1"][2] == 2 and r["k2"][0] == [] and r["k2"][1] == 1


def test_audit_sees_a_stray_vmcnt_wait_behind_the_first_piece_of_a_gemm4_tile():
    """round 6: a value reloaded at an item's setup made hipcc wait vmcnt(0) in front of the first W piece of EVERY K tile of the 256 x 384
    kernel (its pieces issued since the tile's head included); a wait in front of the tile's first piece is harmless and not counted"""
    import kloop_spill_audit as A
    mf = "  v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]"
    text = "\n".join(["0000000000001000 <gemm4_kernel_x>:", mf, "  s_waitcnt vmcnt(0)", "  global_load_lds_dwordx4 v1, s[2:3]", mf,
                      "  s_waitcnt vmcnt(0)", "  global_load_lds_dwordx4 v2, s[4:5]", mf,
                      "0000000000002000 <gemm3_kernel_x>:", mf, "  global_load_lds_dwordx4 v1, s[2:3]", "  s_waitcnt vmcnt(2)", mf])
    r = A.audit(text)
    assert len(r["gemm4_kernel_x"][0]) == 1 and r["gemm3_kernel_x"][0] == []
