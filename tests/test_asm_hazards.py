"""CPU (hipcc cross-compiles without a GPU): static audit of the inline-asm MFMAs of the attention kernels for the two hazards
hipcc does not pad around an `asm` statement (scripts/asm_hazard_audit.py; /opt/skills/guides/cdna_hip_programming.md 5.7):
a compiler-scheduled VALU write of an MFMA source right in front of the asm MFMA, and compiler code touching an asm MFMA's result
before it has landed.  Round 2 found both kinds by wrong results on the GPU (head dims 56 / 64); the audit reproduces the
producer-side findings on the sources of that time and must stay at zero."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_attention_forward_has_no_unpadded_asm_mfma_hazard():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asm_hazard_audit.py"),
                        os.path.join(ROOT, "flash_diffusion_amd", "csrc", "attn.hip"), "--kernel", "attn_fwd"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "total findings: 0", r.stdout[-3000:]
    assert sum(1 for l in lines if "attn_fwd_kernel" in l) >= 20   # every instantiation was looked at


def test_audit_flags_a_value_written_right_before_an_asm_mfma():
    """the rule itself, on a hand-written instruction stream"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import asm_hazard_audit as A
    bad = [";;#ASMSTART", "s_nop 4", ";;#ASMEND", "v_mov_b64_e32 v[28:29], s[12:13]", ";;#ASMSTART",
           "v_mfma_f32_16x16x32_bf16 v[28:31], v[60:63], v[44:47], v[28:31]", ";;#ASMEND", "s_nop 15", "s_endpgm"]
    f = A.audit(bad)
    assert [x[0] for x in f] == ["A"]
    ok = list(bad)
    ok.insert(4, "s_nop 1")
    assert A.audit(ok) == []
    late = [";;#ASMSTART", "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]", ";;#ASMEND", "v_mov_b64_e32 v[20:21], v[0:1]"]
    assert [x[0] for x in A.audit(late)] == ["B"]
    late.insert(3, "s_nop 11")
    assert A.audit(late) == []
