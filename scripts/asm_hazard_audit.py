#!/usr/bin/env python
"""Static audit of the inline-asm MFMAs of a HIP source for the two hazards hipcc does not pad around an `asm` statement
(/opt/skills/guides/cdna_hip_programming.md section 5.7, item 2) -- no GPU needed:

  (A) producer -> MFMA source: a compiler-scheduled instruction that WRITES a VGPR which an inline-asm MFMA reads as A, B or C
      must be followed by wait states before that MFMA (a just-written "v" operand: >= 2 states; for an LDS / global load the
      s_waitcnt the compiler places is what counts, so only VALU writers are checked);
  (B) MFMA result -> compiler code: an instruction outside asm statements that READS or WRITES the destination of an inline-asm
      MFMA needs the MFMA to have landed (8-pass XDL: >= 12 states after the LAST asm MFMA that wrote it; the next MFMA taking it
      whole as C is exempt).

  python scripts/asm_hazard_audit.py flash_diffusion_amd/csrc/attn.hip [--kernel attn_fwd] [--verbose]

The source is compiled to device assembly (`hipcc -S --cuda-device-only`), every kernel is split into basic blocks, and the two
rules are checked along straight-line code (a block and, for (B), its fall-through / branch successors up to the state budget).
Wait states: `s_nop N` = N + 1, any other instruction = 1 (a lower bound: issue takes at least one cycle).  A finding is a
place to look at, not a proof of a wrong result -- and no finding is not a proof of safety either (rule (A) only sees VALU
writers, loops are followed once)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEED_A = 2     # VALU write -> MFMA source read
NEED_B = 12    # MFMA (8 passes) result -> any other reader / writer


def regs_of(tok):
    """'v[4:7]' -> {4,5,6,7}; 'v12' -> {12}; anything else -> empty"""
    m = re.fullmatch(r"-?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    return {int(m.group(1))} if m else set()


def parse_operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])]
    return parts[0], [o.split()[0] if o else o for o in ops]


def writes_reads(ins):
    """(written VGPRs, read VGPRs) of one instruction -- first operand is the destination for v_* / ds_read / *_load"""
    op, ops = parse_operands(ins)
    if not ops:
        return set(), set()
    if op.startswith(("v_cmp", "v_cmpx")):
        return set(), set().union(*[regs_of(o) for o in ops])
    if op.startswith(("ds_write", "ds_store", "global_store", "buffer_store", "scratch_store", "global_atomic", "ds_add", "ds_bpermute")) \
            and not op.startswith("ds_bpermute"):
        return set(), set().union(*[regs_of(o) for o in ops])
    if op.startswith(("v_", "ds_read", "ds_load", "ds_bpermute", "global_load", "buffer_load", "scratch_load")):
        w = regs_of(ops[0])
        r = set().union(*[regs_of(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")) or (op.startswith("v_mfma") and len(ops) == 4):
            r |= regs_of(ops[-1]) if op.startswith("v_mfma") else w
        return w, r
    return set(), set().union(*[regs_of(o) for o in ops])


def states(ins):
    m = re.fullmatch(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def kernels(asm_text):
    cur, body = None, []
    for l in asm_text.split("\n"):
        m = re.match(r"^(_Z\S+):\s*(;.*)?$", l)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                yield cur, body
                cur, body = None, []
            else:
                body.append(l.strip())
    if cur:
        yield cur, body


def audit(body):
    """-> list of (rule, index, text)"""
    items, inasm = [], False
    for l in body:
        if "ASMSTART" in l:
            inasm = True
            continue
        if "ASMEND" in l:
            inasm = False
            continue
        if not l or l.startswith(";") or l.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", l):
                items.append((False, l, True))
            continue
        items.append((inasm, l.split(";")[0].strip(), False))
    findings = []
    n = len(items)
    for i, (a, ins, is_label) in enumerate(items):
        if not (a and ins.startswith("v_mfma")):
            continue
        op, ops = parse_operands(ins)
        dst = regs_of(ops[0])
        src = set().union(*[regs_of(o) for o in ops[1:]])
        # (A) look back inside the straight-line code in front of the MFMA
        budget, k = 0, i - 1
        while k >= 0 and budget < NEED_A:
            ka, kins, klabel = items[k]
            if klabel or kins.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            if not ka and kins.startswith("v_") and not kins.startswith("v_mfma"):
                w, _ = writes_reads(kins)
                if w & src:
                    findings.append(("A", i, f"{kins}   -> {budget} state(s) ->   {ins}"))
                    break
            budget += states(kins)
            k -= 1
        # (B) look forward: compiler instructions touching the destination before it has landed
        budget, k = 0, i + 1
        while k < n and budget < NEED_B:
            ka, kins, klabel = items[k]
            if klabel:
                k += 1
                continue
            if kins.startswith(("s_endpgm", "s_setpc")):
                break
            if ka and kins.startswith("v_mfma"):
                o2, ops2 = parse_operands(kins)
                if regs_of(ops2[0]) & dst:   # the accumulate chain (or a later MFMA re-defining it): that one is audited itself
                    break
            elif not ka or not kins.startswith("v_mfma"):
                w, r = writes_reads(kins)
                if (w | r) & dst and not kins.startswith(("s_nop", "s_waitcnt")):
                    findings.append(("B", i, f"{ins}   -> {budget} state(s) ->   {kins}"))
                    break
            budget += states(kins)
            k += 1
    return findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", default="", help="substring of the (mangled) kernel names to audit")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                        os.path.abspath(a.source)], check=True, stderr=subprocess.DEVNULL,
                       cwd=os.path.dirname(os.path.abspath(a.source)))
        text = open(out).read()
    total = 0
    for name, body in kernels(text):
        if a.kernel and a.kernel not in name:
            continue
        nasm = sum(1 for l in body if "ASMSTART" in l)
        if not nasm:
            continue
        f = audit(body)
        total += len(f)
        print(f"{name}: {len(f)} finding(s)  [{sum(1 for x in f if x[0] == 'A')} A, {sum(1 for x in f if x[0] == 'B')} B]")
        if a.verbose or f:
            for rule, idx, txt in f[:12]:
                print(f"    ({rule}) @{idx}: {txt}")
    print(f"total findings: {total}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
