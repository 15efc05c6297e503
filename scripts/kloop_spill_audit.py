#!/usr/bin/env python
"""No register spill inside a GEMM kernel's K loop (static, no GPU): disassembles the DEVICE code of built objects
(flash_diffusion_amd/csrc/gemm3.o, gemm4.o) and reports every `scratch_*` instruction that lies between a kernel's first and last
MFMA.

Why it matters here: the LDS-DMA ring of these kernels is issued from inline asm and waited for with counted `s_waitcnt vmcnt(n)`;
a spill reload is a load the compiler DOES track -- it waits `vmcnt(0)` for it, i.e. for the whole ring, once per K tile.  The
kernels sit at the 256-VGPR cap, so an innocent edit of an epilogue variant can push a loop-resident value (an LDS offset, a
pointer) into scratch; the kernel stays correct and silently loses its pipelining (round 4: the GroupNorm-sum conv kernel after its
epilogue gained the line-wide variants).  Spills OUTSIDE the MFMA span (epilogues, item set-up) are reported as a count only.

  python scripts/kloop_spill_audit.py [objects...]      exit status 1 when a K loop contains a scratch instruction"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_disassembly(obj):
    """the gfx950 code object bundled in a hipcc-built .o, disassembled"""
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True, check=True)
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        assert dev, f"no device bundle in {obj}"
        return subprocess.run([OBJDUMP, "-d", os.path.join(d, dev[0])], capture_output=True, text=True, check=True).stdout


def audit(text):
    """{kernel: (scratch instructions inside the MFMA span, outside it, MFMA count)}"""
    out, name, ins = {}, None, []

    def close():
        if name and ins:
            mf = [i for i, l in enumerate(ins) if "v_mfma" in l]
            sc = [i for i, l in enumerate(ins) if "scratch_" in l]
            if mf:
                inside = [ins[i].strip() for i in sc if mf[0] <= i <= mf[-1]]
                # gemm4's 2-slot ring has ONE vmcnt wait per K tile, placed by hand in front of the tile's first MFMA: a vmcnt wait between
                # the first and the last MFMA is hipcc waiting for one of ITS loads (a spill reloaded at the item's setup, say) -- and with it
                # for the LDS-DMA pieces issued since the tile's head (round 6: the 256 x 384 kernel lost 6 % of its family that way)
                # (a wait in front of the tile's FIRST piece is harmless: nothing has been issued since the head's own vmcnt(0))
                if "gemm4_kernel" in name:
                    dma = [i for i, l in enumerate(ins) if "global_load_lds" in l and mf[0] < i < mf[-1]]
                    if dma:
                        inside += [ins[i].strip() for i, l in enumerate(ins) if "s_waitcnt" in l and "vmcnt" in l and dma[0] < i < mf[-1]]
                out[name] = (inside, len(sc) - len([x for x in inside if "scratch_" in x]), len(mf))
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            close()
            name, ins = m.group(1), []
        elif name and line.strip():
            ins.append(line)
    close()
    return out


def main():
    objs = sys.argv[1:] or [os.path.join(ROOT, "flash_diffusion_amd", "csrc", f) for f in ("gemm3.o", "gemm4.o", "gemm5.o")]
    bad = 0
    for o in objs:
        res = audit(device_disassembly(o))
        for k, (inside, outside, nmf) in sorted(res.items()):
            dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
            print(f"{os.path.basename(o)}: {dn[:70]:70s} {nmf:4d} MFMAs, scratch / stray vmcnt waits inside the K loop: {len(inside)}, elsewhere: {outside}")
            for l in inside:
                print("      " + l)
            bad += len(inside)
    print(f"total in-loop spills: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
