#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel of a HIP source as hipcc reports it (no GPU needed; dev tool):
  python scripts/kernel_resources.py flash_diffusion_amd/csrc/gemm4.hip [filter]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash_diffusion_amd", "csrc")


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                          "-Rpass-analysis=kernel-resource-usage", "-I", CSRC, "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur, rows = None, {}
    for l in out.splitlines():
        if " error" in l:
            print(l)
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", l)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for n, r in rows.items():
        dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        if flt and flt not in dn:
            continue
        print(f"{dn[:80]:80s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} SGPR {r.get('TotalSGPRs'):>3s} scratch "
              f"{r.get('ScratchSize [bytes/lane]'):>4s} occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}")


if __name__ == "__main__":
    main()
