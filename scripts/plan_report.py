#!/usr/bin/env python
"""Work list of a UNet forward (+ backward) as the planner sees it, without a GPU: every GEMM / conv of a workspace-query walk with
the kernel, tile and split-K the launcher would pick (FDMI_PLAN_LOG=1), grouped by problem, with algorithmic GFLOP and share.

  python scripts/plan_report.py [sd15|sdxl|pixart|sd3] [B] [hw] [--save] [--lora r]
  (defaults: sd15 32 64 = one teacher CFG forward of C2; pixart / sd3: the transformer plans, --lora r declares rank-r adapters on
  the examples' target modules, --save walks the taped forward and its backward)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import sys; sys.path.insert(0, %r)
import torch
from flash_diffusion_amd import _lib
from flash_diffusion_amd.unet import MiUNet2DConditionModel
from flash_diffusion_amd import workloads
arch, B, hw, flags, lora = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
L = _lib.lib()
if arch in ("pixart", "sd3"):      # the transformer denoisers' plans (csrc/dit_plan.h); --lora r declares the adapters (no GPU needed)
    from flash_diffusion_amd import dit
    with torch.device("meta"):
        m = (dit.MiTransformer2DModel if arch == "pixart" else dit.MiSD3Transformer2DModel)(**getattr(workloads, arch.upper()))
        if lora:
            m.add_adapter(lora)
    p = m._plan()
    for n, mod in m._lora_modules():
        assert L.fdmi_unet_declare_lora(p.handle, n.encode(), mod.rank) == 0, L.fdmi_last_error()
    assert L.fdmi_dit_workspace_bytes(p.handle, B, hw, hw, 120 if arch == "pixart" else 333, 0, flags & 1) > 0
else:
    with torch.device("meta"):
        m = MiUNet2DConditionModel(**getattr(workloads, arch.upper()))
    assert L.fdmi_unet_workspace_bytes(m._plan().handle, B, hw, hw, 77, flags) > 0
"""


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--lora"]
    arch = args[0] if args else "sd15"
    B = int(args[1]) if len(args) > 1 else 32
    hw = int(args[2]) if len(args) > 2 else 64
    flags = 1 if "--save" in sys.argv else 8
    env = dict(os.environ, FDMI_PLAN_LOG="1")
    lora = int(sys.argv[sys.argv.index("--lora") + 1]) if "--lora" in sys.argv else 0
    err = subprocess.run([sys.executable, "-c", CHILD % ROOT, arch, str(B), str(hw), str(flags), str(lora)], env=env,
                         capture_output=True, text=True, check=True).stderr
    rows = collections.Counter()
    for l in err.splitlines():
        if l.startswith("PLANGEMM"):
            rows[tuple(int(v) for v in re.findall(r"=(-?\d+)", l))] += 1
    tot = sum(2.0 * k[1] * k[2] * k[3] * n for k, n in rows.items())
    names = {0: "tile128", 1: "gemm3", 2: "gemm4"}
    print(f"{arch} B={B} {hw}x{hw} flags={flags}: {sum(rows.values())} GEMM/conv launches, {tot / 1e12:.2f} TFLOP")
    print("count  mode  M       N      K      act res dgrad  kernel        splitk   GFLOP each   share")
    by_kernel = collections.Counter()
    for k, n in sorted(rows.items(), key=lambda kv: -2.0 * kv[0][1] * kv[0][2] * kv[0][3] * kv[1]):
        mode, M, N, K, act, res, dgrad, atomic, kern, BM, BN, sk = k
        fl = 2.0 * M * N * K
        by_kernel[(names[kern], BN, sk > 1)] += fl * n
        print(f"{n:5d}  {'conv' if mode else 'row ':4s}  {M:6d}  {N:5d}  {K:5d}  {act:3d} {res:3d} {dgrad:5d}  {names[kern]:7s} {BM:3d}x{BN:<3d}  {sk:5d}  {fl / 1e9:11.2f}  {fl * n / tot:6.1%}")
    print("\nshare of the FLOPs by kernel (tile, split-K):")
    for (kn, bn, split), fl in sorted(by_kernel.items(), key=lambda kv: -kv[1]):
        print(f"  {kn:7s} BN={bn:<3d} {'split-K' if split else '       '}  {fl / tot:6.1%}")


if __name__ == "__main__":
    main()
