#!/usr/bin/env python
"""HBM-bound kernel families of the C2 step priced against the HBM roofline ("report their GB/s separately").

Algorithmic bytes per step come from the plan's own counters in workspace-query mode (no GPU needed:
fdmi_unet_last_hbm_bytes) for the step's composition -- one student forward + backward at B=16 and four teacher CFG forwards at
2B=32; the time per family is read from a committed rocprofv3 summary (profiles/rN_kernel_stats*.csv).  The LoRA wgrad
transposes are not in the byte count (LoRA tensors need device memory to be registered), so `transpose2d` is left out.

  python scripts/hbm_table.py [profiles/r1_kernel_stats_final.csv]"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 8.0e12   # B/s, /opt/skills/guides/MI355X_MICROARCH.md
FAMILIES = ["gn_reduce", "gn_apply", "layernorm", "transpose2d", "transpose_heads", "copy2d", "geglu_bwd", "pool2x2", "splitk_finalize"]
KERNELS = {"gn_reduce": ["gn_reduce_kernel"], "gn_apply": ["gn_apply_kernel"], "layernorm": ["ln_kernel"],
           "transpose_heads": ["transpose_heads_kernel"], "copy2d": ["copy2d_kernel"], "geglu_bwd": ["geglu_bwd_kernel"],
           "splitk_finalize": ["gemm_finalize_kernel"]}


def step_bytes(teacher_steps=4, B=16, hw=64, L=77):
    import torch
    from flash_diffusion_amd import _lib
    from flash_diffusion_amd.unet import FDMI_UNET_CTX_FILL, FDMI_UNET_CTX_REUSE, FDMI_UNET_SAVE, MiUNet2DConditionModel
    from flash_diffusion_amd.workloads import SD15
    lib = _lib.lib()
    with torch.device("meta"):
        m = MiUNet2DConditionModel(**SD15)
    plan = m._plan()
    tot = [0.0] * len(FAMILIES)

    def add(Bn, flags, times=1):
        assert lib.fdmi_unet_workspace_bytes(plan.handle, Bn, hw, hw, L, flags) > 0
        for i in range(len(FAMILIES)):
            tot[i] += times * lib.fdmi_unet_last_hbm_bytes(plan.handle, i)

    add(B, FDMI_UNET_SAVE)                              # student forward + backward
    add(2 * B, FDMI_UNET_CTX_FILL)                      # teacher CFG step 0
    add(2 * B, FDMI_UNET_CTX_REUSE, teacher_steps - 1)  # (the dry walk counts the cached K/V head transposes as if recomputed)
    return dict(zip(FAMILIES, tot))


def kernel_ms(path):
    ms = {}
    with open(path) as fh:
        rows = [r for r in csv.reader(l for l in fh if not l.startswith("#"))]
    for r in rows[1:]:
        for fam, keys in KERNELS.items():
            if any(k in r[0] for k in keys):
                ms[fam] = ms.get(fam, 0.0) + float(r[2])
    return ms


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r1_kernel_stats_final.csv")
    b, ms = step_bytes(), kernel_ms(path)
    print(f"| family | algorithmic GB/step | ms/step ({os.path.basename(path)}) | TB/s | of {PEAK / 1e12:.0f} TB/s |")
    print("|---|---|---|---|---|")
    for fam in FAMILIES:
        if fam not in ms or b[fam] == 0:
            continue
        rate = b[fam] / (ms[fam] * 1e-3)
        print(f"| {fam} | {b[fam] / 1e9:.1f} | {ms[fam]:.2f} | {rate / 1e12:.2f} | {rate / PEAK:.0%} |")
    tb, tm = sum(b[f] for f in ms if b[f]), sum(ms[f] for f in ms if b[f])
    print(f"| all of the above | {tb / 1e9:.1f} | {tm:.2f} | {tb / tm / 1e9:.2f} | {tb / (tm * 1e-3) / PEAK:.0%} |")


if __name__ == "__main__":
    main()
