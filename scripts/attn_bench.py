import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops
from kbench import timeit
from flash_diffusion_amd import _lib
L = _lib.lib()
BF = torch.bfloat16
for (B, S, Skv, H, d) in [(16, 4096, 4096, 8, 40), (32, 4096, 4096, 8, 40), (16, 4096, 77, 8, 40), (16, 1024, 1024, 8, 80), (16, 256, 256, 8, 160),
                          (8, 4096, 4096, 10, 64), (16, 1024, 1024, 20, 64), (8, 4096, 4096, 16, 72), (16, 4096, 4096, 16, 72), (8, 4096, 120, 16, 72),
                          (32, 1024, 1024, 8, 80), (8, 4429, 4429, 24, 64)]:
    q = torch.randn(B, S, H * d, device="cuda").to(BF)
    k = torch.randn(B, Skv, H * d, device="cuda").to(BF)
    v = torch.randn(B, Skv, H * d, device="cuda").to(BF)
    res = []
    # default: 32x32x16 forward for d < 64, two query fragments for 64 < d <= 96; the switches restore round 2's choices
    variants = (("default", ()), ("16x16x32 (26=1)", ((26, 1),)), ("16x16x32 QF=1 (26=1,27=1)", ((26, 1), (27, 1))), ("default again", ()))
    for name, kn in variants:
        for k_, v_ in kn:
            L.fdmi_tune_set(k_, v_)
        timeit(lambda: ops.attn_fwd(q, k, v, H, d ** -0.5), 5)
        us = timeit(lambda: ops.attn_fwd(q, k, v, H, d ** -0.5), 10)
        for k_, v_ in kn:
            L.fdmi_tune_set(k_, 0)
        res.append(f"{name} {us:8.1f} us {4.0 * B * H * S * Skv * d / us / 1e6:6.1f} TF/s")
    print(f"attn fwd B={B} S={S} Skv={Skv} d={d}: " + " | ".join(res), flush=True)

# ---- backward (dq + dkv kernels + the delta / transpose helpers ops.attn_bwd launches) ----
for (B, S, Skv, H, d) in [(16, 4096, 4096, 8, 40), (8, 4096, 4096, 16, 72), (8, 4096, 120, 16, 72), (16, 1024, 1024, 8, 80), (4, 4429, 4429, 24, 64)]:
    q, k, v, do = (torch.randn(B, n, H * d, device="cuda").to(BF) for n in (S, Skv, Skv, S))
    o, lse = ops.attn_fwd(q, k, v, H, d ** -0.5, need_lse=True)
    res = []
    for name, kn in (("default", ()), ("16x16x32 dq (35=1)", ((35, 1),)), ("16x16x32 dq + dkv (35=1,36=1)", ((35, 1), (36, 1))), ("default again", ())):
        for k_, v_ in kn:
            L.fdmi_tune_set(k_, v_)
        timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d ** -0.5), 3)
        us = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d ** -0.5), 6)
        for k_, v_ in kn:
            L.fdmi_tune_set(k_, 0)
        res.append(f"{name} {us:8.1f} us {10.0 * B * H * S * Skv * d / us / 1e6:6.1f} TF/s")
    print(f"attn bwd B={B} S={S} Skv={Skv} d={d}: " + " | ".join(res), flush=True)
