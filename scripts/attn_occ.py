"""attention fwd at forced occupancies (dev tool): extra LDS per block limits resident blocks per CU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops
from flash_diffusion_amd._lib import lib
from kbench import timeit
BF = torch.bfloat16
B, S, H, d = 16, 4096, 8, 40
q = torch.randn(B, S, H * d, device="cuda").to(BF)
k = torch.randn(B, S, H * d, device="cuda").to(BF)
v = torch.randn(B, S, H * d, device="cuda").to(BF)
for extra, label in ((0, "as built (regs allow 2 blocks/CU... per SIMD 2 waves)"), (60000, "<=2 blocks/CU by LDS"), (120000, "1 block/CU by LDS")):
    lib().fdmi_tune_set(10, extra)
    us = timeit(lambda: ops.attn_fwd(q, k, v, H, d ** -0.5), 10)
    print(f"extra LDS {extra:6d} ({label}): {us:9.1f} us", flush=True)
lib().fdmi_tune_set(10, 0)
