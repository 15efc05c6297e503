"""bit-reproducibility of attention fwd (dev tool): same inputs, repeated launches must give identical bytes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops
from flash_diffusion_amd._lib import lib
BF = torch.bfloat16
for knob in (0, 1):
    lib().fdmi_tune_set(11, knob)
    for (B, S, Skv, H, d) in [(2, 256, 256, 2, 16), (2, 256, 7, 2, 16), (2, 64, 64, 2, 32), (2, 64, 7, 2, 32), (4, 4096, 4096, 8, 40), (4, 4096, 77, 8, 40), (2, 1024, 1024, 8, 80)]:
        q = torch.randn(B, S, H * d, device="cuda").to(BF)
        k = torch.randn(B, Skv, H * d, device="cuda").to(BF)
        v = torch.randn(B, Skv, H * d, device="cuda").to(BF)
        outs = [ops.attn_fwd(q, k, v, H, d ** -0.5).clone() for _ in range(6)]
        same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
        md = max(float((outs[0].float() - o.float()).abs().max()) for o in outs[1:])
        print(f"knob {knob} B={B} S={S} Skv={Skv} d={d}: identical={same} maxdiff={md:.3e}", flush=True)
