import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops
BF = torch.bfloat16
B, S, H, d = 16, 4096, 8, 40
q = torch.randn(B, S, H * d, device="cuda").to(BF)
k = torch.randn(B, S, H * d, device="cuda").to(BF)
v = torch.randn(B, S, H * d, device="cuda").to(BF)
for _ in range(3):
    ops.attn_fwd(q, k, v, H, d ** -0.5)
torch.cuda.synchronize()
