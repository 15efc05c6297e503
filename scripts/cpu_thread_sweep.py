#!/usr/bin/env python
"""Host-thread sweep for bench.py's cpu_baseline leg: one fp32 SD1.5 oracle UNet forward (B=1, 64x64 latents) per thread count."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle.unet_cpu import UNet2DConditionRef, sd15_config  # noqa: E402

torch.manual_seed(0)
net = UNet2DConditionRef(sd15_config()).eval()
x, t, c = torch.randn(1, 4, 64, 64), torch.tensor([999]), {"cond": {"crossattn": torch.randn(1, 77, 768)}}
try:
    import psutil
    print("logical", psutil.cpu_count(), "physical", psutil.cpu_count(logical=False), flush=True)
except Exception as e:
    print("psutil:", e)
for n in (16, 32, 64, 96, 128):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    with torch.no_grad():
        net(x, t, c)
        t0 = time.perf_counter()
        for _ in range(2):
            net(x, t, c)
        dt = (time.perf_counter() - t0) / 2
    print(f"threads {n:4d}: {dt:.2f} s per UNet forward ({0.8033 / dt:.2f} TFLOP/s)", flush=True)
