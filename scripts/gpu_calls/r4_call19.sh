#!/bin/bash
# Round 4, call 19 (the last GPU seconds): the register-transposing wgrad_tn -- its parity tests, the LoRA-gradient fixtures that run
# through it, and its rate at the C4 / C2 shapes.
mkdir -p gpurun_out/c19
timeout 70 python -m pytest tests/test_zz_dit_gpu.py -q -x -k "wgrad_tn or plan_matches or dit_lora" > gpurun_out/c19/tests.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/c19/tests.txt | cut -c1-300
timeout 25 python - <<'PY' 2>&1 | tail -8
import torch
from flash_diffusion_amd import ops
def rate(M, N1, N2):
    x = torch.randn(M, N1, device="cuda").bfloat16(); y = torch.randn(M, N2, device="cuda").bfloat16()
    c = torch.zeros(N1, N2, device="cuda")
    for _ in range(3): ops.wgrad_tn(x, y, c)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.wgrad_tn(x, y, c)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"wgrad_tn M={M} N1={N1} N2={N2}: {us:.1f} us, {2 * M * (N1 + N2) / us / 1e6:.2f} TB/s of operand bytes")
for s in [(32768, 1152, 64), (32768, 64, 1152), (32768, 4608, 64), (65536, 320, 128), (65536, 128, 320), (16384, 1536, 64)]:
    rate(*s)
PY
