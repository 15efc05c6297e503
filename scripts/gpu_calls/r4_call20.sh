#!/bin/bash
# Round 4, call 20: wgrad_tn old staging (2-byte scatter) vs new (register transpose), operand swap on / off, same box.
cat > rate_tmp.py <<'PY'
import sys, torch
from flash_diffusion_amd import ops, _lib
def rate(M, N1, N2):
    x = torch.randn(M, N1, device="cuda").bfloat16(); y = torch.randn(M, N2, device="cuda").bfloat16()
    c = torch.zeros(N1, N2, device="cuda")
    for _ in range(3): ops.wgrad_tn(x, y, c)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.wgrad_tn(x, y, c)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3
S = [(32768, 1152, 64), (32768, 64, 1152), (32768, 4608, 64), (32768, 64, 4608), (65536, 320, 128), (65536, 128, 320), (16384, 640, 128), (16384, 1536, 64)]
for k in (0, 1):
    _lib.lib().fdmi_tune_set(43, k)
    print(sys.argv[1], "no swap" if k else "swap   ", " ".join(f"{rate(*s):7.1f}" for s in S), flush=True)
PY
echo "us per launch at (M,N1,N2) = (32768,1152,64) (32768,64,1152) (32768,4608,64) (32768,64,4608) (65536,320,128) (65536,128,320) (16384,640,128) (16384,1536,64)"
FDMI_LIB=$PWD/scripts/ab/libfdmi_old_wgrad.so timeout 14 python rate_tmp.py "old staging" 2>&1 | tail -3
timeout 14 python rate_tmp.py "new staging" 2>&1 | tail -3
