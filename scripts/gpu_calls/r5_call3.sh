#!/bin/bash
# Round-5 third GPU call: s_setprio experiment on the d = 40 attention forward (knob 47), the weight-gradient kernel's row-split rule.
set -u
out=gpurun_out/r5c3
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
cat > /tmp/attn_prio.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import torch
from flash_diffusion_amd import ops, _lib
from kbench import timeit
L = _lib.lib()
BF = torch.bfloat16
for (B, S, H, d) in [(32, 4096, 8, 40), (16, 4096, 8, 40)]:
    q, k, v = (torch.randn(B, S, H * d, device="cuda").to(BF) for _ in range(3))
    ref = None
    res = []
    for prio in (0, 1, 2, 0, 1, 2):
        L.fdmi_tune_set(47, prio)
        timeit(lambda: ops.attn_fwd(q, k, v, H, d ** -0.5), 5)
        us = timeit(lambda: ops.attn_fwd(q, k, v, H, d ** -0.5), 20)
        o = ops.attn_fwd(q, k, v, H, d ** -0.5)
        torch.cuda.synchronize()
        if ref is None:
            ref = o.clone()
        res.append(f"prio{prio}: {us:7.1f} us {4.0 * B * H * S * S * d / us / 1e6:6.1f} TF{'' if torch.equal(o, ref) else ' MISMATCH'}")
    L.fdmi_tune_set(47, 0)
    print(f"attn fwd B={B} S={S} d={d}: " + " | ".join(res), flush=True)
PY
run 01_attn_prio timeout 300 python /tmp/attn_prio.py
cat "$out/01_attn_prio.log"
run 02_wgrad_rates timeout 300 python scripts/wgrad_rates.py
cat "$out/02_wgrad_rates.log" | cut -c1-400
