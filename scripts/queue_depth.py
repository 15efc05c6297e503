"""How far can the host run ahead of the GPU?  (dev tool, round 6)  Issues N launches of a ~100 us kernel back to back on one stream
and prints the host time at which launch k returned: the knee where the per-launch issue time jumps from a few us to the kernel's
duration is the number of launches the runtime lets a stream hold in flight -- the depth the step's two-stream schedule has to
live with ("Streams")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops

x = torch.randn(64, 4096, 320, device="cuda").to(torch.bfloat16)      # 168 MB: gn_apply ~ 80 us
g, b = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
y, st = ops.groupnorm_fwd(x, g, b, 32, 1e-5, True)
torch.cuda.synchronize()
for use_torch in (False, True):
    N = 6000
    stamps = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        if use_torch:
            x.mul_(1.0)
        else:
            ops.groupnorm_apply(x, g, b, st, 1e-5, True)
        if k % 250 == 249:
            stamps.append((k + 1, time.perf_counter() - t0))
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(("torch mul_" if use_torch else "fdmi gn_apply"), f"{N} launches: issue done after {t_issue * 1e3:.1f} ms, GPU done after {t_all * 1e3:.1f} ms "
          f"({t_all / N * 1e6:.1f} us per kernel)")
    prev = (0, 0.0)
    for k, t in stamps:
        print(f"   launches {prev[0]:5d}-{k:5d}: {(t - prev[1]) / (k - prev[0]) * 1e6:7.1f} us per launch (host)")
        prev = (k, t)
