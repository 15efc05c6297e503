#!/usr/bin/env python
"""One leg of the C2 step in isolation, for `rocprofv3 --kernel-trace --stats` (run on the GPU box):
  python scripts/leg_prof.py --leg student|teacher|student_fwd [--iters 4] [--arch sd15]
student = B=16 forward (saved) + backward of the LoRA student; teacher = the frozen 2B=32 CFG forward with the context cache
reused (what 3 of the 4 loop steps run); student_fwd = no-grad student forward (the sampler's call)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", default="student")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--hw", type=int, default=64)
    args = ap.parse_args()
    import torch
    from flash_diffusion_amd.workloads import SD15, build_flash, synthetic_batch
    model = build_flash(SD15, lora_rank=128, n_teacher_steps=4, device="cuda", seed=0)
    B = args.batch
    b = synthetic_batch(B, args.hw, 768)
    if args.leg == "teacher":
        cond = {"cond": {"crossattn": torch.cat([b["crossattn"], torch.zeros_like(b["crossattn"])], 0)}}
        x = torch.randn(2 * B, 4, args.hw, args.hw, device="cuda")
        t = torch.full((2 * B,), 999.0, device="cuda")
        with torch.no_grad():
            model.teacher_denoiser(x, t, cond, ctx_cache="fill")

        def fn():
            with torch.no_grad():
                model.teacher_denoiser(x, t, cond, ctx_cache="reuse")
    else:
        x = torch.randn(B, 4, args.hw, args.hw, device="cuda")
        t = torch.full((B,), 999.0, device="cuda")
        c = {"cond": {"crossattn": b["crossattn"]}}
        if args.leg == "student_fwd":
            def fn():
                with torch.no_grad():
                    model.student_denoiser(x, t, c)
        else:
            def fn():
                y = model.student_denoiser(x, t, c)
                y.backward(torch.ones_like(y))
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        fn()
    torch.cuda.synchronize()
    print(f"LEG {args.leg} {(time.perf_counter() - t0) / args.iters * 1e3:.2f} ms/iter ({args.iters} iters + 1 warm-up)")


if __name__ == "__main__":
    main()
