#!/usr/bin/env python
"""In-process interleaved A/B of developer switches on the headline workload C2 (run on the GPU box):

  python scripts/knob_ab.py [--rounds 3] [--steps 3] [--variants base,k14,k15,...]

One model / trainer is built once; every round runs each variant for `--steps` generator iterations back to back (the switches are
flipped through fdmi_tune_set / os.environ between them), so box-to-box and process-to-process spread cancels.  Prints the median
and minimum ms/step per variant, and -- with --legs -- the split of the step into the teacher's 2B forward and the student's
forward + backward (HIP events)."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "base": ({}, {}),
    "unfused_qkv": ({17: 1}, {}),
    "no_gn_epilogue": ({14: 1}, {}),
    "no_dedup": ({13: 1}, {"FDMI_CFG_DEDUP": "0"}),
    "no_tloop": ({}, {"FDMI_TEACHER_LOOP": "0"}),
    "no_side_stream": ({}, {"FDMI_TEACHER_STREAM": "0"}),
    "no_defer": ({}, {"FDMI_DEFER_BACKWARD": "0"}),
    "serial": ({}, {"FDMI_DEFER_BACKWARD": "0", "FDMI_TEACHER_STREAM": "0"}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--variants", default="base,unfused_qkv")
    ap.add_argument("--legs", action="store_true")
    ap.add_argument("--extra", default="", help="extra variants: name:knob=val+knob=val;...")
    args = ap.parse_args()
    import torch
    from flash_diffusion_amd import _lib
    from flash_diffusion_amd.trainer import TrainingConfig, TrainingPipeline
    from flash_diffusion_amd.workloads import SD15, build_flash, synthetic_batch
    for spec in filter(None, args.extra.split(";")):
        name, kv = spec.split(":")
        VARIANTS[name] = ({int(k): int(v) for k, v in (p.split("=") for p in kv.split("+"))}, {})
    L = _lib.lib()
    model = build_flash(SD15, lora_rank=128, n_teacher_steps=4, device="cuda", seed=0)
    pipe = TrainingPipeline(model, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-5],
                                                  trainable_params=[["student_denoiser"]]))
    pipe.configure_optimizers()
    batches = [synthetic_batch(16, 64, 768, seed=1234 + 1000 * i) for i in range(4)]
    names = [v for v in args.variants.split(",") if v] + [s.split(":")[0] for s in filter(None, args.extra.split(";"))]
    all_knobs = sorted({k for n in names for k in VARIANTS[n][0]})
    all_env = sorted({k for n in names for k in VARIANTS[n][1]})

    def apply(name):
        knobs, env = VARIANTS[name]
        for k in all_knobs:
            L.fdmi_tune_set(k, knobs.get(k, 0))
        for k in all_env:
            if k in env:
                os.environ[k] = env[k]
            else:
                os.environ.pop(k, None)

    host_ms = {}

    def run(n, tag=None):
        h = []
        for i in range(n):
            t = time.perf_counter()
            pipe.training_step(batches[i % 4], i)
            h.append((time.perf_counter() - t) * 1e3)   # host time to ISSUE the step (no device wait inside = it runs ahead)
        pipe.finish()
        torch.cuda.synchronize()
        if tag is not None:
            host_ms.setdefault(tag, []).extend(h)

    res = {n: [] for n in names}
    for n in names:   # warm every variant once (allocations, first-launch attribute calls)
        apply(n)
        try:
            run(1)
        except Exception as e:   # a broken variant must not cost the others
            print(f"variant {n} FAILED in warm-up: {e!r}", flush=True)
            res.pop(n)
    for r in range(args.rounds):
        for n in list(res):
            apply(n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps, n)
            res[n].append((time.perf_counter() - t0) / args.steps * 1e3)
    apply("base")
    out = {n: {"median_ms": round(statistics.median(v), 2), "min_ms": round(min(v), 2), "all": [round(x, 2) for x in v]}
           for n, v in res.items()}
    for n, v in out.items():
        v["host_issue_ms_median"] = round(statistics.median(host_ms[n]), 2)
    for n, v in out.items():
        print(f"{n:14s} host-issue {v['host_issue_ms_median']:7.2f}  median {v['median_ms']:8.2f}  min {v['min_ms']:8.2f}  {v['all']}", flush=True)
    if args.legs:
        b = batches[0]
        cond = {"cond": {"crossattn": torch.cat([b["crossattn"], torch.zeros_like(b["crossattn"])], 0)}}
        x2 = torch.randn(32, 4, 64, 64, device="cuda")
        t2 = torch.full((32,), 999.0, device="cuda")

        def ev_time(fn, n=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        with torch.no_grad():
            out["leg_teacher_2B_fwd_ms"] = round(ev_time(lambda: model.teacher_denoiser(x2, t2, cond)), 2)
            out["leg_teacher_2B_fwd_ctx_reuse_ms"] = round(ev_time(lambda: model.teacher_denoiser(x2, t2, cond, ctx_cache="reuse")), 2)
        x1 = torch.randn(16, 4, 64, 64, device="cuda")
        t1 = torch.full((16,), 999.0, device="cuda")
        c1 = {"cond": {"crossattn": b["crossattn"]}}
        with torch.no_grad():
            out["leg_student_fwd_nograd_ms"] = round(ev_time(lambda: model.student_denoiser(x1, t1, c1)), 2)

        def fb():
            y = model.student_denoiser(x1, t1, c1)
            y.backward(torch.ones_like(y))
        out["leg_student_fwd_bwd_ms"] = round(ev_time(fb), 2)
        for k in ("leg_teacher_2B_fwd_ms", "leg_teacher_2B_fwd_ctx_reuse_ms", "leg_student_fwd_nograd_ms", "leg_student_fwd_bwd_ms"):
            print(k, out[k], flush=True)
    print("KNOB_AB_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
