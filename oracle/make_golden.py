"""TEST INFRASTRUCTURE (oracle) -- generate tests/golden/*.npz with the REAL reference.

Runs the reference's own ``flash.models.flash.FlashDiffusion`` (imported unmodified from
/root/reference/src through oracle/shim_import.py) on a tiny SD1.5-shaped UNet (oracle
restatement of diffusers' UNet2DConditionModel -- the fork is absent, see oracle/__init__.py),
records every random draw it consumed, and stores inputs / draws / outputs / losses / grads.

Only runs in the build container (needs /root/reference).  Usage:  python -m oracle.make_golden
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import shim_import
from .flash_ref import Draws, TensorConditioner
from .golden_cases import CASES, SCHEDS, build_models, make_batch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _record_draws(seed, model_fn):
    """Run the oracle restatement once in rng mode under the same seed to capture the draw
    values the reference consumed (the two are bit-identical, tests/test_oracle_vs_reference.py)."""
    torch.manual_seed(seed)
    return model_fn()


def main():
    from .flash_ref import FlashConfigRef, FlashDiffusionRef
    FD, FDC = shim_import.import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, (kw, sched, step, seed) in CASES.items():
        # 1) the real reference
        teacher, student, disc = build_models()
        ref = FD(FDC(**kw), student_denoiser=student, teacher_denoiser=teacher,
                 teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(),
                 discriminator=disc)
        batch = make_batch()
        torch.manual_seed(seed)
        out = ref(batch, step=step, device="cpu")
        loss = out["loss"][step]
        loss.backward()
        grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        # 2) the restatement, to capture the draws (and to double-check bit-identity)
        teacher, student, disc = build_models()
        ora = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(),
                                discriminator=disc)
        torch.manual_seed(seed)
        out2 = ora(make_batch(), step=step, device="cpu")
        for k in ("teacher_output", "student_output", "noisy_sample"):
            assert torch.equal(out[k], out2[k]), (name, k)
        blob = {"z": batch["image"].numpy(), "crossattn": batch["crossattn"].numpy(),
                "step": np.int64(step), "start_timestep": np.int64(out["start_timestep"])}
        for k, v in ora.last_draws.values.items():
            blob["draw:" + k] = v.numpy()
        for k in ("teacher_output", "student_output", "noisy_sample"):
            blob["out:" + k] = out[k].detach().numpy()
        for i in (0, 1):
            blob[f"loss:{i}"] = np.float64(float(out["loss"][i]))
        for k, v in ora.terms.items():
            blob["term:" + k] = np.float64(float(v))
        for n, g in grads.items():
            blob["grad:" + n] = g.numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "loss", blob["loss:0"], blob["loss:1"], "ngrads", len(grads),
              "start_t", out["start_timestep"], os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
