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


def main(cases=None):
    from .flash_ref import FlashConfigRef, FlashDiffusionRef
    from .golden_cases import ADAPTER_CASES, LPIPS_CASES, LPIPS_REAL_CASES, build_lpips_real, make_edge, make_pixel_batch
    from .unet_cpu import TinyLPIPS as _TinyLPIPS, TinyT2IAdapter, TinyVAE as _TinyVAE, tiny_config
    FD, FDC = shim_import.import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, (kw, sched, step, seed) in (cases if cases is not None else CASES).items():
        with_adapter = name in ADAPTER_CASES
        with_vae = name in LPIPS_CASES or name in LPIPS_REAL_CASES
        # the real architectures (restated AutoencoderKL decoder + VGG16 LPIPS) or the toy stand-ins
        TinyVAE = (lambda: build_lpips_real()[0]) if name in LPIPS_REAL_CASES else _TinyVAE
        TinyLPIPS = (lambda: build_lpips_real()[1]) if name in LPIPS_REAL_CASES else _TinyLPIPS

        def make_batch_():
            b = make_pixel_batch() if with_vae else make_batch()
            if with_adapter:
                b["edge"] = make_edge()
            return b
        # 1) the real reference
        teacher, student, disc = build_models()
        ref = FD(FDC(**kw), student_denoiser=student, teacher_denoiser=teacher,
                 teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(),
                 discriminator=disc, adapter=TinyT2IAdapter(tiny_config()) if with_adapter else None,
                 vae=TinyVAE() if with_vae else None)
        if with_vae:
            ref.lpips = TinyLPIPS()       # FD:102-103 built the shim's lpips.LPIPS stub (package and VGG weights are absent)
        batch = make_batch_()
        torch.manual_seed(seed)
        out = ref(batch, step=step, device="cpu")
        loss = out["loss"][step]
        loss.backward()
        grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        # 2) the restatement, to capture the draws (and to double-check bit-identity)
        teacher, student, disc = build_models()
        ora = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(),
                                discriminator=disc, adapter=TinyT2IAdapter(tiny_config()) if with_adapter else None,
                                vae=TinyVAE() if with_vae else None, lpips_model=TinyLPIPS() if with_vae else None)
        torch.manual_seed(seed)
        out2 = ora(make_batch_(), step=step, device="cpu")
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
        for n, p_ in ref.discriminator.named_parameters():   # (wgan clamps the weights inside the forward, FD:573-585)
            blob["post:discriminator." + n] = p_.detach().numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "loss", blob["loss:0"], blob["loss:1"], "ngrads", len(grads),
              "start_t", out["start_timestep"], os.path.getsize(path) // 1024, "KiB")


def make_sample_golden():
    """FlashDiffusion.sample (FD:754-915) of the REAL reference: 4-step LCM student sampler with CFG and the teacher's own
    4-step DPM-Solver++ sampler next to it; the LCM re-noising draws are recorded so the HIP path can replay them."""
    from .sched_cpu import DPMSolverMultistepSchedulerRef, LCMSchedulerRef
    FD, FDC = shim_import.import_reference()
    teacher, student, disc = build_models()
    g = torch.Generator().manual_seed(21)
    for p_ in student.parameters():          # peft leaves LoRA B = 0: make the student differ from the teacher
        if p_.requires_grad and p_.abs().max() == 0:
            p_.data.copy_(torch.randn(p_.shape, generator=g) * 0.05)
    ref = FD(FDC(K=[4], num_iterations_per_K=[10]), student_denoiser=student, teacher_denoiser=teacher,
             teacher_noise_scheduler=DPMSolverMultistepSchedulerRef(), conditioner=TensorConditioner(),
             discriminator=disc)
    ref.sampling_noise_scheduler = LCMSchedulerRef()
    ref.teacher_sampling_noise_scheduler = DPMSolverMultistepSchedulerRef()
    noises = []

    def noise_fn(shape):
        n = torch.randn(shape, generator=g)
        noises.append(n)
        return n

    ref.sampling_noise_scheduler.noise_fn = noise_fn
    batch = make_batch()
    z = torch.randn(2, 4, 32, 32, generator=g)
    un = {"crossattn": torch.randn(2, 77, 64, generator=g), "text": ["", ""]}
    s, sr = ref.sample(z, num_steps=4, guidance_scale=1.3, teacher_guidance_scale=5.0, conditioner_inputs=batch,
                       uncond_conditioner_inputs=un, log_teacher_samples=True)
    blob = {"z": z.numpy(), "crossattn": batch["crossattn"].numpy(), "uncond_crossattn": un["crossattn"].numpy(),
            "student_sample": s.numpy(), "teacher_sample": sr.numpy(), "guidance_scale": np.float64(1.3),
            "teacher_guidance_scale": np.float64(5.0), "num_steps": np.int64(4),
            "lcm_timesteps": ref.sampling_noise_scheduler.timesteps.numpy()}
    for i, n in enumerate(noises):
        blob[f"lcm_noise:{i}"] = n.numpy()
    for n_, p_ in student.named_parameters():
        if "lora" in n_:
            blob["lora:" + n_] = p_.detach().numpy()
    path = os.path.join(OUT, "sample_lcm4.npz")
    np.savez_compressed(path, **blob)
    print("sample_lcm4", float(s.abs().mean()), float(sr.abs().mean()), len(noises), os.path.getsize(path) // 1024, "KiB")


def make_sd3_golden():
    """FlashDiffusionSD3.forward (+ backward) of the REAL reference on the seeded test denoisers; draws recorded through
    the bit-identical restatement (oracle/flash_sd3_ref.py)."""
    from .flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from .golden_cases import SD3_CASES, build_sd3_models
    from .sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    FD3, FD3C = shim_import.import_reference_sd3()
    for name, (kw, step, seed) in SD3_CASES.items():
        teacher, student, disc, pipe, batch = build_sd3_models()
        ref = FD3(FD3C(**kw), student_denoiser=student, teacher_denoiser=teacher,
                  teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=disc, pipeline=pipe)
        torch.manual_seed(seed)
        out = ref(batch, step=step)
        out["loss"][step].backward()
        grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        teacher, student, disc, pipe, batch2 = build_sd3_models()
        ora = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                   teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=disc,
                                   pipeline=pipe)
        torch.manual_seed(seed)
        out2 = ora(batch2, step=step)
        for k in ("teacher_output", "student_output", "noisy_sample"):
            assert torch.equal(out[k], out2[k]), (name, k)
        blob = {"z": batch["image"].numpy(), "step": np.int64(step), "start_timestep": np.float64(out["start_timestep"])}
        for k, v in ora.last_draws.values.items():
            blob["draw:" + k] = v.numpy()
        for k in ("teacher_output", "student_output", "noisy_sample"):
            blob["out:" + k] = out[k].detach().numpy()
        for i in (0, 1):
            blob[f"loss:{i}"] = np.float64(float(out["loss"][i]))
        for n, g in grads.items():
            blob["grad:" + n] = g.numpy()
        for n, p_ in ref.discriminator.named_parameters():   # (wgan clamps the weights inside the forward, FD:573-585)
            blob["post:discriminator." + n] = p_.detach().numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "loss", blob["loss:0"], blob["loss:1"], "ngrads", len(grads), "start_t", out["start_timestep"],
              os.path.getsize(path) // 1024, "KiB")


def make_dit_golden():
    """PixArt / SD3 denoiser fixtures: the reference's REAL wrapper classes (on the restated diffusers bases,
    oracle/shim_import.py) with the seeded weights of golden_cases.build_dit / build_mmdit -- frozen forward, LoRA forward and
    the LoRA gradients of sum(out * w)"""
    from . import dit_cpu
    from .golden_cases import DIT_CASES, MMDIT_CASES, build_dit, build_mmdit
    PixWrapper, _ = shim_import.import_reference_dit()
    SD3Wrapper = shim_import.import_reference_sd3_wrapper()
    for name in list(DIT_CASES) + list(MMDIT_CASES):
        Wrapper, build = (PixWrapper, build_dit) if name in DIT_CASES else (SD3Wrapper, build_mmdit)
        blob = {}
        for r in (0, 8):
            cfg, ora, (x, t, cond), w = build(name, lora_r=r)
            real = Wrapper(**cfg)
            if r:
                dit_cpu.add_lora_(real, r, seed=4, b_std=0.05)
            real.load_state_dict(ora.state_dict())
            out = real(x, t, cond)
            assert torch.equal(out, ora(x, t, cond)), name
            if not r:
                blob["out:frozen"] = out.detach().numpy()
                continue
            blob["out:lora"] = out.detach().numpy()
            (out * w).sum().backward()
            for n, p in real.named_parameters():
                if p.grad is not None:
                    blob["grad:" + n.replace(".base_layer.", ".")] = p.grad.numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "ngrads", sum(k.startswith("grad:") for k in blob), os.path.getsize(path) // 1024, "KiB")


def make_sd3_mmdit_golden():
    """FlashDiffusionSD3 (the REAL class) over the REAL DiffusersSD3Transformer2DWrapper (restated diffusers base) holding the
    seeded tiny-MMDiT weights: fixtures tests/golden/sd3_mmdit_*.npz for the step over the HIP MMDiT."""
    from . import dit_cpu
    from .flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from .golden_cases import SD3_MMDIT_CASES, build_sd3_mmdit_inputs
    from .sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    FD3, FD3C = shim_import.import_reference_sd3()
    Wrapper = shim_import.import_reference_sd3_wrapper()
    for name, (kw, case, step, seed) in SD3_MMDIT_CASES.items():
        def models():
            cfg, t_o, s_o, head, pipe, batch = build_sd3_mmdit_inputs(case)
            teacher = Wrapper(**cfg)
            teacher.load_state_dict(t_o.state_dict())
            teacher.freeze()
            student = Wrapper(**cfg)
            dit_cpu.add_lora_(student, 8, seed=4, b_std=0.05)
            student.load_state_dict(s_o.state_dict())
            return teacher, student, head, pipe, batch
        teacher, student, head, pipe, batch = models()
        ref = FD3(FD3C(**kw), student_denoiser=student, teacher_denoiser=teacher,
                  teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=head, pipeline=pipe)
        torch.manual_seed(seed)
        out = ref(batch, step=step)
        out["loss"][step].backward()
        grads = {n.replace(".base_layer.", "."): p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        teacher, student, head, pipe, batch2 = models()
        ora = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                   teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=head, pipeline=pipe)
        torch.manual_seed(seed)
        out2 = ora(batch2, step=step)
        for k in ("teacher_output", "student_output", "noisy_sample"):
            assert torch.equal(out[k], out2[k]), (name, k)
        blob = {"z": batch["image"].numpy(), "step": np.int64(step), "start_timestep": np.float64(out["start_timestep"])}
        for k, v in ora.last_draws.values.items():
            blob["draw:" + k] = v.numpy()
        for k in ("teacher_output", "student_output", "noisy_sample"):
            blob["out:" + k] = out[k].detach().numpy()
        for i in (0, 1):
            blob[f"loss:{i}"] = np.float64(float(out["loss"][i]))
        for n, g in grads.items():
            blob["grad:" + n] = g.numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "loss", blob["loss:0"], blob["loss:1"], "ngrads", len(grads), "start_t", out["start_timestep"],
              os.path.getsize(path) // 1024, "KiB")


def make_pixart_step_golden():
    """FlashDiffusion (the REAL class) over the REAL DiffusersTransformer2DWrapper (TW:9-100) with the PixArt example's head and
    loss recipe: fixtures tests/golden/pixart_*.npz for the epsilon-prediction step over the HIP DiT (BASELINE C4's workload)."""
    from . import dit_cpu
    from .flash_ref import FlashConfigRef, FlashDiffusionRef
    from .golden_cases import PIXART_STEP_CASES, PromptTableConditioner, build_pixart_step_inputs
    FD, FDC = shim_import.import_reference()
    Wrapper, _ = shim_import.import_reference_dit()
    for name, (kw, step, seed) in PIXART_STEP_CASES.items():
        def models():
            cfg, t_o, s_o, head, batch = build_pixart_step_inputs()
            teacher = Wrapper(**cfg)
            teacher.load_state_dict(t_o.state_dict())
            teacher.freeze()
            student = Wrapper(**cfg)
            dit_cpu.add_lora_(student, 8, seed=4, b_std=0.05)
            student.load_state_dict(s_o.state_dict())
            return teacher, student, head, batch
        teacher, student, head, batch = models()
        ref = FD(FDC(**kw), student_denoiser=student, teacher_denoiser=teacher, teacher_noise_scheduler=SCHEDS["dpm"](),
                 conditioner=PromptTableConditioner(), discriminator=head)
        torch.manual_seed(seed)
        out = ref(batch, step=step, device="cpu")
        out["loss"][step].backward()
        grads = {n.replace(".base_layer.", "."): p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        teacher, student, head, batch2 = models()
        ora = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=PromptTableConditioner(), discriminator=head)
        torch.manual_seed(seed)
        out2 = ora(batch2, step=step, device="cpu")
        for k in ("teacher_output", "student_output", "noisy_sample"):
            assert torch.equal(out[k], out2[k]), (name, k)
        assert float(out["loss"][step]) == float(out2["loss"][step])
        blob = {"z": batch["image"].numpy(), "step": np.int64(step), "start_timestep": np.int64(out["start_timestep"])}
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
        print(name, "loss", blob["loss:0"], blob["loss:1"], "ngrads", len(grads), "start_t", out["start_timestep"],
              {k: float(blob[k]) for k in blob if k.startswith("term:")}, os.path.getsize(path) // 1024, "KiB")


def make_fullsize_golden(names=None):
    """fp32 oracle output of one full-size B = 1 forward per C3 / C4 / C5 denoiser (tests/golden/full_*.npz); the weights and
    inputs are rebuilt from oracle/hash_init.py wherever the fixture is replayed"""
    from .golden_cases import FULL_CASES, build_full_oracle, full_inputs
    import time
    for name in (names or FULL_CASES):
        t0 = time.time()
        m = build_full_oracle(name)
        x, t, cond = full_inputs(name)
        t1 = time.time()
        with torch.no_grad():
            out = m(x, t, cond)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, out=out.numpy(), nparams=np.int64(sum(p.numel() for p in m.parameters())),
                            psum=np.float64(sum(float(p.double().sum()) for p in m.parameters())))
        print(name, tuple(out.shape), "mean |out|", float(out.abs().mean()), "std", float(out.std()),
              f"init {t1 - t0:.0f} s fwd {time.time() - t1:.0f} s", os.path.getsize(path) // 1024, "KiB")
        del m


from .golden_cases import C1_KW, C1_SEED, build_c1_models, c1_grad_probe  # noqa: E402


def make_fullstep_golden(names=None):
    """Full-width B = 1 steps of C3 / C4 / C5 (tests/golden/step_{sdxl,pixart,sd3}.npz; golden_cases.FULLSTEP_CASES): the REAL
    FlashDiffusion / FlashDiffusionSD3 over the fp32 oracle denoisers at their real widths, forward and backward; stored like the
    C1 / C2 fixtures (draws, outputs, every loss term, per-tensor gradient norm + seeded projection, four LoRA tensors in full).
    Host memory: the materialised attention probabilities of 28 blocks x 4096 tokens do not fit beside the models, so the oracle
    denoisers run fp32 SDPA (log-sum-exp saved) in BOTH runs -- the same arithmetic in a different association order."""
    import gc
    import time
    from . import dit_cpu, unet_cpu
    from .flash_ref import FlashConfigRef, FlashDiffusionRef
    from .flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from .flash_ref import timestep_pmf
    from .golden_cases import FULLSTEP4_HW, FULLSTEP_CASES, FULLSTEP_CASES_ALL, build_fullstep_models, fullstep_inputs
    from .sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    unet_cpu.FUSED_ATTENTION = True
    dit_cpu.FUSED_ATTENTION = True
    for name in (names or FULLSTEP_CASES):
        kind, kw, seed = FULLSTEP_CASES_ALL[name]
        t0 = time.time()
        four = name.startswith("step4_")
        if four:
            # `step4_*`: all four teacher steps.  The reference draws the start index (FD:167 / FD3:177, after
            # `randn_like(z)`): take the first seed >= the case's whose draw is index 0.  FDMI_STEP4_HW=<n> overrides the latent size.
            if os.environ.get("FDMI_STEP4_HW"):
                FULLSTEP4_HW[name] = int(os.environ["FDMI_STEP4_HW"])
            z0 = fullstep_inputs(name)[0]["image"]
            pmf = timestep_pmf(FlashConfigRef(**{k: v for k, v in kw.items() if k not in ("ucg_keys", "use_empty_prompt")}), 4, 0)
            for s in range(seed, seed + 100):
                torch.manual_seed(s)
                torch.randn_like(z0)
                if int(torch.multinomial(pmf, 1)) == 0:
                    seed = s
                    break

        def build(real):
            teacher, student, disc = build_fullstep_models(name)
            batch, cond = fullstep_inputs(name)
            if kind == "fd":
                FD, FDC = shim_import.import_reference()
                cls, cfg = (FD, FDC) if real else (FlashDiffusionRef, FlashConfigRef)
                m = cls(cfg(**kw), student_denoiser=student, teacher_denoiser=teacher, teacher_noise_scheduler=SCHEDS["dpm"](),
                        conditioner=cond, discriminator=disc)
            else:
                FD3, FD3C = shim_import.import_reference_sd3()
                cls, cfg = (FD3, FD3C) if real else (FlashDiffusionSD3Ref, FlashSD3ConfigRef)
                m = cls(cfg(**kw), student_denoiser=student, teacher_denoiser=teacher,
                        teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=disc, pipeline=cond)
            return m, batch
        ref, batch = build(True)
        torch.manual_seed(seed)
        out = ref(batch, step=0, device="cpu") if kind == "fd" else ref(batch, step=0)
        out["loss"][0].backward()
        grads = {n.replace(".base_layer.", "."): p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        keep = {k: out[k].detach().clone() for k in ("teacher_output", "student_output", "noisy_sample")}
        losses = [float(out["loss"][i]) for i in (0, 1)]
        start_t = float(out["start_timestep"])
        del ref, out
        gc.collect()
        t1 = time.time()
        ora, batch2 = build(False)
        torch.manual_seed(seed)
        out2 = ora(batch2, step=0, device="cpu") if kind == "fd" else ora(batch2, step=0)
        for k in keep:
            assert torch.equal(keep[k], out2[k]), (name, k)
        assert float(out2["loss"][0]) == losses[0], (name, float(out2["loss"][0]), losses[0])
        blob = {"step": np.int64(0), "start_timestep": np.float64(start_t), "seed": np.int64(seed)}
        if four:
            assert int(ora.last_draws.values["start_idx"]) == 0, ora.last_draws.values["start_idx"]   # all four teacher steps
            blob["B"] = np.int64(batch2["image"].shape[0])
            blob["hw"] = np.int64(batch2["image"].shape[-1])
        for k, v in ora.last_draws.values.items():
            blob["draw:" + k] = v.numpy()
        for k, v in keep.items():
            blob["out:" + k] = v.numpy()
        for i in (0, 1):
            blob[f"loss:{i}"] = np.float64(losses[i])
        for k, v in getattr(ora, "terms", {}).items():
            blob["term:" + k] = np.float64(float(v))
        names_ = sorted(grads)
        blob["gradnames"] = np.array(names_)
        blob["gradnorm"] = np.array([float(grads[n].double().norm()) for n in names_])
        blob["gradproj"] = np.array([float(grads[n].double().flatten() @ c1_grad_probe(grads[n].numel(), 1000 + i).double())
                                     for i, n in enumerate(names_)])
        lora = [n for n in names_ if ".lora_" in n]
        for n in lora[:2] + lora[-2:]:
            blob["grad:" + n] = grads[n].numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "loss", losses, "terms", {k: float(blob[k]) for k in blob if k.startswith("term:")}, "ngrads", len(names_),
              f"start_t {start_t} reference run {t1 - t0:.0f} s, restatement run {time.time() - t1:.0f} s", os.path.getsize(path) // 1024,
              "KiB", flush=True)
        del ora, out2, grads
        gc.collect()



def make_bf16_anchor(tag):
    """The reference's OWN training precision as a yardstick: the pinned oracle replays a full-width step
    fixture's draws under `torch.autocast("cpu", bfloat16)` -- PyTorch's implementation of the `precision="bf16-mixed"` the
    reference trains with (examples/train_flash_sd.py:405) -- and the distances of that run to the fixture's fp32 run are stored
    in tests/golden/<tag>_bf16ref.npz.  The GPU tests hold the HIP bf16 path to 1.5 x these distances instead of to bars derived
    from its own history.  Forward only (outputs and loss terms).  tag: c2_sd15_r128_n4[_b16] or step4_{sdxl,pixart,sd3}."""
    import time
    from . import dit_cpu, unet_cpu
    from .flash_ref import Draws, FlashConfigRef, FlashDiffusionRef
    from .flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from .golden_cases import (C2_KW, FULLSTEP_CASES_ALL, build_c2_models, build_fullstep_models, c2_batch, fullstep_inputs)
    from .sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    unet_cpu.FUSED_ATTENTION = True
    dit_cpu.FUSED_ATTENTION = True
    blob = np.load(os.path.join(OUT, tag + ".npz"))
    draws = {k[5:]: torch.from_numpy(blob[k]) for k in blob.files if k.startswith("draw:")}
    t0 = time.time()
    if tag.startswith("c2_"):
        B = int(blob["B"]) if "B" in blob.files else 2
        teacher, student, disc = build_c2_models()
        m = FlashDiffusionRef(FlashConfigRef(**C2_KW), student_denoiser=student, teacher_denoiser=teacher,
                              teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=TensorConditioner(), discriminator=disc)
        batch, call = c2_batch(B=B), (lambda mm, b: mm(b, step=0, device="cpu"))
    else:
        kind, kw, _ = FULLSTEP_CASES_ALL[tag]
        teacher, student, disc = build_fullstep_models(tag)
        batch, cond = fullstep_inputs(tag, hw=int(blob["hw"]) if "hw" in blob.files else None)
        if kind == "fd":
            m = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                  teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=cond, discriminator=disc)
            call = lambda mm, b: mm(b, step=0, device="cpu")
        else:
            m = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                     teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=disc, pipeline=cond)
            call = lambda mm, b: mm(b, step=0)
    m.draws = Draws(draws)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        out = call(m, batch)

    def rel(a, b):
        a, b = a.detach().double(), torch.from_numpy(b).double()
        return float((a - b).norm() / (b.norm() + 1e-30))
    rec = {"teacher_output_rel": rel(out["teacher_output"].float(), blob["out:teacher_output"]),
           "student_output_rel": rel(out["student_output"].float(), blob["out:student_output"]),
           "loss_rel": abs(float(out["loss"][0]) - float(blob["loss:0"])) / abs(float(blob["loss:0"]))}
    for k, v in getattr(m, "terms", {}).items():
        if "term:" + k in blob.files and k not in ("K_step", "guidance") and float(blob["term:" + k]) != 0:
            rec["term_rel:" + k] = abs(float(v) - float(blob["term:" + k])) / abs(float(blob["term:" + k]))
    np.savez(os.path.join(OUT, tag + "_bf16ref.npz"), **{k: np.float64(v) for k, v in rec.items()},
             mode=np.array("torch.autocast(cpu, bfloat16), no_grad, oracle restatement (bit-pinned to the reference class)"))
    print(tag, "reference bf16-mixed vs its fp32 run:", {k: f"{v:.3e}" for k, v in rec.items()}, f"{time.time() - t0:.0f} s", flush=True)


def make_c2_golden(B=2):
    """One C2-shaped step (fixture tests/golden/c2_sd15_r128_n4.npz): the REAL reference class, full-size SD1.5, r128, all four
    teacher CFG steps, B = 2; stored like the C1 fixture (outputs, losses, per-tensor gradient norm + seeded projection, four
    LoRA tensors in full).  B = 16 (`python -m oracle.make_golden c2 16`, fixture c2_sd15_r128_n4_b16.npz) is BASELINE.json
    configs[1] at its own batch: the shapes bench.py times (about a quarter of an hour on 8 host cores)."""
    from .flash_ref import FlashConfigRef, FlashDiffusionRef, timestep_pmf
    from .golden_cases import C2_KW, build_c2_models, c2_batch
    FD, FDC = shim_import.import_reference()
    batch = c2_batch(B=B)
    tag = "c2_sd15_r128_n4" + ("" if B == 2 else f"_b{B}")
    if B > 2:      # host memory: fp32 SDPA (log-sum-exp saved) instead of materialised probabilities, in BOTH runs below
        from . import unet_cpu
        unet_cpu.FUSED_ATTENTION = True
    # the reference draws the start index from the pmf (FD:167): pick the seed whose draw is index 0 (= all four teacher steps)
    pmf = timestep_pmf(FlashConfigRef(**C2_KW), 4, 0)
    seed = None
    for s in range(100, 200):
        torch.manual_seed(s)
        torch.randn_like(batch["image"])
        if int(torch.multinomial(pmf, 1)) == 0:
            seed = s
            break
    teacher, student, disc = build_c2_models()
    ref = FD(FDC(**C2_KW), student_denoiser=student, teacher_denoiser=teacher,
             teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=TensorConditioner(), discriminator=disc)
    torch.manual_seed(seed)
    out = ref(batch, step=0, device="cpu")
    out["loss"][0].backward()
    grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    del ref
    teacher, student, disc = build_c2_models()
    ora = FlashDiffusionRef(FlashConfigRef(**C2_KW), student_denoiser=student, teacher_denoiser=teacher,
                            teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=TensorConditioner(), discriminator=disc)
    torch.manual_seed(seed)
    out2 = ora(c2_batch(B=B), step=0, device="cpu")
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert torch.equal(out[k], out2[k]), k
    assert int(out["start_timestep"]) == 999, out["start_timestep"]    # start index 0 of the trailing K = 4 schedule: 4 teacher steps
    blob = {"step": np.int64(0), "start_timestep": np.int64(out["start_timestep"]), "seed": np.int64(seed), "B": np.int64(B)}
    for k, v in ora.last_draws.values.items():
        blob["draw:" + k] = v.numpy()
    for k in ("teacher_output", "student_output", "noisy_sample"):
        blob["out:" + k] = out[k].detach().numpy()
    for i in (0, 1):
        blob[f"loss:{i}"] = np.float64(float(out["loss"][i]))
    for k, v in ora.terms.items():
        blob["term:" + k] = np.float64(float(v))
    names = sorted(grads)
    blob["gradnames"] = np.array(names)
    blob["gradnorm"] = np.array([float(grads[n].double().norm()) for n in names])
    blob["gradproj"] = np.array([float(grads[n].double().flatten() @ c1_grad_probe(grads[n].numel(), 1000 + i).double())
                                 for i, n in enumerate(names)])
    lora = [n for n in names if ".lora_" in n]
    for n in lora[:2] + lora[-2:]:
        blob["grad:" + n] = grads[n].numpy()
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **blob)
    print(tag, "seed", seed, "loss", blob["loss:0"], "terms", {k: float(blob[k]) for k in blob if k.startswith("term:")},
          "ngrads", len(names), os.path.getsize(path) // 1024, "KiB")


def make_c1_golden():
    """One full-size step (fixture tests/golden/c1_sd15_full.npz): inputs, draws, outputs, every loss term, and for each of
    the 256 LoRA / discriminator gradient tensors its norm and its projection on a seeded random direction (plus the first
    and last LoRA pair in full)."""
    from .flash_ref import FlashConfigRef, FlashDiffusionRef
    FD, FDC = shim_import.import_reference()
    batch = make_batch(B=1, hw=64, ctx_dim=768, seed=7)
    teacher, student, disc = build_c1_models()
    ref = FD(FDC(**C1_KW), student_denoiser=student, teacher_denoiser=teacher,
             teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=TensorConditioner(), discriminator=disc)
    torch.manual_seed(C1_SEED)
    out = ref(batch, step=0, device="cpu")
    out["loss"][0].backward()
    grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    teacher, student, disc = build_c1_models()
    ora = FlashDiffusionRef(FlashConfigRef(**C1_KW), student_denoiser=student, teacher_denoiser=teacher,
                            teacher_noise_scheduler=SCHEDS["dpm"](), conditioner=TensorConditioner(), discriminator=disc)
    torch.manual_seed(C1_SEED)
    out2 = ora(make_batch(B=1, hw=64, ctx_dim=768, seed=7), step=0, device="cpu")
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert torch.equal(out[k], out2[k]), k
    blob = {"z": batch["image"].numpy(), "crossattn": batch["crossattn"].numpy(), "step": np.int64(0),
            "start_timestep": np.int64(out["start_timestep"])}
    for k, v in ora.last_draws.values.items():
        blob["draw:" + k] = v.numpy()
    for k in ("teacher_output", "student_output", "noisy_sample"):
        blob["out:" + k] = out[k].detach().numpy()
    for i in (0, 1):
        blob[f"loss:{i}"] = np.float64(float(out["loss"][i]))
    for k, v in ora.terms.items():
        blob["term:" + k] = np.float64(float(v))
    names = sorted(grads)
    blob["gradnames"] = np.array(names)
    blob["gradnorm"] = np.array([float(grads[n].double().norm()) for n in names])
    blob["gradproj"] = np.array([float(grads[n].double().flatten() @ c1_grad_probe(grads[n].numel(), 1000 + i).double())
                                 for i, n in enumerate(names)])
    lora = [n for n in names if ".lora_" in n]
    for n in lora[:2] + lora[-2:]:
        blob["grad:" + n] = grads[n].numpy()
    path = os.path.join(OUT, "c1_sd15_full.npz")
    np.savez_compressed(path, **blob)
    print("c1_sd15_full loss", blob["loss:0"], "terms", {k: float(blob[k]) for k in blob if k.startswith("term:")},
          "ngrads", len(names), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "c1":
        make_c1_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "sd3_mmdit":
        make_sd3_mmdit_golden()
        make_pixart_step_golden()
        make_fullsize_golden()
        make_c2_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "pixart_step":
        make_pixart_step_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "full":
        make_fullsize_golden(sys.argv[2:] or None)
    elif len(sys.argv) > 1 and sys.argv[1] == "fullstep":
        import resource                  # an allocation beyond the host's memory should fail here, not take the container down
        resource.setrlimit(resource.RLIMIT_AS, (int(os.environ.get("FDMI_GOLDEN_MEM_GB", "58")) << 30,) * 2)
        make_fullstep_golden(sys.argv[2:] or None)
    elif len(sys.argv) > 1 and sys.argv[1] == "bf16anchor":
        for tag in sys.argv[2:] or ("c2_sd15_r128_n4", "c2_sd15_r128_n4_b16", "step4_sdxl", "step4_pixart", "step4_sd3"):
            make_bf16_anchor(tag)
    elif len(sys.argv) > 1 and sys.argv[1] == "c2":
        make_c2_golden(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
    elif len(sys.argv) > 1 and sys.argv[1] == "gan":
        from .golden_cases import CASES as _C
        want = sys.argv[2:] or ("g_wgan", "d_wgan", "d_lsgan", "d_vanilla", "d_nonsat")
        main({k: v for k, v in _C.items() if k in want})
    elif len(sys.argv) > 1 and sys.argv[1] == "sample":
        make_sample_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "sd3":
        make_sd3_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "dit":
        make_dit_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "adapter":
        from .golden_cases import ADAPTER_CASES
        main(ADAPTER_CASES)
    elif len(sys.argv) > 1 and sys.argv[1] == "lpips":
        from .golden_cases import LPIPS_CASES
        main(LPIPS_CASES)
    elif len(sys.argv) > 1 and sys.argv[1] == "lpips_real":
        from .golden_cases import LPIPS_REAL_CASES
        main(LPIPS_REAL_CASES)
    else:
        main()
        from .golden_cases import ADAPTER_CASES, LPIPS_CASES
        main(ADAPTER_CASES)
        main(LPIPS_CASES)
        from .golden_cases import LPIPS_REAL_CASES
        main(LPIPS_REAL_CASES)
        make_sample_golden()
        make_sd3_golden()
        make_dit_golden()
        make_c1_golden()
        make_sd3_mmdit_golden()
        make_pixart_step_golden()
        make_fullsize_golden()
        make_c2_golden()
