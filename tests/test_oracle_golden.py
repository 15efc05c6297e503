"""CPU: the oracle restatement, fed the recorded draws, reproduces the fixtures that the REAL
reference FlashDiffusion produced (tests/golden/*.npz)."""
import pytest
import torch

from oracle.flash_ref import Draws, FlashConfigRef, FlashDiffusionRef, TensorConditioner
from oracle.golden_cases import CASES, SCHEDS, build_models
from tests.golden_util import load_case, rel_err


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_golden(name):
    kw, sched, step, _seed = CASES[name]
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(),
                          discriminator=disc)
    m.draws = Draws(g["draws"])
    batch = {"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * g["z"].shape[0]}
    out = m(batch, step=step)
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-5, k
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-5 * max(1.0, abs(g["loss"][i]))
    out["loss"][step].backward()
    n = 0
    for pn, p in m.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad, g["grads"][pn]) < 1e-4, pn
            n += 1
    assert n == len(g["grads"])
    # reference invariants (tests/test_flash/test_flash_diffusion.py:146-153)
    if step == 0:
        assert g["loss"][0] > 0 and g["loss"][1] == 0.0
    else:
        # (a Wasserstein critic loss has no sign: the reference's `> 0` invariant is for its own lsgan test configuration)
        assert g["loss"][1] > 0 or (kw.get("gan_loss_type") == "wgan" and g["loss"][1] != 0)


def test_scheduler_trailing_timesteps():
    from oracle.sched_cpu import DPMSolverMultistepSchedulerRef
    s = DPMSolverMultistepSchedulerRef()
    s.set_timesteps(4)
    assert s.timesteps.tolist() == [999, 749, 499, 249]
    s.set_timesteps(1)
    assert s.timesteps.tolist() == [999]
    assert float(s.sigmas[-1]) == 0.0
    # last step of dpmsolver++ with final sigma 0 returns the x0 prediction
    order, cs, c0, c1 = s.step_coefficients(0, 0)
    assert order == 1 and abs(cs) < 1e-12 and abs(c0 - 1.0) < 1e-6


def test_oracle_sampler_reproduces_golden():
    """FlashDiffusion.sample (FD:754-915): 4-step LCM student sampler + the teacher's DPM sampler, fixture made by the real
    reference; the LCM re-noising draws are replayed."""
    from oracle.sched_cpu import DPMSolverMultistepSchedulerRef, LCMSchedulerRef
    from tests.golden_util import load_sample_case, sampler_models_from_golden
    g = load_sample_case()
    teacher, student, disc = sampler_models_from_golden(g)
    m = FlashDiffusionRef(FlashConfigRef(K=[4], num_iterations_per_K=[10]), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=DPMSolverMultistepSchedulerRef(), conditioner=TensorConditioner(),
                          discriminator=disc, sampling_noise_scheduler=LCMSchedulerRef(),
                          teacher_sampling_noise_scheduler=DPMSolverMultistepSchedulerRef())
    it = iter(g["noises"])
    m.sampling_noise_scheduler.noise_fn = lambda shape: next(it)
    B = g["z"].shape[0]
    s, sr = m.sample(g["z"], num_steps=int(g["num_steps"]), guidance_scale=float(g["guidance_scale"]),
                     teacher_guidance_scale=float(g["teacher_guidance_scale"]),
                     conditioner_inputs={"crossattn": g["crossattn"], "text": ["a"] * B},
                     uncond_conditioner_inputs={"crossattn": g["uncond_crossattn"], "text": [""] * B},
                     log_teacher_samples=True)
    assert m.sampling_noise_scheduler.timesteps.tolist() == g["lcm_timesteps"].tolist() == [999, 749, 499, 249]
    assert rel_err(s, g["student_sample"]) < 1e-5 and rel_err(sr, g["teacher_sample"]) < 1e-5
    assert rel_err(s, g["teacher_sample"]) > 1e-2   # the LoRA student is a different sampler


def test_lcm_default_schedule():
    from oracle.sched_cpu import LCMSchedulerRef
    s = LCMSchedulerRef()
    s.set_timesteps(4)
    assert s.timesteps.tolist() == [999, 759, 499, 259]   # the well-known 4-step LCM schedule
    s.set_timesteps(timesteps=[999, 749, 499, 249])
    assert s.timesteps.tolist() == [999, 749, 499, 249] and s.num_inference_steps == 4


@pytest.mark.parametrize("name", ["sd3_g_dmd_lsgan", "sd3_d_hinge"])
def test_oracle_sd3_reproduces_golden(name):
    """FlashDiffusionSD3.forward (flow matching): fixtures made by the real reference class"""
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.golden_cases import SD3_CASES, build_sd3_models
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    kw, step, _seed = SD3_CASES[name]
    g = load_case(name)
    teacher, student, disc, pipe, batch = build_sd3_models()
    m = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                             teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=disc, pipeline=pipe)
    m.draws = Draws(g["draws"])
    assert torch.equal(batch["image"], g["z"])
    out = m(batch, step=step)
    assert abs(out["start_timestep"] - g["start_timestep"]) < 1e-4
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-5, k
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-5 * max(1.0, abs(g["loss"][i]))
    out["loss"][step].backward()
    n = 0
    for pn, p in m.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad, g["grads"][pn]) < 1e-4, pn
            n += 1
    assert n == len(g["grads"]) and n > 0


@pytest.mark.parametrize("name", ["dit_tiny", "dit_hd72_masked"])
def test_dit_oracle_reproduces_reference_fixture(name):
    """oracle/dit_cpu.PixartTransformerRef vs the fixture written by the reference's real wrapper class (runs on any host:
    tolerance 1e-5 relative for a different CPU's fp32 summation order)"""
    from oracle.golden_cases import build_dit
    from tests.golden_util import load_case, rel_err
    g = load_case(name)
    _, ora, (x, t, cond), w = build_dit(name)
    assert rel_err(ora(x, t, cond), g["out"]["frozen"]) < 1e-5
    _, ora, (x, t, cond), w = build_dit(name, lora_r=8)
    out = ora(x, t, cond)
    assert rel_err(out, g["out"]["lora"]) < 1e-5
    (out * w).sum().backward()
    n = 0
    for k, p in ora.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad, g["grads"][k.replace(".base_layer.", ".")]) < 1e-4, k
            n += 1
    assert n == len(g["grads"])


@pytest.mark.parametrize("name", ["mmdit_tiny", "mmdit_hd64"])
def test_mmdit_oracle_reproduces_reference_fixture(name):
    from oracle.golden_cases import build_mmdit
    g = load_case(name)
    _, ora, (x, t, cond), w = build_mmdit(name)
    assert rel_err(ora(x, t, cond), g["out"]["frozen"]) < 1e-5
    _, ora, (x, t, cond), w = build_mmdit(name, lora_r=8)
    out = ora(x, t, cond)
    assert rel_err(out, g["out"]["lora"]) < 1e-5
    (out * w).sum().backward()
    n = 0
    for k, p in ora.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad, g["grads"][k.replace(".base_layer.", ".")]) < 1e-4, k
            n += 1
    assert n == len(g["grads"])


def test_oracle_reproduces_adapter_golden():
    """T2I-adapter residuals on every denoiser call (FD:207-218, 264-310, 436-450, 555-567): fixture by the real reference"""
    from oracle.golden_cases import ADAPTER_CASES, make_edge
    from oracle.unet_cpu import TinyT2IAdapter, tiny_config
    (name, (kw, sched, step, _)), = ADAPTER_CASES.items()
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(), discriminator=disc,
                          adapter=TinyT2IAdapter(tiny_config()))
    m.draws = Draws(g["draws"])
    batch = {"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * g["z"].shape[0], "edge": make_edge()}
    out = m(batch, step=step)
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-5, k
    assert abs(float(out["loss"][0]) - g["loss"][0]) <= 1e-5 * max(1.0, abs(g["loss"][0]))
    out["loss"][step].backward()
    n = sum(1 for pn, p in m.named_parameters() if p.grad is not None and rel_err(p.grad, g["grads"][pn]) < 1e-4)
    assert n == len(g["grads"])


def test_oracle_reproduces_lpips_golden():
    """VAE-encoded pixel batch + LPIPS distillation term (FD:128-133, 182-185, 383-397): fixture by the real reference with the
    frozen stand-in VAE / LPIPS networks (the pretrained ones are not available offline)"""
    from oracle.golden_cases import LPIPS_CASES
    from oracle.unet_cpu import TinyLPIPS, TinyVAE
    (name, (kw, sched, step, _)), = LPIPS_CASES.items()
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(), discriminator=disc,
                          vae=TinyVAE(), lpips_model=TinyLPIPS())
    m.draws = Draws(g["draws"])
    assert g["z"].shape == (2, 3, 64, 64)                       # pixels: the latents come out of vae.encode
    out = m({"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * g["z"].shape[0]}, step=step)
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert out[k].shape == (2, 4, 32, 32) and rel_err(out[k], g["out"][k]) < 1e-5, k
    assert abs(float(out["loss"][0]) - g["loss"][0]) <= 1e-5 * max(1.0, abs(g["loss"][0]))
    out["loss"][step].backward()
    n = sum(1 for pn, p in m.named_parameters() if p.grad is not None and rel_err(p.grad, g["grads"][pn]) < 1e-4)
    assert n == len(g["grads"]) and n > 0
