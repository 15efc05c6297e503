"""Host logic of the step orchestration on CPU: the PRODUCT classes flash_diffusion_amd.flash.FlashDiffusion and
flash_sd3.FlashDiffusionSD3 (with the product schedulers) are run with tests/fake_ops.py standing in for the HIP launches and
with the oracle's fp32 denoisers in the denoiser slots (the classes are denoiser-agnostic: anything honouring the wrapper
signature), every random draw injected, and compared with the fixtures the REAL reference produced (tests/golden/*.npz) --
everything is fp32 here, so the bar is 1e-4 relative.  This pins the orchestration (draw order, timestep selection, CFG loop,
scheduler coefficients, loss assembly, which call receives which conditioning / adapter residual, gradient flow to LoRA and
discriminator) without a GPU; the kernels themselves and the HIP denoisers are covered by the -m gpu tests."""
import copy

import pytest
import torch

from oracle.golden_cases import ADAPTER_CASES, CASES, SD3_CASES, build_models, build_sd3_models, make_edge
from tests import fake_ops
from tests.golden_util import load_case, rel_err


class _Head(torch.nn.Module):   # not an nn.Sequential: FlashDiffusion keeps it as a plain torch module
    def __init__(self, seq):
        super().__init__()
        self.seq = seq

    def forward(self, x):
        return self.seq(x)


def _patch(monkeypatch):
    from flash_diffusion_amd import flash, flash_sd3, schedulers
    for mod in (flash, flash_sd3, schedulers):
        monkeypatch.setattr(mod, "ops", fake_ops)
    for mod in (flash, flash_sd3):
        monkeypatch.setattr(mod, "_DistillLoss", fake_ops.FakeDistillLoss)
        monkeypatch.setattr(mod, "_DmdLoss", fake_ops.FakeDmdLoss)


def _check_grads(m, g, rename=lambda k: k):
    n = 0
    for pn, p in m.named_parameters():
        key = rename(pn)
        if p.grad is None:
            assert key not in g["grads"] or float(g["grads"][key].abs().max()) == 0.0, pn
            continue
        assert key in g["grads"], pn
        ref = g["grads"][key]
        if float(ref.norm()) > 1e-12:
            assert rel_err(p.grad, ref) < 2e-3, (pn, rel_err(p.grad, ref))
            n += 1
    assert n == sum(1 for v in g["grads"].values() if float(v.norm()) > 1e-12) and n > 0


@pytest.mark.parametrize("name", list(CASES) + list(ADAPTER_CASES))
def test_flash_step_orchestration_matches_reference_golden(monkeypatch, name):
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DDPMScheduler, DPMSolverMultistepScheduler
    from oracle.unet_cpu import TinyT2IAdapter, tiny_config
    _patch(monkeypatch)
    kw, sched, step, _ = (CASES.get(name) or ADAPTER_CASES[name])
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler() if sched == "dpm" else DDPMScheduler(),
                       conditioner=TensorConditioner(), discriminator=_Head(disc),
                       adapter=TinyT2IAdapter(tiny_config()) if name in ADAPTER_CASES else None)
    m.draws = Draws(g["draws"])
    B = g["z"].shape[0]
    batch = {"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * B}
    if name in ADAPTER_CASES:
        batch["edge"] = make_edge()
    calls = []
    orig = type(teacher).forward
    monkeypatch.setattr(type(teacher), "forward", lambda self, *a, **k: (calls.append(k["sample"].shape[0]), orig(self, *a, **k))[1])
    out = m(batch, step=step, device="cpu")
    assert 2 * B in calls                                   # the teacher's CFG pair really went out as ONE 2B call
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-4, (k, rel_err(out[k], g["out"][k]))
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-4 * max(1.0, abs(g["loss"][i])), i
    named = dict(m.named_parameters())
    for pn, ref in g["post"].items():     # wgan clamps the discriminator weights inside the forward (FD:573-585)
        assert torch.equal(named[pn.replace("discriminator.", "discriminator.seq.")].detach(), ref), pn
    if torch.is_tensor(out["loss"][step]) and out["loss"][step].requires_grad:
        out["loss"][step].backward()
        _check_grads(m, g, lambda k: k.replace("discriminator.seq.", "discriminator."))


@pytest.mark.parametrize("name", list(SD3_CASES))
def test_sd3_step_orchestration_matches_reference_golden(monkeypatch, name):
    from flash_diffusion_amd.flash import Draws
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler
    _patch(monkeypatch)
    kw, step, _ = SD3_CASES[name]
    g = load_case(name)
    teacher, student, disc, pipe, batch = build_sd3_models()
    m = FlashDiffusionSD3(FlashDiffusionSD3Config(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=FlowMatchEulerDiscreteScheduler(), discriminator=_Head(disc), pipeline=pipe)
    m.draws = Draws(g["draws"])
    out = m(batch, step=step)
    assert abs(out["start_timestep"] - g["start_timestep"]) < 1e-3
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-4, (k, rel_err(out[k], g["out"][k]))
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-4 * max(1.0, abs(g["loss"][i])), i
    out["loss"][step].backward()
    _check_grads(m, g, lambda k: k.replace("discriminator.seq.", "discriminator."))
    # pinned start index / guidance (what bench.py and the data-parallel trainer set) and the trainer's hook
    m2 = FlashDiffusionSD3(FlashDiffusionSD3Config(**kw), student_denoiser=student, teacher_denoiser=teacher,
                           teacher_noise_scheduler=FlowMatchEulerDiscreteScheduler(), discriminator=_Head(disc), pipeline=pipe)
    m2.fixed_start_idx, m2.fixed_guidance = 0, 4.0
    seen = []
    m2.before_student = lambda: seen.append(1)
    o2 = m2(batch, step=0)
    assert seen == [1] and m2.terms["guidance"] == 4.0 and abs(o2["start_timestep"] - float(m2.teacher_noise_scheduler.timesteps[0])) < 1e-3


def test_sd3_lpips_step_with_vae_matches_the_pinned_oracle(monkeypatch):
    """FlashDiffusionSD3 with a VAE attached and distill_loss_type="lpips" (FD3:138-144, 190-191, 391-411): the product on the
    stand-in ops against the oracle restatement (itself bit-identical to the real class, tests/test_oracle_vs_reference.py), the
    oracle's draws replayed"""
    from flash_diffusion_amd.flash import Draws
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    from oracle.unet_cpu import TinyLPIPS, TinyVAE
    _patch(monkeypatch)
    kw = dict(SD3_CASES["sd3_g_dmd_lsgan"][0], distill_loss_type="lpips")
    teacher, student, disc, pipe, _ = build_sd3_models()
    gp = torch.Generator().manual_seed(5)
    batch = {"image": torch.randn(2, 3, 32, 32, generator=gp) * 0.5, "text": ["a", "b"]}
    ora = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kw), student_denoiser=copy.deepcopy(student), teacher_denoiser=teacher,
                               teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=copy.deepcopy(disc),
                               pipeline=pipe, vae=TinyVAE(), lpips_model=TinyLPIPS())
    torch.manual_seed(11)
    ref = ora(batch, step=0)
    ref["loss"][0].backward()
    m = FlashDiffusionSD3(FlashDiffusionSD3Config(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=FlowMatchEulerDiscreteScheduler(), discriminator=_Head(disc), pipeline=pipe,
                          vae=TinyVAE(), lpips_model=TinyLPIPS())
    m.draws = Draws(dict(ora.last_draws.values))
    out = m(batch, step=0)
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert out[k].shape == (2, 4, 16, 16) and rel_err(out[k], ref[k]) < 1e-4, (k, rel_err(out[k], ref[k]))
    assert abs(float(out["loss"][0]) - float(ref["loss"][0])) <= 1e-4 * max(1.0, abs(float(ref["loss"][0])))
    out["loss"][0].backward()
    want = {n: p.grad for n, p in ora.named_parameters() if p.grad is not None and float(p.grad.norm()) > 1e-12}
    got = {n.replace("discriminator.seq.", "discriminator."): p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert len(want) > 0 and all(rel_err(got[n], want[n]) < 2e-3 for n in want), [n for n in want if n not in got]
    with pytest.raises(ValueError, match="vae"):
        FlashDiffusionSD3(FlashDiffusionSD3Config(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=FlowMatchEulerDiscreteScheduler(), pipeline=pipe, lpips_model=TinyLPIPS())


def test_lpips_step_with_vae_matches_reference_golden(monkeypatch):
    """distill_loss_type="lpips" with a VAE attached (FD:128-133, 182-185, 383-397): the product encodes the pixel batch, runs the
    step on the latents, decodes the centre crop of both outputs through the caller's VAE and perceptual network (torch modules;
    the frozen stand-ins of the fixture) and sends the gradient back into the student -- fixture by the real reference"""
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import LPIPS_CASES
    from oracle.unet_cpu import TinyLPIPS, TinyVAE
    _patch(monkeypatch)
    (name, (kw, sched, step, _)), = LPIPS_CASES.items()
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(),
                       discriminator=_Head(disc), vae=TinyVAE(), lpips_model=TinyLPIPS())
    m.draws = Draws(g["draws"])
    out = m({"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * g["z"].shape[0]}, step=step, device="cpu")
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-4, (k, rel_err(out[k], g["out"][k]))
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-4 * max(1.0, abs(g["loss"][i])), i
    out["loss"][step].backward()
    _check_grads(m, g, lambda k: k.replace("discriminator.seq.", "discriminator."))
    assert all(p.grad is None for n, p in m.named_parameters() if n.startswith(("vae.", "lpips.")))
    # the constructor's contract
    with pytest.raises(ValueError, match="vae"):
        FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), lpips_model=TinyLPIPS())
    # without lpips_model the reference builds lpips.LPIPS(net="vgg") with its PRETRAINED weights (FD:102-103).  The product copies
    # them into its HIP twin -- and refuses to train against placeholder weights when the package is absent
    import sys
    import types
    if "lpips" not in sys.modules:
        monkeypatch.setitem(sys.modules, "lpips", None)         # (import lpips -> ImportError, whatever the image has)
        with pytest.raises(ImportError, match="lpips"):
            FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                           teacher_noise_scheduler=DPMSolverMultistepScheduler(), vae=TinyVAE())
    from flash_diffusion_amd.nets import MiLPIPS
    donor = MiLPIPS()
    sd = {k: torch.full_like(v, 0.25) for k, v in donor.state_dict().items()}
    sd.update({f"lins.{l}.model.1.weight": sd[f"lin{l}.model.1.weight"] for l in range(5)})   # lpips 0.1.4's ModuleList duplicates

    class _FakeLPIPS:
        def __init__(self, net="vgg"):
            assert net == "vgg"

        def state_dict(self):
            return sd
    monkeypatch.setitem(sys.modules, "lpips", types.SimpleNamespace(LPIPS=_FakeLPIPS))
    m_def = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                           teacher_noise_scheduler=DPMSolverMultistepScheduler(), vae=TinyVAE())
    assert type(m_def.lpips).__name__ == "MiLPIPS" and not any(p.requires_grad for p in m_def.lpips.parameters())
    got = m_def.lpips.state_dict()
    assert "net.slice1.0.weight" in got and "lin4.model.1.weight" in got and not any(k.startswith("lins.") for k in got)
    assert all(torch.equal(got[k], sd[k]) for k in got if not k.startswith("scaling_layer."))   # the donor's weights, not placeholders
    # the sampler decodes, log_samples infers the latent shape from the VAE (FD:865-868, 977-984)
    from flash_diffusion_amd.schedulers import LCMScheduler
    m.sampling_noise_scheduler = LCMScheduler()
    logs = m.log_samples({"image": g["z"], "crossattn": g["crossattn"], "text": ["a", "a"]}, num_steps=2, max_samples=2)
    (k, v), = logs.items()
    assert v.shape == (2, 3, 64, 64) and torch.isfinite(v).all()


def test_sampler_orchestration_matches_reference_golden(monkeypatch):
    """FlashDiffusion.sample (FD:754-915) with the product LCM / DPM-Solver++ schedulers on the stand-in ops: 4-step student
    sampler + the teacher's own sampler, fixture by the real reference (LCM re-noising draws replayed)"""
    from flash_diffusion_amd.flash import FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler, LCMScheduler
    from tests.golden_util import load_sample_case, sampler_models_from_golden
    _patch(monkeypatch)
    g = load_sample_case()
    teacher, student, _ = sampler_models_from_golden(g)
    m = FlashDiffusion(FlashDiffusionConfig(K=[4], num_iterations_per_K=[10]), student_denoiser=student,
                       teacher_denoiser=teacher, teacher_noise_scheduler=DPMSolverMultistepScheduler(),
                       conditioner=TensorConditioner(), discriminator=None, sampling_noise_scheduler=LCMScheduler(),
                       teacher_sampling_noise_scheduler=DPMSolverMultistepScheduler())
    it = iter(g["noises"])
    m.sampling_noise_scheduler.noise_fn = lambda shape: next(it)
    B = g["z"].shape[0]
    s, sr = m.sample(g["z"], num_steps=int(g["num_steps"]), guidance_scale=float(g["guidance_scale"]),
                     teacher_guidance_scale=float(g["teacher_guidance_scale"]),
                     conditioner_inputs={"crossattn": g["crossattn"], "text": ["a"] * B},
                     uncond_conditioner_inputs={"crossattn": g["uncond_crossattn"], "text": [""] * B},
                     log_teacher_samples=True)
    assert m.sampling_noise_scheduler.timesteps.tolist() == g["lcm_timesteps"].tolist()
    assert rel_err(s, g["student_sample"]) < 1e-4 and rel_err(sr, g["teacher_sample"]) < 1e-4


def test_teacher_loop_route_is_taken_only_when_asked(monkeypatch):
    """FDMI_TEACHER_LOOP=1 hands the whole CFG loop to the denoiser's `teacher_loop` (fdmi_teacher_loop) with the scheduler's
    coefficient table; here a recording stand-in that replays the table on the oracle denoiser checks the hand-over (timesteps,
    [cond | uncond] context, coefficients) gives the step-by-step result"""
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    _patch(monkeypatch)
    name = "g_noreg_vanilla"
    kw, sched, step, _ = CASES[name]
    g = load_case(name)
    outs = []
    for route in ("0", "1"):
        monkeypatch.setenv("FDMI_TEACHER_LOOP", route)
        teacher, student, disc = build_models()
        used = []

        def teacher_loop(x, timesteps, ctx2, vec2, coeffs, _t=teacher, _used=used):
            _used.append((list(timesteps), tuple(ctx2.shape), len(coeffs)))
            B = x.shape[0]
            prev = None
            for t, a in zip(timesteps, coeffs):
                e = _t(sample=torch.cat([x, x]), timestep=torch.full((2 * B,), t), conditioning={"cond": {"crossattn": ctx2}})
                e_c, e_u = e.chunk(2)
                x0 = a[0] * x + a[1] * e_c + a[2] * e_u
                x = a[3] * x + a[4] * x0 + (a[5] * prev if prev is not None else 0.0)
                prev = x0
            return x
        teacher.teacher_loop = teacher_loop
        m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                           teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(),
                           discriminator=_Head(disc))
        m.draws = Draws(g["draws"])
        B = g["z"].shape[0]
        outs.append((m({"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * B}, step=step, device="cpu"), used))
    (o0, u0), (o1, u1) = outs
    assert u0 == [] and len(u1) == 1 and u1[0][1][0] == 2 * g["z"].shape[0] and u1[0][2] == len(u1[0][0])
    assert rel_err(o1["teacher_output"], o0["teacher_output"]) < 1e-5
    assert rel_err(o1["teacher_output"], g["out"]["teacher_output"]) < 1e-4


@pytest.mark.parametrize("name", ["pixart_g_dmd_lsgan", "pixart_d_lsgan"])
def test_pixart_step_orchestration_matches_reference_golden(monkeypatch, name):
    """BASELINE C4's workload in miniature on the host: the PRODUCT FlashDiffusion over the oracle's PixArt DiT (key mask,
    vector conditioning, `use_empty_prompt` unconditional embeddings, the six-conv-recipe head on the epsilon prediction --
    the DiT ignores return_intermediate) against the fixture of the REAL reference class over its REAL DiT wrapper"""
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import PIXART_STEP_CASES, PromptTableConditioner, build_pixart_step_inputs
    _patch(monkeypatch)
    kw, step, _ = PIXART_STEP_CASES[name]
    g = load_case(name)
    cfg, teacher, student, head, batch = build_pixart_step_inputs()
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=PromptTableConditioner(),
                       discriminator=_Head(head))
    m.draws = Draws(g["draws"])
    calls = []
    orig = type(teacher).forward
    monkeypatch.setattr(type(teacher), "forward",
                        lambda self, *a, **k: (calls.append((k["sample"] if "sample" in k else a[0]).shape[0]), orig(self, *a, **k))[1])
    out = m(batch, step=step, device="cpu")
    assert 4 in calls                                       # [cond | uncond] (with their key masks) as ONE 2B call
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-4, (k, rel_err(out[k], g["out"][k]))
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-4 * max(1.0, abs(g["loss"][i])), i
    out["loss"][step].backward()
    _check_grads(m, g, lambda k: k.replace("discriminator.seq.", "discriminator.").replace(".base_layer.", "."))
