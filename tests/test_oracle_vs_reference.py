"""Pins oracle/flash_ref.py against the reference's OWN FlashDiffusion (imported unmodified
from /root/reference/src via oracle/shim_import.py).  Skipped where the reference is absent
(the GPU box); there tests/golden/*.npz (made by the real reference) carry the pin."""
import copy

import pytest
import torch

from oracle import shim_import
from oracle.flash_ref import FlashConfigRef, FlashDiffusionRef, TensorConditioner
from oracle.sched_cpu import DDPMSchedulerRef, DPMSolverMultistepSchedulerRef
from oracle.unet_cpu import UNet2DConditionRef, make_discriminator, seeded_init_, tiny_config

pytestmark = pytest.mark.skipif(not shim_import.reference_available(), reason="reference absent")


def _build(cls, cfg_cls, sched_cls, **cfg_kw):
    torch.manual_seed(0)
    teacher = seeded_init_(UNet2DConditionRef(tiny_config()), 1)
    student = copy.deepcopy(teacher)
    student.add_adapter(8)
    seeded_init_(student, 2)
    student.load_state_dict({k: v for k, v in teacher.state_dict().items()}, strict=False)
    teacher.freeze()
    disc = seeded_init_(make_discriminator("sd15", color_dim=64, feat=16, last_k=2), 3)
    cfg = cfg_cls(**cfg_kw)
    m = cls(cfg, student_denoiser=student, teacher_denoiser=teacher,
            teacher_noise_scheduler=sched_cls(), conditioner=TensorConditioner(), discriminator=disc)
    return m


def _batch():
    g = torch.Generator().manual_seed(5)
    return {"image": torch.randn(2, 4, 32, 32, generator=g),
            "crossattn": torch.randn(2, 77, 64, generator=g), "text": ["a", "b"]}


CASES = [
    dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="l2",
         gan_loss_type="lsgan", use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=13.0),
    dict(K=[8], num_iterations_per_K=[10], timestep_distribution="mixture", distill_loss_type="l1",
         gan_loss_type="hinge", use_dmd_loss=False, mixture_num_components=4, mixture_var=0.5,
         mode_probs=[[0.1, 0.3, 0.3, 0.3]]),
    dict(K=[6], num_iterations_per_K=[10], timestep_distribution="gaussian", distill_loss_type="l2",
         gan_loss_type="non-saturating", use_dmd_loss=True, use_teacher_as_real=True),
    dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="wgan"),
    dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="vanilla"),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("step", [0, 1])
@pytest.mark.parametrize("sched", [DPMSolverMultistepSchedulerRef, DDPMSchedulerRef])
def test_restatement_is_bit_identical(case, step, sched):
    FD, FDC = shim_import.import_reference()
    kw = CASES[case]
    ref = _build(FD, FDC, sched, **kw)
    ora = _build(FlashDiffusionRef, FlashConfigRef, sched, **kw)
    outs = []
    for m in (ref, ora):
        torch.manual_seed(1234 + case)
        out = m(_batch(), step=step, device="cpu")
        loss = out["loss"][step]
        loss.backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        outs.append((out, grads))
    (o1, g1), (o2, g2) = outs
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert torch.equal(o1[k], o2[k]), k
    assert o1["start_timestep"] == o2["start_timestep"]
    for i in (0, 1):
        a, b = o1["loss"][i], o2["loss"][i]
        assert float(a) == float(b)
    assert set(g1) == set(g2) and len(g1) > 0
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


# ---- FlashDiffusion.sample / log_samples (FD:754-1019): the few-step sampler ("next" row 1 of SURVEY 8f) ----
def _build_sampler(cls, cfg_cls):
    from oracle.sched_cpu import LCMSchedulerRef
    m = _build(cls, cfg_cls, DPMSolverMultistepSchedulerRef, K=[4], num_iterations_per_K=[10])
    m.sampling_noise_scheduler = LCMSchedulerRef()
    m.teacher_sampling_noise_scheduler = DPMSolverMultistepSchedulerRef()
    return m


@pytest.mark.parametrize("kw", [dict(num_steps=4, guidance_scale=1.0), dict(num_steps=3, guidance_scale=1.7, max_samples=1),
                                dict(num_steps=4, guidance_scale=1.3, log_teacher_samples=True, teacher_guidance_scale=5.0,
                                     with_uncond=True)])
def test_sample_restatement_is_bit_identical(kw):
    FD, FDC = shim_import.import_reference()
    kw = dict(kw)
    with_uncond = kw.pop("with_uncond", False)
    outs = []
    for cls, ccls in ((FD, FDC), (FlashDiffusionRef, FlashConfigRef)):
        m = _build_sampler(cls, ccls)
        b = _batch()
        g = torch.Generator().manual_seed(11)
        z = torch.randn(2, 4, 32, 32, generator=g)
        un = {"crossattn": torch.randn(2, 77, 64, generator=g), "text": ["", ""]} if with_uncond else None
        torch.manual_seed(7)
        outs.append(m.sample(z, conditioner_inputs=b, uncond_conditioner_inputs=un, **kw))
    (a, ar), (o, orf) = outs
    assert torch.equal(a, o)
    assert (ar is None) == (orf is None)
    if ar is not None:
        assert torch.equal(ar, orf)


def test_log_samples_restatement_is_bit_identical():
    FD, FDC = shim_import.import_reference()
    logs = []
    for cls, ccls in ((FD, FDC), (FlashDiffusionRef, FlashConfigRef)):
        m = _build_sampler(cls, ccls)
        torch.manual_seed(3)
        logs.append(m.log_samples(_batch(), input_shape=(4, 32, 32), guidance_scale=1.5, max_samples=8, num_steps=[2, 4],
                                  log_teacher_samples=True))
    assert list(logs[0].keys()) == list(logs[1].keys()) and len(logs[0]) == 4
    for k in logs[0]:
        assert torch.equal(logs[0][k], logs[1][k]), k


# ---- FlashDiffusionSD3.forward (flow matching): oracle/flash_sd3_ref.py vs the real class ----
def _build_sd3(cls, cfg_cls, with_disc=True, extra=None, **cfg_kw):
    from oracle.flash_sd3_ref import EmbeddingPipeline, TinyFlowDenoiser
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    teacher = TinyFlowDenoiser(seed=1)
    student = copy.deepcopy(teacher)
    g = torch.Generator().manual_seed(2)
    for p in student.parameters():
        p.data.add_(torch.randn(p.shape, generator=g) * 0.02)
    teacher.freeze()
    disc = None
    if with_disc:
        disc = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 4, 2, 1), torch.nn.SiLU(), torch.nn.Conv2d(8, 1, 8, 1, 0),
                                   torch.nn.Flatten())   # [B,4,16,16] -> [B,1]
        g3 = torch.Generator().manual_seed(3)
        for p in disc.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g3) * 0.1)
    ge = torch.Generator().manual_seed(9)
    pipe = EmbeddingPipeline(torch.randn(2, 5, 10, generator=ge), torch.randn(2, 12, generator=ge),
                             torch.randn(2, 5, 10, generator=ge), torch.randn(2, 12, generator=ge))
    m = cls(cfg_cls(**cfg_kw), student_denoiser=student, teacher_denoiser=teacher,
            teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), discriminator=disc, pipeline=pipe, **(extra or {}))
    return m


SD3_CASES = [
    dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="lsgan", use_dmd_loss=True),
    dict(K=[8], num_iterations_per_K=[10], timestep_distribution="mixture", gan_loss_type="hinge", distill_loss_type="l2",
         mixture_num_components=4, mixture_var=0.5, mode_probs=[[0.1, 0.3, 0.3, 0.3]]),
    dict(K=[6], num_iterations_per_K=[10], timestep_distribution="gaussian", gan_loss_type="non-saturating",
         use_dmd_loss=True, use_teacher_as_real=True),
    dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="wgan"),
    dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="vanilla"),
]


@pytest.mark.parametrize("case", range(len(SD3_CASES)))
@pytest.mark.parametrize("step", [0, 1])
def test_sd3_restatement_is_bit_identical(case, step):
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    FD3, FD3C = shim_import.import_reference_sd3()
    kw = SD3_CASES[case]
    outs = []
    for cls, ccls in ((FD3, FD3C), (FlashDiffusionSD3Ref, FlashSD3ConfigRef)):
        for seed in (0, 1, 2):   # several start indices (incl. start_idx == 0: the pure-noise branch FD3:264-268)
            m = _build_sd3(cls, ccls, **kw)
            g = torch.Generator().manual_seed(5)
            batch = {"image": torch.randn(2, 4, 16, 16, generator=g), "text": ["a", "b"]}
            torch.manual_seed(100 + seed)
            out = m(batch, step=step)
            loss = out["loss"][step]
            grads = None
            if torch.is_tensor(loss) and loss.requires_grad:
                loss.backward()
                grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
            outs.append((cls.__name__, seed, out, grads))
    ref, ora = outs[:3], outs[3:]
    for (_, seed, a, ga), (_, _, b, gb) in zip(ref, ora):
        assert a["start_timestep"] == b["start_timestep"]
        for k in ("teacher_output", "student_output", "noisy_sample"):
            assert torch.equal(a[k], b[k]), (seed, k)
        for i in (0, 1):
            la, lb = a["loss"][i], b["loss"][i]
            assert (torch.is_tensor(la) == torch.is_tensor(lb)) and float(la) == float(lb), (seed, i)
        assert (ga is None) == (gb is None)
        if ga is not None:
            assert set(ga) == set(gb) and len(ga) > 0
            for n in ga:
                assert torch.equal(ga[n], gb[n]), (seed, n)


def test_sd3_without_discriminator_returns_scalar_loss():
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    FD3, FD3C = shim_import.import_reference_sd3()
    vals = []
    for cls, ccls in ((FD3, FD3C), (FlashDiffusionSD3Ref, FlashSD3ConfigRef)):
        m = _build_sd3(cls, ccls, with_disc=False, K=[4], num_iterations_per_K=[10], timestep_distribution="uniform",
                       use_dmd_loss=True)
        g = torch.Generator().manual_seed(5)
        torch.manual_seed(7)
        out = m({"image": torch.randn(2, 4, 16, 16, generator=g), "text": ["a", "b"]})
        assert torch.is_tensor(out["loss"]) and out["loss"].dim() == 0     # FD3:357-364: a scalar, not a list
        vals.append(out)
    assert float(vals[0]["loss"]) == float(vals[1]["loss"]) and torch.equal(vals[0]["student_output"], vals[1]["student_output"])


@pytest.mark.parametrize("kw", [dict(num_steps=4, guidance_scale=1.0), dict(num_steps=3, guidance_scale=2.5, max_samples=1),
                                dict(num_steps=4, guidance_scale=1.5, log_teacher_samples=True, teacher_guidance_scale=4.0)])
def test_sd3_sample_restatement_is_bit_identical(kw):
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    FD3, FD3C = shim_import.import_reference_sd3()
    outs = []
    for cls, ccls in ((FD3, FD3C), (FlashDiffusionSD3Ref, FlashSD3ConfigRef)):
        m = _build_sd3(cls, ccls, with_disc=False, K=[4], num_iterations_per_K=[10], timestep_distribution="uniform")
        m.sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
        m.teacher_sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
        g = torch.Generator().manual_seed(13)
        z = torch.randn(2, 4, 16, 16, generator=g)
        outs.append(m.sample(z, conditioner_inputs={"text": ["a", "b"]}, **kw))
    (a, ar), (o, orf) = outs
    assert torch.equal(a, o) and (ar is None) == (orf is None)
    if ar is not None:
        assert torch.equal(ar, orf) and not torch.equal(ar, a)


def test_sd3_product_asks_the_pipeline_what_the_reference_asks(monkeypatch):
    """FD3:203-217 (forward) and 722-736 (sample): the PRODUCT's ``pipeline.encode_prompt`` calls carry exactly the reference's
    keyword arguments -- the fixed negative prompts and clip_skip=False included (an unconditional branch fed with "" instead
    would change every teacher CFG target).  The stub pipeline records what it is asked."""
    from flash_diffusion_amd import flash, flash_sd3, schedulers
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    from tests import fake_ops
    for mod in (flash, flash_sd3, schedulers):
        monkeypatch.setattr(mod, "ops", fake_ops)
    for mod in (flash, flash_sd3):
        monkeypatch.setattr(mod, "_DistillLoss", fake_ops.FakeDistillLoss)
        monkeypatch.setattr(mod, "_DmdLoss", fake_ops.FakeDmdLoss)
    FD3, FD3C = shim_import.import_reference_sd3()
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform")
    real = _build_sd3(FD3, FD3C, with_disc=False, **kw)
    real.sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
    mine = _build_sd3(FlashDiffusionSD3, FlashDiffusionSD3Config, with_disc=False, **kw)
    mine.teacher_noise_scheduler = FlowMatchEulerDiscreteScheduler()
    mine.sampling_noise_scheduler = FlowMatchEulerDiscreteScheduler()
    gb = torch.Generator().manual_seed(5)
    batch = {"image": torch.randn(2, 4, 16, 16, generator=gb), "text": ["a", "b"]}
    z = torch.randn(2, 4, 16, 16, generator=gb)
    for m in (real, mine):
        torch.manual_seed(0)
        m(batch, step=0)
        m.sample(z, num_steps=2, conditioner_inputs={"text": ["a", "b"]})
    norm = lambda calls: [{k: (str(v) if k == "device" else v) for k, v in c.items()} for c in calls]
    a, b = norm(real.pipeline.calls), norm(mine.pipeline.calls)
    assert len(a) == len(b) == 2 and a == b
    assert a[0]["negative_prompt"].startswith("deformed, distorted") and a[0]["clip_skip"] is False and a[0]["_args"] == ()


# ---- PixArt DiT wrapper: the reference's REAL wrapper + AdaLayerNormSingle on the restated base ----------
@pytest.mark.parametrize("masked,concat_vec", [(False, True), (True, True), (False, False)])
def test_dit_wrapper_restatement_is_bit_identical(masked, concat_vec):
    from oracle import dit_cpu
    Wrapper, AdaLN = shim_import.import_reference_dit()
    cfg = dict(dit_cpu.TINY_DIT)
    if not concat_vec:
        cfg.update(use_concat_vector_conditioning=False, num_vector_conditionings=None, projection_class_embeddings_input_dim=32)
    real, mine = Wrapper(**cfg), dit_cpu.PixartTransformerRef(**cfg)
    assert isinstance(real.adaln_single, AdaLN) and issubclass(Wrapper, dit_cpu.Transformer2DModelRef)
    assert {k: v.shape for k, v in real.state_dict().items()} == {k: v.shape for k, v in mine.state_dict().items()}
    dit_cpu.seeded_init_(real, 3)
    mine.load_state_dict(real.state_dict())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 16, 16, generator=g)
    t = torch.tensor([999.0, 250.0])
    cond = {"crossattn": torch.randn(2, 7, 48, generator=g), "vector": torch.randn(2, 32, generator=g)}
    if masked:
        cond["attention_mask"] = torch.tensor([[1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1]])
    a, b = real(x, t, {"cond": cond}), mine(x, t, {"cond": cond})
    assert a.shape == (2, 4, 16, 16) and torch.equal(a, b)
    real.freeze()
    assert not any(p.requires_grad for p in real.parameters()) and not real.training


# ---- SD3 MMDiT wrapper: the reference's REAL wrapper on the restated SD3Transformer2DModel --------------
@pytest.mark.parametrize("name", ["mmdit_tiny", "mmdit_hd64"])
def test_sd3_wrapper_restatement_is_bit_identical(name):
    from oracle import dit_cpu, mmdit_cpu
    from oracle.golden_cases import MMDIT_CASES, build_mmdit
    Wrapper = shim_import.import_reference_sd3_wrapper()
    cfg, mine, (x, t, cond), _ = build_mmdit(name)
    real = Wrapper(**cfg)
    assert issubclass(Wrapper, mmdit_cpu.SD3Transformer2DModelRef)
    real.load_state_dict(mine.state_dict(), strict=True)
    a, b = real(x, t, cond), mine(x, t, cond)
    assert a.shape == x.shape and torch.equal(a, b)
    # concat conditioning (TW:141-142): channels appended to the sample, output sliced back (TW:154)
    cond2 = {"cond": dict(cond["cond"], concat=x[:, 8:])}
    assert torch.equal(real(x[:, :8], t, cond2), mine(x[:, :8], t, cond2)) and real(x[:, :8], t, cond2).shape[1] == 8
    real.freeze()
    assert not any(p.requires_grad for p in real.parameters()) and not real.training


def test_sd3_log_samples_restatement_is_bit_identical():
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    FD3, FD3C = shim_import.import_reference_sd3()
    logs = []
    for cls, ccls in ((FD3, FD3C), (FlashDiffusionSD3Ref, FlashSD3ConfigRef)):
        m = _build_sd3(cls, ccls, with_disc=False, K=[4], num_iterations_per_K=[10], timestep_distribution="uniform")
        m.sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
        m.teacher_sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
        torch.manual_seed(21)
        logs.append(m.log_samples({"text": ["a", "b"]}, input_shape=(4, 16, 16), guidance_scale=1.5, max_samples=8,
                                  num_steps=[2, 3], log_teacher_samples=True))
        with pytest.raises(ValueError):
            m.log_samples({"text": ["a", "b"]})
    a, b = logs
    assert list(a) == list(b) and len(a) == 4
    for k in a:
        assert a[k].shape[0] == 2 and torch.equal(a[k], b[k]), k


# ---- T2I-adapter residuals (FD:207-218, 555-560, 820-829) ------------------------------------------------
@pytest.mark.parametrize("step", [0, 1])
def test_adapter_residuals_restatement_is_bit_identical(step):
    from oracle.unet_cpu import TinyT2IAdapter
    FD, FDC = shim_import.import_reference()
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="l2", gan_loss_type="lsgan",
              use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=13.0, adapter_input_key="edge",
              adapter_conditioning_scale=0.7)
    outs = []
    for cls, ccls in ((FD, FDC), (FlashDiffusionRef, FlashConfigRef)):
        m = _build(cls, ccls, DPMSolverMultistepSchedulerRef, **kw)
        m.adapter = TinyT2IAdapter(tiny_config())
        batch = _batch()
        batch["edge"] = torch.randn(2, 1, 32, 32, generator=torch.Generator().manual_seed(8))
        torch.manual_seed(77)
        out = m(batch, step=step, device="cpu")
        out["loss"][step].backward()
        outs.append((out, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    (o1, g1), (o2, g2) = outs
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert torch.equal(o1[k], o2[k]), k
    assert float(o1["loss"][0]) == float(o2["loss"][0]) and float(o1["loss"][1]) == float(o2["loss"][1])
    assert set(g1) == set(g2) and len(g1) > 0 and all(torch.equal(g1[n], g2[n]) for n in g1)
    # the residuals really change the result
    m = _build(FlashDiffusionRef, FlashConfigRef, DPMSolverMultistepSchedulerRef, **kw)
    torch.manual_seed(77)
    plain = m(_batch(), step=step, device="cpu")
    assert not torch.equal(plain["teacher_output"], o2["teacher_output"])


def test_adapter_residuals_in_the_sampler_are_bit_identical():
    from oracle.sched_cpu import LCMSchedulerRef
    from oracle.unet_cpu import TinyT2IAdapter
    FD, FDC = shim_import.import_reference()
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", adapter_input_key="edge")
    outs = []
    for cls, ccls in ((FD, FDC), (FlashDiffusionRef, FlashConfigRef)):
        m = _build(cls, ccls, DPMSolverMultistepSchedulerRef, **kw)
        m.adapter = TinyT2IAdapter(tiny_config())
        m.sampling_noise_scheduler = LCMSchedulerRef()
        m.teacher_sampling_noise_scheduler = DPMSolverMultistepSchedulerRef()
        ci = {k: v for k, v in _batch().items() if k != "image"}
        ci["edge"] = torch.randn(2, 1, 32, 32, generator=torch.Generator().manual_seed(8))
        z = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(9))
        torch.manual_seed(3)
        outs.append(m.sample(z, num_steps=3, guidance_scale=1.5, conditioner_inputs=ci, log_teacher_samples=True,
                             adapter_conditioning_scale=0.5))
    (a, ar), (b, br) = outs
    assert torch.equal(a, b) and torch.equal(ar, br)


# ---- VAE in the loop + LPIPS distillation loss (FD:128-133, 182-185, 383-397, 865-868, 910-913, 977-984) -------
def _build_lpips(cls, cfg_cls, **kw):
    """the pretrained AutoencoderKL / lpips.LPIPS are absent offline: both classes get the same frozen stand-ins
    (oracle.unet_cpu.TinyVAE / TinyLPIPS); the reference builds its own `self.lpips` (FD:102-103: the shim's stub), replaced here"""
    from oracle.unet_cpu import TinyLPIPS, TinyVAE
    torch.manual_seed(0)
    teacher = seeded_init_(UNet2DConditionRef(tiny_config()), 1)
    student = copy.deepcopy(teacher)
    student.add_adapter(8)
    seeded_init_(student, 2)
    student.load_state_dict(dict(teacher.state_dict()), strict=False)
    teacher.freeze()
    disc = seeded_init_(make_discriminator("sd15", color_dim=64, feat=16, last_k=2), 3)
    common = dict(student_denoiser=student, teacher_denoiser=teacher, teacher_noise_scheduler=DPMSolverMultistepSchedulerRef(),
                  conditioner=TensorConditioner(), discriminator=disc, vae=TinyVAE())
    if cls is FlashDiffusionRef:
        return cls(cfg_cls(**kw), lpips_model=TinyLPIPS(), **common)
    m = cls(cfg_cls(**kw), **common)
    if kw.get("distill_loss_type") == "lpips":
        m.lpips = TinyLPIPS()
    return m


def _pixel_batch(px):
    g = torch.Generator().manual_seed(5)
    return {"image": torch.randn(2, 3, px, px, generator=g) * 0.5, "crossattn": torch.randn(2, 77, 64, generator=g),
            "text": ["a", "b"]}


# latents 32x32 (the slice's negative start selects the trailing 16 rows / columns) on both steps; 72x72 (a real centre crop) once
@pytest.mark.parametrize("step,px", [(0, 64), (1, 64), (0, 144)])
def test_lpips_distill_with_vae_restatement_is_bit_identical(step, px):
    FD, FDC = shim_import.import_reference()
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="lpips", gan_loss_type="lsgan",
              use_dmd_loss=px == 64, guidance_scale_min=3.0, guidance_scale_max=13.0)
    outs = []
    for cls, ccls in ((FD, FDC), (FlashDiffusionRef, FlashConfigRef)):
        m = _build_lpips(cls, ccls, **kw)
        torch.manual_seed(77)
        out = m(_pixel_batch(px), step=step, device="cpu")
        out["loss"][step].backward()
        outs.append((out, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    (o1, g1), (o2, g2) = outs
    assert o2["student_output"].shape[-1] == px // 2
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert torch.equal(o1[k], o2[k]), k
    assert float(o1["loss"][0]) == float(o2["loss"][0]) and float(o1["loss"][1]) == float(o2["loss"][1])
    assert set(g1) == set(g2) and len(g1) > 0 and all(torch.equal(g1[n], g2[n]) for n in g1)
    assert not any(n.startswith(("vae.", "lpips.")) for n in g2)            # both networks stay frozen
    if step == 0:
        assert float(o2["loss"][0]) > 0 and any(float(g.abs().max()) > 0 for n, g in g2.items() if "lora" in n)


def test_sampler_decodes_through_the_vae_bit_identically():
    from oracle.sched_cpu import LCMSchedulerRef
    FD, FDC = shim_import.import_reference()
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform")
    outs, logs = [], []
    for cls, ccls in ((FD, FDC), (FlashDiffusionRef, FlashConfigRef)):
        m = _build_lpips(cls, ccls, **kw)
        m.sampling_noise_scheduler = LCMSchedulerRef()
        m.teacher_sampling_noise_scheduler = DPMSolverMultistepSchedulerRef()
        ci = {k: v for k, v in _pixel_batch(64).items() if k != "image"}
        z = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(9))
        torch.manual_seed(3)
        outs.append(m.sample(z, num_steps=3, guidance_scale=1.5, conditioner_inputs=ci, log_teacher_samples=True))
        torch.manual_seed(4)
        logs.append(m.log_samples(_pixel_batch(64), num_steps=2, max_samples=2))      # latent shape inferred from the VAE
    (a, ar), (b, br) = outs
    assert a.shape == (2, 3, 64, 64) and torch.equal(a, b) and torch.equal(ar, br)
    assert list(logs[0]) == list(logs[1]) and all(torch.equal(logs[0][k], logs[1][k]) for k in logs[0])


@pytest.mark.parametrize("step,px", [(0, 32), (1, 32), (0, 144)])   # latents 16x16 (crop clamps to the whole map) and 72x72 (centre crop)
def test_sd3_lpips_distill_with_vae_restatement_is_bit_identical(step, px):
    """FlashDiffusionSD3 with a VAE attached and distill_loss_type="lpips" (FD3:138-144, 190-191, 391-411), same stand-ins"""
    from oracle.flash_sd3_ref import FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.unet_cpu import TinyLPIPS, TinyVAE
    FD3, FD3C = shim_import.import_reference_sd3()
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="lsgan", use_dmd_loss=True,
              distill_loss_type="lpips")
    outs = []
    for cls, ccls in ((FD3, FD3C), (FlashDiffusionSD3Ref, FlashSD3ConfigRef)):
        extra = dict(vae=TinyVAE(), lpips_model=TinyLPIPS()) if cls is FlashDiffusionSD3Ref else dict(vae=TinyVAE())
        m = _build_sd3(cls, ccls, with_disc=px == 32, extra=extra, **kw)      # (the test head's geometry fits 16x16 latents only)
        m.lpips = TinyLPIPS()        # (the real class built the shim's lpips.LPIPS stub, FD3:130-131)
        g = torch.Generator().manual_seed(5)
        batch = {"image": torch.randn(2, 3, px, px, generator=g) * 0.5, "text": ["a", "b"]}
        torch.manual_seed(101)
        out = m(batch, step=step)
        loss = out["loss"][step] if isinstance(out["loss"], (list, tuple)) else out["loss"]
        loss.backward()
        outs.append((out, float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    (o1, l1, g1), (o2, l2, g2) = outs
    assert o2["student_output"].shape[-1] == px // 2 and l1 == l2
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert torch.equal(o1[k], o2[k]), k
    assert set(g1) == set(g2) and len(g1) > 0 and all(torch.equal(g1[n], g2[n]) for n in g1)
    # and the samplers decode (FD3:794-797, 838-841), log_samples infers the latent shape (FD3:904-912)
    if step == 0 and px == 32:
        from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
        res = []
        for cls, ccls in ((FD3, FD3C), (FlashDiffusionSD3Ref, FlashSD3ConfigRef)):
            extra = dict(vae=TinyVAE(), lpips_model=TinyLPIPS()) if cls is FlashDiffusionSD3Ref else dict(vae=TinyVAE())
            m = _build_sd3(cls, ccls, extra=extra, **kw)
            m.lpips = TinyLPIPS()
            m.sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
            m.teacher_sampling_noise_scheduler = FlowMatchEulerDiscreteSchedulerRef()
            torch.manual_seed(7)
            res.append(m.log_samples({"image": torch.zeros(2, 3, 32, 32), "text": ["a", "b"]}, num_steps=2, max_samples=2,
                                     log_teacher_samples=True))
        assert list(res[0]) == list(res[1]) and len(res[0]) == 2
        for k in res[0]:
            assert res[0][k].shape == (2, 3, 32, 32) and torch.equal(res[0][k], res[1][k]), k


def test_tiled_decode_restatement_matches_the_reference_tiler():
    """oracle/tiler_ref.py (the checker of the HIP VAE's tiled decode) against the reference's OWN Tiler / pad
    (/root/reference/src/flash/models/utils.py, imported unmodified) driven as autoencoderKL.py:86-123 drives them: bit-identical,
    including the trailing partial tiles and the asymmetric gaussian midpoints"""
    shim_import.import_reference()
    from flash.models.utils import Tiler, pad
    from oracle.tiler_ref import tiled_decode_ref
    torch.manual_seed(0)
    mix = torch.randn(3, 4)

    def decode(t):      # a toy decoder: x4 nearest upsample, channel mix, a position-dependent term
        u = torch.nn.functional.interpolate(t, scale_factor=4, mode="nearest")
        o = torch.einsum("oc,bchw->bohw", mix, u)
        return o + 0.01 * torch.arange(o.shape[-1])[None, None, None, :]

    def reference(z, ts, ov, scale):
        samples = []
        for i in range(z.shape[0]):
            tiler = Tiler()
            tiles = tiler.get_tiles(input=z[i].unsqueeze(0), tile_size=ts, overlap_size=ov, scale=scale, out_channels=3)
            for a, row in enumerate(tiles):
                for b, tile in enumerate(row):
                    shp = tile.shape
                    d = decode(pad(tile, base_h=ts[0], base_w=ts[1]))
                    tiles[a][b] = d[0, :, :int(shp[2] * scale), :int(shp[3] * scale)].cpu().unsqueeze(0)
            samples.append(tiler.merge_tiles(tiles=tiles))
        return torch.cat(samples, 0)
    for (H, W, ts, ov) in [(40, 52, (16, 16), (4, 4)), (32, 32, (16, 16), (4, 4)), (20, 16, (16, 16), (6, 2)), (33, 47, (16, 24), (5, 7))]:
        z = torch.randn(2, 4, H, W)
        assert torch.equal(reference(z, ts, ov, 4), tiled_decode_ref(z, decode, ts, ov, 4)), (H, W, ts, ov)
