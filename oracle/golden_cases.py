"""TEST INFRASTRUCTURE (oracle) -- shared definition of the golden-fixture cases.

Used by oracle/make_golden.py (runs the REAL reference FlashDiffusion, build container only)
and by the tests that replay the fixtures (oracle restatement on CPU; HIP path on the GPU)."""
from __future__ import annotations

import copy

import numpy as np
import torch

from .sched_cpu import DDPMSchedulerRef, DPMSolverMultistepSchedulerRef
from .unet_cpu import UNet2DConditionRef, make_discriminator, seeded_init_, tiny_config

LORA_RANK = 8
SCHEDS = {"dpm": DPMSolverMultistepSchedulerRef, "ddpm": DDPMSchedulerRef}

CASES = {
    # name: (config kwargs, scheduler, step, seed)
    "g_dmd_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform",
                         distill_loss_type="l2", gan_loss_type="lsgan", use_dmd_loss=True,
                         guidance_scale_min=3.0, guidance_scale_max=13.0,
                         dmd_loss_scale=0.3, adversarial_loss_scale=0.1), "dpm", 0, 11),
    "d_hinge": (dict(K=[8], num_iterations_per_K=[10], timestep_distribution="mixture",
                     distill_loss_type="l1", gan_loss_type="hinge", use_dmd_loss=False,
                     mixture_num_components=4, mixture_var=0.5,
                     mode_probs=[[0.1, 0.3, 0.3, 0.3]]), "dpm", 1, 12),
    "g_nonsat_teacher_real": (dict(K=[6], num_iterations_per_K=[10], timestep_distribution="gaussian",
                                   distill_loss_type="l2", gan_loss_type="non-saturating",
                                   use_dmd_loss=True, use_teacher_as_real=True), "dpm", 0, 13),
    "g_noreg_vanilla": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform",
                             distill_loss_scale=0.0, gan_loss_type="vanilla"), "dpm", 0, 14),
    # the remaining GAN branches of FD:573-662 -- wgan (its weight clamp +-0.01 mutates the discriminator IN
    # the forward, FD:573-585; the fixtures also carry the post-forward discriminator weights) on both steps, and the
    # discriminator step of lsgan / vanilla / non-saturating
    "g_wgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="wgan",
                    adversarial_loss_scale=0.2), "dpm", 0, 17),
    "d_wgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="wgan"), "dpm", 1, 18),
    "d_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="gaussian", gan_loss_type="lsgan"), "dpm", 1, 19),
    "d_vanilla": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="vanilla",
                       use_teacher_as_real=True), "dpm", 1, 20),
    "d_nonsat": (dict(K=[6], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="non-saturating"),
                 "dpm", 1, 21),
}


# T2I-adapter residuals threaded through every denoiser call (FD:207-218, 264-310, 436-450, 555-567): (config, sched, step, seed)
ADAPTER_CASES = {
    "g_adapter_dmd_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="l2",
                                 gan_loss_type="lsgan", use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=13.0,
                                 dmd_loss_scale=0.3, adversarial_loss_scale=0.1, adapter_input_key="edge",
                                 adapter_conditioning_scale=0.7), "dpm", 0, 15),
}


# VAE in the loop + LPIPS distillation loss (FD:128-133, 182-185, 383-397): the batch carries PIXELS, the frozen stand-in VAE
# (oracle.unet_cpu.TinyVAE, downsampling 2) encodes them to the 32x32 latents of the other cases, both outputs are decoded again
# for the perceptual term (oracle.unet_cpu.TinyLPIPS)
LPIPS_CASES = {
    "g_lpips_dmd_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="lpips",
                               gan_loss_type="lsgan", use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=13.0,
                               dmd_loss_scale=0.3, adversarial_loss_scale=0.1), "dpm", 0, 16),
}


def make_pixel_batch(B=2, px=64, ctx_dim=64, seed=6):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy((0.5 * rs.standard_normal((B, 3, px, px))).astype(np.float32))
    c = torch.from_numpy(rs.standard_normal((B, 77, ctx_dim)).astype(np.float32))
    return {"image": x, "crossattn": c, "text": ["a"] * B}


def make_edge(B=2, hw=32, seed=8):
    return torch.randn(B, 1, hw, hw, generator=torch.Generator().manual_seed(seed))


def build_models(unet_cfg=None, lora_rank=LORA_RANK, disc_kw=None):
    unet_cfg = unet_cfg or tiny_config()
    teacher = seeded_init_(UNet2DConditionRef(unet_cfg), 1)
    student = copy.deepcopy(teacher)
    student.add_adapter(lora_rank)
    seeded_init_(student, 2)
    student.load_state_dict(dict(teacher.state_dict()), strict=False)
    teacher.freeze()
    disc_kw = disc_kw or dict(kind="sd15", color_dim=unet_cfg.block_out_channels[-1], feat=16, last_k=2)
    disc = seeded_init_(make_discriminator(**disc_kw), 3)
    return teacher, student, disc


def make_batch(B=2, hw=32, ctx_dim=64, seed=5):
    rs = np.random.RandomState(seed)
    z = torch.from_numpy(rs.standard_normal((B, 4, hw, hw)).astype(np.float32))
    c = torch.from_numpy(rs.standard_normal((B, 77, ctx_dim)).astype(np.float32))
    return {"image": z, "crossattn": c, "text": ["a"] * B}


# ---- FlashDiffusionSD3 (flow matching) fixtures: models and inputs are fully seeded ---------------------------------
SD3_CASES = {
    # name: (config kwargs, step, seed)
    "sd3_g_dmd_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="lsgan",
                             use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=7.0), 0, 11),
    "sd3_d_hinge": (dict(K=[8], num_iterations_per_K=[10], timestep_distribution="mixture", gan_loss_type="hinge",
                         mixture_num_components=4, mixture_var=0.5, mode_probs=[[0.1, 0.3, 0.3, 0.3]]), 1, 12),
}


def build_sd3_models():
    """teacher / perturbed student test denoisers, PatchGAN-style head, text-embedding pipeline stub, batch"""
    from .flash_sd3_ref import EmbeddingPipeline, TinyFlowDenoiser
    teacher = TinyFlowDenoiser(seed=1)
    student = copy.deepcopy(teacher)
    g = torch.Generator().manual_seed(2)
    for p in student.parameters():
        p.data.add_(torch.randn(p.shape, generator=g) * 0.02)
    teacher.freeze()
    disc = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 4, 2, 1), torch.nn.SiLU(), torch.nn.Conv2d(8, 1, 8, 1, 0),
                               torch.nn.Flatten())
    g3 = torch.Generator().manual_seed(3)
    for p in disc.parameters():
        p.data.copy_(torch.randn(p.shape, generator=g3) * 0.1)
    ge = torch.Generator().manual_seed(9)
    pipe = EmbeddingPipeline(torch.randn(2, 5, 10, generator=ge), torch.randn(2, 12, generator=ge),
                             torch.randn(2, 5, 10, generator=ge), torch.randn(2, 12, generator=ge))
    gb = torch.Generator().manual_seed(5)
    batch = {"image": torch.randn(2, 4, 16, 16, generator=gb), "text": ["a", "b"]}
    return teacher, student, disc, pipe, batch


# ---- PixArt DiT denoiser: name -> (config overrides on dit_cpu.TINY_DIT, masked) -----------------------
DIT_CASES = {
    "dit_tiny": (dict(), False),
    "dit_hd72_masked": (dict(attention_head_dim=72, num_attention_heads=4, cross_attention_dim=288, time_embed_dim=288,
                             caption_channels=64, num_vector_conditionings=3), True),
}


def build_dit(name, lora_r=0):
    """(oracle-restated PixArt denoiser with seeded weights [+ LoRA], inputs) for a DIT_CASES entry"""
    from . import dit_cpu
    over, masked = DIT_CASES[name]
    cfg = {**dit_cpu.TINY_DIT, **over}
    m = dit_cpu.seeded_init_(dit_cpu.PixartTransformerRef(**cfg), 3)
    if lora_r:
        dit_cpu.add_lora_(m, lora_r, seed=4, b_std=0.05)
    g = torch.Generator().manual_seed(1)
    nv = cfg["num_vector_conditionings"]
    x = torch.randn(2, 4, 16, 16, generator=g)
    t = torch.tensor([999.0, 250.0])
    cond = {"crossattn": torch.randn(2, 7, cfg["caption_channels"], generator=g),
            "vector": torch.randn(2, nv * cfg["projection_class_embeddings_input_dim"], generator=g)}
    if masked:
        cond["attention_mask"] = torch.tensor([[1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1]])
    w = torch.randn(2, 4, 16, 16, generator=g)
    return cfg, m, (x, t, {"cond": cond}), w


# ---- SD3 MMDiT denoiser: name -> config overrides on mmdit_cpu.TINY_MMDIT ------------------------------
MMDIT_CASES = {
    "mmdit_tiny": dict(),
    "mmdit_hd64": dict(attention_head_dim=64, num_attention_heads=2, caption_projection_dim=128, num_layers=3,
                       pos_embed_max_size=10),
}


def build_mmdit(name, lora_r=0):
    """(oracle-restated SD3 denoiser with seeded weights [+ LoRA], inputs) for an MMDIT_CASES entry"""
    from . import dit_cpu, mmdit_cpu
    cfg = {**mmdit_cpu.TINY_MMDIT, **MMDIT_CASES[name]}
    m = dit_cpu.seeded_init_(mmdit_cpu.SD3TransformerRef(**cfg), 3)
    if lora_r:
        dit_cpu.add_lora_(m, lora_r, seed=4, b_std=0.05)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 16, 16, generator=g)
    t = torch.tensor([999.0, 250.0])
    cond = {"crossattn": torch.randn(2, 7, cfg["joint_attention_dim"], generator=g),
            "vector": torch.randn(2, cfg["pooled_projection_dim"], generator=g)}
    w = torch.randn(2, 16, 16, 16, generator=g)
    return cfg, m, (x, t, {"cond": cond}), w


# ---- the full-size C1 step (BASELINE.json configs[0]): fixture tests/golden/c1_sd15_full.npz (oracle/make_golden.py c1) ----
C1_LORA_RANK = 16
C1_SEED = 31


def build_c1_models():
    """BASELINE.json configs[0] / SURVEY 8a "C1": the full-size SD1.5 UNet (examples/train_flash_sd.py:56-114), B = 1, one
    teacher step (K = [1]), seeded weights; the student carries a small-rank LoRA with non-zero B so that every gradient is
    exercised, the PatchGAN head is the reference's SD1.5 one (examples/train_flash_sd.py:225-240)."""
    from .unet_cpu import sd15_config
    teacher = seeded_init_(UNet2DConditionRef(sd15_config()), 1)
    student = copy.deepcopy(teacher)
    student.add_adapter(C1_LORA_RANK)
    seeded_init_(student, 2)
    student.load_state_dict(dict(teacher.state_dict()), strict=False)
    teacher.freeze()
    disc = seeded_init_(make_discriminator(kind="sd15", color_dim=1280, feat=64, last_k=4), 3)
    return teacher, student, disc


C1_KW = dict(K=[1], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="l2", gan_loss_type="lsgan",
             use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=13.0, dmd_loss_scale=0.3, adversarial_loss_scale=0.1)


def c1_grad_probe(numel, seed):
    """fixed pseudo-random direction a gradient tensor is projected on (fixtures cannot carry 3 M LoRA gradient values)"""
    return torch.randn(numel, generator=torch.Generator().manual_seed(seed))


# ---- FlashDiffusionSD3 over the MMDiT denoiser itself (BASELINE.json configs[4] in miniature): the
# reference's REAL step class on the reference's REAL wrapper (restated diffusers base), DMD + lsgan head ----------------------
SD3_MMDIT_CASES = {
    # name: (config kwargs, mmdit case, step, seed)
    "sd3_mmdit_g_dmd_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="lsgan",
                                   use_dmd_loss=True, guidance_scale_min=3.0, guidance_scale_max=7.0), "mmdit_tiny", 0, 41),
    "sd3_mmdit_d_lsgan": (dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", gan_loss_type="lsgan"),
                          "mmdit_tiny", 1, 42),
}


def sd3_mmdit_head(in_ch=16):
    """examples/train_flash_sd3.py:148-183 in miniature: strided 4x4 convs + GroupNorm + SiLU on the [B, 16, 16, 16] prediction"""
    d = torch.nn.Sequential(torch.nn.Conv2d(in_ch, 16, 4, 2, 1, bias=False), torch.nn.SiLU(),
                            torch.nn.Conv2d(16, 32, 4, 2, 1, bias=False), torch.nn.GroupNorm(4, 32), torch.nn.SiLU(),
                            torch.nn.Conv2d(32, 1, 4, 1, 0, bias=False), torch.nn.Flatten())
    g3 = torch.Generator().manual_seed(3)
    for p in d.parameters():
        p.data.copy_(torch.randn(p.shape, generator=g3) * 0.1 + (1.0 if p.dim() == 1 and p is d[3].weight else 0.0))
    return d


def build_sd3_mmdit_inputs(case="mmdit_tiny"):
    """(cfg, teacher oracle module (frozen), student oracle module with LoRA, head, embedding pipeline, batch)"""
    from . import dit_cpu
    from .flash_sd3_ref import EmbeddingPipeline
    cfg, teacher, (x, t, cond), _ = build_mmdit(case, lora_r=0)
    _, student, _, _ = build_mmdit(case, lora_r=8)
    for p in teacher.parameters():
        p.requires_grad = False
    teacher.eval()
    ge = torch.Generator().manual_seed(9)
    L, D, P = 7, cfg["joint_attention_dim"], cfg["pooled_projection_dim"]
    pipe = EmbeddingPipeline(torch.randn(2, L, D, generator=ge), torch.randn(2, P, generator=ge),
                             torch.randn(2, L, D, generator=ge), torch.randn(2, P, generator=ge))
    gb = torch.Generator().manual_seed(5)
    batch = {"image": torch.randn(2, 16, 16, 16, generator=gb), "text": ["a", "b"]}
    return cfg, teacher, student, sd3_mmdit_head(), pipe, batch

