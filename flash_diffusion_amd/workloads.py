"""Synthetic BASELINE.json workloads: random-init weights of the pinned
architectures, synthetic latents / text embeddings, pinned teacher-step count."""
from __future__ import annotations

import copy

import torch

from .flash import FlashDiffusion, FlashDiffusionConfig, TensorConditioner
from .schedulers import DPMSolverMultistepScheduler
from .unet import MiUNet2DConditionModel

SD15 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
            cross_attention_dim=768, attention_head_dim=8, transformer_layers_per_block=1)
SDXL = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
            up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), cross_attention_dim=2048,
            attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10), class_embed_type="projection",
            projection_class_embeddings_input_dim=2816)   # examples/train_flash_sdxl.py:66-118
PIXART = dict(sample_size=128, num_layers=28, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2,
              attention_bias=True, num_attention_heads=16, cross_attention_dim=1152, activation_fn="gelu-approximate",
              num_embeds_ada_norm=1000, norm_type="ada_norm_single", norm_elementwise_affine=False, norm_eps=1e-6,
              caption_channels=4096, projection_class_embeddings_input_dim=256, time_embed_dim=1152,
              timesteps_embedding_num_channels=256, use_concat_vector_conditioning=True,
              num_vector_conditionings=3)                  # examples/train_flash_pixart.py:63-86 (PixArt-alpha XL/2)
TINY_PIXART = dict(PIXART, sample_size=16, num_layers=2, attention_head_dim=8, num_attention_heads=4, cross_attention_dim=32,
                   caption_channels=48, projection_class_embeddings_input_dim=16, time_embed_dim=32,
                   timesteps_embedding_num_channels=16, num_vector_conditionings=2)
SD3 = dict(sample_size=128, patch_size=2, in_channels=16, num_layers=24, attention_head_dim=64, num_attention_heads=24,
           joint_attention_dim=4096, caption_projection_dim=1536, pooled_projection_dim=2048, out_channels=16,
           pos_embed_max_size=192)                            # examples/train_flash_sd3.py:65-77 (SD3-medium)
TINY_SD3 = dict(SD3, sample_size=16, num_layers=2, attention_head_dim=8, num_attention_heads=4, joint_attention_dim=48,
                caption_projection_dim=32, pooled_projection_dim=24, pos_embed_max_size=12)
TINY = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64, 64, 64), layers_per_block=2,
            cross_attention_dim=64, attention_head_dim=2, transformer_layers_per_block=1)


def sd15_discriminator(color_dim=1280, feat=64):
    """the PatchGAN head of examples/train_flash_sd.py:225-240 on the teacher's mid-block features [B,1280,8,8]"""
    import torch.nn as nn
    return nn.Sequential(nn.Conv2d(color_dim, feat, 3, 1, 1), nn.SiLU(True), nn.Conv2d(feat, feat * 2, 4, 2, 1, bias=False),
                         nn.SiLU(True), nn.GroupNorm(4, feat * 2), nn.Conv2d(feat * 2, 1, 4, 1, 0, bias=False), nn.Flatten())


def build_flash(arch=SD15, lora_rank=128, n_teacher_steps=4, device="cuda", seed=0, discriminator=None,
                use_dmd_loss=False, gan_loss_type="lsgan", guidance=8.0, distill_loss_type="l2", vae=None, lpips_model=None):
    """teacher (frozen) + student = copy + LoRA (peft init: A gaussian, B = 0), DPM-Solver++ trailing schedule
    with K = n_teacher_steps and start index pinned to 0 (so every step runs exactly n teacher CFG steps)."""
    torch.manual_seed(seed)
    if arch.get("norm_type") == "ada_norm_single":          # PixArt DiT denoiser
        from .dit import MiTransformer2DModel
        teacher = MiTransformer2DModel(**arch)
    else:
        teacher = MiUNet2DConditionModel(**arch)
    student = copy.deepcopy(teacher)
    teacher = teacher.to(device)
    teacher.freeze()
    student = student.to(device)
    student.add_adapter(lora_rank)
    cfg = FlashDiffusionConfig(K=[n_teacher_steps], num_iterations_per_K=[10 ** 9], timestep_distribution="uniform",
                               distill_loss_type=distill_loss_type, use_dmd_loss=use_dmd_loss, gan_loss_type=gan_loss_type,
                               guidance_scale_min=3.0, guidance_scale_max=13.0, adversarial_loss_scale=0.1,
                               dmd_loss_scale=0.3)
    m = FlashDiffusion(cfg, student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(),
                       discriminator=discriminator, vae=vae, lpips_model=lpips_model).to(device)
    m.fixed_start_idx = 0
    m.fixed_guidance = guidance
    return m


def sd_vae(device="cuda", seed=0):
    """the SD1.5 AutoencoderKL decoder (random-init weights of the architecture: no network) behind the reference wrapper's
    surface, on the HIP path; the batch carries latents, so the (no-grad, pre-hot-path) encode is an identity stand-in"""
    from .nets import MiAutoencoderKL, MiAutoencoderKLDiffusers

    class _LatentsIn(torch.nn.Module):
        def encode(self, x):
            return x
    torch.manual_seed(seed)
    with torch.device(device):
        v = MiAutoencoderKL()
    w = MiAutoencoderKLDiffusers(v.to(device), encoder=_LatentsIn())
    w.freeze()
    return w


def synthetic_batch(B, hw, ctx_dim, device="cuda", seed=1234, L=77, vector_dim=0, attention_mask=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    b = {"image": torch.randn(B, 4, hw, hw, generator=g).to(device),
         "crossattn": torch.randn(B, L, ctx_dim, generator=g).to(device), "text": ["synthetic"] * B}
    if vector_dim:
        b["vector"] = torch.randn(B, vector_dim, generator=g).to(device)
    if attention_mask:                                      # "mask of ones" for the PixArt T5 context
        b["attention_mask"] = torch.ones(B, L, dtype=torch.long, device=device)
        b["attention_mask_lens"] = [L] * B                  # what a tokenizer knows on the host: no device read per step
    return b


class SyntheticPromptEncoder:
    """Stands where ``StableDiffusion3Pipeline`` stands in FlashDiffusionSD3 (FD3:196-229): ``encode_prompt`` returns
    (prompt_embeds, negative_prompt_embeds, pooled, negative_pooled) -- synthetic embeddings of the SD3 shapes
    ([B, 333, 4096] = 77 CLIP + 256 T5 tokens, pooled [B, 2048]); the unconditional ones are zeros.  The bench
    measures the step, not the encoders (``clip.py`` / ``t5.py`` run them on the same op layer when a caller wants text in)."""

    def __init__(self, B, L, ctx_dim, pooled_dim, device="cuda", seed=4321):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.pe = torch.randn(B, L, ctx_dim, generator=g).to(device)
        self.pp = torch.randn(B, pooled_dim, generator=g).to(device)

    def to(self, *a, **k):
        return self

    def encode_prompt(self, *a, **k):
        return self.pe, torch.zeros_like(self.pe), self.pp, torch.zeros_like(self.pp)


def sd3_discriminator(color_dim=16, feat=64):
    """the PatchGAN head of examples/train_flash_sd3.py:145-183 on the [B, 16, 128, 128] prediction (the DiT wrappers ignore
    return_intermediate, so the head sees the full model output): four strided 4x4 convs + GroupNorm + SiLU -> 5x5 logits"""
    import torch.nn as nn
    return nn.Sequential(nn.Conv2d(color_dim, feat, 4, 2, 1, bias=False), nn.SiLU(True),
                         nn.Conv2d(feat, feat * 2, 4, 2, 1, bias=False), nn.GroupNorm(4, feat * 2), nn.SiLU(True),
                         nn.Conv2d(feat * 2, feat * 4, 4, 2, 1, bias=False), nn.GroupNorm(4, feat * 4), nn.SiLU(True),
                         nn.Conv2d(feat * 4, feat * 8, 4, 2, 1, bias=False), nn.GroupNorm(4, feat * 8), nn.SiLU(True),
                         nn.Conv2d(feat * 8, 1, 4, 1, 0, bias=False), nn.Flatten())


def build_flash_sd3(arch=SD3, lora_rank=64, n_teacher_steps=4, B=4, L=333, device="cuda", seed=0, guidance=5.0,
                    use_dmd_loss=True, discriminator="sd3", gan_loss_type="lsgan"):
    """BASELINE.json configs[4] (C5) on one GPU: frozen MMDiT teacher + LoRA student (examples/train_flash_sd3.py:100-121),
    flow-matching Euler teacher loop with K = n_teacher_steps and the start index pinned to 0, l2 distillation PLUS the DMD
    term (three more denoiser calls, FD3:415-497) and the lsgan GAN term (the full teacher at 2B as backbone + the example's
    PatchGAN head, FD3:498-560) -- what C5 names.  discriminator=None / use_dmd_loss=False give the distillation-only step."""
    from .dit import MiSD3Transformer2DModel
    from .flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler
    torch.manual_seed(seed)
    teacher = MiSD3Transformer2DModel(**arch)
    student = copy.deepcopy(teacher)
    teacher = teacher.to(device)
    teacher.freeze()
    student = student.to(device)
    student.add_adapter(lora_rank)
    if discriminator == "sd3":
        discriminator = sd3_discriminator(arch["in_channels"]).to(device) if arch["sample_size"] >= 64 else None
    cfg = FlashDiffusionSD3Config(K=[n_teacher_steps], num_iterations_per_K=[10 ** 9], timestep_distribution="uniform",
                                  distill_loss_type="l2", guidance_scale_min=3.0, guidance_scale_max=7.0,
                                  use_dmd_loss=use_dmd_loss, gan_loss_type=gan_loss_type, dmd_loss_scale=0.3,
                                  adversarial_loss_scale=0.1)
    m = FlashDiffusionSD3(cfg, student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=FlowMatchEulerDiscreteScheduler(), discriminator=discriminator,
                          pipeline=SyntheticPromptEncoder(B, L, arch["joint_attention_dim"], arch["pooled_projection_dim"],
                                                          device))
    m.fixed_start_idx = 0
    m.fixed_guidance = guidance
    return m
