"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's flow-matching distillation step
``FlashDiffusionSD3.forward`` (/root/reference/src/flash/models/flash_sd3/flash_diffusion_model.py, "FD3" below; SURVEY.md
section 8a row a18), pinned bit-identically against the reference's own class (imported unmodified through
oracle/shim_import.py) in tests/test_oracle_vs_reference.py.  Only tests / golden generation may import this module.

What differs from FlashDiffusion (oracle/flash_ref.py): rectified-flow noising ``x_t = sigma eps + (1 - sigma) z``
(FD3:262-270), an Euler teacher loop on the scheduler's sigmas (FD3:281-314), ``x0_hat = x_t - sigma v`` (FD3:325), a DMD
term without the alpha-bar weighting whose "predicted x0" is literally the real velocity (FD3:483-485 -- restated as
written), GAN timesteps picked from a 1000-step copy of the scheduler (FD3:520-528), conditioning taken from a
``pipeline.encode_prompt`` call instead of a conditioner (FD3:198-229), the discriminator backbone called with
``return_post_mid_blocks=True`` (FD3:563), and a scalar ``loss`` (not a list) when no discriminator is given (FD3:357-364).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .flash_ref import Draws, distill_loss, gan_losses, timestep_pmf


@dataclass
class FlashSD3ConfigRef:
    """FlashDiffusionSD3Config (flash_sd3/flash_diffusion_config.py) with its __post_init__ list expansion."""
    K: List[int] = field(default_factory=lambda: [32, 32, 32, 32, 32])
    num_iterations_per_K: List[int] = field(default_factory=lambda: [5000, 10000, 15000, 20000, 25000])
    guidance_scale_min: Any = 3.0
    guidance_scale_max: Any = 7.0
    distill_loss_type: str = "l2"
    ucg_keys: List[str] = field(default_factory=lambda: ["text"])
    timestep_distribution: str = "mixture"
    mixture_num_components: Any = 4
    mixture_var: Any = 0.5
    use_dmd_loss: bool = False
    dmd_loss_scale: Any = 1.0
    distill_loss_scale: Any = 1.0
    adversarial_loss_scale: Any = 1.0
    gan_loss_type: str = "hinge"
    mode_probs: Optional[List[List[float]]] = None
    use_teacher_as_real: bool = False
    input_key: str = "image"

    def __post_init__(self):
        n = len(self.K)
        for k in ("mixture_num_components", "guidance_scale_min", "guidance_scale_max", "mixture_var",
                  "distill_loss_scale", "dmd_loss_scale", "adversarial_loss_scale"):
            v = getattr(self, k)
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                setattr(self, k, [v] * n)
        if self.mode_probs is None:
            self.mode_probs = [None] * n   # FD3:169 indexes mode_probs[K_step]; gaussian_mixture() treats None as uniform


class EmbeddingPipeline:
    """Duck type of the ``DiffusionPipeline`` FlashDiffusionSD3 asks for text embeddings (FD3:196-219): returns the
    embeddings carried by the batch (what a frozen text-encoder stack would produce) and fixed negative ones."""

    def __init__(self, prompt_embeds, pooled, neg_prompt_embeds, neg_pooled):
        self.e = (prompt_embeds, neg_prompt_embeds, pooled, neg_pooled)

    def to(self, *a, **k):
        return self

    def encode_prompt(self, *a, **k):
        self.calls = getattr(self, "calls", []) + [dict(k, _args=a)]   # what the caller asked for (kwargs parity tests)
        return self.e


def get_sigmas(scheduler, timesteps, n_dim=4, dtype=torch.float32, device="cpu"):
    """FD3:947-958"""
    sigmas = scheduler.sigmas.to(device=device, dtype=dtype)
    st = scheduler.timesteps.to(device)
    idx = [(st == t).nonzero().item() for t in timesteps.to(device)]
    s = sigmas[idx].flatten()
    while len(s.shape) < n_dim:
        s = s.unsqueeze(-1)
    return s


class FlashDiffusionSD3Ref(torch.nn.Module):
    def __init__(self, config, student_denoiser, teacher_denoiser=None, teacher_noise_scheduler=None,
                 teacher_sampling_noise_scheduler=None, sampling_noise_scheduler=None, vae=None, conditioner=None,
                 discriminator=None, pipeline=None, cpu_offload=False, lpips_model=None):
        super().__init__()
        self.vae = vae                     # FD3:93: any module with the AutoencoderKLDiffusers surface the step touches
        if config.distill_loss_type == "lpips":                                          # FD3:130-131 builds lpips.LPIPS itself
            assert vae is not None and lpips_model is not None                           # (package / weights absent here)
            self.lpips = lpips_model
        self.config = config
        self.input_key = config.input_key
        self.student_denoiser = student_denoiser
        self.teacher_denoiser = teacher_denoiser
        self.teacher_noise_scheduler = teacher_noise_scheduler
        self.teacher_noise_scheduler_copy = copy.deepcopy(teacher_noise_scheduler)      # FD3:116
        self.teacher_sampling_noise_scheduler = teacher_sampling_noise_scheduler
        self.sampling_noise_scheduler = sampling_noise_scheduler
        self.discriminator = discriminator
        self.use_adversarial_loss = discriminator is not None                            # FD3:120-128
        self.pipeline = pipeline
        self.iter_steps = 0
        self.disc_update_counter = 0
        self.K_steps = np.cumsum(config.num_iterations_per_K)
        self.K_prev = config.K[0]
        self.draws: Optional[Draws] = None
        self.last_draws: Optional[Draws] = None
        self.terms: Dict[str, Any] = {}

    def forward(self, batch, batch_idx=0, step=0, *args, **kwargs):
        cfg = self.config
        sch = self.teacher_noise_scheduler
        d = self.draws if self.draws is not None else Draws()
        self.last_draws = d
        self.iter_steps += 1
        if self.vae is not None:                                                         # FD3:138-144, 190-191
            with torch.no_grad():
                z = self.vae.encode(batch[self.vae.config.input_key])
        else:
            z = batch[self.input_key]
        with torch.no_grad():                                                            # FD3:196-219
            pe, npe, ppe, nppe = self.pipeline.encode_prompt(prompt=batch["text"], device=z.device)
        cond = {"cond": {"vector": ppe, "crossattn": pe}}
        uncond = {"cond": {"vector": nppe, "crossattn": npe}}
        if self.iter_steps > self.K_steps[-1]:                                           # FD3:237-243
            K_step = len(cfg.K) - 1
        else:
            K_step = int(np.argmax(self.iter_steps < self.K_steps))
        K = cfg.K[K_step]
        g_min, g_max = cfg.guidance_scale_min[K_step], cfg.guidance_scale_max[K_step]
        assert K == self.K_prev, "K switching (teacher <- student copy, FD3:244-249) is not restated"
        noise = d.randn_like("noise", z)                                                  # FD3:252
        sch.set_timesteps(K)                                                              # FD3:135-177
        prob = timestep_pmf(cfg, K, K_step)
        start_idx = d.multinomial("start_idx", prob, 1)
        start_t = sch.timesteps[start_idx].to(z.device).repeat(z.shape[0])
        sig = get_sigmas(sch, start_t, device=z.device)                                   # FD3:260-262
        if start_idx == 0:                                                                # FD3:264-272
            x_init = noise
            if hasattr(sch, "init_noise_sigma"):
                x_init *= sch.init_noise_sigma
            x_student = noise
        else:
            x_init = sig * noise + (1.0 - sig) * z
            x_student = x_init
        x = x_init.clone().detach()
        g = d.rand1("guidance").to(z.device) * (g_max - g_min) + g_min                     # FD3:278-280
        with torch.no_grad():                                                             # FD3:282-314
            for t in sch.timesteps[start_idx:]:
                tt = torch.tensor([t], device=z.device).repeat(z.shape[0])
                e_c = self.teacher_denoiser(sample=x, timestep=tt, conditioning=cond, *args, **kwargs)
                e_u = self.teacher_denoiser(sample=x, timestep=tt, conditioning=uncond, *args, **kwargs)
                e = g * e_c + (1 - g) * e_u
                x = sch.step(e, t, x, return_dict=False)[0]
        teacher_output = x
        v_s = self.student_denoiser(sample=x_student, timestep=start_t, conditioning=cond)    # FD3:318-323
        student_output = x_student - v_s * sig                                            # FD3:325
        if cfg.distill_loss_type == "lpips":                                             # FD3:391-411: clamped crop bounds
            so, to = student_output, teacher_output
            crop_h = max((so.shape[2] - 64) // 2, 0)
            crop_w = max((so.shape[3] - 64) // 2, 0)
            so = so[:, :, crop_h:min(crop_h + 64, so.shape[2]), crop_w:min(crop_w + 64, so.shape[3])]
            to = to[:, :, crop_h:min(crop_h + 64, to.shape[2]), crop_w:min(crop_w + 64, to.shape[3])]
            l_distill = self.lpips(self.vae.decode(so).clamp(-1, 1), self.vae.decode(to).clamp(-1, 1)).mean()
        else:
            l_distill = distill_loss(cfg.distill_loss_type, student_output, teacher_output)
        loss = l_distill * cfg.distill_loss_scale[K_step]
        self.terms = {"distill": l_distill.detach(), "K_step": K_step, "guidance": float(g)}
        if cfg.use_dmd_loss:
            l_dmd = self.dmd_loss(d, student_output, cond, cond, uncond, K_step)
            self.terms["dmd"] = l_dmd.detach()
            loss = loss + l_dmd * cfg.dmd_loss_scale[K_step]
        if self.use_adversarial_loss:
            gan = self.gan_loss(d, z, student_output, teacher_output, cond, step)
            loss = loss + cfg.adversarial_loss_scale[K_step] * gan[0]
            self.terms["gan_G"], self.terms["gan_D"] = gan
            return {"loss": [loss, gan[1]], "teacher_output": teacher_output, "student_output": student_output,
                    "noisy_sample": x_init, "start_timestep": start_t[0].item()}
        return {"loss": loss.mean(), "teacher_output": teacher_output, "student_output": student_output,
                "noisy_sample": x_init, "start_timestep": start_t[0].item()}

    def dmd_loss(self, d, s, student_cond, cond, uncond, K_step):
        """FD3:416-499 (restated as written, including pred_x_0_student = real_noise_pred)"""
        cfg, sc = self.config, self.teacher_noise_scheduler_copy
        noise = d.randn_like("dmd_noise", s)
        ti = d.randint("dmd_t", 0, self.teacher_noise_scheduler.config.num_train_timesteps, (s.shape[0],), "cpu")
        t = sc.timesteps[ti].to(s.device)
        sig = get_sigmas(sc, t, device=s.device)
        x = sig * noise + (1.0 - sig) * s
        with torch.no_grad():
            r_c = self.teacher_denoiser(sample=x, timestep=t, conditioning=cond)
            r_u = self.teacher_denoiser(sample=x, timestep=t, conditioning=uncond)
            f_c = self.student_denoiser(sample=x, timestep=t, conditioning=student_cond)
            g = (d.rand1("dmd_guidance").to(s.device) * (cfg.guidance_scale_max[K_step] - cfg.guidance_scale_min[K_step])
                 + cfg.guidance_scale_min[K_step])
        real = g * r_c + (1 - g) * r_u
        coeff = (-f_c) - (-real)
        w = 1.0 / ((s - real).abs().mean([1, 2, 3], keepdim=True) + 1e-5).detach()
        return F.mse_loss(s, (s - w * coeff).detach(), reduction="mean")

    def gan_loss(self, d, z, s, teacher_output, cond, step):
        """FD3:501-667"""
        cfg, sc = self.config, self.teacher_noise_scheduler_copy
        self.disc_update_counter += 1
        noise = d.randn_like("gan_noise", s)
        real = teacher_output if cfg.use_teacher_as_real else z
        sel = [float(sc.timesteps[-10]), float(sc.timesteps[-250]), float(sc.timesteps[-500]), float(sc.timesteps[-750])]
        idx = d.multinomial("gan_t", torch.tensor([0.25, 0.25, 0.25, 0.25]), s.shape[0], replacement=True).to(s.device)
        t = torch.tensor(sel, device=s.device)[idx]
        sig = get_sigmas(sc, t, device=s.device)
        x = torch.cat([sig * noise + (1.0 - sig) * s, sig * noise + (1.0 - sig) * real], dim=0)
        c2 = {"cond": {k: torch.cat([v, v], dim=0) for k, v in cond["cond"].items()}} if cond is not None else None
        feat = self.teacher_denoiser(sample=x, timestep=torch.cat([t, t], dim=0), conditioning=c2,
                                     return_post_mid_blocks=True)
        ff, fr = feat.chunk(2, dim=0)
        return gan_losses(cfg.gan_loss_type, self.discriminator, ff, fr, step, s.size(0), noise.device)


    # ---- FD3:682-843: Euler sampler of the student (and, optionally, of the teacher next to it) ----
    @torch.no_grad()
    def sample(self, z, num_steps=20, guidance_scale=1.0, teacher_guidance_scale=5.0, conditioner_inputs=None,
               uncond_conditioner_inputs=None, max_samples=None, verbose=False, log_teacher_samples=False):
        self.teacher_noise_scheduler.set_timesteps(num_steps)                                   # FD3:707
        ss = self.sampling_noise_scheduler
        ss.set_timesteps(num_steps)                                                             # FD3:709
        sample = z
        pe, npe, ppe, nppe = self.pipeline.encode_prompt(prompt=conditioner_inputs["text"], device=z.device)
        cond = {"cond": {"vector": ppe, "crossattn": pe}}
        uncond = {"cond": {"vector": nppe, "crossattn": npe}}
        if max_samples is not None:                                                             # FD3:751-762
            sample = sample[:max_samples]
            cond["cond"] = {k: v[:max_samples] for k, v in cond["cond"].items()}
            uncond["cond"] = {k: v[:max_samples] for k, v in uncond["cond"].items()}
        sample_init = sample
        if hasattr(ss, "init_noise_sigma"):
            sample = sample * ss.init_noise_sigma
        for t in ss.timesteps:                                                                  # FD3:767-795
            tt = t.to(z.device).repeat(sample.shape[0])
            e_c = self.student_denoiser(sample=sample, timestep=tt, conditioning=cond)
            e_u = self.student_denoiser(sample=sample, timestep=tt, conditioning=uncond)
            e = guidance_scale * e_c + (1 - guidance_scale) * e_u
            sample = ss.step(e, t, sample, return_dict=False)[0]
        decoded = self.vae.decode(sample) if self.vae is not None else sample                   # FD3:794-797
        ref = None
        if log_teacher_samples:                                                                 # FD3:804-841
            ts = self.teacher_sampling_noise_scheduler
            ts.set_timesteps(num_steps)
            ref = sample_init * ts.init_noise_sigma if hasattr(ts, "init_noise_sigma") else sample_init
            for t in ts.timesteps:
                tt = t.to(z.device).repeat(ref.shape[0])
                e_c = self.teacher_denoiser(sample=ref, timestep=tt, conditioning=cond)
                e_u = self.teacher_denoiser(sample=ref, timestep=tt, conditioning=uncond)
                e = teacher_guidance_scale * e_c + (1 - teacher_guidance_scale) * e_u
                ref = ts.step(e, t, ref, return_dict=False)[0]
            if self.vae is not None:                                                            # FD3:838-841
                ref = self.vae.decode(ref)
        return decoded, ref


    def log_samples(self, batch, input_shape=None, guidance_scale=1.0, teacher_guidance_scale=5.0, max_samples=8,
                    num_steps=20, device="cpu", log_teacher_samples=False, conditioner_inputs=None,
                    conditioner_uncond_inputs=None):
        """FD3:845-945 (no VAE: `input_shape` = latent shape is mandatory, the reference's ValueError branch)"""
        if isinstance(num_steps, int):
            num_steps = [num_steps]
        logs = {}
        N = max_samples
        if batch is not None:
            N = min(N, min(len(batch[k]) for k in batch))
        if conditioner_inputs is not None:
            m = min(len(conditioner_inputs[k]) for k in conditioner_inputs)
            conditioner_inputs.update({k: v.to(device) for k, v in conditioner_inputs.items() if torch.is_tensor(v)})
            batch.update(conditioner_inputs)
            N = min(N, m)
        if conditioner_uncond_inputs is not None:
            m = min(len(conditioner_uncond_inputs[k]) for k in conditioner_uncond_inputs)
            conditioner_uncond_inputs.update({k: v.to(device) for k, v in conditioner_uncond_inputs.items()
                                              if torch.is_tensor(v)})
            batch_uncond = copy.deepcopy(batch)
            batch_uncond.update(conditioner_uncond_inputs)
            N = min(N, m)
        else:
            batch_uncond = None
        if input_shape is None:
            if self.vae is not None:                                                               # FD3:904-912
                px = batch[self.vae.config.input_key].shape[2:]
                input_shape = (self.vae.latent_channels, px[0] // self.vae.downsampling_factor,
                               px[1] // self.vae.downsampling_factor)
            else:
                raise ValueError("input_shape must be passed when no VAE is used in the model")       # FD3:913-916
        for n in num_steps:
            z = torch.randn(N, *input_shape).to(device)                                            # FD3:911
            samples, samples_ref = self.sample(z, num_steps=n, conditioner_inputs=batch,
                                               uncond_conditioner_inputs=batch_uncond, guidance_scale=guidance_scale,
                                               teacher_guidance_scale=teacher_guidance_scale, max_samples=N,
                                               log_teacher_samples=log_teacher_samples)
            logs[f"samples_{n}_steps/{self.sampling_noise_scheduler.__class__.__name__}_{guidance_scale}_cfg/student"] = samples
            if samples_ref is not None:
                logs[f"samples_{n}_steps/{self.teacher_sampling_noise_scheduler.__class__.__name__}"
                     f"_{teacher_guidance_scale}_cfg/teacher"] = samples_ref
        return logs


class TinyFlowDenoiser(torch.nn.Module):
    """A small velocity model honouring the reference's transformer-wrapper contract (TW:113-155: sample [B,C,H,W],
    timestep [B], conditioning {"cond": {"vector", "crossattn"}}, unknown kwargs swallowed -- e.g. the
    ``return_post_mid_blocks=True`` of FD3:563).  Test double for FlashDiffusionSD3's denoisers; NOT an SD3 transformer."""

    def __init__(self, channels=4, feat=16, vector_dim=12, ctx_dim=10, seed=0):
        super().__init__()
        self.cin = torch.nn.Conv2d(channels, feat, 3, 1, 1)
        self.t1 = torch.nn.Linear(8, feat)
        self.vec = torch.nn.Linear(vector_dim, feat)
        self.ctx = torch.nn.Linear(ctx_dim, feat)
        self.cout = torch.nn.Conv2d(feat, channels, 3, 1, 1)
        g = torch.Generator().manual_seed(seed)
        for p in self.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.05))

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, sample, timestep, conditioning=None, *args, **kwargs):
        t = timestep.to(sample.dtype).reshape(-1, 1) / 1000.0
        fr = torch.arange(1, 5, dtype=sample.dtype, device=sample.device).reshape(1, -1)
        emb = torch.cat([torch.sin(t * fr), torch.cos(t * fr)], dim=1)
        h = self.t1(emb)
        if conditioning is not None:
            h = h + self.vec(conditioning["cond"]["vector"]) + self.ctx(conditioning["cond"]["crossattn"].mean(1))
        x = F.silu(self.cin(sample) + h[:, :, None, None])
        return self.cout(x)
