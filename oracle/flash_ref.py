"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the Flash-Diffusion distillation step.

Follows /root/reference/src/flash/models/flash/flash_diffusion_model.py ("FD"):
  forward FD:179-366, _get_timesteps FD:135-177 (+gaussian_mixture FD:23-35),
  _distill_loss FD:368-399, _dmd_loss FD:401-499, _gan_loss FD:501-667,
  _scalings_for_boundary_conditions FD:710-716, _predicted_x_0 FD:718-752.

PINNED: tests/test_oracle_vs_reference.py runs this file and the reference's own
``FlashDiffusion`` (imported unmodified through oracle/shim_import.py) on the same seeded
global RNG stream and requires bit-identical outputs; tests/golden/*.npz were produced by the
real reference class (oracle/make_golden.py).

Every random draw goes through a ``Draws`` object: in "rng" mode it consumes torch's global
RNG in exactly the reference's order (SURVEY.md appendix A "RNG draw order") and records the
values; in "inject" mode it replays recorded values, which is how the HIP path and this
oracle are made to see identical noised latents.
"""
from __future__ import annotations

from copy import deepcopy
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
class Draws:
    def __init__(self, values: Optional[Dict[str, torch.Tensor]] = None):
        self.inject = values is not None
        self.values: Dict[str, torch.Tensor] = dict(values) if values else {}
        self._count: Dict[str, int] = {}

    def _key(self, name):
        i = self._count.get(name, 0)
        self._count[name] = i + 1
        return name if i == 0 else f"{name}#{i}"

    def _get(self, name, make):
        k = self._key(name)
        if self.inject:
            return self.values[k]
        v = make()
        self.values[k] = v.detach().clone()
        return v

    def randn_like(self, name, x):
        return self._get(name, lambda: torch.randn_like(x)).to(x.device, x.dtype)

    def multinomial(self, name, prob, n, replacement=False):
        return self._get(name, lambda: torch.multinomial(prob, n, replacement=replacement))

    def rand1(self, name):
        return self._get(name, lambda: torch.rand(1))

    def randint(self, name, lo, hi, shape, device):
        return self._get(name, lambda: torch.randint(lo, hi, shape, device=device)).to(device)


@dataclass
class FlashConfigRef:
    """Field-for-field mirror of FlashDiffusionConfig (flash_diffusion_config.py:9-105)."""
    K: List[int] = field(default_factory=lambda: [32, 32, 32, 32, 32])
    num_iterations_per_K: List[int] = field(default_factory=lambda: [5000, 10000, 15000, 20000, 25000])
    guidance_scale_min: Any = 3.0
    guidance_scale_max: Any = 7.0
    distill_loss_type: str = "l2"
    ucg_keys: List[str] = field(default_factory=lambda: ["text"])
    timestep_distribution: str = "mixture"
    mixture_num_components: Any = 4
    mixture_var: Any = 0.5
    adapter_conditioning_scale: float = 1.0
    adapter_input_key: Optional[str] = None
    use_dmd_loss: bool = False
    dmd_loss_scale: Any = 1.0
    distill_loss_scale: Any = 1.0
    adversarial_loss_scale: Any = 1.0
    gan_loss_type: str = "hinge"
    mode_probs: Optional[List[List[float]]] = None
    use_teacher_as_real: bool = False
    use_empty_prompt: bool = False
    input_key: str = "image"

    def __post_init__(self):
        n = len(self.K)
        for f in ("mixture_num_components",):
            if isinstance(getattr(self, f), int):
                setattr(self, f, [getattr(self, f)] * n)
        for f in ("guidance_scale_min", "guidance_scale_max", "mixture_var", "distill_loss_scale",
                  "dmd_loss_scale", "adversarial_loss_scale"):
            if isinstance(getattr(self, f), float):
                setattr(self, f, [getattr(self, f)] * n)
        if self.mode_probs is None:
            self.mode_probs = [[1 / m] * m for m in self.mixture_num_components]
        assert len(self.K) == len(self.num_iterations_per_K)
        assert len(self.K) == len(self.mode_probs)


class TensorConditioner(torch.nn.Module):
    """Synthetic stand-in for ConditionerWrapper (embedders/conditioners_wrapper.py:39-91):
    the batch already carries the embeddings (``crossattn`` [B,L,D], optional ``vector``
    [B,V]); keys listed in ``ucg_keys`` are zeroed, which is what force_zero_embedding
    yields for a dropped text conditioner (clip_embedder_model.py:93-94)."""

    def __init__(self, input_key="text"):
        super().__init__()
        self.input_key = input_key

    def forward(self, batch, ucg_keys=None, set_ucg_rate_zero=False, *args, **kwargs):
        drop = ucg_keys is not None and self.input_key in ucg_keys
        cond = {}
        for k in ("crossattn", "vector", "concat"):
            if k in batch and batch[k] is not None:
                cond[k] = torch.zeros_like(batch[k]) if drop else batch[k]
        return {"cond": cond}


# --------------------------------------------------------------------------------------
def timestep_pmf(cfg, K, K_step):
    """FD:141-165."""
    if cfg.timestep_distribution == "uniform":
        return torch.ones(K) / K
    if cfg.timestep_distribution == "gaussian":
        p = [torch.exp(-torch.tensor([(i - K / 2) ** 2 / K])) for i in range(K)]
        p = torch.tensor(p)
        return p / torch.sum(p)
    M = cfg.mixture_num_components[K_step]
    mp = cfg.mode_probs[K_step]
    var = cfg.mixture_var[K_step]
    locs = [i * (K // M) for i in range(M)]
    p = [sum(mp[j] * torch.exp(-torch.tensor([(i - loc) ** 2 / var])) for j, loc in enumerate(locs))
         for i in range(K)]
    p = torch.tensor(p)
    return p / torch.sum(p)


def boundary_scalings(t, sigma_data=0.5):
    """FD:710-716."""
    c_skip = sigma_data ** 2 / ((t / 0.1) ** 2 + sigma_data ** 2)
    c_out = (t / 0.1) / ((t / 0.1) ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out


def predicted_x0_eps(eps, t, x_t, sqrt_ac, sigmas, fallback):
    """FD:731-742 (epsilon branch): (x_t - sigma_t eps) / alpha_t where alpha_t > 0, else fallback."""
    shape = (x_t.shape[0],) + (1,) * (x_t.ndim - 1)
    sg = sigmas.to(x_t.device)[t].reshape(shape)
    al = sqrt_ac.to(x_t.device)[t].reshape(shape)
    pos = (al > 0).reshape(-1)
    zero = (al == 0).reshape(-1)
    out = torch.zeros_like(x_t)
    out[pos] = (x_t[pos] - sg[pos] * eps[pos]) / al[pos]
    out[zero] = fallback[zero]
    return out


def distill_loss(kind, s, t, vae=None, lpips_model=None):
    """FD:368-399.  lpips: the centre 64x64 latent crop of both outputs (the reference's slice expression verbatim -- for latents
    smaller than 64 its negative start selects the trailing rows / columns) is decoded by the VAE, clamped to [-1, 1] and
    compared by the LPIPS network; the pretrained VAE / VGG weights are not available offline, so the two modules are the
    caller's (tests: oracle.unet_cpu.TinyVAE / TinyLPIPS)."""
    if kind == "l2":
        return torch.mean(((s - t) ** 2).reshape(s.shape[0], -1), 1).mean()
    if kind == "l1":
        return torch.mean(torch.abs(s - t).reshape(s.shape[0], -1), 1).mean()
    if kind == "lpips":
        crop_h = (s.shape[2] - 64) // 2                                                  # FD:385-386
        crop_w = (s.shape[3] - 64) // 2
        s = s[:, :, crop_h:crop_h + 64, crop_w:crop_w + 64]
        t = t[:, :, crop_h:crop_h + 64, crop_w:crop_w + 64]
        ds = vae.decode(s).clamp(-1, 1)                                                  # FD:394-395
        dt = vae.decode(t).clamp(-1, 1)
        return lpips_model(ds, dt).mean()                                                # FD:397
    raise NotImplementedError(kind)


def gan_losses(kind, disc, feat_fake, feat_real, step, B, device):
    """FD:573-662. Returns [loss_G, loss_D] (python 0 for the inactive one, as the reference)."""
    gen = step % 2 == 0
    if kind == "wgan":
        for p in disc.parameters():
            p.data.clamp_(-0.01, 0.01)
        if gen:
            return [-disc(feat_fake).mean(), 0]
        return [0, -disc(feat_real).mean() + disc(feat_fake.detach()).mean()]
    if kind == "lsgan":
        valid = torch.ones(B, 1, device=device)
        fake = torch.zeros(B, 1, device=device)
        if gen:
            return [F.mse_loss(torch.sigmoid(disc(feat_fake)), valid), 0]
        return [0, 0.5 * (F.mse_loss(torch.sigmoid(disc(feat_real)), valid)
                          + F.mse_loss(torch.sigmoid(disc(feat_fake.detach())), fake))]
    if kind == "hinge":
        if gen:
            return [-disc(feat_fake).mean(), 0]
        return [0, F.relu(1.0 - disc(feat_real)).mean() + F.relu(1.0 + disc(feat_fake.detach())).mean()]
    if kind == "non-saturating":
        if gen:
            return [-torch.mean(torch.log(torch.sigmoid(disc(feat_fake)) + 1e-8)), 0]
        return [0, -torch.mean(torch.log(torch.sigmoid(disc(feat_real)) + 1e-8)
                               + torch.log(1 - torch.sigmoid(disc(feat_fake.detach())) + 1e-8))]
    valid = torch.ones(B, 1, device=device)
    if gen:
        return [F.binary_cross_entropy_with_logits(disc(feat_fake), valid), 0]
    fake = torch.zeros(B, 1, device=device)
    return [0, F.binary_cross_entropy_with_logits(disc(feat_real), valid)
            + F.binary_cross_entropy_with_logits(disc(feat_fake.detach()), fake)]


# --------------------------------------------------------------------------------------
class FlashDiffusionRef(torch.nn.Module):
    """Same constructor / forward contract as the reference FlashDiffusion (FD:40-60, 179)."""

    def __init__(self, config, student_denoiser, teacher_denoiser=None, teacher_noise_scheduler=None,
                 teacher_sampling_noise_scheduler=None, sampling_noise_scheduler=None, vae=None,
                 conditioner=None, adapter=None, discriminator=None, lpips_model=None):
        super().__init__()
        self.adapter = adapter                                               # FD:91-94 (T2I adapter: image -> residual list)
        self.adapter_conditioning_scale = config.adapter_conditioning_scale
        self.adapter_input_key = config.adapter_input_key
        self.config = config
        self.input_key = config.input_key
        self.student_denoiser = student_denoiser
        self.teacher_denoiser = teacher_denoiser
        self.teacher_noise_scheduler = teacher_noise_scheduler
        self.teacher_sampling_noise_scheduler = teacher_sampling_noise_scheduler
        self.sampling_noise_scheduler = sampling_noise_scheduler
        # FD:67: any module with the AutoencoderKLDiffusers surface the step touches (config.input_key, encode, decode,
        # latent_channels, downsampling_factor; vae/autoencoderKL.py:11-128); the pretrained network itself is out of scope
        self.vae = vae
        if config.distill_loss_type == "lpips":                              # FD:102-103 builds lpips.LPIPS(net="vgg") itself;
            assert vae is not None and lpips_model is not None               # that package / its weights are absent here
            self.lpips = lpips_model
        self.conditioner = conditioner
        self.discriminator = discriminator
        self.iter_steps = 0
        self.disc_update_counter = 0
        self.K_steps = np.cumsum(config.num_iterations_per_K)
        self.K_prev = config.K[0]
        ac = teacher_noise_scheduler.alphas_cumprod
        self.register_buffer("sqrt_alpha_cumprod", torch.sqrt(ac))
        self.register_buffer("sigmas", torch.sqrt(1 - ac))
        self.draws: Optional[Draws] = None          # set to Draws(values) to inject
        self.last_draws: Optional[Draws] = None
        self.terms: Dict[str, Any] = {}

    def _cond(self, batch, ucg_keys=None, set_ucg_rate_zero=False, *a, **k):
        if self.conditioner is None:
            return None
        return self.conditioner(batch, ucg_keys=ucg_keys, set_ucg_rate_zero=set_ucg_rate_zero,
                                vae=self.vae, *a, **k)

    def forward(self, batch, batch_idx=0, step=0, *args, **kwargs):
        cfg = self.config
        sch = self.teacher_noise_scheduler
        d = self.draws if self.draws is not None else Draws()
        self.last_draws = d
        self.iter_steps += 1                                                  # FD:181
        if self.vae is not None:                                              # FD:128-133, 182-183
            with torch.no_grad():
                z = self.vae.encode(batch[self.vae.config.input_key])
        else:
            z = batch[self.input_key]                                         # FD:185
        conditioning = self._cond(batch, set_ucg_rate_zero=True, *args, **kwargs)   # FD:188
        student_conditioning = self._cond(batch, *args, **kwargs)                   # FD:192
        if cfg.use_empty_prompt and "text" in cfg.ucg_keys:                   # FD:194-199
            ub = deepcopy(batch)
            ub["text"] = [""] * len(batch["text"])
            uncond = self._cond(ub, set_ucg_rate_zero=True, *args, **kwargs)
        else:
            uncond = self._cond(batch, ucg_keys=cfg.ucg_keys, *args, **kwargs)      # FD:203
        if self.adapter:                                                      # FD:207-218
            res = self.adapter(batch[self.adapter_input_key])
            for k, v in enumerate(res):
                res[k] = v * self.adapter_conditioning_scale
        else:
            res = None
        if self.iter_steps > self.K_steps[-1]:                                # FD:221-227
            K_step = len(cfg.K) - 1
        else:
            K_step = int(np.argmax(self.iter_steps < self.K_steps))
        K = cfg.K[K_step]
        g_min, g_max = cfg.guidance_scale_min[K_step], cfg.guidance_scale_max[K_step]

        noise = d.randn_like("noise", z)                                      # FD:236
        sch.set_timesteps(K)                                                  # FD:139
        prob = timestep_pmf(cfg, K, K_step)
        start_idx = d.multinomial("start_idx", prob, 1)                       # FD:167
        start_t = sch.timesteps[start_idx].to(z.device).repeat(z.shape[0])    # FD:171-175
        if start_idx == 0:                                                    # FD:243-246
            x_init = noise
            x_init *= sch.init_noise_sigma
        else:
            x_init = sch.add_noise(z, noise, start_t)                         # FD:250
        x_in = sch.scale_model_input(x_init, start_t)
        eps_s = self.student_denoiser(sample=x_in, timestep=start_t, conditioning=student_conditioning,
                                      down_intrablock_additional_residuals=res)    # FD:260
        c_skip, c_out = boundary_scalings(start_t)                            # FD:267
        shp = (z.shape[0],) + (1,) * (z.ndim - 1)
        c_skip, c_out = c_skip.reshape(shp), c_out.reshape(shp)
        x0_s = predicted_x0_eps(eps_s, start_t.type(torch.int64), x_init, self.sqrt_alpha_cumprod,
                                self.sigmas, z)                               # FD:272
        x = x_init.clone().detach()
        g = d.rand1("guidance").to(z.device) * (g_max - g_min) + g_min        # FD:284
        with torch.no_grad():                                                 # FD:288-324
            for t in sch.timesteps[int(start_idx):]:
                tt = torch.tensor([t], device=z.device).repeat(z.shape[0])
                x_ = sch.scale_model_input(x, t)
                e_c = self.teacher_denoiser(sample=x_, timestep=tt, conditioning=conditioning,
                                            down_intrablock_additional_residuals=res, *args, **kwargs)
                e_u = self.teacher_denoiser(sample=x_, timestep=tt, conditioning=uncond,
                                            down_intrablock_additional_residuals=res, *args, **kwargs)
                e = g * e_c + (1 - g) * e_u
                x = sch.step(e, t, x, return_dict=False)[0]
        teacher_output = x
        student_output = c_skip * x_init + c_out * x0_s                       # FD:328
        l_distill = distill_loss(cfg.distill_loss_type, student_output, teacher_output, self.vae,
                                 getattr(self, "lpips", None))
        loss = l_distill * cfg.distill_loss_scale[K_step]
        self.terms = {"distill": l_distill.detach(), "K_step": K_step, "guidance": float(g)}
        if cfg.use_dmd_loss:                                                  # FD:335-345
            l_dmd = self.dmd_loss(d, student_output, student_conditioning, conditioning, uncond, K_step, res)
            self.terms["dmd"] = l_dmd.detach()
            loss = loss + l_dmd * cfg.dmd_loss_scale[K_step]
        if self.discriminator is not None:
            gan = self.gan_loss(d, z, student_output, teacher_output, conditioning, step, res)   # FD:347
        else:
            gan = [0, 0]   # the reference crashes here without a discriminator; used by bench.py's cpu_baseline
        self.terms["gan_G"] = gan[0].detach() if torch.is_tensor(gan[0]) else gan[0]
        self.terms["gan_D"] = gan[1].detach() if torch.is_tensor(gan[1]) else gan[1]
        loss = loss + cfg.adversarial_loss_scale[K_step] * gan[0]             # FD:357
        return {"loss": [loss, gan[1]], "teacher_output": teacher_output,
                "student_output": student_output, "noisy_sample": x_init,
                "start_timestep": start_t[0].item()}

    def dmd_loss(self, d, s, student_cond, cond, uncond, K_step, res=None):
        """FD:401-499."""
        cfg, sch = self.config, self.teacher_noise_scheduler
        noise = d.randn_like("dmd_noise", s)
        t = d.randint("dmd_t", 0, sch.config.num_train_timesteps, (s.shape[0],), s.device)
        noisy = sch.add_noise(s, noise, t)
        with torch.no_grad():
            e_c = self.teacher_denoiser(sample=noisy, timestep=t, conditioning=cond,
                                        down_intrablock_additional_residuals=res)
            e_u = self.teacher_denoiser(sample=noisy, timestep=t, conditioning=uncond,
                                        down_intrablock_additional_residuals=res)
            e_f = self.student_denoiser(sample=noisy, timestep=t, conditioning=student_cond,
                                        down_intrablock_additional_residuals=res)
            g = (d.rand1("dmd_guidance").to(s.device)
                 * (cfg.guidance_scale_max[K_step] - cfg.guidance_scale_min[K_step])
                 + cfg.guidance_scale_min[K_step])
        real = g * e_c + (1 - g) * e_u
        score_real, score_fake = -real, -e_f
        a = sch.alphas_cumprod.to(device=s.device, dtype=s.dtype)[t]
        b = 1.0 - a
        coeff = (score_fake - score_real) * b.view(-1, 1, 1, 1) ** 0.5 / a.view(-1, 1, 1, 1) ** 0.5
        x0 = predicted_x0_eps(real, t, noisy, self.sqrt_alpha_cumprod, self.sigmas, s)
        w = 1.0 / ((s - x0).abs().mean([1, 2, 3], keepdim=True) + 1e-5).detach()
        return F.mse_loss(s, (s - w * coeff).detach(), reduction="mean")

    def gan_loss(self, d, z, s, teacher_output, conditioning, step, res=None):
        """FD:501-667."""
        cfg, sch = self.config, self.teacher_noise_scheduler
        self.disc_update_counter += 1
        noise = d.randn_like("gan_noise", s)
        real = teacher_output if cfg.use_teacher_as_real else z
        prob = torch.tensor([0.25, 0.25, 0.25, 0.25])
        idx = d.multinomial("gan_idx", prob, s.shape[0], replacement=True).to(s.device)
        ts = torch.tensor([10, 250, 500, 750], device=s.device, dtype=torch.long)[idx]
        noisy_fake = sch.add_noise(s, noise, ts)
        noisy_real = sch.add_noise(real, noise, ts)
        x = torch.cat([noisy_fake, noisy_real], dim=0)
        if conditioning is not None:
            conditioning = {"cond": {k: torch.cat([v, v], dim=0) for k, v in conditioning["cond"].items()}}
        t2 = torch.cat([ts, ts], dim=0)
        if self.adapter:                                                      # FD:555-560 (mutates the caller's list)
            for k, v in enumerate(res):
                res[k] = torch.cat([v, v], dim=0)
        else:
            res = None
        feat = self.teacher_denoiser(sample=x, timestep=t2, conditioning=conditioning,
                                     down_intrablock_additional_residuals=res, return_intermediate=True)
        f_fake, f_real = feat.chunk(2, dim=0)
        return gan_losses(cfg.gan_loss_type, self.discriminator, f_fake, f_real, step, s.size(0),
                          noise.device)

    # ---- FD:754-915: few-step sampler (student; optionally the teacher's own sampler next to it) ----
    @torch.no_grad()
    def sample(self, z, num_steps=20, guidance_scale=1.0, teacher_guidance_scale=5.0, conditioner_inputs=None,
               uncond_conditioner_inputs=None, max_samples=None, verbose=False, log_teacher_samples=False,
               adapter_conditioning_scale=1.0):
        self.teacher_noise_scheduler.set_timesteps(num_steps)                      # FD:781
        ss = self.sampling_noise_scheduler
        try:                                                                       # FD:783-788
            ss.set_timesteps(timesteps=self.teacher_noise_scheduler.timesteps)
        except Exception:
            ss.set_timesteps(num_steps)
        sample = z
        cond = self._cond(conditioner_inputs, set_ucg_rate_zero=True, device=z.device)        # FD:793-795
        if uncond_conditioner_inputs is not None:                                              # FD:798-805
            uncond = self._cond(uncond_conditioner_inputs, set_ucg_rate_zero=True, device=z.device)
        else:
            uncond = self._cond(conditioner_inputs, ucg_keys=self.config.ucg_keys, device=z.device)
        if max_samples is not None:                                                            # FD:807-818
            sample = sample[:max_samples]
            if cond:
                cond["cond"] = {k: v[:max_samples] for k, v in cond["cond"].items()}
                uncond["cond"] = {k: v[:max_samples] for k, v in uncond["cond"].items()}
        if self.adapter:                                                                       # FD:820-829
            res = self.adapter(conditioner_inputs[self.adapter_input_key])
            for k, v in enumerate(res):
                res[k] = v * adapter_conditioning_scale
        else:
            res = None
        sample_init = sample
        sample = sample * ss.init_noise_sigma                                                  # FD:833
        for t in ss.timesteps:                                                                 # FD:834-867
            x = ss.scale_model_input(sample, t)
            tt = t.to(z.device).repeat(x.shape[0])
            e_c = self.student_denoiser(sample=x, timestep=tt, conditioning=cond,
                                        down_intrablock_additional_residuals=res)
            e_u = self.student_denoiser(sample=x, timestep=tt, conditioning=uncond,
                                        down_intrablock_additional_residuals=res)
            e = guidance_scale * e_c + (1 - guidance_scale) * e_u
            sample = ss.step(e, t, sample, return_dict=False)[0]
        decoded = self.vae.decode(sample) if self.vae is not None else sample                  # FD:865-868
        decoded_ref = None
        if log_teacher_samples:                                                                # FD:876-913
            ts = self.teacher_sampling_noise_scheduler
            ts.set_timesteps(num_steps)
            ref = sample_init * ts.init_noise_sigma
            for t in ts.timesteps:
                x = ts.scale_model_input(ref, t)
                tt = t.to(z.device).repeat(x.shape[0])
                e_c = self.teacher_denoiser(sample=x, timestep=tt, conditioning=cond,
                                            down_intrablock_additional_residuals=res)
                e_u = self.teacher_denoiser(sample=x, timestep=tt, conditioning=uncond,
                                            down_intrablock_additional_residuals=res)
                e = teacher_guidance_scale * e_c + (1 - teacher_guidance_scale) * e_u
                ref = ts.step(e, t, ref, return_dict=False)[0]
                decoded_ref = self.vae.decode(ref) if self.vae is not None else ref            # FD:910-913 (every step)
        return decoded, decoded_ref

    # ---- FD:917-1019 ----
    def log_samples(self, batch, input_shape=None, guidance_scale=1.0, teacher_guidance_scale=5.0, max_samples=8,
                    num_steps=20, device="cpu", log_teacher_samples=False, conditioner_inputs=None,
                    conditioner_uncond_inputs=None, adapter_conditioning_scale=1.0):
        import copy as _copy
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
            batch_uncond = _copy.deepcopy(batch)
            batch_uncond.update(conditioner_uncond_inputs)
            N = min(N, m)
        else:
            batch_uncond = None
        if input_shape is None:
            if self.vae is not None:                                                           # FD:977-984
                px = batch[self.vae.config.input_key].shape[2:]
                input_shape = (self.vae.latent_channels, px[0] // self.vae.downsampling_factor,
                               px[1] // self.vae.downsampling_factor)
            else:
                raise ValueError("input_shape must be passed when no VAE is used in the model")   # FD:985-988
        for n in num_steps:
            z = torch.randn(N, *input_shape).to(device)                                        # FD:992
            samples, samples_ref = self.sample(z, num_steps=n, conditioner_inputs=batch,
                                               uncond_conditioner_inputs=batch_uncond, guidance_scale=guidance_scale,
                                               teacher_guidance_scale=teacher_guidance_scale, max_samples=N,
                                               log_teacher_samples=log_teacher_samples,
                                               adapter_conditioning_scale=adapter_conditioning_scale)
            logs[f"samples_{n}_steps/{self.sampling_noise_scheduler.__class__.__name__}_{guidance_scale}_cfg/student"] = samples
            if samples_ref is not None:
                logs[f"samples_{n}_steps/{self.teacher_sampling_noise_scheduler.__class__.__name__}"
                     f"_{teacher_guidance_scale}_cfg/teacher"] = samples_ref
        return logs
