"""FlashDiffusion -- MI355X drop-in for the reference's distillation model
(/root/reference/src/flash/models/flash/flash_diffusion_model.py:38, "FD"): same constructor
(FD:40-60), same forward(batch, batch_idx=0, step=0, *args, **kwargs) -> dict contract (FD:179,
360-366), same loss semantics (FD:368-667), same config fields (flash_diffusion_config.py:9-105).

What differs is where the work runs: the denoisers are MiUNet2DConditionModel plans (hand-written
HIP kernels behind libfdmi.so), the teacher loop fuses CFG + DPM-Solver++ update into one kernel per
step, and the teacher pass is issued BEFORE the student pass (legal: the teacher is frozen and its
inputs do not depend on the student, FD:282-324; the SD3 twin of the reference already orders it this
way, flash_sd3/flash_diffusion_model.py:281-323) so that a data-parallel trainer can overlap the
previous step's gradient all-reduce + AdamW with it.

Random draws go through `Draws` so tests can inject the exact values the reference consumed."""
from __future__ import annotations

import logging
from copy import deepcopy
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import copy

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# ----------------------------------------------------------------------------------------------------
_log = logging.getLogger(__name__)
_torch_disc_warned = False


def _warn_torch_discriminator(reason: str) -> None:
    """The discriminator head stays a torch module (its convolutions run on torch's ROCm kernels, not libfdmi.so): log it once."""
    global _torch_disc_warned
    if not _torch_disc_warned:
        _torch_disc_warned = True
        _log.warning("flash_diffusion_amd: the discriminator head is NOT on the HIP path and runs as the caller's torch module (%s)", reason)



@dataclass
class FlashDiffusionConfig:
    """Field-for-field mirror of the reference FlashDiffusionConfig (flash_diffusion_config.py:9-105),
    including the scalar -> per-stage list broadcasting of its __post_init__."""
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
        assert self.distill_loss_type in ("l2", "l1", "lpips")
        assert self.timestep_distribution in ("gaussian", "uniform", "mixture")
        assert self.gan_loss_type in ("hinge", "vanilla", "non-saturating", "wgan", "lsgan")
        if isinstance(self.mixture_num_components, int):
            self.mixture_num_components = [self.mixture_num_components] * n
        for f in ("guidance_scale_min", "guidance_scale_max", "mixture_var", "distill_loss_scale",
                  "dmd_loss_scale", "adversarial_loss_scale"):
            if isinstance(getattr(self, f), float):
                setattr(self, f, [getattr(self, f)] * n)
        if self.mode_probs is None:
            self.mode_probs = [[1 / m] * m for m in self.mixture_num_components]
        for i in range(n):
            assert len(self.mode_probs[i]) == self.mixture_num_components[i], \
                f"Number of mode probabilities must match number of mixture components for stage {i}"
        assert len(self.K) == len(self.num_iterations_per_K), "Number of timesteps must match number of iterations"
        assert len(self.K) == len(self.mode_probs), "Number of timesteps must match number of mode probabilities"

    def to_dict(self):
        return dict(self.__dict__)


class Draws:
    """Source of every random draw of one forward.  values=None: draw from torch's RNG (noise on the
    input's device, categorical / uniform scalars on the host like the reference, FD:167, 285, 454, 528)
    and record; values=dict: replay (parity tests inject what the reference consumed)."""

    def __init__(self, values: Optional[Dict[str, torch.Tensor]] = None):
        self.inject = values is not None
        self.values: Dict[str, torch.Tensor] = dict(values) if values else {}
        self._count: Dict[str, int] = {}

    def _get(self, name, make):
        i = self._count.get(name, 0)
        self._count[name] = i + 1
        k = name if i == 0 else f"{name}#{i}"
        if self.inject:
            return self.values[k]
        v = make()
        self.values[k] = v
        return v

    def randn_like(self, name, x):
        return self._get(name, lambda: torch.randn_like(x)).to(device=x.device, dtype=x.dtype)

    def multinomial(self, name, prob, n, replacement=False, generator=None):
        return self._get(name, lambda: torch.multinomial(prob, n, replacement=replacement, generator=generator))

    def rand1(self, name):
        return self._get(name, lambda: torch.rand(1))

    def randint(self, name, lo, hi, shape, device):
        return ops.upload(self._get(name, lambda: torch.randint(lo, hi, shape)), device)


def shared_start_seed_for(seed: int, step: int) -> int:
    """seed of the per-forward start-index generator of data-parallel training: a splitmix64-style hash of (the seed every rank
    shares, the step counter) -- consecutive integers do not go into the Mersenne twister as they are"""
    m = (1 << 64) - 1
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(step) + 0x632BE59BD9B4E019) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return (z ^ (z >> 31)) & ((1 << 62) - 1)



class TensorConditioner(nn.Module):
    """Stand-in for ConditionerWrapper (embedders/conditioners_wrapper.py:39-91) when the batch already
    carries the embeddings (`crossattn` [B,L,D], optional `vector`, `concat`).  Keys listed in `ucg_keys`
    are zeroed -- what force_zero_embedding yields for a dropped conditioner
    (clip_embedder_model.py:93-94).  Text encoders themselves are out of scope."""

    def __init__(self, input_key="text"):
        super().__init__()
        self.input_key = input_key

    def forward(self, batch, ucg_keys=None, set_ucg_rate_zero=False, *args, **kwargs):
        drop = ucg_keys is not None and self.input_key in ucg_keys
        cond = {}
        for k in ("crossattn", "vector", "concat"):
            if batch.get(k, None) is not None:
                cond[k] = torch.zeros_like(batch[k]) if drop else batch[k]
        if batch.get("attention_mask", None) is not None:   # T5 key mask of the PixArt path (TW:75): never dropped
            cond["attention_mask"] = batch["attention_mask"]
            if batch.get("attention_mask_lens", None) is not None:     # its host-side prefix lengths (no device read per step)
                cond["attention_mask_lens"] = batch["attention_mask_lens"]
        return {"cond": cond}


# ----------------------------------------------------------------------------------------------------
def _tensors_of(obj):
    """every tensor inside nested dicts / lists / tuples"""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


class _PerSampleAffine(torch.autograd.Function):
    """out[b] = ca[b] * x[b] + cb[b] * eps[b]  (fused; gradient flows to eps only)."""

    @staticmethod
    def forward(ctx, eps, x, ca, cb):
        ctx.save_for_backward(cb)
        return ops.add_noise(x.contiguous(), eps.contiguous(), ca, cb)

    @staticmethod
    def backward(ctx, g):
        (cb,) = ctx.saved_tensors
        g = g.contiguous()
        return ops.add_noise(g, g, cb, torch.zeros_like(cb)), None, None, None


class _DistillLoss(torch.autograd.Function):
    """FD:368-382 as one fused reduction kernel (+ one element-wise kernel for the gradient)."""

    @staticmethod
    def forward(ctx, s, t, l1):
        from ._lib import check, lib, ptr, stream_ptr
        s, t = s.contiguous(), t.contiguous()
        out = torch.empty(1, dtype=torch.float32, device=s.device)
        check(lib().fdmi_distill_loss(ptr(s), ptr(t), s.numel(), int(l1), ptr(out), stream_ptr()))
        ctx.save_for_backward(s, t)
        ctx.l1 = int(l1)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        from ._lib import check, lib, ptr, stream_ptr
        s, t = ctx.saved_tensors
        ds = torch.empty_like(s)
        check(lib().fdmi_distill_grad(ptr(s), ptr(t), s.numel(), ctx.l1, 1.0 / s.numel(), ptr(ds), stream_ptr()))
        return ds * g, None, None   # g: device scalar (no host sync)


class _DmdLoss(torch.autograd.Function):
    """FD:459-499 fused: per-sample weight 1/(mean|s - x0(real)| + 1e-5), coefficient
    (real - fake) sqrt(1-abar)/sqrt(abar), loss mean((w coeff)^2) and dL/ds = 2 w coeff / N."""

    @staticmethod
    def forward(ctx, s, noisy, real, fake, inv_a, ms_a, kb):
        from ._lib import check, lib, ptr, stream_ptr
        B = s.shape[0]
        s = s.contiguous()
        w = torch.empty(B, dtype=torch.float32, device=s.device)
        grad = torch.empty_like(s)
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        check(lib().fdmi_dmd_loss(ptr(s), ptr(noisy.contiguous()), ptr(real.contiguous()), ptr(fake.contiguous()),
                                  ptr(inv_a), ptr(ms_a), ptr(kb), ptr(w), ptr(grad), ptr(loss), B, s.numel() // B,
                                  stream_ptr()))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None


def gaussian_mixture_pmf(K, locs, var, mode_probs):
    p = [sum(mode_probs[j] * torch.exp(-torch.tensor([(i - loc) ** 2 / var])) for j, loc in enumerate(locs))
         for i in range(K)]
    p = torch.tensor(p)
    return p / torch.sum(p)


def ss_name(s):
    return s.__class__.__name__


class FlashDiffusion(nn.Module):
    calls_before_student = True   # forward() calls the trainer's before_student hook right before the student call

    def __init__(self, config: FlashDiffusionConfig, student_denoiser, teacher_denoiser=None,
                 teacher_noise_scheduler=None, teacher_sampling_noise_scheduler=None, sampling_noise_scheduler=None,
                 vae=None, conditioner=None, adapter=None, discriminator: nn.Module = None, lpips_model: nn.Module = None):
        super().__init__()
        self.config = config
        self.input_key = config.input_key
        self.student_denoiser = student_denoiser
        self.teacher_denoiser = teacher_denoiser
        self.teacher_noise_scheduler = teacher_noise_scheduler
        self.teacher_sampling_noise_scheduler = teacher_sampling_noise_scheduler
        self.sampling_noise_scheduler = sampling_noise_scheduler
        # VAE (FD:67): optional, any frozen module with the surface of the reference's AutoencoderKLDiffusers that the step touches
        # (vae/autoencoderKL.py:11-128: config.input_key, encode, decode, latent_channels, downsampling_factor).  On the HIP path:
        # nets.MiAutoencoderKLDiffusers (decoder = a plan of libfdmi.so); any other module works as the
        # caller's torch module.  With vae=None the batch carries latents (the benchmark's case, FD:184-185).
        self.vae = vae
        # T2I adapter (FD:91-94): any frozen module mapping batch[adapter_input_key] to one residual per UNet down block
        # (on the HIP path: nets.MiT2IAdapter); its features are threaded to every denoiser
        # call exactly as the reference does (FD:207-218, 264, 301, 310, 436-450, 555-567, 820-899)
        self.adapter = adapter
        self.adapter_conditioning_scale = config.adapter_conditioning_scale
        self.adapter_input_key = config.adapter_input_key
        self.conditioner = conditioner
        if isinstance(discriminator, nn.Sequential) and not hasattr(discriminator, "convert"):
            from .discriminator import MiDiscriminator
            # run the reference's PatchGAN heads on the HIP kernels; anything exotic stays the caller's torch module -- and says so
            # ONCE (a head that silently trained on MIOpen was indistinguishable from one on libfdmi.so)
            if all(isinstance(m, (nn.Conv2d, nn.SiLU, nn.GroupNorm, nn.Flatten)) for m in discriminator):
                try:
                    discriminator = MiDiscriminator.convert(discriminator)
                except Exception as e:   # an unsupported layer geometry (MiDiscriminator.convert raises with the reason)
                    _warn_torch_discriminator(f"MiDiscriminator.convert failed: {e!r}")
            else:
                _warn_torch_discriminator("layers outside Conv2d / SiLU / GroupNorm / Flatten: " + ", ".join(
                    sorted({type(m).__name__ for m in discriminator if not isinstance(m, (nn.Conv2d, nn.SiLU, nn.GroupNorm, nn.Flatten))})))
        if getattr(discriminator, "precision", None) is not None and \
                getattr(student_denoiser, "config_dict", {}).get("precision") == "fp32":
            discriminator.precision = "fp32"   # an fp32 validation student: the head runs the validation kernels too
        self.discriminator = discriminator
        for f in ("guidance_scale_min", "guidance_scale_max", "ucg_keys", "K", "num_iterations_per_K",
                  "distill_loss_type", "timestep_distribution", "mixture_num_components", "mixture_var",
                  "use_dmd_loss", "dmd_loss_scale", "distill_loss_scale", "adversarial_loss_scale", "gan_loss_type",
                  "mode_probs", "use_teacher_as_real", "use_empty_prompt"):
            setattr(self, f, getattr(config, f))
        if self.distill_loss_type == "lpips":                  # FD:102-103: self.lpips = lpips.LPIPS(net="vgg")
            if vae is None:
                raise ValueError("distill_loss_type='lpips' decodes both outputs: a vae is required (FD:394-395)")
            if lpips_model is None:
                # the reference builds lpips.LPIPS(net="vgg") with its PRETRAINED weights (FD:102-103, and fails at import time
                # without the package).  The HIP twin (nets.MiLPIPS: same architecture, same state_dict names) takes those
                # weights; without the `lpips` package there is nothing meaningful to train against, so this raises as the
                # reference does -- pass lpips_model=MiLPIPS() explicitly for a run on placeholder weights (bench.py does)
                from .nets import MiLPIPS
                try:
                    import lpips as _lpips
                except ImportError as e:
                    raise ImportError("distill_loss_type='lpips' with lpips_model=None needs the `lpips` package for the pretrained "
                                      "VGG16 / linear-layer weights (FD:102-103); pass lpips_model=nets.MiLPIPS() loaded from a "
                                      "checkpoint (or left on placeholder weights, explicitly) instead") from e
                ref_lpips = _lpips.LPIPS(net="vgg")
                lpips_model = MiLPIPS(precision="fp32" if getattr(student_denoiser, "config_dict", {}).get("precision") == "fp32"
                                      else "bf16")
                lpips_model.load_state_dict(ref_lpips.state_dict())      # (drops lpips' duplicate `lins.*` entries)
                lpips_model.freeze()
            self.lpips = lpips_model
        self.iter_steps = 0
        self.disc_update_counter = 0
        self.disc_backbone = self.teacher_denoiser
        self.K_steps = np.cumsum(self.num_iterations_per_K)
        self.K_prev = self.K[0]
        ac = teacher_noise_scheduler.alphas_cumprod
        self.register_buffer("sqrt_alpha_cumprod", torch.sqrt(ac))
        self.register_buffer("sigmas", torch.sqrt(1 - ac))
        self._alpha_all_positive = bool((ac > 0).all())   # checked once on the host table (no per-step device round trip)
        self.draws: Optional[Draws] = None
        self.last_draws: Optional[Draws] = None
        self.terms: Dict[str, Any] = {}
        self.fixed_start_idx: Optional[int] = None     # benchmark: pin the teacher-step count
        self.fixed_guidance: Optional[float] = None
        self.shared_start_seed: Optional[int] = None   # data-parallel training: see share_start_idx()

    # ---- helpers -----------------------------------------------------------------------------------
    def share_start_idx(self, seed: Optional[int]):
        """Data-parallel training: the reference draws the start index per rank (FD:167), so every step would wait
        for the rank that drew the longest teacher loop.  With a host generator seeded IDENTICALLY on every rank (the trainer
        broadcasts rank 0's seed once) all ranks draw the same index from the same pmf at every step -- no communication in the
        step, and each rank's marginal distribution of start indices is the reference's.  Noise, guidance scale and GAN draws
        stay per rank.  seed=None switches back to per-rank draws.  The generator of a forward holds no state between forwards: it
        is seeded with a hash of (seed, step), where `step` is `self.shared_start_step` when the trainer sets it -- TrainingPipeline
        sets its own count of training_step calls before every forward, a counter a rank-local forward (validation or sample
        logging through the model on rank 0 only, a retried batch) does not advance -- and otherwise the model's forward counter
        (FD:181), which such a forward DOES advance: without a trainer-owned counter the ranks then draw different indices from that
        point on and every step waits for the longest teacher loop, with no error raised.  Neither counter is part of
        a checkpoint; a resumed job restarts the sequence on all ranks alike.  DEVIATION from the reference, on by default in
        TrainingPipeline when world > 1: the global batch of an optimizer step sees ONE start timestep instead of `world`."""
        self.shared_start_seed = None if seed is None else int(seed)

    def _shared_start_generator(self):
        if getattr(self, "shared_start_seed", None) is None:
            return None
        step = getattr(self, "shared_start_step", None)
        return torch.Generator().manual_seed(shared_start_seed_for(self.shared_start_seed, self.iter_steps if step is None else step))

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def on_train_batch_end(self, batch, *a, **k):
        pass

    def _get_conditioning(self, batch, ucg_keys=None, set_ucg_rate_zero=False, *args, **kwargs):
        if self.conditioner is None:
            return None
        return self.conditioner(batch, ucg_keys=ucg_keys, set_ucg_rate_zero=set_ucg_rate_zero, vae=self.vae,
                                *args, **kwargs)

    def _timestep_pmf(self, K, K_step):
        """FD:141-165"""
        if self.timestep_distribution == "uniform":
            return torch.ones(K) / K
        if self.timestep_distribution == "gaussian":
            p = torch.tensor([torch.exp(-torch.tensor([(i - K / 2) ** 2 / K])) for i in range(K)])
            return p / torch.sum(p)
        M = self.mixture_num_components[K_step]
        locs = [i * (K // M) for i in range(M)]
        return gaussian_mixture_pmf(K, locs, self.mixture_var[K_step], self.mode_probs[K_step])

    def _get_timesteps(self, d: Draws, num_samples, K, K_step, device):
        """FD:135-177: ONE start index per batch, drawn on the host."""
        self.teacher_noise_scheduler.set_timesteps(K)
        if self.fixed_start_idx is not None:
            start_idx = torch.tensor([self.fixed_start_idx])
        else:
            start_idx = d.multinomial("start_idx", self._timestep_pmf(K, K_step), 1, generator=self._shared_start_generator())
        t0 = self.teacher_noise_scheduler.timesteps[start_idx]
        self._start_t_host = int(t0.reshape(-1)[0])   # forward() reports it without a device round trip
        return start_idx, ops.upload(t0.reshape(-1)[:1].repeat(num_samples), device)   # (no blocking copy: ops.upload)

    @staticmethod
    def _scalings_for_boundary_conditions(timestep, sigma_data=0.5):
        """FD:710-716"""
        c_skip = sigma_data ** 2 / ((timestep / 0.1) ** 2 + sigma_data ** 2)
        c_out = (timestep / 0.1) / ((timestep / 0.1) ** 2 + sigma_data ** 2) ** 0.5
        return c_skip, c_out

    @staticmethod
    def _cat_cond(cond, uncond):
        if cond is None or uncond is None or set(cond["cond"]) != set(uncond["cond"]):
            return None
        # ("attention_mask_lens": host integers beside a key mask -- lists concatenate)
        return {"cond": {k: (torch.cat([cond["cond"][k], uncond["cond"][k]], dim=0) if torch.is_tensor(cond["cond"][k])
                             else list(cond["cond"][k]) + list(uncond["cond"][k])) for k in cond["cond"]}}

    def _adapter_residuals(self, inputs, scale):
        """FD:207-218 / 820-829: list of residual tensors (scaled), or None without an adapter"""
        if not self.adapter:
            return None
        with torch.no_grad():
            return [v * scale for v in self.adapter(inputs[self.adapter_input_key])]

    @staticmethod
    def _dup(res):
        """[v | v] per residual: the 2B-batched calls see the same control features in both halves (FD:555-557)"""
        return None if res is None else [torch.cat([v, v], dim=0) for v in res]

    # ---- few-step sampler (FD:754-915) and its logging wrapper (FD:917-1019): SURVEY 8(f) "next" row 1 ----
    def _cfg_pair(self, net, x, tt, cond, uncond, g, ctx_cache=None, res=None):
        """g * eps(cond) + (1 - g) * eps(uncond).  The reference makes two calls per step (FD:838-858); every layer is
        per-sample, so ONE call on [x | x] with [cond | uncond] gives the same two predictions.  g == 1 multiplies the
        unconditional branch by zero: it is skipped."""
        if float(g) == 1.0:
            return net(sample=x, timestep=tt, conditioning=cond, down_intrablock_additional_residuals=res)
        both = self._cat_cond(cond, uncond)
        if both is None:  # conditionings with different keys cannot share a batch: two calls, as the reference does
            e_c = net(sample=x, timestep=tt, conditioning=cond, down_intrablock_additional_residuals=res)
            e_u = net(sample=x, timestep=tt, conditioning=uncond, down_intrablock_additional_residuals=res)
            return ops.axpby(e_c.contiguous(), float(g), e_u.contiguous(), 1.0 - float(g))
        kw = {}
        if ctx_cache is not None and getattr(net, "supports_ctx_cache", False) and not getattr(net, "lora_rank", 0):
            kw["ctx_cache"] = ctx_cache
        e = net(sample=torch.cat([x, x], dim=0), timestep=torch.cat([tt, tt], dim=0),
                conditioning=both, down_intrablock_additional_residuals=self._dup(res), **kw)
        e_c, e_u = e.chunk(2, dim=0)
        return ops.axpby(e_c.contiguous(), float(g), e_u.contiguous(), 1.0 - float(g))

    @torch.no_grad()
    def sample(self, z, num_steps=20, guidance_scale=1.0, teacher_guidance_scale=5.0, conditioner_inputs=None,
               uncond_conditioner_inputs=None, max_samples=None, verbose=False, log_teacher_samples=False,
               adapter_conditioning_scale=1.0):
        """Same contract as the reference (FD:755-915): returns (samples, teacher_samples or None) -- latents, or images when a
        vae is attached.  Student: `sampling_noise_scheduler` (LCM) on the teacher scheduler's schedule when it accepts
        custom timesteps; teacher (optional): `teacher_sampling_noise_scheduler` with its own CFG scale."""
        assert self.sampling_noise_scheduler is not None, "sample() needs a sampling_noise_scheduler (e.g. LCMScheduler)"
        self.teacher_noise_scheduler.set_timesteps(num_steps)
        ss = self.sampling_noise_scheduler
        try:
            ss.set_timesteps(timesteps=self.teacher_noise_scheduler.timesteps)
        except Exception:
            ss.set_timesteps(num_steps)
        sample = z
        cond = self._get_conditioning(conditioner_inputs, set_ucg_rate_zero=True, device=z.device)
        if uncond_conditioner_inputs is not None:
            uncond = self._get_conditioning(uncond_conditioner_inputs, set_ucg_rate_zero=True, device=z.device)
        else:
            uncond = self._get_conditioning(conditioner_inputs, ucg_keys=self.ucg_keys, device=z.device)
        if max_samples is not None:
            sample = sample[:max_samples]
            if cond:
                cond["cond"] = {k: v[:max_samples] for k, v in cond["cond"].items()}
                uncond["cond"] = {k: v[:max_samples] for k, v in uncond["cond"].items()}
        res = self._adapter_residuals(conditioner_inputs, adapter_conditioning_scale)        # FD:820-829
        sample_init = sample
        sample = (sample * ss.init_noise_sigma).float().contiguous()
        for t in ss.timesteps:
            x = ss.scale_model_input(sample, t)
            tt = torch.full((x.shape[0],), float(t), device=z.device)
            e = self._cfg_pair(self.student_denoiser, x, tt, cond, uncond, guidance_scale, res=res)
            sample = ss.step(e, t, sample, return_dict=False)[0]
        if self.vae is not None:                                                 # FD:865-868
            sample = self.vae.decode(sample)
        decoded_ref = None
        if log_teacher_samples:
            ts = self.teacher_sampling_noise_scheduler
            assert ts is not None, "log_teacher_samples needs a teacher_sampling_noise_scheduler"
            ts.set_timesteps(num_steps)
            ref = (sample_init * ts.init_noise_sigma).float().contiguous()
            for it, t in enumerate(ts.timesteps):
                x = ts.scale_model_input(ref, t)
                tt = torch.full((x.shape[0],), float(t), device=z.device)
                e = self._cfg_pair(self.teacher_denoiser, x, tt, cond, uncond, teacher_guidance_scale,
                                   ctx_cache="fill" if it == 0 else "reuse", res=res)
                ref = ts.step(e, t, ref, return_dict=False)[0]
            decoded_ref = self.vae.decode(ref) if self.vae is not None else ref  # FD:910-913 (the reference decodes every step)
        return sample, decoded_ref

    def log_samples(self, batch, input_shape=None, guidance_scale=1.0, teacher_guidance_scale=5.0, max_samples=8,
                    num_steps=20, device="cpu", log_teacher_samples=False, conditioner_inputs=None,
                    conditioner_uncond_inputs=None, adapter_conditioning_scale=1.0):
        """FD:917-1019 (`input_shape` = latent shape; inferred from the VAE when one is attached, else mandatory as in the
        reference's ValueError branch)."""
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
            if self.vae is not None:                                             # FD:977-984
                px = batch[self.vae.config.input_key].shape[2:]
                input_shape = (self.vae.latent_channels, px[0] // self.vae.downsampling_factor,
                               px[1] // self.vae.downsampling_factor)
            else:
                raise ValueError("input_shape must be passed when no VAE is used in the model")
        for n in num_steps:
            z = torch.randn(N, *input_shape).to(device)
            samples, samples_ref = self.sample(z, num_steps=n, conditioner_inputs=batch,
                                               uncond_conditioner_inputs=batch_uncond, guidance_scale=guidance_scale,
                                               teacher_guidance_scale=teacher_guidance_scale, max_samples=N,
                                               log_teacher_samples=log_teacher_samples,
                                               adapter_conditioning_scale=adapter_conditioning_scale)
            logs[f"samples_{n}_steps/{ss_name(self.sampling_noise_scheduler)}_{guidance_scale}_cfg/student"] = samples
            if samples_ref is not None:
                logs[f"samples_{n}_steps/{ss_name(self.teacher_sampling_noise_scheduler)}"
                     f"_{teacher_guidance_scale}_cfg/teacher"] = samples_ref
        return logs

    def _teacher_stream(self, like):
        """side stream for the teacher loop (one per model and device); None off the GPU"""
        if not (torch.is_tensor(like) and like.is_cuda):
            return None
        st = getattr(self, "_side_streams", None)
        if st is None:
            st = self._side_streams = {}
        if like.device not in st:
            # (dev A/B: FDMI_TEACHER_PRIORITY = -1 gives the loop's queue the higher hardware priority, 0 = default)
            st[like.device] = torch.cuda.Stream(device=like.device, priority=int(os.environ.get("FDMI_TEACHER_PRIORITY", "0")))
        return st[like.device]

    def _teacher_cfg(self, x, tt, cond, uncond, cfg_cond, *args, ctx_cache=None, res=None, **kwargs):
        """The reference evaluates the frozen teacher twice per step, once per conditioning (FD:297-313).
        Every layer of the UNet is per-sample (GroupNorm included), so ONE call on the 2B batch
        [x | x] with [cond | uncond] gives the same two predictions with half the launches and twice
        the rows per GEMM."""
        if cfg_cond is None or not getattr(self, "batch_cfg", True):
            e_c = self.teacher_denoiser(sample=x, timestep=tt, conditioning=cond,
                                        down_intrablock_additional_residuals=res, *args, **kwargs)
            e_u = self.teacher_denoiser(sample=x, timestep=tt, conditioning=uncond,
                                        down_intrablock_additional_residuals=res, *args, **kwargs)
            return e_c, e_u
        if (ctx_cache is not None and getattr(self.teacher_denoiser, "supports_ctx_cache", False)
                and not os.environ.get("FDMI_NO_CTX_CACHE")):
            kwargs = dict(kwargs, ctx_cache=ctx_cache)  # same [cond | uncond] context at every step of this loop
        if (os.environ.get("FDMI_CFG_DEDUP", "1") == "1" and getattr(self.teacher_denoiser, "supports_cfg_halves", False)
                and res is None):   # [x | x]: layers before the first cross-attention once (A/B switch: FDMI_CFG_DEDUP=0)
            kwargs = dict(kwargs, cfg_halves=True)
        e = self.teacher_denoiser(sample=torch.cat([x, x], dim=0), timestep=torch.cat([tt, tt], dim=0),
                                  conditioning=cfg_cond, down_intrablock_additional_residuals=self._dup(res), *args,
                                  **kwargs)
        e_c, e_u = e.chunk(2, dim=0)
        return e_c.contiguous(), e_u.contiguous()

    def _x0_coeffs(self, t):
        """per-sample (1/alpha_t, -sigma_t/alpha_t) of the epsilon-branch of _predicted_x_0 (FD:731-742)."""
        al = self.sqrt_alpha_cumprod[t]
        sg = self.sigmas[t]
        assert self._alpha_all_positive or bool((al > 0).all()), \
            "alpha_t == 0 never occurs on the trailing schedules of the reference configs"
        return 1.0 / al, -sg / al

    # ---- forward (FD:179-366) ------------------------------------------------------------------------
    def forward(self, batch: Dict[str, Any], batch_idx=0, step=0, *args, **kwargs):
        sch = self.teacher_noise_scheduler
        d = self.draws if self.draws is not None else Draws()
        self.last_draws = d
        self.iter_steps += 1
        if self.vae is not None:                                                # FD:128-133, 182-183
            with torch.no_grad():
                z = self.vae.encode(batch[self.vae.config.input_key]).float().contiguous()
        else:
            z = batch[self.input_key].float().contiguous()
        B = z.shape[0]
        conditioning = self._get_conditioning(batch, set_ucg_rate_zero=True, *args, **kwargs)
        student_conditioning = self._get_conditioning(batch, *args, **kwargs)
        if self.use_empty_prompt and "text" in self.ucg_keys:
            ub = dict(batch)
            ub["text"] = [""] * len(batch["text"])
            uncond = self._get_conditioning(ub, set_ucg_rate_zero=True, *args, **kwargs)
        else:
            uncond = self._get_conditioning(batch, ucg_keys=self.ucg_keys, *args, **kwargs)
        res = self._adapter_residuals(batch, self.adapter_conditioning_scale)          # FD:207-218
        if self.iter_steps > self.K_steps[-1]:
            K_step = len(self.K) - 1
        else:
            K_step = int(np.argmax(self.iter_steps < self.K_steps))
        K = self.K[K_step]
        g_min, g_max = self.guidance_scale_min[K_step], self.guidance_scale_max[K_step]
        if K != self.K_prev:
            self.K_prev = K
            if getattr(self, "switch_teacher", False):  # the reference reads an attribute it never sets (FD:230)
                if getattr(self, "before_student", None) is not None:
                    self.before_student()   # the copy reads the student: outstanding backward / AdamW first
                self.teacher_denoiser = deepcopy(self.student_denoiser)
                self.teacher_denoiser.freeze()

        noise = d.randn_like("noise", z)
        start_idx, start_t = self._get_timesteps(d, B, K, K_step, z.device)
        si = int(start_idx)
        if si == 0:
            x_init = noise if sch.init_noise_sigma == 1.0 else noise * sch.init_noise_sigma
        else:
            with torch.no_grad():
                x_init = sch.add_noise(z, noise, start_t)
        x_in = sch.scale_model_input(x_init, start_t)
        if self.fixed_guidance is not None:
            g = float(self.fixed_guidance)
        else:
            g = float(d.rand1("guidance")) * (g_max - g_min) + g_min

        # ---- teacher: n = K - start_idx CFG denoising steps, no grad (FD:288-324) ----
        def run_teacher():
            with torch.no_grad():
                x = x_init
                fused = hasattr(sch, "fused_cfg_step")
                cfg_cond = self._cat_cond(conditioning, uncond)
                one_call = (os.environ.get("FDMI_TEACHER_LOOP", "1") == "1" and cfg_cond is not None
                            and getattr(self, "batch_cfg", True) and hasattr(sch, "loop_coefficients")
                            and hasattr(self.teacher_denoiser, "teacher_loop") and not args and set(kwargs) <= {"device"}
                            and res is None
                            and set(cfg_cond["cond"]) <= ({"crossattn", "vector", "attention_mask", "attention_mask_lens"}
                                                          if getattr(self.teacher_denoiser, "teacher_loop_takes_mask", False)
                                                          else {"crossattn", "vector"})
                            and (not hasattr(self.teacher_denoiser, "_use_plan") or self.teacher_denoiser._use_plan(x)))
                if one_call:   # the whole loop inside the library (fdmi_teacher_loop / fdmi_dit_teacher_loop; A/B switch: FDMI_TEACHER_LOOP=0)
                    kw = {k: cfg_cond["cond"][k] for k in ("attention_mask", "attention_mask_lens") if k in cfg_cond["cond"]}
                    x = self.teacher_denoiser.teacher_loop(x, [float(t) for t in sch.timesteps[si:]],
                                                           cfg_cond["cond"]["crossattn"], cfg_cond["cond"].get("vector"),
                                                           sch.loop_coefficients(si, g), **kw)
                for it, t in enumerate(sch.timesteps[si:] if not one_call else []):
                    x_ = sch.scale_model_input(x, t)
                    e_c, e_u = self._teacher_cfg(x_, torch.full((B,), float(t), device=z.device), conditioning, uncond,
                                                 cfg_cond, *args, ctx_cache="fill" if it == 0 else "reuse", res=res, **kwargs)
                    if fused:
                        x = sch.fused_cfg_step(e_c, e_u, g, t, x)
                    else:
                        e = ops.axpby(e_c, g, e_u, 1.0 - g)
                        x = sch.step(e, t, x, return_dict=False)[0]
                return x

        # The teacher loop and the student's forward are independent (the teacher is frozen, both start from x_init): the loop
        # is issued on a SIDE HIP stream and the student's forward on the current one, so the many launches of the B-row
        # student that under-fill the chip (deep UNet levels, rank-r LoRA GEMMs) run beside the teacher's 2B-row kernels instead
        # of after them; the streams join before the first loss that needs both.  (A/B switch: FDMI_TEACHER_STREAM=0.)
        side = self._teacher_stream(z) if os.environ.get("FDMI_TEACHER_STREAM", "1") == "1" else None
        cur = torch.cuda.current_stream() if side is not None else None
        if side is not None:
            side.wait_stream(cur)
            # tensors allocated on the current stream and read by the side stream: the caching allocator must not hand their
            # blocks to a later current-stream allocation before the side stream is done with them
            for t_ in _tensors_of((x_init, conditioning, uncond, res)):
                if t_.is_cuda:
                    t_.record_stream(side)
        try:
            if side is not None:
                with torch.cuda.stream(side):
                    teacher_output = run_teacher()
            else:
                teacher_output = run_teacher()

            # ---- student: one step with grad (FD:255-280, 328) ----
            hook = getattr(self, "before_student", None)
            if hook is not None:
                hook()  # data-parallel trainer: wait for the deferred all-reduce + AdamW of the previous step
            eps_s = self.student_denoiser(sample=x_in, timestep=start_t, conditioning=student_conditioning,
                                          down_intrablock_additional_residuals=res)
            c_skip, c_out = self._scalings_for_boundary_conditions(start_t.float())
            inv_a, ms_a = self._x0_coeffs(start_t.long())
            # student_output = c_skip x + c_out (x - sigma eps)/alpha
            ca = (c_skip + c_out * inv_a).float().contiguous()
            cb = (c_out * ms_a).float().contiguous()
            student_output = _PerSampleAffine.apply(eps_s, x_init, ca, cb)
        finally:
            # join -- also when the student raised: the teacher's plan slot / workspace may not be reused (a later no-grad
            # teacher call of the DMD / GAN branches, the next forward) while the side stream still runs in it
            if side is not None:
                cur.wait_stream(side)
        if side is not None:   # everything below reads the teacher's result on the current stream
            teacher_output.record_stream(cur)

        l_distill = self._distill_loss(student_output, teacher_output)
        loss = l_distill * self.distill_loss_scale[K_step]
        self.terms = {"distill": l_distill.detach(), "K_step": K_step, "guidance": g, "n_teacher_steps": K - si}
        if self.use_dmd_loss:
            l_dmd = self._dmd_loss(d, student_output, student_conditioning, conditioning, uncond, K_step, res)
            self.terms["dmd"] = l_dmd.detach()
            loss = loss + l_dmd * self.dmd_loss_scale[K_step]
        if self.discriminator is not None:
            gan = self._gan_loss(d, z, student_output, teacher_output, conditioning, step, res)
        else:
            gan = [0, 0]  # the reference cannot run without a discriminator (FD:347); we degrade gracefully
        self.terms["gan_G"] = gan[0].detach() if torch.is_tensor(gan[0]) else gan[0]
        self.terms["gan_D"] = gan[1].detach() if torch.is_tensor(gan[1]) else gan[1]
        loss = loss + self.adversarial_loss_scale[K_step] * gan[0]
        return {"loss": [loss, gan[1]], "teacher_output": teacher_output, "student_output": student_output,
                "noisy_sample": x_init, "start_timestep": self._start_t_host}

    # ---- losses --------------------------------------------------------------------------------------
    def _distill_loss(self, s, t):
        """FD:368-399.  l2 / l1: one fused HIP launch (+ one for the gradient).  lpips: the centre 64x64 latent crop of both
        outputs (the reference's slice expression verbatim) goes through the VAE decoder and the perceptual network -- on the
        HIP path nets.MiAutoencoderKLDiffusers / nets.MiLPIPS (plans of libfdmi.so: implicit-GEMM convs, GroupNorm, the LPIPS
        distance kernels, taped input gradients); the student's decode and its half of the VGG stack are
        taped, the teacher's run without a tape."""
        if self.distill_loss_type == "lpips":
            crop_h = (s.shape[2] - 64) // 2
            crop_w = (s.shape[3] - 64) // 2
            s = s[:, :, crop_h:crop_h + 64, crop_w:crop_w + 64]
            t = t[:, :, crop_h:crop_h + 64, crop_w:crop_w + 64]
            return self.lpips(self.vae.decode(s).clamp(-1, 1), self.vae.decode(t).clamp(-1, 1)).mean()
        return _DistillLoss.apply(s, t.detach(), self.distill_loss_type == "l1")

    def _dmd_loss(self, d, s, student_cond, cond, uncond, K_step, res=None):
        """FD:401-499"""
        sch = self.teacher_noise_scheduler
        B = s.shape[0]
        noise = d.randn_like("dmd_noise", s)
        t = d.randint("dmd_t", 0, sch.config.num_train_timesteps, (B,), s.device)
        noisy = sch.add_noise(s, noise, t)
        with torch.no_grad():
            tf = t.float()
            e_c, e_u = self._teacher_cfg(noisy.detach(), tf, cond, uncond, self._cat_cond(cond, uncond), res=res)
            e_f = self.student_denoiser(sample=noisy, timestep=tf, conditioning=student_cond,
                                        down_intrablock_additional_residuals=res)
            g = (float(d.rand1("dmd_guidance")) * (self.guidance_scale_max[K_step] - self.guidance_scale_min[K_step])
                 + self.guidance_scale_min[K_step])
            real = ops.axpby(e_c, g, e_u, 1.0 - g)
            ac_dev = getattr(sch, "_ac_dev", None)      # (the table goes to the device once, not per call: a blocking copy)
            if ac_dev is None or ac_dev.device != s.device:
                ac_dev = sch._ac_dev = sch.alphas_cumprod.to(s.device)
            a = ac_dev[t]
            kb = ((1.0 - a) ** 0.5 / a ** 0.5).float().contiguous()     # (score_fake - score_real) = real - fake
            inv_a, ms_a = self._x0_coeffs(t)
        return _DmdLoss.apply(s, noisy.detach(), real, e_f, inv_a.float().contiguous(), ms_a.float().contiguous(), kb)

    def _gan_loss(self, d, z, s, teacher_output, conditioning, step, res=None):
        """FD:501-667"""
        sch = self.teacher_noise_scheduler
        self.disc_update_counter += 1
        B = s.shape[0]
        noise = d.randn_like("gan_noise", s)
        real = teacher_output if self.use_teacher_as_real else z
        idx = d.multinomial("gan_idx", torch.tensor([0.25, 0.25, 0.25, 0.25]), B, replacement=True)
        ts = ops.upload(torch.tensor([10, 250, 500, 750], dtype=torch.long)[idx.cpu()], s.device)   # (host gather, one non-blocking upload)
        gen = step % 2 == 0
        s_in = s if gen else s.detach()          # D-step: fake branch is detached (FD:583, 602, 616, 638, 659)
        noisy_fake = sch.add_noise(s_in, noise, ts)
        with torch.no_grad():
            noisy_real = sch.add_noise(real, noise, ts)
        x = torch.cat([noisy_fake, noisy_real], dim=0)
        if conditioning is not None:
            conditioning = {"cond": {k: (torch.cat([v, v], dim=0) if torch.is_tensor(v) else list(v) + list(v))
                                     for k, v in conditioning["cond"].items()}}
        t2 = torch.cat([ts, ts], dim=0).float()
        feat = self.disc_backbone(sample=x, timestep=t2, conditioning=conditioning,
                                  down_intrablock_additional_residuals=self._dup(res), return_intermediate=True)
        f_fake, f_real = feat.chunk(2, dim=0)
        disc = self.discriminator
        kind = self.gan_loss_type
        dev = s.device
        if kind == "wgan":
            for p in disc.parameters():
                p.data.clamp_(-0.01, 0.01)
            if gen:
                return [-disc(f_fake).mean(), 0]
            return [0, -disc(f_real).mean() + disc(f_fake.detach()).mean()]
        if kind == "lsgan":
            valid, fake = torch.ones(B, 1, device=dev), torch.zeros(B, 1, device=dev)
            if gen:
                return [F.mse_loss(torch.sigmoid(disc(f_fake)), valid), 0]
            return [0, 0.5 * (F.mse_loss(torch.sigmoid(disc(f_real)), valid)
                              + F.mse_loss(torch.sigmoid(disc(f_fake.detach())), fake))]
        if kind == "hinge":
            if gen:
                return [-disc(f_fake).mean(), 0]
            return [0, F.relu(1.0 - disc(f_real)).mean() + F.relu(1.0 + disc(f_fake.detach())).mean()]
        if kind == "non-saturating":
            if gen:
                return [-torch.mean(torch.log(torch.sigmoid(disc(f_fake)) + 1e-8)), 0]
            return [0, -torch.mean(torch.log(torch.sigmoid(disc(f_real)) + 1e-8)
                                   + torch.log(1 - torch.sigmoid(disc(f_fake.detach())) + 1e-8))]
        valid = torch.ones(B, 1, device=dev)
        if gen:
            return [F.binary_cross_entropy_with_logits(disc(f_fake), valid), 0]
        fake = torch.zeros(B, 1, device=dev)
        return [0, F.binary_cross_entropy_with_logits(disc(f_real), valid)
                + F.binary_cross_entropy_with_logits(disc(f_fake.detach()), fake)]
