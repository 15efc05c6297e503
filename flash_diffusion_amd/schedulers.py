"""Scheduler duck-types for the distillation hot path, device-side math through libfdmi.so.

The reference consumes a diffusers scheduler through this surface only
(/root/reference/src/flash/models/flash/flash_diffusion_model.py): set_timesteps FD:139, .timesteps
FD:172/289, .init_noise_sigma FD:245, add_noise FD:250/426/536, scale_model_input FD:255/292, step
FD:322, .alphas_cumprod FD:111/469, .config.num_train_timesteps FD:420.  Coefficients are computed on
the host in fp32 exactly as upstream diffusers does; the latent update itself is ONE fused axpby
kernel per step (fdmi_axpby4) instead of ~10 tiny element-wise launches.

DPMSolverMultistepScheduler: dpmsolver++ / order 2 / midpoint / lower_order_final /
final_sigmas_type="zero" built from the SDXL scheduler config with timestep_spacing="trailing"
(examples/train_flash_sd.py:204-208, configs/flash_sd.yaml:37).  add_noise follows the DDPM definition
for any integer timestep (identical to upstream on schedule timesteps)."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import ops


def _betas(n, beta_start, beta_end, schedule):
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(schedule)


class _Base:
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 timestep_spacing="trailing", steps_offset=1, prediction_type="epsilon"):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset,
                                      prediction_type=prediction_type)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.num_inference_steps = None
        self._dev_tables = {}

    def _spaced(self, n):
        T, sp = self.config.num_train_timesteps, self.config.timestep_spacing
        if sp == "trailing":
            ts = np.arange(T, 0, -T / n).round() - 1
        elif sp == "leading":
            ts = (np.arange(0, n) * (T // n)).round()[::-1].copy() + self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, n).round()[::-1].copy()
        else:
            raise NotImplementedError(sp)
        return ts.astype(np.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _tables(self, device):
        t = self._dev_tables.get(device)
        if t is None:
            ac = self.alphas_cumprod.to(device)
            t = (ac ** 0.5, (1 - ac) ** 0.5)
            self._dev_tables[device] = t
        return t

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(abar_t) x + sqrt(1-abar_t) eps.  Differentiable w.r.t. `original_samples` (the DMD / GAN
        branches noise the student output, FD:426, 536): gradient = sqrt(abar_t) * grad."""
        sa_t, sb_t = self._tables(original_samples.device)
        ts = timesteps.to(original_samples.device).reshape(-1)
        sa, sb = sa_t[ts].contiguous(), sb_t[ts].contiguous()
        if ts.numel() == 1 and original_samples.shape[0] != 1:
            sa, sb = sa.expand(original_samples.shape[0]).contiguous(), sb.expand(original_samples.shape[0]).contiguous()
        return _AddNoise.apply(original_samples, noise, sa, sb)


class _AddNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise, sa, sb):
        ctx.save_for_backward(sa)
        return ops.add_noise(x.float().contiguous(), noise.float().contiguous(), sa, sb)

    @staticmethod
    def backward(ctx, g):
        (sa,) = ctx.saved_tensors
        zero = torch.zeros_like(sa)
        gx = ops.add_noise(g.contiguous(), g.contiguous(), sa, zero)
        return gx, None, None, None


class DPMSolverMultistepScheduler(_Base):
    solver_order = 2

    def __init__(self, **kw):
        super().__init__(**kw)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.model_outputs = [None, None]
        self.lower_order_nums = 0
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None):
        ts = self._spaced(num_inference_steps)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None, None]
        self.lower_order_nums = 0
        self._step_index = None

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def index_for_timestep(self, timestep):
        cand = (self.timesteps == int(timestep)).nonzero()
        if len(cand) == 0:
            return len(self.timesteps) - 1
        return int(cand[1 if len(cand) > 1 else 0].item())

    def step_coefficients(self, i, lower_order_nums):
        """x_next = c_s x + c_d0 m0 + c_d1 (m0 - m1) with m = x0 predictions (upstream fp32 arithmetic)."""
        n = len(self.timesteps)
        final = i == n - 1
        sig = self.sigmas
        alpha_t, sigma_t = self._alpha_sigma(sig[i + 1])
        alpha_s0, sigma_s0 = self._alpha_sigma(sig[i])
        lam_t = torch.log(alpha_t) - torch.log(sigma_t)
        lam_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lam_t - lam_s0
        c_s = sigma_t / sigma_s0
        c_d0 = -(alpha_t * (torch.exp(-h) - 1.0))
        if lower_order_nums < 1 or final:
            return 1, float(c_s), float(c_d0), 0.0
        alpha_s1, sigma_s1 = self._alpha_sigma(sig[i - 1])
        lam_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
        r0 = (lam_s0 - lam_s1) / h
        return 2, float(c_s), float(c_d0), float(-0.5 * (alpha_t * (torch.exp(-h) - 1.0)) / r0)

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        i = self._step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        # x0 = (x - sigma_t eps) / alpha_t
        x0 = ops.axpby(sample, float(1.0 / alpha_t), model_output, float(-sigma_t / alpha_t))
        self.model_outputs = [self.model_outputs[1], x0]
        order, c_s, c_d0, c_d1 = self.step_coefficients(i, self.lower_order_nums)
        if order == 1:
            prev = ops.axpby(sample, c_s, x0, c_d0)
        else:
            prev = ops.axpby(sample, c_s, x0, c_d0 + c_d1, self.model_outputs[0], -c_d1)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,)

    def loop_coefficients(self, start_index, guidance):
        """[n][6] host coefficients of the whole CFG loop from step `start_index` of a fresh schedule (right after
        set_timesteps): x0 = a0 x + a1 eps_c + a2 eps_u ; x = a3 x + a4 x0 + a5 x0_prev -- the numbers fused_cfg_step
        would use step by step, for the C-ABI's fdmi_teacher_loop (no host round trip between steps)."""
        rows, lower = [], 0
        for i in range(start_index, len(self.timesteps)):
            alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
            k = float(-sigma_t / alpha_t)
            order, c_s, c_d0, c_d1 = self.step_coefficients(i, lower)
            if order == 1:
                upd = (c_s, c_d0, 0.0)
            else:
                upd = (c_s, c_d0 + c_d1, -c_d1)
            rows.append((float(1.0 / alpha_t), k * guidance, k * (1.0 - guidance)) + upd)
            lower = min(lower + 1, self.solver_order)
        return rows

    def fused_cfg_step(self, eps_c, eps_u, guidance, timestep, sample):
        """CFG combine (FD:316-319) + step (FD:322-324) with the guidance folded into the x0 prediction:
        x0 = x/alpha - (sigma/alpha)(g eps_c + (1-g) eps_u)  -- one kernel instead of two."""
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        i = self._step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        k = float(-sigma_t / alpha_t)
        x0 = ops.axpby(sample, float(1.0 / alpha_t), eps_c, k * guidance, eps_u, k * (1.0 - guidance))
        self.model_outputs = [self.model_outputs[1], x0]
        order, c_s, c_d0, c_d1 = self.step_coefficients(i, self.lower_order_nums)
        if order == 1:
            prev = ops.axpby(sample, c_s, x0, c_d0)
        else:
            prev = ops.axpby(sample, c_s, x0, c_d0 + c_d1, self.model_outputs[0], -c_d1)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return prev


class DDPMScheduler(_Base):
    """Upstream DDPMScheduler defaults (the class the reference's own test uses,
    tests/test_flash/test_flash_diffusion.py:93-98): linear betas, leading spacing, fixed_small variance,
    clip_sample=True."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 timestep_spacing="leading", steps_offset=0, clip_sample=True, clip_sample_range=1.0, **kw):
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                         beta_schedule=beta_schedule, timestep_spacing=timestep_spacing, steps_offset=steps_offset, **kw)
        self.config.clip_sample = clip_sample
        self.config.clip_sample_range = clip_sample_range
        self.variance_noise_fn = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(self._spaced(num_inference_steps))

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, **kw):
        t = int(timestep)
        n = self.num_inference_steps or self.config.num_train_timesteps
        prev_t = t - self.config.num_train_timesteps // n
        ac = self.alphas_cumprod
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        x0 = ops.axpby(sample, float(1 / a_t ** 0.5), model_output, float(-(b_t ** 0.5) / a_t ** 0.5))
        if self.config.clip_sample:
            x0 = x0.clamp_(-self.config.clip_sample_range, self.config.clip_sample_range)
        c0 = float((a_prev ** 0.5 * cur_beta) / b_t)
        c1 = float(cur_alpha ** 0.5 * b_prev / b_t)
        if t > 0:
            if self.variance_noise_fn is not None:
                noise = self.variance_noise_fn(model_output.shape).to(model_output)
            else:
                noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                    dtype=model_output.dtype)
            std = float(torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20) ** 0.5)
            return (ops.axpby(x0, c0, sample, c1, noise, std),)
        return (ops.axpby(x0, c0, sample, c1),)


class LCMScheduler(_Base):
    """The sampling scheduler of the few-step student sampler (FlashDiffusion.sample, FD:754-915; shipped configs:
    SAMPLING_SCHEDULER: LCMScheduler).  Upstream LCMScheduler semantics: epsilon prediction, boundary-condition scalings
    c_skip / c_out on timestep * timestep_scaling, re-noising with fresh Gaussian noise on every step but the last;
    `set_timesteps` takes either `num_inference_steps` (floor(linspace) picks out of the reversed 50-step training
    schedule) or custom `timesteps=` -- the reference hands over the teacher scheduler's schedule (FD:783-788).
    The latent update  prev = A x + B eps + C noise  is ONE fused HIP launch (fdmi_axpby4); coefficients in host fp32."""

    def __init__(self, original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5, **kw):
        super().__init__(**kw)
        self.config.original_inference_steps = original_inference_steps
        self.config.timestep_scaling = timestep_scaling
        self.sigma_data = sigma_data
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self._step_index = None
        self.noise_fn = None   # tests inject the reference's re-noising draws here

    def set_timesteps(self, num_inference_steps=None, device=None, original_inference_steps=None, timesteps=None,
                      strength=1.0):
        if num_inference_steps is None and timesteps is None:
            raise ValueError("Must pass exactly one of `num_inference_steps` or `custom_timesteps`.")
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        T = self.config.num_train_timesteps
        original_steps = original_inference_steps or self.config.original_inference_steps
        if original_steps > T:
            raise ValueError("original_inference_steps cannot exceed num_train_timesteps")
        k = T // original_steps
        origin = np.asarray(list(range(1, int(original_steps * strength) + 1))) * k - 1
        if timesteps is not None:
            ts = np.asarray(timesteps.cpu().numpy() if torch.is_tensor(timesteps) else timesteps, dtype=np.int64)
            if any(ts[i] >= ts[i - 1] for i in range(1, len(ts))):
                raise ValueError("`custom_timesteps` must be in descending order.")
            if ts[0] >= T:
                raise ValueError("`timesteps` must start before `num_train_timesteps`.")
        else:
            if num_inference_steps > T or num_inference_steps > len(origin):
                raise ValueError("`num_inference_steps` is larger than the available schedule")
            rev = origin[::-1].copy()
            idx = np.floor(np.linspace(0, len(rev), num=num_inference_steps, endpoint=False)).astype(np.int64)
            ts = rev[idx]
        self.timesteps = torch.from_numpy(np.asarray(ts, dtype=np.int64))
        self.num_inference_steps = len(ts)
        self._step_index = None

    def boundary_scalings(self, timestep):
        st = float(timestep) * self.config.timestep_scaling
        sd = self.sigma_data
        return sd ** 2 / (st ** 2 + sd ** 2), st / (st ** 2 + sd ** 2) ** 0.5

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, **kw):
        if self._step_index is None:
            cand = (self.timesteps == int(timestep)).nonzero()
            self._step_index = int(cand[0].item()) if len(cand) else len(self.timesteps) - 1
        i = self._step_index
        prev_t = int(self.timesteps[i + 1]) if i + 1 < len(self.timesteps) else int(timestep)
        a_t = self.alphas_cumprod[int(timestep)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        c_skip, c_out = self.boundary_scalings(int(timestep))
        # denoised = c_out (x - sqrt(b_t) eps) / sqrt(a_t) + c_skip x
        dx = float(c_out / a_t.sqrt() + c_skip)
        de = float(-c_out * b_t.sqrt() / a_t.sqrt())
        x = sample.float().contiguous()
        e = model_output.float().contiguous()
        if i != self.num_inference_steps - 1:
            if self.noise_fn is not None:
                noise = self.noise_fn(model_output.shape).to(device=x.device, dtype=torch.float32)
            else:
                noise = torch.randn(model_output.shape, generator=generator, device=x.device, dtype=torch.float32)
            sp = float(a_prev.sqrt())
            prev = ops.axpby(x, sp * dx, e, sp * de, noise.contiguous(), float(b_prev.sqrt()))
            denoised = None
        else:
            prev = ops.axpby(x, dx, e, de)
            denoised = prev
        self._step_index += 1
        if denoised is None and kw.get("return_denoised", False):
            denoised = ops.axpby(x, dx, e, de)
        return (prev, denoised)
