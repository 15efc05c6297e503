"""TEST INFRASTRUCTURE (oracle) -- scheduler duck-types used by the reference hot path.

The reference calls (flash_diffusion_model.py, "FD"):
  set_timesteps(K) FD:139, .timesteps FD:172/289, .init_noise_sigma FD:245,
  add_noise FD:250/426/536, scale_model_input FD:255/292, step FD:322,
  .alphas_cumprod FD:111/469, .config.num_train_timesteps FD:420.

The schedulers themselves live in diffusers (third-party, absent here:
requirements.txt:1 pins an un-vendored fork branch).  This file restates the
published upstream algorithms:
  * DPMSolverMultistepScheduler -- dpmsolver++ / order 2 / midpoint /
    lower_order_final / final_sigmas_type="zero", the class the shipped configs
    select (examples/configs/flash_sd.yaml:37) built from the SDXL scheduler
    config with timestep_spacing="trailing" (examples/train_flash_sd.py:204-208).
  * DDPMScheduler -- the class the reference's own test uses for every
    scheduler role (tests/test_flash/test_flash_diffusion.py:93-98).

Stated assumption: add_noise follows the DDPM definition
sqrt(abar_t) x + sqrt(1-abar_t) eps for *any* integer timestep.  For timesteps
on the current schedule this is identical to upstream DPMSolverMultistep's
sigma-indexed add_noise; off-schedule timesteps (DMD / GAN draws, FD:418, 524)
are only well defined under the DDPM form, which is what T-FD exercises.

PARITY UNPINNED: diffusers is not installed and the reference holds no golden
values for its schedulers, so this restatement of the published algorithms could
not be checked against the reference's own dependency (DESIGN.md section 2 says
the same; the steps built on it are pinned only against the reference's
FlashDiffusion class running over THESE schedulers).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                              dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


class _SchedulerBase:
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", timestep_spacing="trailing", steps_offset=1,
                 prediction_type="epsilon"):
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, timestep_spacing=timestep_spacing,
            steps_offset=steps_offset, prediction_type=prediction_type)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.timesteps = torch.from_numpy(
            np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.num_inference_steps = None

    # ---- shared -----------------------------------------------------------------
    def _spaced_timesteps(self, n, last_timestep=None):
        T = self.config.num_train_timesteps
        sp = self.config.timestep_spacing
        if sp == "trailing":
            last = T if last_timestep is None else last_timestep
            step_ratio = last / n
            ts = np.arange(last, 0, -step_ratio).round() - 1
        elif sp == "leading":
            step_ratio = T // n
            ts = (np.arange(0, n) * step_ratio).round()[::-1].copy()
            ts = ts + self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, n).round()[::-1].copy()
        else:
            raise NotImplementedError(sp)
        return ts.astype(np.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = ac[timesteps] ** 0.5
        sb = (1 - ac[timesteps]) ** 0.5
        sa = sa.flatten()
        sb = sb.flatten()
        while sa.ndim < original_samples.ndim:
            sa = sa.unsqueeze(-1)
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise


class DPMSolverMultistepSchedulerRef(_SchedulerBase):
    """dpmsolver++ (2M, midpoint), epsilon prediction, final sigma = 0."""

    solver_order = 2

    def __init__(self, **kw):
        super().__init__(**kw)
        ac = self.alphas_cumprod
        self.alpha_t = torch.sqrt(ac)
        self.sigma_t = torch.sqrt(1 - ac)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.sigmas = ((1 - ac) / ac) ** 0.5
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None):
        ts = self._spaced_timesteps(num_inference_steps)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32)  # final_sigmas_type="zero"
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    @staticmethod
    def _sigma_to_alpha_sigma_t(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def index_for_timestep(self, timestep):
        cand = (self.timesteps == int(timestep)).nonzero()
        if len(cand) == 0:
            return len(self.timesteps) - 1
        pos = 1 if len(cand) > 1 else 0
        return int(cand[pos].item())

    def step_coefficients(self, step_index, lower_order_nums):
        """(order, c_sample, c_d0, c_d1) with x_next = c_sample*x + c_d0*m0 + c_d1*(m0-m1);
        m = x0 prediction.  Pure python floats in fp32 torch arithmetic like upstream."""
        n = len(self.timesteps)
        lower_order_final = step_index == n - 1  # final_sigmas_type == "zero"
        sig = self.sigmas
        sigma_t, sigma_s0 = sig[step_index + 1], sig[step_index]
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(sigma_t)
        alpha_s0, sigma_s0 = self._sigma_to_alpha_sigma_t(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        c_sample = sigma_t / sigma_s0
        c_d0 = -(alpha_t * (torch.exp(-h) - 1.0))
        if lower_order_nums < 1 or lower_order_final:
            return 1, float(c_sample), float(c_d0), 0.0
        sigma_s1 = sig[step_index - 1]
        alpha_s1, sigma_s1 = self._sigma_to_alpha_sigma_t(sigma_s1)
        lambda_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
        h_0 = lambda_s0 - lambda_s1
        r0 = h_0 / h
        c_d1 = -0.5 * (alpha_t * (torch.exp(-h) - 1.0)) / r0
        return 2, float(c_sample), float(c_d0), float(c_d1)

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        i = self._step_index
        sigma = self.sigmas[i]
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(sigma)
        x0_pred = (sample - sigma_t * model_output) / alpha_t
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0_pred
        sample = sample.to(torch.float32)
        order, c_s, c_d0, c_d1 = self.step_coefficients(i, self.lower_order_nums)
        if order == 1:
            prev = c_s * sample + c_d0 * x0_pred
        else:
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            prev = c_s * sample + c_d0 * m0 + c_d1 * (m0 - m1)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        prev = prev.to(model_output.dtype)
        return (prev,)


class DDPMSchedulerRef(_SchedulerBase):
    """Upstream DDPMScheduler defaults: linear betas 1e-4..0.02, leading spacing,
    steps_offset 0, variance_type fixed_small, clip_sample=True (range 1.0)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                 beta_schedule="linear", timestep_spacing="leading", steps_offset=0,
                 clip_sample=True, clip_sample_range=1.0, **kw):
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                         beta_end=beta_end, beta_schedule=beta_schedule,
                         timestep_spacing=timestep_spacing, steps_offset=steps_offset, **kw)
        self.config.clip_sample = clip_sample
        self.config.clip_sample_range = clip_sample_range
        self.one = torch.tensor(1.0)
        self.variance_noise_fn = None  # injectable: fn(shape) -> tensor

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ts = self._spaced_timesteps(num_inference_steps)
        self.timesteps = torch.from_numpy(ts).to(device=device, dtype=torch.int64)

    def previous_timestep(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, **kw):
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        ac = self.alphas_cumprod
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        c0 = (a_prev ** 0.5 * cur_beta) / b_t
        c1 = cur_alpha ** 0.5 * b_prev / b_t
        prev = c0 * x0 + c1 * sample
        if t > 0:
            if self.variance_noise_fn is not None:
                noise = self.variance_noise_fn(model_output.shape).to(model_output)
            else:
                noise = torch.randn(model_output.shape, generator=generator,
                                    dtype=model_output.dtype).to(model_output.device)
            var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
            prev = prev + (var ** 0.5) * noise
        return (prev,)


class LCMSchedulerRef(_SchedulerBase):
    """Upstream LCMScheduler step (epsilon prediction) with the boundary-condition
    scalings, used only by FlashDiffusion.sample (FD:754-915, a 'next' row)."""

    def __init__(self, original_inference_steps=50, timestep_scaling=10.0, **kw):
        super().__init__(**kw)
        self.config.original_inference_steps = original_inference_steps
        self.config.timestep_scaling = timestep_scaling
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self._step_index = None
        self.noise_fn = None

    def set_timesteps(self, num_inference_steps=None, device=None, original_inference_steps=None, timesteps=None,
                      strength=1.0):
        """Upstream LCMScheduler.set_timesteps (restated): custom `timesteps=` (what FD:783-788 tries first, handing over
        the teacher scheduler's trailing schedule) or `num_inference_steps` picked by floor(linspace) out of the reversed
        original 50-step training schedule {k, 2k, ...} - 1 with k = num_train_timesteps // original_inference_steps."""
        if num_inference_steps is None and timesteps is None:
            raise ValueError("Must pass exactly one of `num_inference_steps` or `custom_timesteps`.")
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        original_steps = original_inference_steps or self.config.original_inference_steps
        T = self.config.num_train_timesteps
        if original_steps > T:
            raise ValueError("original_inference_steps cannot exceed num_train_timesteps")
        k = T // original_steps
        origin = np.asarray(list(range(1, int(original_steps * strength) + 1))) * k - 1
        if timesteps is not None:
            ts = np.asarray(torch.as_tensor(timesteps).cpu().numpy() if torch.is_tensor(timesteps) else timesteps,
                            dtype=np.int64)
            for i in range(1, len(ts)):
                if ts[i] >= ts[i - 1]:
                    raise ValueError("`custom_timesteps` must be in descending order.")
            if ts[0] >= T:
                raise ValueError("`timesteps` must start before `num_train_timesteps`.")
        else:
            if num_inference_steps > T:
                raise ValueError("`num_inference_steps` cannot be larger than `num_train_timesteps`")
            if num_inference_steps > len(origin):
                raise ValueError("`num_inference_steps` cannot be larger than the original schedule")
            rev = origin[::-1].copy()
            idx = np.floor(np.linspace(0, len(rev), num=num_inference_steps, endpoint=False)).astype(np.int64)
            ts = rev[idx]
        self.timesteps = torch.from_numpy(np.asarray(ts, dtype=np.int64)).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(ts)
        self._step_index = None

    def boundary_scalings(self, timestep, sigma_data=0.5):
        st = timestep * self.config.timestep_scaling
        c_skip = sigma_data ** 2 / (st ** 2 + sigma_data ** 2)
        c_out = st / (st ** 2 + sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, **kw):
        if self._step_index is None:
            cand = (self.timesteps == int(timestep)).nonzero()
            self._step_index = int(cand[0].item()) if len(cand) else len(self.timesteps) - 1
        i = self._step_index
        prev_i = i + 1
        prev_t = self.timesteps[prev_i] if prev_i < len(self.timesteps) else timestep
        a_t = self.alphas_cumprod[int(timestep)]
        a_prev = self.alphas_cumprod[int(prev_t)] if int(prev_t) >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        c_skip, c_out = self.boundary_scalings(float(timestep))
        x0 = (sample - b_t.sqrt() * model_output) / a_t.sqrt()
        denoised = c_out * x0 + c_skip * sample
        if i != self.num_inference_steps - 1:
            if self.noise_fn is not None:
                noise = self.noise_fn(model_output.shape).to(model_output)
            else:
                noise = torch.randn(model_output.shape, generator=generator,
                                    dtype=denoised.dtype).to(model_output.device)
            prev = a_prev.sqrt() * denoised + b_prev.sqrt() * noise
        else:
            prev = denoised
        self._step_index += 1
        return (prev, denoised)


class FlowMatchEulerDiscreteSchedulerRef:
    """Upstream FlowMatchEulerDiscreteScheduler (rectified flow, SD3) restated: sigma-shifted linear schedule, Euler step
    x <- x + (sigma_next - sigma) * v.  The surface FlashDiffusionSD3 uses (FD3:150, 262-270, 281-314, 431-447, 520-536,
    947-958): set_timesteps, .timesteps (float), .sigmas (with a trailing 0), step, .config.num_train_timesteps.
    (diffusers is absent offline: parity of these internals is unpinned, like the other schedulers; the orchestration
    around them is pinned against the real FlashDiffusionSD3 in tests/test_oracle_vs_reference.py.)"""

    def __init__(self, num_train_timesteps=1000, shift=3.0):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift)
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.sigma_min = float(sig[-1])
        self.sigma_max = float(sig[0])
        self._step_index = None
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps, device=None):
        N, shift = self.config.num_train_timesteps, self.config.shift
        ts = np.linspace(self.sigma_max * N, self.sigma_min * N, num_inference_steps)
        sig = ts / N
        sig = shift * sig / (1 + (shift - 1) * sig)
        sig = torch.from_numpy(sig).to(dtype=torch.float32, device=device)
        self.timesteps = sig * N
        self.sigmas = torch.cat([sig, torch.zeros(1, device=sig.device)])
        self.num_inference_steps = num_inference_steps
        self._step_index = None

    def index_for_timestep(self, timestep):
        cand = (self.timesteps == timestep).nonzero()
        pos = 1 if len(cand) > 1 else 0
        return int(cand[pos].item())

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        if self._step_index is None:
            t = timestep if torch.is_tensor(timestep) else torch.tensor(timestep)
            self._step_index = self.index_for_timestep(t.to(self.timesteps.device))
        sigma, sigma_next = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        prev = sample.to(torch.float32) + (sigma_next - sigma) * model_output
        self._step_index += 1
        return (prev.to(model_output.dtype),)
