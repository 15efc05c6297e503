"""CPU tests of the host-side logic of the product package (no GPU, no kernels): config broadcasting,
timestep pmf, scheduler coefficients vs the oracle, optimizer parameter selection (TR:76-139), and the
data-parallel gradient exchange on a 2-process gloo group."""
import os
import re

import pytest
import torch
import torch.nn as nn

from flash_diffusion_amd.flash import FlashDiffusion, FlashDiffusionConfig, TensorConditioner
from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
from flash_diffusion_amd.trainer import TrainingConfig, TrainingPipeline
from oracle.flash_ref import FlashConfigRef, timestep_pmf
from oracle.sched_cpu import DPMSolverMultistepSchedulerRef


def test_config_broadcast_matches_reference_post_init():
    c = FlashDiffusionConfig(K=[8, 8], num_iterations_per_K=[5, 5], guidance_scale_min=3.0, mixture_var=0.5)
    assert c.guidance_scale_min == [3.0, 3.0] and c.mixture_num_components == [4, 4]
    assert c.mode_probs == [[0.25] * 4] * 2 and c.distill_loss_type == "l2" and c.gan_loss_type == "hinge"
    with pytest.raises(AssertionError):
        FlashDiffusionConfig(K=[8], num_iterations_per_K=[5, 5])


class _Dummy(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(3))


@pytest.mark.parametrize("dist,kw", [("uniform", {}), ("gaussian", {}),
                                     ("mixture", dict(mixture_num_components=4, mixture_var=0.5, mode_probs=[[0.1, 0.3, 0.3, 0.3]]))])
def test_timestep_pmf_matches_oracle(dist, kw):
    cfg = FlashDiffusionConfig(K=[16], num_iterations_per_K=[3], timestep_distribution=dist, **kw)
    m = FlashDiffusion(cfg, _Dummy(), _Dummy(), DPMSolverMultistepScheduler())
    ref = timestep_pmf(FlashConfigRef(K=[16], num_iterations_per_K=[3], timestep_distribution=dist, **kw), 16, 0)
    assert torch.equal(m._timestep_pmf(16, 0), ref)


@pytest.mark.parametrize("K", [1, 4, 8, 32])
def test_dpm_coefficients_match_oracle(K):
    a, b = DPMSolverMultistepScheduler(), DPMSolverMultistepSchedulerRef()
    a.set_timesteps(K)
    b.set_timesteps(K)
    assert a.timesteps.tolist() == b.timesteps.tolist()
    assert torch.equal(a.sigmas, b.sigmas)
    for i in range(K):
        for lo in (0, 1):
            if lo == 1 and i == 0:
                continue
            assert a.step_coefficients(i, lo) == b.step_coefficients(i, lo)
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod)


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.student_denoiser = nn.Linear(4, 4)
        self.teacher_denoiser = nn.Linear(4, 4)
        self.discriminator = nn.Sequential(nn.Linear(4, 1))
        for p in self.teacher_denoiser.parameters():
            p.requires_grad = False

    def on_train_batch_end(self, b):
        pass

    def forward(self, batch, step=0, **kw):
        s = self.student_denoiser(batch["x"])
        t = self.teacher_denoiser(batch["x"]).detach()
        lg = ((s - t) ** 2).mean() - self.discriminator(s).mean()
        ld = self.discriminator(s.detach()).mean() - self.discriminator(t).mean()
        return {"loss": [lg, 0] if step % 2 == 0 else [0, ld], "start_timestep": 1}


def test_configure_optimizers_regex_selection_and_manual_loop():
    m = _Toy()
    pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW", "AdamW"], learning_rates=[1e-2, 1e-2],
                                              trainable_params=[["student_denoiser"], ["discriminator."]]), overlap=False)
    opts = pipe.configure_optimizers()
    assert not pipe.automatic_optimization and len(opts) == 2
    assert sum(p.numel() for p in opts[0].param_groups[0]["params"]) == 20
    assert sum(p.numel() for p in opts[1].param_groups[0]["params"]) == 5
    before = {k: v.clone() for k, v in m.state_dict().items()}
    out = pipe.training_step({"x": torch.randn(8, 4)}, 0)
    assert "loss_optimizer_0" in out and "loss_optimizer_1" in out
    after = m.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in after if k.startswith("teacher"))
    assert any(not torch.equal(before[k], after[k]) for k in after if k.startswith("student"))
    assert any(not torch.equal(before[k], after[k]) for k in after if k.startswith("discriminator"))


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = _Toy()  # identical replicas
    pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-2],
                                              trainable_params=[["student_denoiser"]]), overlap=False)
    pipe.configure_optimizers()
    g = torch.Generator().manual_seed(100 + rank)  # each rank its own shard
    pipe.training_step({"x": torch.randn(8, 4, generator=g)}, 0)
    q.put((rank, {k: v.numpy().copy() for k, v in m.student_denoiser.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_all_reduce_gloo_world2():
    """world_size-2 gloo run: replicas stay identical and equal a single process that averages both shards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in res.items()}
    for p in procs:
        p.join(60)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), "replicas diverged"
    # single-process reference: mean of the two shard gradients
    torch.manual_seed(0)
    m = _Toy()
    opt = torch.optim.AdamW(m.student_denoiser.parameters(), lr=1e-2)
    grads = []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        opt.zero_grad()
        m({"x": torch.randn(8, 4, generator=g)})["loss"][0].backward()
        grads.append([p.grad.clone() for p in m.student_denoiser.parameters()])
    for p, a, b in zip(m.student_denoiser.parameters(), *grads):
        p.grad = (a + b) / 2
    opt.step()
    for k, v in m.student_denoiser.state_dict().items():
        assert torch.allclose(v, res[0][k], atol=1e-6), k


def _ddp_sgd_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = _Toy()
    pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["SGD"], learning_rates=[0.5],
                                              trainable_params=[["student_denoiser"]]), overlap=False)
    pipe.configure_optimizers()
    seen = []
    pipe.reduced_grad_hook = lambda opt, scale: seen.append(
        [p.grad.detach().clone() * scale for g in opt.param_groups for p in g["params"]])
    g = torch.Generator().manual_seed(100 + rank)
    pipe.training_step({"x": torch.randn(8, 4, generator=g)}, 0)
    q.put((rank, {k: v.numpy().copy() for k, v in m.student_denoiser.state_dict().items()}, [t.numpy().copy() for t in seen[0]]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_exchange_is_a_mean_of_the_rank_gradients():
    """AdamW is scale-invariant in the gradient, so comparing parameters after AdamW cannot tell a sum from a
    mean.  Here (a) the gradient the step receives -- after the all-reduce, times the factor the step applies -- is compared with
    the single-process mean of the two shard gradients, and (b) the optimizer is SGD (lr 0.5), whose update IS the gradient: the
    two-rank parameters equal the mean-gradient step and are far from the sum-gradient step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_sgd_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    res = {r: ({k: torch.from_numpy(v) for k, v in sd.items()}, [torch.from_numpy(t) for t in gr]) for r, sd, gr in got}
    torch.manual_seed(0)
    m = _Toy()
    params = list(m.student_denoiser.parameters())
    before = [p.detach().clone() for p in params]
    grads = []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        for p in params:
            p.grad = None
        m({"x": torch.randn(8, 4, generator=g)})["loss"][0].backward()
        grads.append([p.grad.clone() for p in params])
    mean = [(a + b) / 2 for a, b in zip(*grads)]
    for r in (0, 1):
        for gm, gg in zip(mean, res[r][1]):
            assert torch.allclose(gg, gm, atol=1e-7, rtol=1e-6), "the step's gradient is not the mean over the ranks"
    names = [k for k, _ in m.student_denoiser.named_parameters()]
    for k, b, gm in zip(names, before, mean):
        step_mean, step_sum = b - 0.5 * gm, b - 0.5 * 2 * gm
        assert torch.allclose(res[0][0][k], step_mean, atol=1e-6), k
        assert torch.equal(res[0][0][k], res[1][0][k])
        assert (res[0][0][k] - step_sum).abs().max() > 10 * (res[0][0][k] - step_mean).abs().max() + 1e-5, "cannot tell sum from mean"


def _start_idx_worker(rank, world, port, q, share):
    """one rank: the PRODUCT FlashDiffusion's timestep selection (FD:135-177) under a TrainingPipeline; torch's global RNG is
    seeded differently per rank, as in a real data-parallel job"""
    import torch.distributed as dist
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import build_models
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)
    teacher, student, _ = build_models()
    m = FlashDiffusion(FlashDiffusionConfig(K=[8], num_iterations_per_K=[100], timestep_distribution="uniform"),
                       student_denoiser=student, teacher_denoiser=teacher, teacher_noise_scheduler=DPMSolverMultistepScheduler())
    pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-3],
                                              trainable_params=[["student_denoiser"]]), overlap=False,
                            **({} if share is None else {"share_start_idx": share}))
    assert pipe.share_start_idx == (True if share is None else share)
    idx = []
    for i in range(24):
        torch.randn(3)                                          # the ranks' global RNG streams drift apart, as noise draws do
        m.iter_steps += 1                                       # what forward() does first (FD:181)
        idx.append(int(m._get_timesteps(Draws(), 2, 8, 0, "cpu")[0]))
        if rank == 0 and i % 5 == 2:
            # a rank-LOCAL extra draw (sample logging / validation on rank 0 only): the shared index is a function of (seed,
            # forward counter) and holds no generator state, so the ranks cannot drift apart
            m._get_timesteps(Draws(), 2, 8, 0, "cpu")
    # round 6: a rank-local FORWARD advances the model's forward counter -- with the counter the trainer owns
    # (TrainingPipeline.training_step sets model.shared_start_step) the ranks still agree
    idx2 = []
    for i in range(12):
        m.shared_start_step = 100 + i                           # what training_step does before the forward
        m.iter_steps += 1 + (rank == 0 and i % 3 == 1)          # rank 0 ran an extra forward (validation) in between
        idx2.append(int(m._get_timesteps(Draws(), 2, 8, 0, "cpu")[0]))
    q.put((rank, idx + [-1] + idx2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("share", [None, False])
def test_ranks_share_the_start_index_by_default(share):
    """SURVEY 8e / VERDICT r3 item 8: with more than one rank the trainer makes all ranks draw the SAME start index per step (same
    teacher-loop length, nobody waits), from identically seeded host generators; share_start_idx=False is the reference's per-rank
    draw.  Either way every rank's own sequence covers the K start indices (the marginal is the reference's pmf)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + (0 if share is None else 7)
    procs = [ctx.Process(target=_start_idx_worker, args=(r, 2, port, q, share)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
    tail = {r: v[v.index(-1) + 1:] for r, v in res.items()}     # the draws under the trainer-owned step counter
    res = {r: v[:v.index(-1)] for r, v in res.items()}
    assert all(0 <= i < 8 for i in res[0] + res[1]) and len(set(res[0])) >= 4
    if share is None:
        assert tail[0] == tail[1] and len(set(tail[0])) >= 3, (tail[0], tail[1])
    if share is None:
        assert res[0] == res[1], (res[0], res[1])
    else:
        assert res[0] != res[1]                                 # 24 independent uniform draws over 8 values


def test_lcm_scheduler_host_side_matches_oracle():
    """schedule selection and boundary scalings of the product LCMScheduler (host fp32 math) vs the oracle restatement"""
    from flash_diffusion_amd.schedulers import LCMScheduler
    from oracle.sched_cpu import LCMSchedulerRef
    a, b = LCMScheduler(), LCMSchedulerRef()
    for n in (1, 2, 4, 8, 50):
        a.set_timesteps(n)
        b.set_timesteps(n)
        assert a.timesteps.tolist() == b.timesteps.tolist()
    a.set_timesteps(timesteps=[999, 749, 499, 249])
    b.set_timesteps(timesteps=torch.tensor([999, 749, 499, 249]))
    assert a.timesteps.tolist() == b.timesteps.tolist() and a.num_inference_steps == 4
    for t in (999, 500, 3, 0):
        cs, co = a.boundary_scalings(t)
        rs, ro = b.boundary_scalings(float(t))
        assert abs(cs - rs) < 1e-12 and abs(co - ro) < 1e-12
    import pytest as _pt
    with _pt.raises(ValueError):
        a.set_timesteps()
    with _pt.raises(ValueError):
        a.set_timesteps(4, timesteps=[9, 5])
    with _pt.raises(ValueError):
        a.set_timesteps(timesteps=[5, 9])
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod) and a.init_noise_sigma == b.init_noise_sigma == 1.0


def test_flow_match_scheduler_host_side_matches_oracle():
    """sigma-shifted schedule of the product FlowMatchEulerDiscreteScheduler (host fp32 math) vs the oracle restatement"""
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler, get_sigmas
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    a, b = FlowMatchEulerDiscreteScheduler(), FlowMatchEulerDiscreteSchedulerRef()
    assert torch.equal(a.timesteps, b.timesteps) and torch.equal(a.sigmas, b.sigmas)       # the 1000-step training schedule
    for n in (1, 4, 8, 32):
        a.set_timesteps(n)
        b.set_timesteps(n)
        assert torch.equal(a.timesteps, b.timesteps) and torch.equal(a.sigmas, b.sigmas) and float(a.sigmas[-1]) == 0.0
    a.set_timesteps(4)
    sig = get_sigmas(a, a.timesteps[[0, 2, 3]])
    assert torch.equal(sig, a.sigmas[[0, 2, 3]])
    d0 = a.step_delta(a.timesteps[0])
    d1 = a.step_delta(a.timesteps[1])
    assert abs(d0 - float(a.sigmas[1] - a.sigmas[0])) < 1e-12 and abs(d1 - float(a.sigmas[2] - a.sigmas[1])) < 1e-12
    c = FlashDiffusionSD3Config(K=[4, 2], num_iterations_per_K=[10, 20], guidance_scale_min=2.0)
    assert c.guidance_scale_min == [2.0, 2.0] and c.mixture_num_components == [4, 4] and c.distill_loss_scale == [1.0, 1.0]


@pytest.mark.parametrize("K,si,g", [(4, 0, 8.0), (4, 1, 3.5), (8, 3, 5.0), (1, 0, 7.0)])
def test_dpm_loop_coefficients_reproduce_the_stepwise_scheduler(monkeypatch, K, si, g):
    """schedulers.DPMSolverMultistepScheduler.loop_coefficients -- the [n][6] host table handed to the C-ABI's
    fdmi_teacher_loop (x0 = a0 x + a1 e_c + a2 e_u ; x = a3 x + a4 x0 + a5 x0_prev) -- against the step-by-step
    fused_cfg_step it replaces, both evaluated on CPU with the stand-in axpby of tests/fake_ops.py"""
    from flash_diffusion_amd import schedulers
    from tests import fake_ops
    monkeypatch.setattr(schedulers, "ops", fake_ops)
    gen = torch.Generator().manual_seed(K * 10 + si)
    x = torch.randn(2, 4, 8, 8, generator=gen)
    eps = [(torch.randn(2, 4, 8, 8, generator=gen), torch.randn(2, 4, 8, 8, generator=gen)) for _ in range(K - si)]
    sch = schedulers.DPMSolverMultistepScheduler()
    sch.set_timesteps(K)
    want = x
    for (e_c, e_u), t in zip(eps, sch.timesteps[si:]):
        want = sch.fused_cfg_step(e_c, e_u, g, t, want)
    sch.set_timesteps(K)
    rows = sch.loop_coefficients(si, g)
    assert len(rows) == K - si and all(len(r) == 6 for r in rows)
    got, prev = x, None
    for (e_c, e_u), a in zip(eps, rows):
        x0 = a[0] * got + a[1] * e_c + a[2] * e_u
        got = a[3] * got + a[4] * x0 + (a[5] * prev if prev is not None else 0.0)
        assert prev is not None or a[5] == 0.0
        prev = x0
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), float((got - want).abs().max())


def test_lr_schedulers_are_built_and_stepped_like_the_reference_recipe():
    """TR:140-166: a torch.optim.lr_scheduler class by name per optimizer; one-optimizer (automatic) mode steps an
    interval="step" scheduler every `frequency` optimizer steps; a None entry means no scheduler."""
    m = _Toy()
    pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-2], trainable_params=[["student_denoiser"]],
                                              lr_schedulers_name=["StepLR"], lr_schedulers_kwargs=[dict(step_size=1, gamma=0.5)],
                                              lr_schedulers_interval=["step"], lr_schedulers_frequency=[2]), overlap=False)
    ret = pipe.configure_optimizers()
    assert isinstance(ret, tuple) and ret[1][0]["interval"] == "step" and ret[1][0]["frequency"] == 2
    lrs = []
    for i in range(4):
        pipe.training_step({"x": torch.randn(8, 4)}, i)
        lrs.append(pipe.optims[0].param_groups[0]["lr"])
    assert lrs == [1e-2, 5e-3, 5e-3, 2.5e-3]
    p2 = TrainingPipeline(_Toy(), TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-2],
                                                 trainable_params=[["student_denoiser"]]), overlap=False)
    assert isinstance(p2.configure_optimizers(), list) and p2.lr_schedulers == [None]


def test_fused_adamw_refuses_keywords_it_does_not_implement():
    from flash_diffusion_amd.trainer import FusedAdamW
    p = [torch.nn.Parameter(torch.zeros(4))]
    FusedAdamW(p, lr=1e-3, amsgrad=False, foreach=None, fused=None)       # torch.optim.AdamW defaults are accepted
    for kw in (dict(amsgrad=True), dict(maximize=True)):
        with pytest.raises(NotImplementedError):
            FusedAdamW(p, lr=1e-3, **kw)
    with pytest.raises(TypeError):
        FusedAdamW(p, lr=1e-3, nesterov=True)


class _ToyHook(nn.Module):
    """a model that calls the trainer's before_student hook like FlashDiffusion.forward: after the (frozen) teacher, before
    the student reads its parameters"""
    calls_before_student = True

    def __init__(self):
        super().__init__()
        self.student_denoiser = nn.Linear(4, 4)
        self.teacher_denoiser = nn.Linear(4, 4)
        self.discriminator = None
        self.order = []
        for p in self.teacher_denoiser.parameters():
            p.requires_grad = False

    def forward(self, batch, **kw):
        t = self.teacher_denoiser(batch["x"]).detach()
        self.order.append("teacher")
        hook = getattr(self, "before_student", None)
        if hook is not None:
            hook()
        self.order.append("student")
        s = self.student_denoiser(batch["x"])
        return {"loss": [((s - t) ** 2).mean(), 0], "start_timestep": 1}


def test_deferred_backward_is_the_same_training_run():
    """the backward + AdamW of iteration i issued from the hook of iteration i+1 (what the GPU path does to run it beside the
    next teacher loop): parameters after N steps + finish() equal the immediate schedule bit for bit, the step happens
    before the next student forward reads the parameters, and finish() / state_dict() drain the outstanding one"""
    xs = [torch.randn(8, 4, generator=torch.Generator().manual_seed(i)) for i in range(4)]

    def train(defer):
        torch.manual_seed(0)
        m = _ToyHook()
        pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-2],
                                                  lr_schedulers_name=["StepLR"], lr_schedulers_kwargs=[{"step_size": 2, "gamma": 0.5}],
                                                  trainable_params=[["student_denoiser"]]), overlap=False)
        pipe.configure_optimizers()
        if defer:   # what configure_optimizers sets up on a GPU (no streams on this box: the comm stream is a placeholder)
            pipe._comm_stream = object()
            m.before_student = pipe._before_student
            assert pipe._defer_ok()
        snaps = []
        for i, x in enumerate(xs):
            w0 = m.student_denoiser.weight.detach().clone()
            pipe.training_step({"x": x}, i)
            snaps.append((w0, m.student_denoiser.weight.detach().clone(), pipe._deferred is not None))
        sd = pipe.state_dict()   # drains
        assert pipe._deferred is None
        return m, snaps, {k: v.clone() for k, v in sd.items()}, pipe.optims[0].param_groups[0]["lr"]

    m0, s0, sd0, lr0 = train(False)
    m1, s1, sd1, lr1 = train(True)
    assert all(torch.equal(sd0[k], sd1[k]) for k in sd0) and lr0 == lr1
    assert all(not pend for _, _, pend in s0) and all(pend for _, _, pend in s1)
    # deferred: training_step(i) returns with the parameters of step i-1 applied (its own update is outstanding) ...
    assert torch.equal(s1[0][0], s1[0][1]) and torch.equal(s1[1][1], s0[0][1]) and torch.equal(s1[3][1], s0[2][1])
    # ... and the student forward of step i+1 saw them updated: same losses means same trajectory (checked through sd above)
    assert m1.order == ["teacher", "student"] * 4
    os.environ["FDMI_DEFER_BACKWARD"] = "0"
    try:
        torch.manual_seed(0)
        m = _ToyHook()
        pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-2],
                                                  trainable_params=[["student_denoiser"]]), overlap=False)
        pipe.configure_optimizers()
        pipe._comm_stream = object()
        assert not pipe._defer_ok()
    finally:
        del os.environ["FDMI_DEFER_BACKWARD"]


class _ToyHook2(_ToyHook):
    """+ a discriminator and the reference's step-selected losses: loss[0] (generator: distillation + a term through the
    discriminator) at step 0, loss[1] (discriminator) at step 1"""

    def __init__(self):
        super().__init__()
        self.discriminator = nn.Sequential(nn.Linear(4, 1))

    def forward(self, batch, step=0, **kw):
        t = self.teacher_denoiser(batch["x"]).detach()
        hook = getattr(self, "before_student", None)
        if hook is not None:
            hook()
        s = self.student_denoiser(batch["x"])
        if step % 2 == 0:
            return {"loss": [((s - t) ** 2).mean() - self.discriminator(s).mean(), 0], "start_timestep": 1}
        return {"loss": [0, self.discriminator(s.detach()).mean() - self.discriminator(t).mean()], "start_timestep": 1}


def test_deferred_backward_two_optimizers_is_the_same_training_run():
    """manual (two-optimizer) loop: each forward's backward issued from the NEXT forward's hook, with the optimizer toggle applied
    around it -- same parameters (student and discriminator) as the immediate schedule after 3 batches, bit for bit; the
    generator's backward leaves the discriminator's gradients untouched"""
    xs = [torch.randn(8, 4, generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]

    def train(defer):
        torch.manual_seed(0)
        m = _ToyHook2()
        pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW", "AdamW"], learning_rates=[1e-2, 3e-2],
                                                  trainable_params=[["student_denoiser"], ["discriminator."]]), overlap=False)
        pipe.configure_optimizers()
        assert not pipe.automatic_optimization
        if defer:
            pipe._comm_stream = object()
            m.before_student = pipe._before_student
        for i, x in enumerate(xs):
            out = pipe.training_step({"x": x}, i)
            assert "loss_optimizer_0" in out and "loss_optimizer_1" in out
            assert (pipe._deferred is not None) == defer
        pipe.finish()
        assert pipe._deferred is None and not pipe._toggled
        assert all(p.requires_grad for p in m.discriminator.parameters())
        return {k: v.clone() for k, v in m.state_dict().items()}

    a, b = train(False), train(True)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert any(k.startswith("discriminator") for k in a)


def test_epoch_interval_scheduler_steps_after_the_deferred_optimizer_step():
    """with the deferred backward the last optimizer step of an epoch is still outstanding when
    on_train_epoch_end() runs -- it must be issued with THIS epoch's lr before an interval="epoch" scheduler advances, and
    readers that bypass the pipeline (optimizers(), optimizer.state_dict(), model.state_dict()) must drain it as well:
    parameters, lr trajectory and optimizer state equal the immediate schedule bit for bit."""
    xs = [torch.randn(8, 4, generator=torch.Generator().manual_seed(40 + i)) for i in range(6)]

    def train(defer, reader):
        torch.manual_seed(0)
        m = _ToyHook()
        pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-1],
                                                  lr_schedulers_name=["StepLR"], lr_schedulers_kwargs=[{"step_size": 1, "gamma": 0.1}],
                                                  lr_schedulers_interval=["epoch"], trainable_params=[["student_denoiser"]]),
                                overlap=False)
        pipe.configure_optimizers()
        if defer:
            pipe._comm_stream = object()
            m.before_student = pipe._before_student
        lrs, seen = [], None
        for epoch in range(2):
            for i in range(3):
                pipe.training_step({"x": xs[epoch * 3 + i]}, i)
                lrs.append(pipe.optims[0].param_groups[0]["lr"])
            if epoch == 0:
                assert (pipe._deferred is not None) == defer
            pipe.on_train_epoch_end()
            assert pipe._deferred is None                   # drained BEFORE the scheduler moved the lr
        pipe.training_step({"x": xs[0]}, 0)
        if reader == "optimizers":
            seen = pipe.optimizers()[0].state_dict()
        elif reader == "opt_state":
            seen = pipe.optims[0].state_dict()
        else:
            seen = m.state_dict()
        assert pipe._deferred is None
        return {k: v.clone() for k, v in m.state_dict().items()}, lrs, seen

    for reader in ("optimizers", "opt_state", "model_state"):
        (a, la, sa), (b, lb, sb) = train(False, reader), train(True, reader)
        assert la == lb and all(torch.equal(a[k], b[k]) for k in a), reader
        if reader != "model_state":
            ea, eb = sa["state"], sb["state"]      # (torch.optim.AdamW on this CPU box: exp_avg / exp_avg_sq / step)
            assert ea.keys() == eb.keys() and len(ea) > 0, reader
            for k in ea:
                assert all(torch.equal(torch.as_tensor(ea[k][n]), torch.as_tensor(eb[k][n])) for n in ea[k]), reader
                assert float(ea[k]["step"]) == 7.0


def test_scheduler_lists_are_sized_by_lr_schedulers_name_like_the_reference():
    """training_config.py:108-131 sizes the scheduler lists by len(lr_schedulers_name), TR:140-166 iterates
    over that: two optimizers with ONE named scheduler is a valid reference config (optimizer 1 gets none)."""
    cfg = TrainingConfig(optimizers_name=["AdamW", "AdamW"], learning_rates=[1e-2, 1e-2],
                         trainable_params=[["student_denoiser"], ["discriminator."]],
                         lr_schedulers_name=["StepLR"], lr_schedulers_kwargs=[dict(step_size=1, gamma=0.5)])
    assert cfg.lr_schedulers_interval == ["step"] and cfg.lr_schedulers_frequency == [1]
    pipe = TrainingPipeline(_ToyHook2(), cfg, overlap=False)
    pipe.configure_optimizers()
    assert len(pipe.lr_schedulers) == 1 and pipe.lr_schedulers[0]["scheduler"].optimizer is pipe.optims[0]
    pipe.training_step({"x": torch.randn(8, 4)}, 0)        # the manual loop: index 1 has no scheduler entry
    pipe.on_train_epoch_end()
    # mismatched explicit lists are refused with the reference's message
    with pytest.raises(AssertionError, match="lr_schedulers_kwargs"):
        TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-2], lr_schedulers_name=["StepLR", None],
                       lr_schedulers_kwargs=[dict(step_size=1)])
    with pytest.raises(AssertionError, match="lr_schedulers_frequency"):
        TrainingConfig(optimizers_name=["AdamW", "AdamW"], learning_rates=[1e-2, 1e-2], lr_schedulers_name=["StepLR", None],
                       lr_schedulers_kwargs=[dict(step_size=1), {}], lr_schedulers_frequency=[2])
    # explicit values equal to nothing special are kept as given (no silent broadcast of a user's [2])
    c2 = TrainingConfig(optimizers_name=["AdamW", "AdamW"], learning_rates=[1e-2, 1e-2], lr_schedulers_name=["StepLR", "StepLR"],
                        lr_schedulers_kwargs=[dict(step_size=1), dict(step_size=2)], lr_schedulers_frequency=[2, 3])
    assert c2.lr_schedulers_frequency == [2, 3] and c2.lr_schedulers_interval == ["step", "step"]


def test_t5_relative_position_buckets_match_transformers():
    """host part of the T5 text encoder (flash_diffusion_amd/t5.py): the bidirectional bucketing of key - query positions"""
    from transformers.models.t5.modeling_t5 import T5Attention
    from flash_diffusion_amd.t5 import relative_position_bucket
    for S, nb, md in ((120, 32, 128), (512, 32, 128), (77, 16, 20)):
        pos = torch.arange(S)
        rel = pos[None, :] - pos[:, None]
        ref = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=nb, max_distance=md)
        assert torch.equal(relative_position_bucket(rel, nb, md), ref)


def test_tiled_vae_decode_merges_like_the_reference():
    """MiAutoencoderKLDiffusers._decode_tiled (tiles batched through the decoder, merged where they are) with a toy decoder on
    the host against oracle/tiler_ref.py (pinned to the reference's Tiler): partial trailing tiles, asymmetric overlaps, batch"""
    from types import SimpleNamespace
    from flash_diffusion_amd.nets import MiAutoencoderKLDiffusers
    from oracle.tiler_ref import tiled_decode_ref
    torch.manual_seed(1)
    mix = torch.randn(3, 4)

    def decode(t):
        u = torch.nn.functional.interpolate(t, scale_factor=4, mode="nearest")
        o = torch.einsum("oc,bchw->bohw", mix, u)
        return o + 0.01 * torch.arange(o.shape[-1])[None, None, None, :]
    fake = SimpleNamespace(_up=4, config=SimpleNamespace(latent_channels=4, scaling_factor=1.0, latents_mean=None, latents_std=None))
    for (H, W, ts, ov, tb) in [(40, 52, (16, 16), (4, 4), 8), (32, 32, (16, 16), (4, 4), 3), (33, 47, (16, 24), (5, 7), 1)]:
        w = MiAutoencoderKLDiffusers.__new__(MiAutoencoderKLDiffusers)
        torch.nn.Module.__init__(w)
        w.vae_model, w.tiling_size, w.tiling_overlap, w.tile_batch, w.downsampling_factor = fake, ts, ov, tb, 4
        z = torch.randn(3, 4, H, W)
        got = w._decode_tiled(z, decode_fn=decode)
        ref = tiled_decode_ref(z, decode, ts, ov, 4)
        assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), (H, W)
