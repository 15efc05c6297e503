"""Host logic of the PixArt DiT path (flash_diffusion_amd/dit.py) on CPU: the module's launch sequence and
its hand-written backward formulas are run with tests/fake_ops.py standing in for the HIP op layer (same signatures, bf16
storage mimicked) and compared with the fp32 oracle (oracle/dit_cpu.py).  This checks composition and gradient plumbing
only -- which operands reach which op, strides / leading dimensions, LoRA and adaLN gradient formulas, parameter naming --
not kernel numerics (those are the -m gpu tests, which call the real libfdmi.so)."""
import copy

import pytest
import torch

from oracle import dit_cpu
from tests import fake_ops


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def _inputs(B=2, L=7, masked=False, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([999.0, 250.0])[:B]
    cond = {"crossattn": torch.randn(B, L, 48, generator=g), "vector": torch.randn(B, 32, generator=g)}
    if masked:
        m = torch.ones(B, L, dtype=torch.long)
        m[0, 4:] = 0
        cond["attention_mask"] = m
    return x, t, {"cond": cond}


def _product_from(oracle_model, monkeypatch, lora_r=0):
    from flash_diffusion_amd import dit
    monkeypatch.setattr(dit, "ops", fake_ops)
    m = dit.MiTransformer2DModel(**dit_cpu.TINY_DIT)
    if lora_r:
        m.add_adapter(lora_r)
    sd = {k.replace(".base_layer.", "."): v for k, v in oracle_model.state_dict().items()}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m


def test_parameter_names_match_the_reference_state_dict():
    from flash_diffusion_amd import dit
    ref = dit_cpu.PixartTransformerRef(**dit_cpu.TINY_DIT)
    mine = dit.MiTransformer2DModel(**dit_cpu.TINY_DIT)
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert a == b
    names = dit_cpu.add_lora_(ref, 8)
    mine.add_adapter(8)
    a = {k.replace(".base_layer.", "."): tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert a == b and len(names) == len(mine.lora_parameters()) // 2
    assert all(p.requires_grad == (".lora_" in n) for n, p in mine.named_parameters())


@pytest.mark.parametrize("masked", [False, True])
def test_frozen_forward_and_input_gradient(monkeypatch, masked):
    ref = dit_cpu.seeded_init_(dit_cpu.PixartTransformerRef(**dit_cpu.TINY_DIT), 3)
    ref.freeze()
    mine = _product_from(ref, monkeypatch)
    mine.freeze()
    x, t, cond = _inputs(masked=masked)
    with torch.no_grad():
        want, got = ref(x, t, cond), mine(x, t, cond)
    assert got.shape == want.shape == (2, 4, 16, 16) and got.dtype == torch.float32
    assert _rel(got, want) < 2e-2, _rel(got, want)
    assert mine.last_flops > 0
    # float / int timesteps are broadcast (UW/TW accept them for the UNets; here a convenience)
    with torch.no_grad():
        assert _rel(mine(x, 999.0, cond), ref(x, torch.full((2,), 999.0), cond)) < 2e-2
    # gradient w.r.t. the input through the frozen network (the GAN generator step back-propagates through the teacher)
    w = torch.randn(want.shape, generator=torch.Generator().manual_seed(5))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    (ref(xa, t, cond) * w).sum().backward()
    (mine(xb, t, cond) * w).sum().backward()
    assert _cos(xb.grad, xa.grad) > 0.995 and _rel(xb.grad, xa.grad) < 8e-2, (_cos(xb.grad, xa.grad), _rel(xb.grad, xa.grad))


@pytest.mark.parametrize("masked", [False, True])
def test_lora_student_gradients(monkeypatch, masked):
    ref = dit_cpu.seeded_init_(dit_cpu.PixartTransformerRef(**dit_cpu.TINY_DIT), 3)
    dit_cpu.add_lora_(ref, 8, seed=4, b_std=0.05)
    mine = _product_from(ref, monkeypatch, lora_r=8)
    x, t, cond = _inputs(masked=masked)
    w = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    want, got = ref(x, t, cond), mine(x, t, cond)
    assert _rel(got, want) < 2e-2, _rel(got, want)
    (want * w).sum().backward()
    (got * w).sum().backward()
    refg = {k.replace(".base_layer.", "."): p.grad for k, p in ref.named_parameters()}
    n = 0
    worst = (1.0, "")
    for k, p in mine.named_parameters():
        if ".lora_" not in k:
            assert p.grad is None, k
            continue
        assert p.grad is not None and p.grad.shape == p.shape, k
        r = refg[k]
        if float(r.norm()) < 1e-9:
            continue
        c = _cos(p.grad, r)
        worst = min(worst, (c, k))
        assert c > 0.999 and _rel(p.grad, r) < 6e-2, (k, c, _rel(p.grad, r))
        n += 1
    assert n == len(mine.lora_parameters()), (n, worst)
    # a second forward after an in-place update of the LoRA masters must see the new weights
    with torch.no_grad():
        for p in mine.lora_parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(9)))
        for (k, p) in ref.named_parameters():
            if ".lora_" in k:
                p.copy_(dict(mine.named_parameters())[k.replace(".base_layer.", ".")])
        assert _rel(mine(x, t, cond), ref(x, t, cond)) < 2e-2


def test_deepcopy_gives_an_independent_student(monkeypatch):
    ref = dit_cpu.seeded_init_(dit_cpu.PixartTransformerRef(**dit_cpu.TINY_DIT), 3)
    teacher = _product_from(ref, monkeypatch)
    student = copy.deepcopy(teacher)
    student.add_adapter(8, init_std_b=0.05, generator=torch.Generator().manual_seed(2))
    teacher.freeze()
    x, t, cond = _inputs()
    with torch.no_grad():
        a, b = teacher(x, t, cond), student(x, t, cond)
    assert _rel(a, b) > 1e-3 and teacher.lora_r == 0 and student.lora_r == 8
    assert all(m._owner[0] is student for m in student.modules() if hasattr(m, "_owner"))


def test_unsupported_configurations_raise():
    from flash_diffusion_amd import dit
    with pytest.raises(NotImplementedError):
        dit.MiTransformer2DModel(**{**dit_cpu.TINY_DIT, "norm_type": "layer_norm"})
    with pytest.raises(NotImplementedError):
        dit.MiTransformer2DModel(**{**dit_cpu.TINY_DIT, "double_self_attention": True})


@pytest.mark.parametrize("name", ["dit_tiny", "dit_hd72_masked"])
def test_composition_against_the_reference_fixture(monkeypatch, name):
    """same composition check against the fixtures made by the reference's REAL wrapper (head dim 72, three vector
    conditionings, padded keys) -- the data the -m gpu test uses with the real kernels"""
    from oracle.golden_cases import build_dit
    from tests.golden_util import load_case, rel_err
    g = load_case(name)
    cfg, ora, (x, t, cond), w = build_dit(name, lora_r=8)
    from flash_diffusion_amd import dit
    monkeypatch.setattr(dit, "ops", fake_ops)
    m = dit.MiTransformer2DModel(**cfg).add_adapter(8)
    m.load_state_dict({k.replace(".base_layer.", "."): v for k, v in ora.state_dict().items()})
    out = m(x, t, cond)
    assert rel_err(out, g["out"]["lora"]) < 2e-2
    (out * w).sum().backward()
    for k, p in m.named_parameters():
        if ".lora_" in k:
            assert _cos(p.grad, g["grads"][k]) > 0.999 and rel_err(p.grad, g["grads"][k]) < 6e-2, k


# ---- SD3 MMDiT (row a18) ------------------------------------------------------------------------------------------------------
def test_mmdit_state_dict_keys_match_the_reference():
    from flash_diffusion_amd import dit
    from oracle import mmdit_cpu
    ref = mmdit_cpu.SD3TransformerRef(**mmdit_cpu.TINY_MMDIT)
    mine = dit.MiSD3Transformer2DModel(**mmdit_cpu.TINY_MMDIT)
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert a == b and "pos_embed.pos_embed" in b                      # strict=True load (examples/train_flash_sd3.py:79)
    assert torch.equal(ref.pos_embed.pos_embed, mine.pos_embed.pos_embed)
    names = dit_cpu.add_lora_(ref, 8)
    mine.add_adapter(8)
    a = {k.replace(".base_layer.", "."): tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == {k: tuple(v.shape) for k, v in mine.state_dict().items()} and len(names) == len(mine.lora_parameters()) // 2


@pytest.mark.parametrize("name", ["mmdit_tiny", "mmdit_hd64"])
def test_mmdit_composition_against_the_reference_fixture(monkeypatch, name):
    from flash_diffusion_amd import dit
    from oracle.golden_cases import build_mmdit
    from tests.golden_util import load_case, rel_err
    monkeypatch.setattr(dit, "ops", fake_ops)
    g = load_case(name)
    cfg, ora, (x, t, cond), w = build_mmdit(name)
    m = dit.MiSD3Transformer2DModel(**cfg)
    m.load_state_dict(ora.state_dict(), strict=True)
    m.freeze()
    with torch.no_grad():
        assert rel_err(m(x, t, cond), g["out"]["frozen"]) < 2e-2
        assert rel_err(m(x, 999.0, cond), ora(x, torch.full((2,), 999.0), cond)) < 2e-2
    xa = x.clone().requires_grad_(True)                               # input gradient through the frozen network
    xb = x.clone().requires_grad_(True)
    (ora(xa, t, cond) * w).sum().backward()
    (m(xb, t, cond) * w).sum().backward()
    assert _cos(xb.grad, xa.grad) > 0.995 and _rel(xb.grad, xa.grad) < 8e-2
    cfg, ora, (x, t, cond), w = build_mmdit(name, lora_r=8)
    m = dit.MiSD3Transformer2DModel(**cfg).add_adapter(8)
    m.load_state_dict({k.replace(".base_layer.", "."): v for k, v in ora.state_dict().items()})
    out = m(x, t, cond)
    assert rel_err(out, g["out"]["lora"]) < 2e-2
    (out * w).sum().backward()
    n = 0
    for k, p in m.named_parameters():
        if ".lora_" in k:
            assert _cos(p.grad, g["grads"][k]) > 0.999 and rel_err(p.grad, g["grads"][k]) < 6e-2, k
            n += 1
        else:
            assert p.grad is None, k
    assert n == len(g["grads"])


@pytest.mark.parametrize("kw", [dict(num_steps=3, guidance_scale=2.5), dict(num_steps=2, guidance_scale=1.5, max_samples=1,
                                                                            log_teacher_samples=True)])
def test_sd3_sampler_over_the_mmdit_batches_cfg(monkeypatch, kw):
    """FlashDiffusionSD3.sample over MiSD3Transformer2DModel: the per-sample denoiser is called ONCE per step on [x | x] with
    [cond | uncond] (the reference makes two calls, FD3:767-795) -- result against the oracle sampler over the oracle MMDiT"""
    from flash_diffusion_amd import dit, flash_sd3
    from oracle.flash_sd3_ref import EmbeddingPipeline, FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.golden_cases import build_mmdit
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    monkeypatch.setattr(dit, "ops", fake_ops)
    monkeypatch.setattr(flash_sd3, "ops", fake_ops)
    cfg, ora, (x, t, cond), _ = build_mmdit("mmdit_tiny")
    _, ora_s, _, _ = build_mmdit("mmdit_tiny", lora_r=8)
    g = torch.Generator().manual_seed(3)
    c = cond["cond"]
    pipe = EmbeddingPipeline(c["crossattn"], c["vector"], torch.randn(c["crossattn"].shape, generator=g),
                             torch.randn(c["vector"].shape, generator=g))
    kcfg = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform")
    ref = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kcfg), student_denoiser=ora_s, teacher_denoiser=ora,
                               teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(),
                               sampling_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(),
                               teacher_sampling_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), pipeline=pipe)
    teacher = dit.MiSD3Transformer2DModel(**cfg)
    teacher.load_state_dict(ora.state_dict())
    teacher.freeze()
    student = dit.MiSD3Transformer2DModel(**cfg).add_adapter(8)
    student.load_state_dict({k.replace(".base_layer.", "."): v for k, v in ora_s.state_dict().items()})
    calls = []
    orig = dit.MiSD3Transformer2DModel.forward
    monkeypatch.setattr(dit.MiSD3Transformer2DModel, "forward",
                        lambda self, *a, **k: (calls.append(k["sample"].shape[0]), orig(self, *a, **k))[1])
    mine = flash_sd3.FlashDiffusionSD3(flash_sd3.FlashDiffusionSD3Config(**kcfg), student_denoiser=student,
                                       teacher_denoiser=teacher,
                                       teacher_noise_scheduler=flash_sd3.FlowMatchEulerDiscreteScheduler(),
                                       sampling_noise_scheduler=flash_sd3.FlowMatchEulerDiscreteScheduler(),
                                       teacher_sampling_noise_scheduler=flash_sd3.FlowMatchEulerDiscreteScheduler(),
                                       pipeline=pipe)
    z = torch.randn(2, 16, 16, 16, generator=g)
    want, want_ref = ref.sample(z, conditioner_inputs={"text": ["a", "b"]}, **kw)
    got, got_ref = mine.sample(z, conditioner_inputs={"text": ["a", "b"]}, **kw)
    n = kw.get("max_samples") or 2
    steps = kw["num_steps"] * (2 if kw.get("log_teacher_samples") else 1)
    assert calls == [2 * n] * steps, calls                       # one 2B call per step
    assert got.shape == want.shape and _rel(got, want) < 3e-2, _rel(got, want)
    if want_ref is not None:
        assert _rel(got_ref, want_ref) < 3e-2


@pytest.mark.parametrize("kind", ["pixart", "sd3"])
def test_flat_lora_buffer_and_fused_adamw(monkeypatch, kind):
    """trainer.py's data-parallel path wants the LoRA tensors of the student in ONE flat fp32 buffer with ONE flat gradient
    (one all-reduce, one AdamW launch): _reflatten_lora re-homes them; the backward GEMMs then accumulate straight into the
    gradient views"""
    from flash_diffusion_amd import dit, trainer
    from oracle.golden_cases import build_dit, build_mmdit
    monkeypatch.setattr(dit, "ops", fake_ops)
    monkeypatch.setattr(trainer, "ops", fake_ops)
    name, build, cls = (("dit_tiny", build_dit, dit.MiTransformer2DModel) if kind == "pixart"
                        else ("mmdit_tiny", build_mmdit, dit.MiSD3Transformer2DModel))
    cfg, ora, (x, t, cond), w = build(name, lora_r=8)
    sd = {k.replace(".base_layer.", "."): v for k, v in ora.state_dict().items()}
    plain = cls(**cfg).add_adapter(8)
    plain.load_state_dict(sd)
    flat = cls(**cfg).add_adapter(8)
    flat.load_state_dict(sd)
    assert flat.lora_rank == 8 and not flat._reflatten_lora(torch.device("cpu")) and flat._reflatten_lora(torch.device("cpu"))
    buf, gbuf = flat.lora_flat(), flat.lora_flat_grad()
    assert buf.numel() == sum(p.numel() for p in flat.lora_parameters()) == gbuf.numel()
    off = 0
    for p in flat.lora_parameters():                                   # views, in named_parameters order
        assert p.data_ptr() == buf.data_ptr() + 4 * off and p.grad.data_ptr() == gbuf.data_ptr() + 4 * off
        off += p.numel()
    (plain(x, t, cond) * w).sum().backward()
    (flat(x, t, cond) * w).sum().backward()
    for (k, a), b in zip(plain.named_parameters(), flat.parameters()):
        if ".lora_" in k:
            assert torch.equal(a.grad, b.grad), k
    g1 = gbuf.clone()
    assert float(g1.abs().sum()) > 0
    (flat(x, t, cond) * w).sum().backward()                            # a second backward accumulates
    assert torch.allclose(gbuf, 2 * g1, rtol=1e-5, atol=1e-7)
    gbuf.zero_()                                                       # what TrainingPipeline._zero_grad does
    (flat(x, t, cond) * w).sum().backward()
    assert torch.allclose(gbuf, g1, rtol=1e-5, atol=1e-7)
    # one fused AdamW launch on the flat buffer == torch.optim.AdamW on the separate tensors
    opt = trainer.FusedAdamW(flat.lora_parameters(), lr=1e-2, flat=buf, flat_grad=gbuf)
    ref_opt = torch.optim.AdamW(plain.lora_parameters(), lr=1e-2)
    opt.step()
    ref_opt.step()
    for a, b in zip(plain.lora_parameters(), flat.lora_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    with torch.no_grad():                                              # the next forward sees the updated weights
        assert _rel(flat(x, t, cond), plain(x, t, cond)) < 1e-3
    student = copy.deepcopy(flat)                                      # a copy must not share the buffers
    assert student.lora_flat() is None and all(m._gviews is None for m in student.modules() if hasattr(m, "_gviews"))


# ---- conditioner plumbing against the reference's REAL classes ---------------------------------------------
def test_conditioner_wrapper_matches_the_reference(monkeypatch):
    from oracle import shim_import
    if not shim_import.reference_available():
        pytest.skip("reference sources not present")
    shim_import.import_reference()
    import flash.models.embedders as R
    from flash_diffusion_amd import conditioners as M
    monkeypatch.setattr(M, "ops", fake_ops)
    lin = dict(nn_modules=["torch.nn.Linear", "torch.nn.SiLU"], nn_modules_kwargs=[dict(in_features=6, out_features=5), {}])

    class RefTensor(R.BaseConditioner):
        def forward(self, batch, force_zero_embedding=False, *a, **k):
            x = batch[self.input_key]
            return {self.dim2outputkey[x.dim()]: 0 * x if force_zero_embedding else x}

    def build(ns, tensor_cls):
        cfgk = (lambda cls, **kw: cls(**kw)) if ns is M else None
        if ns is M:
            cs = [M.TensorEmbedder("text", 0.5), M.TimestepsEmbedder(16, input_key="size", unconditional_conditioning_rate=0.3),
                  M.TimestepsEmbedder(16, input_key="crop"), M.TorchNNEmbedder(input_key="pooled", **lin),
                  M.TensorEmbedder("text2", 0.0), M.TensorEmbedder("mask_image", 0.2)]
        else:
            cs = [tensor_cls(R.BaseConditionerConfig(input_key="text", unconditional_conditioning_rate=0.5)),
                  R.TimestepsEmbedder(R.TimestepsEmbedderConfig(num_channels=16, input_key="size",
                                                                unconditional_conditioning_rate=0.3)),
                  R.TimestepsEmbedder(R.TimestepsEmbedderConfig(num_channels=16, input_key="crop")),
                  R.TorchNNEmbedder(R.TorchNNEmbedderConfig(input_key="pooled", **lin)),
                  tensor_cls(R.BaseConditionerConfig(input_key="text2")),
                  tensor_cls(R.BaseConditionerConfig(input_key="mask_image", unconditional_conditioning_rate=0.2))]
        return ns.ConditionerWrapper(cs)

    real, mine = build(R, RefTensor), build(M, None)
    mine.conditioners[3].load_state_dict(real.conditioners[3].state_dict())
    g = torch.Generator().manual_seed(0)
    batch = {"text": torch.randn(3, 7, 12, generator=g), "text2": torch.randn(3, 7, 4, generator=g),
             "size": torch.tensor([[512.0, 512.0], [256.0, 128.0], [1024.0, 768.0]]),
             "crop": torch.tensor([[0.0, 0.0], [16.0, 32.0], [8.0, 0.0]]), "pooled": torch.randn(3, 6, generator=g),
             "mask_image": torch.randn(3, 1, 8, 8, generator=g)}
    for kw in (dict(), dict(set_ucg_rate_zero=True), dict(ucg_keys=["text"]), dict(ucg_keys=["text", "size"], set_ucg_rate_zero=True)):
        for seed in range(6):                                          # same host RNG stream => same dropout decisions
            torch.manual_seed(seed)
            a = real(batch, **kw)
            torch.manual_seed(seed)
            b = mine(batch, **kw)
            assert set(a["cond"]) == set(b["cond"]) == {"crossattn", "vector", "concat"}
            assert a["cond"]["crossattn"].shape == (3, 7, 16) and a["cond"]["vector"].shape == (3, 2 * 32 + 5)
            for k in a["cond"]:
                tol = 8e-3 if k == "vector" else 0.0                    # sinusoids come back from the kernel in bf16
                assert a["cond"][k].shape == b["cond"][k].shape
                assert float((a["cond"][k] - b["cond"][k]).abs().max()) <= tol, (kw, seed, k)
    with pytest.raises(AssertionError):
        M.BaseConditioner("text", 1.5)


# ---- whole steps over the transformer denoisers (C4 / C5 pipelines end to end, on CPU) ---------------------------------------
def _patch_flash(monkeypatch):
    from flash_diffusion_amd import dit, flash, flash_sd3, schedulers
    for mod in (dit, flash, flash_sd3, schedulers):
        monkeypatch.setattr(mod, "ops", fake_ops)
    for mod in (flash, flash_sd3):
        monkeypatch.setattr(mod, "_DistillLoss", fake_ops.FakeDistillLoss)
        monkeypatch.setattr(mod, "_DmdLoss", fake_ops.FakeDmdLoss)


def test_flash_step_over_the_pixart_dit(monkeypatch):
    """C4 end to end: product FlashDiffusion + DPM-Solver++ + MiTransformer2DModel teacher / LoRA student (stand-in ops) against
    the oracle FlashDiffusionRef over the oracle PixArt denoisers, same injected draws; T5-style key mask + vector conditioning
    go through the [cond | uncond] batched teacher call.  bf16 storage is mimicked: 3e-2."""
    from flash_diffusion_amd import dit
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle import flash_ref
    from oracle.golden_cases import build_dit
    from oracle.sched_cpu import DPMSolverMultistepSchedulerRef
    _patch_flash(monkeypatch)
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="l2", use_dmd_loss=True,
              guidance_scale_min=3.0, guidance_scale_max=7.0, dmd_loss_scale=0.3)
    cfg, t_o, (x, t, cond), _ = build_dit("dit_hd72_masked")
    _, s_o, _, _ = build_dit("dit_hd72_masked", lora_r=8)
    t_o.freeze()
    batch = {"image": x, "text": ["a", "b"], **cond["cond"]}

    class OraCond(torch.nn.Module):                        # the oracle's TensorConditioner knows no attention_mask
        def forward(self, b, ucg_keys=None, set_ucg_rate_zero=False, *a, **k):
            drop = ucg_keys is not None and "text" in ucg_keys
            c = {k2: (torch.zeros_like(b[k2]) if drop else b[k2]) for k2 in ("crossattn", "vector")}
            c["attention_mask"] = b["attention_mask"]
            return {"cond": c}
    ref = flash_ref.FlashDiffusionRef(flash_ref.FlashConfigRef(**kw), student_denoiser=s_o, teacher_denoiser=t_o,
                                      teacher_noise_scheduler=DPMSolverMultistepSchedulerRef(), conditioner=OraCond())
    torch.manual_seed(5)
    want = ref(dict(batch), step=0)
    want["loss"][0].backward()
    teacher = dit.MiTransformer2DModel(**cfg)
    teacher.load_state_dict(t_o.state_dict())
    teacher.freeze()
    student = dit.MiTransformer2DModel(**cfg).add_adapter(8)
    student.load_state_dict({k.replace(".base_layer.", "."): v for k, v in s_o.state_dict().items()})
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner())
    m.draws = Draws(ref.last_draws.values)
    got = m(dict(batch), step=0, device="cpu")
    assert got["start_timestep"] == want["start_timestep"]
    for k in ("teacher_output", "student_output"):
        assert _rel(got[k], want[k]) < 3e-2, (k, _rel(got[k], want[k]))
    assert abs(float(got["loss"][0]) - float(want["loss"][0])) < 5e-2 * abs(float(want["loss"][0]))
    got["loss"][0].backward()
    rg = {k.replace(".base_layer.", "."): p.grad for k, p in s_o.named_parameters() if p.grad is not None}
    fa = torch.cat([p.grad.flatten() for k, p in student.named_parameters() if ".lora_" in k])
    fb = torch.cat([rg[k].flatten() for k, p in student.named_parameters() if ".lora_" in k])
    assert _cos(fa, fb) > 0.99, _cos(fa, fb)


def test_sd3_step_over_the_mmdit(monkeypatch):
    """C5 end to end (distillation + DMD): product FlashDiffusionSD3 + flow-match Euler + MiSD3Transformer2DModel teacher / LoRA
    student against the oracle FlashDiffusionSD3Ref over the oracle MMDiT, same injected draws; the teacher loop goes out as
    one 2B call per step"""
    from flash_diffusion_amd import dit, flash_sd3
    from flash_diffusion_amd.flash import Draws
    from oracle.flash_sd3_ref import EmbeddingPipeline, FlashDiffusionSD3Ref, FlashSD3ConfigRef
    from oracle.golden_cases import build_mmdit
    from oracle.sched_cpu import FlowMatchEulerDiscreteSchedulerRef
    _patch_flash(monkeypatch)
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", use_dmd_loss=True, guidance_scale_min=3.0,
              guidance_scale_max=7.0)
    cfg, t_o, (x, t, cond), _ = build_mmdit("mmdit_hd64")
    _, s_o, _, _ = build_mmdit("mmdit_hd64", lora_r=8)
    t_o.freeze()
    c = cond["cond"]
    pipe = EmbeddingPipeline(c["crossattn"], c["vector"], torch.zeros_like(c["crossattn"]), torch.zeros_like(c["vector"]))
    batch = {"image": x, "text": ["a", "b"]}
    ref = FlashDiffusionSD3Ref(FlashSD3ConfigRef(**kw), student_denoiser=s_o, teacher_denoiser=t_o,
                               teacher_noise_scheduler=FlowMatchEulerDiscreteSchedulerRef(), pipeline=pipe)
    torch.manual_seed(6)
    want = ref(dict(batch), step=0)
    wl = want["loss"][0] if isinstance(want["loss"], (list, tuple)) else want["loss"]
    wl.backward()
    teacher = dit.MiSD3Transformer2DModel(**cfg)
    teacher.load_state_dict(t_o.state_dict())
    teacher.freeze()
    student = dit.MiSD3Transformer2DModel(**cfg).add_adapter(8)
    student.load_state_dict({k.replace(".base_layer.", "."): v for k, v in s_o.state_dict().items()})
    m = flash_sd3.FlashDiffusionSD3(flash_sd3.FlashDiffusionSD3Config(**kw), student_denoiser=student, teacher_denoiser=teacher,
                                    teacher_noise_scheduler=flash_sd3.FlowMatchEulerDiscreteScheduler(), pipeline=pipe)
    m.draws = Draws(ref.last_draws.values)
    got = m(dict(batch), step=0)
    gl = got["loss"][0] if isinstance(got["loss"], (list, tuple)) else got["loss"]
    for k in ("teacher_output", "student_output"):
        assert _rel(got[k], want[k]) < 3e-2, (k, _rel(got[k], want[k]))
    assert abs(float(gl) - float(wl)) < 5e-2 * abs(float(wl))
    gl.backward()
    rg = {k.replace(".base_layer.", "."): p.grad for k, p in s_o.named_parameters() if p.grad is not None}
    fa = torch.cat([p.grad.flatten() for k, p in student.named_parameters() if ".lora_" in k])
    fb = torch.cat([rg[k].flatten() for k, p in student.named_parameters() if ".lora_" in k])
    assert _cos(fa, fb) > 0.99, _cos(fa, fb)


def test_full_size_parameter_counts_match_the_published_models():
    """the reference's pinned hyper-parameters (examples/train_flash_pixart.py:63-86, train_flash_sd3.py:65-77) must give the
    published model sizes: PixArt-alpha XL/2 = 0.61 B parameters, SD3-medium MMDiT = 2.03 B (built on the meta device)"""
    from flash_diffusion_amd import dit
    from flash_diffusion_amd.workloads import PIXART, SD3
    with torch.device("meta"):
        pix = dit.MiTransformer2DModel(**PIXART)
        sd3 = dit.MiSD3Transformer2DModel(**SD3)
    assert sum(p.numel() for p in pix.parameters()) == 611_595_680
    assert sum(p.numel() for p in sd3.parameters()) == 2_028_328_000
    assert tuple(sd3.pos_embed.pos_embed.shape) == (1, 192 * 192, 1536)
    n_pix = sum(1 for n, m in pix.named_modules() if isinstance(m, dit.MiLinear)
                and any(n == t or n.endswith("." + t) for t in dit.PIXART_LORA_TARGETS))
    assert n_pix == 293      # 28 blocks x 10 linears + patch conv + 2 + 6 + 2 embedder linears + adaLN + proj_out


@pytest.mark.parametrize("kind", ["pixart", "sd3"])
def test_plan_path_lora_buffers(kind):
    """the host side of the C++ plan path (dit._DenoiserBase): `_reflatten_lora(attach=False)` -- what every plan forward calls --
    makes the flat parameter / gradient buffers exist without touching `.grad`; `_attach_lora_grads` -- what the plan's backward
    calls -- points `.grad` at views of the flat gradient buffer (in named_parameters order: the order `_ensure_packed` binds the
    pairs in) and zeroes it when the gradients were None (optimizer.zero_grad(set_to_none=True)), but keeps accumulated values
    otherwise; the plan binds exactly the modules that carry a LoRA pair, under their module names"""
    from flash_diffusion_amd import dit
    from flash_diffusion_amd.workloads import TINY_PIXART, TINY_SD3
    torch.manual_seed(0)
    m = (dit.MiTransformer2DModel(**TINY_PIXART) if kind == "pixart" else dit.MiSD3Transformer2DModel(**TINY_SD3))
    m.add_adapter(8, init_std_b=0.02)
    lora = [(n, p) for n, p in m.named_parameters() if ".lora_" in n]
    before = {n: p.detach().clone() for n, p in lora}
    assert m._reflatten_lora(torch.device("cpu"), attach=False) is False          # first call: the tensors move into the flat buffer
    assert m._reflatten_lora(torch.device("cpu"), attach=False) is True           # ... and stay there
    assert all(p.grad is None for _, p in lora)
    assert all(torch.equal(p.detach(), before[n]) for n, p in lora)
    flat, g = m.lora_flat(), m.lora_flat_grad()
    assert flat.numel() == g.numel() == sum(p.numel() for _, p in lora)
    off = 0
    for n, p in lora:                                                            # parameters are views, in named_parameters order
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    g.fill_(3.0)                                                                  # stale values of an earlier step
    m._attach_lora_grads()                                                        # grads were None: the buffer is zeroed first
    off = 0
    for n, p in lora:
        assert p.grad is not None and p.grad.shape == p.shape and p.grad.data_ptr() == g.data_ptr() + 4 * off
        off += p.numel()
    assert float(g.abs().max()) == 0.0
    g.fill_(2.0)                                                                  # what a backward accumulated
    m._attach_lora_grads()                                                        # grads attached already: nothing is cleared
    assert float(g.min()) == 2.0
    mods = dict(m._lora_modules())
    assert set(mods) == {n.split(".lora_")[0] for n, _ in lora} and all(v.rank == 8 for v in mods.values())
    assert not m._use_plan(torch.zeros(1))                                        # CPU tensors never take the plan path
