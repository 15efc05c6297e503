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
