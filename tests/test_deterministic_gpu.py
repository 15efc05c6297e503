"""Deterministic mode (round 6): `ops.deterministic()` = developer knob 50 replaces every
accumulation whose order the production kernels leave to the hardware (fp32 atomics of the GroupNorm-sum epilogues, the statistics
passes, the TN weight-gradient row splits, atomic-accumulating GEMMs, column sums, scalar losses) by an ordered one.

  * two runs of the SAME step in that mode are BIT-IDENTICAL -- outputs, both losses, every LoRA / discriminator gradient -- on the tiny
    step fixtures (UNet: generator and discriminator steps, DMD + GAN), on the full-width C2-shaped SD1.5 step, on the PixArt and SD3
    transformer steps;
  * the streaming TN weight-gradient kernels (single and grouped launches) are held to an fp32 X^T Y at 1e-4 on the step's actual
    shapes in BOTH modes -- the stable check of gradient accuracy the bf16 step bars cannot give (they carry bf16 rounding noise of
    the whole network);
  * the mode's kernels agree with the production ones to fp32 rounding (statistics, column sums, losses).
The parity bodies of tests/test_flash_gpu.py, tests/test_unet_gpu.py and the C2-shaped step of tests/test_fullsize_parity_gpu.py run
in this mode as well, against the bars they had before round 5's noise allowances."""
import copy
import os

import pytest
import torch

from flash_diffusion_amd import ops

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "deterministic.txt")


def log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")


def _tiny_step(name):
    """one step of a tiny fixture from a fresh model: (outputs, losses, {parameter: gradient})"""
    from flash_diffusion_amd.flash import Draws
    from oracle.golden_cases import CASES
    from tests.golden_util import load_case
    from tests.test_flash_gpu import build_product
    kw, sched, step, _ = CASES[name]
    g = load_case(name)
    m = build_product(kw, sched)
    m.draws = Draws(g["draws"])
    B = g["z"].shape[0]
    out = m({"image": g["z"].cuda(), "crossattn": g["crossattn"].cuda(), "text": ["a"] * B}, step=step, device="cuda")
    out["loss"][step].backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    return ({k: out[k].detach().clone() for k in ("teacher_output", "student_output")},
            [out["loss"][i].detach().clone() if torch.is_tensor(out["loss"][i]) else out["loss"][i] for i in (0, 1)], grads)


def _identical(a, b, what):
    outs_a, loss_a, grads_a = a
    outs_b, loss_b, grads_b = b
    for k in outs_a:
        assert torch.equal(outs_a[k], outs_b[k]), (what, k, float((outs_a[k].float() - outs_b[k].float()).abs().max()))
    for i in (0, 1):
        assert float(loss_a[i]) == float(loss_b[i]), (what, "loss", i, float(loss_a[i]), float(loss_b[i]))
    assert grads_a.keys() == grads_b.keys() and len(grads_a) > 0
    bad = [n for n in grads_a if not torch.equal(grads_a[n], grads_b[n])]
    assert not bad, (what, len(bad), "of", len(grads_a), bad[:4])
    return len(grads_a)


@pytest.mark.parametrize("name", ["g_dmd_lsgan", "d_lsgan", "g_wgan"])
def test_two_runs_of_a_tiny_step_are_bit_identical(name):
    with ops.deterministic():
        assert ops.deterministic.enabled()
        a = _tiny_step(name)
        b = _tiny_step(name)
    assert not ops.deterministic.enabled()
    n = _identical(a, b, name)
    log(f"tiny step {name}: two deterministic runs bit-identical (2 outputs, 2 losses, {n} gradient tensors)")


def test_production_mode_run_to_run_distance_is_what_the_mode_removes():
    """(documentation of the noise, not a bar: logs how far two PRODUCTION runs of a tiny generator step are apart)"""
    a = _tiny_step("g_dmd_lsgan")
    b = _tiny_step("g_dmd_lsgan")
    d_out = max(float((a[0][k].float() - b[0][k].float()).abs().max()) for k in a[0])
    d_loss = abs(float(a[1][0]) - float(b[1][0])) / max(abs(float(a[1][0])), 1e-12)
    moved = sum(0 if torch.equal(a[2][n], b[2][n]) else 1 for n in a[2])
    log(f"production mode, two runs of g_dmd_lsgan: outputs max |diff| {d_out:.3e}, loss rel diff {d_loss:.3e}, {moved} of {len(a[2])} gradient tensors differ")
    assert d_loss < 5e-2   # (sanity only)


@pytest.mark.gpu_mem(60)
def test_two_runs_of_the_c2_shaped_step_are_bit_identical():
    """full-width SD1.5, rank-128 LoRA, 4 teacher CFG steps, B = 2 (the C2 fixture's models and draws), bf16 production kernels"""
    from tests.isolate import run_isolated
    run_isolated(__name__, "_c2_twice_body", (), timeout=900)


def _c2_once():
    from flash_diffusion_amd import workloads
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from flash_diffusion_amd.unet import MiUNet2DConditionModel
    from oracle.golden_cases import C2_KW, build_c2_models, c2_batch
    from tests.golden_util import load_case
    g = load_case("c2_sd15_r128_n4")

    def make(lora_rank):
        with torch.device("cuda"):
            m = MiUNet2DConditionModel(**workloads.SD15, precision="bf16")
        m = m.cuda()
        if lora_rank:
            m.add_adapter(lora_rank)
        return m
    teacher, student, disc = build_c2_models("cuda", make)
    teacher.freeze()
    m = FlashDiffusion(FlashDiffusionConfig(**C2_KW), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(),
                       discriminator=disc).cuda()
    m.draws = Draws(g["draws"])
    out = m(c2_batch("cuda", B=2), step=0, device="cuda")
    out["loss"][0].backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    res = ({k: out[k].detach().clone() for k in ("teacher_output", "student_output")},
           [out["loss"][i].detach().clone() if torch.is_tensor(out["loss"][i]) else out["loss"][i] for i in (0, 1)], grads)
    del m, teacher, student, disc, out
    torch.cuda.empty_cache()
    return res


def _c2_twice_body():
    ops.deterministic.set(True)
    a = _c2_once()
    b = _c2_once()
    n = _identical(a, b, "c2_sd15_r128_n4")
    log(f"C2-shaped step (SD1.5 full width, r128, 4 teacher steps, B = 2): two deterministic runs bit-identical ({n} gradient tensors)")


@pytest.mark.parametrize("M,N1,N2", [(65536, 320, 128), (65536, 128, 320), (16384, 640, 128), (4096, 1280, 128), (32768, 1152, 64), (8192, 128, 128)])
@pytest.mark.parametrize("det", [False, True])
def test_wgrad_tn_against_fp32_matmul_is_tight_in_both_modes(M, N1, N2, det):
    """C += X^T Y on the LoRA shapes of the C2 / C4 steps against torch's fp32 product of the same bf16 operands: 1e-4 of the
    result's norm (an fp32 accumulation over M terms in ANY order is within ~sqrt(M) 6e-8 of it) -- the streaming kernel, its row
    splits and its atomics add nothing a bf16 step bar could hide.  Deterministic mode additionally repeats bit for bit."""
    g = torch.Generator(device="cuda").manual_seed(M + N1)
    X = torch.randn(M, N1, device="cuda", generator=g).to(BF)
    Y = (torch.randn(M, N2, device="cuda", generator=g) * 0.5).to(BF)
    ref = X.float().T.double() @ Y.float().double()
    with ops.deterministic(det):
        out = torch.zeros(N1, N2, device="cuda")
        ops.wgrad_tn(X, Y, out)
        out2 = torch.zeros(N1, N2, device="cuda")
        ops.wgrad_tn(X, Y, out2)
        torch.cuda.synchronize()
    err = float((out.double() - ref).norm() / ref.norm())
    log(f"wgrad_tn M={M} {N1}x{N2} det={int(det)}: rel err vs fp64 product {err:.2e}; repeat identical: {bool(torch.equal(out, out2))}")
    assert err < 1e-4, err
    if det:
        assert torch.equal(out, out2)


@pytest.mark.parametrize("det", [False, True])
def test_grouped_wgrad_against_fp32_matmul_is_tight_in_both_modes(det):
    """the (dB, dA) pair of a LoRA linear and the six products of a fused q/k/v as ONE launch (fdmi_wgrad_tn_group), M = 65536"""
    g = torch.Generator(device="cuda").manual_seed(7)
    M = 65536
    x = torch.randn(M, 320, device="cuda", generator=g).to(BF)
    probs = []
    for _ in range(3):
        dy = (torch.randn(M, 320, device="cuda", generator=g) * 0.3).to(BF)
        t = (torch.randn(M, 128, device="cuda", generator=g) * 0.3).to(BF)
        du = (torch.randn(M, 128, device="cuda", generator=g) * 0.3).to(BF)
        probs += [(dy, t), (du, x)]
    with ops.deterministic(det):
        outs = [torch.zeros(a.shape[1], b.shape[1], device="cuda") for a, b in probs]
        ops.wgrad_tn_group([(a, b, o) for (a, b), o in zip(probs, outs)])
        outs2 = [torch.zeros(a.shape[1], b.shape[1], device="cuda") for a, b in probs]
        ops.wgrad_tn_group([(a, b, o) for (a, b), o in zip(probs, outs2)])
        torch.cuda.synchronize()
    worst = 0.0
    for (a, b), o in zip(probs, outs):
        ref = a.float().T.double() @ b.float().double()
        worst = max(worst, float((o.double() - ref).norm() / ref.norm()))
    log(f"wgrad_tn_group (6 products, M = {M}) det={int(det)}: worst rel err vs fp64 product {worst:.2e}")
    assert worst < 1e-4, worst
    if det:
        assert all(torch.equal(o, o2) for o, o2 in zip(outs, outs2))


def test_deterministic_statistics_and_sums_agree_with_the_production_kernels():
    """GroupNorm forward / backward (statistics pass), the column sums and the scalar losses: same values to fp32 rounding"""
    g = torch.Generator(device="cuda").manual_seed(3)
    for (B, HW, C) in [(4, 4096, 320), (2, 1024, 640), (2, 64, 2560), (3, 256, 128)]:
        x = torch.randn(B, HW, C, device="cuda", generator=g).to(BF)
        gam, bet = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
        dy = torch.randn(B, HW, C, device="cuda", generator=g).to(BF)
        res = {}
        for det in (False, True):
            with ops.deterministic(det):
                y, stats = ops.groupnorm_fwd(x, gam, bet, 32, 1e-5, True)
                dx = ops.groupnorm_bwd(x, dy, gam, bet, stats, 32, 1e-5, True)
                y2, stats2 = ops.groupnorm_fwd(x, gam, bet, 32, 1e-5, True)
                torch.cuda.synchronize()
            res[det] = (y, stats, dx)
            if det:
                assert torch.equal(y, y2) and torch.equal(stats, stats2)
        e_s = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
        e_y = float((res[True][0].float() - res[False][0].float()).abs().max())
        e_dx = float((res[True][2].float() - res[False][2].float()).abs().max() / res[False][2].float().abs().max())
        log(f"groupnorm B={B} HW={HW} C={C}: statistics rel diff {e_s:.2e}, output max diff {e_y:.2e}, dx rel diff {e_dx:.2e}")
        assert e_s < 1e-5 and e_y <= 0.0625 and e_dx < 2e-2, (e_s, e_y, e_dx)    # (a bf16 rounding may flip where the statistics' last bit moved)


# ---- the transformer denoisers (C++ plans of dit_plan.h: per-sample column sums, grouped weight gradients, fp32 residual stream) ----
def _pixart_once(name):
    from flash_diffusion_amd.dit import MiTransformer2DModel
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import PIXART_STEP_CASES, PromptTableConditioner, build_pixart_step_inputs
    from tests.golden_util import load_case
    kw, step, _ = PIXART_STEP_CASES[name]
    g = load_case(name)
    cfg, t_o, s_o, head, batch = build_pixart_step_inputs()
    teacher = MiTransformer2DModel(**cfg)
    teacher.load_state_dict(t_o.state_dict())
    teacher = teacher.cuda()
    teacher.freeze()
    student = MiTransformer2DModel(**cfg)
    student.add_adapter(8)
    student.load_state_dict({k.replace(".base_layer.", "."): v for k, v in s_o.state_dict().items()})
    student = student.cuda()
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=PromptTableConditioner(),
                       discriminator=copy.deepcopy(head).cuda()).cuda()
    m.draws = Draws(g["draws"])
    out = m({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}, step=step, device="cuda")
    out["loss"][step].backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    return ({k: out[k].detach().clone() for k in ("teacher_output", "student_output")},
            [out["loss"][i].detach().clone() if torch.is_tensor(out["loss"][i]) else out["loss"][i] for i in (0, 1)], grads)


def _sd3_once(name):
    from flash_diffusion_amd.dit import MiSD3Transformer2DModel
    from flash_diffusion_amd.flash import Draws
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler
    from oracle.flash_sd3_ref import EmbeddingPipeline
    from oracle.golden_cases import SD3_MMDIT_CASES, build_sd3_mmdit_inputs
    from tests.golden_util import load_case
    kw, case, step, _ = SD3_MMDIT_CASES[name]
    g = load_case(name)
    cfg, t_o, s_o, head, pipe, batch = build_sd3_mmdit_inputs(case)
    teacher = MiSD3Transformer2DModel(**cfg)
    teacher.load_state_dict(t_o.state_dict())
    teacher = teacher.cuda()
    teacher.freeze()
    student = MiSD3Transformer2DModel(**cfg)
    student.add_adapter(8)
    student.load_state_dict({k.replace(".base_layer.", "."): v for k, v in s_o.state_dict().items()})
    student = student.cuda()
    pipe = EmbeddingPipeline(pipe.e[0].cuda(), pipe.e[2].cuda(), pipe.e[1].cuda(), pipe.e[3].cuda())
    m = FlashDiffusionSD3(FlashDiffusionSD3Config(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=FlowMatchEulerDiscreteScheduler(), discriminator=copy.deepcopy(head).cuda(),
                          pipeline=pipe)
    m.draws = Draws(g["draws"])
    out = m({"image": batch["image"].cuda(), "text": batch["text"]}, step=step)
    out["loss"][step].backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    return ({k: out[k].detach().clone() for k in ("teacher_output", "student_output")},
            [out["loss"][i].detach().clone() if torch.is_tensor(out["loss"][i]) else out["loss"][i] for i in (0, 1)], grads)


def _dit_twice_body(kind, name):
    ops.deterministic.set(True)
    once = _pixart_once if kind == "pixart" else _sd3_once
    a = once(name)
    b = once(name)
    n = _identical(a, b, name)
    log(f"{kind} step {name}: two deterministic runs bit-identical ({n} gradient tensors)")


@pytest.mark.parametrize("kind,name", [("pixart", "pixart_g_dmd_lsgan"), ("sd3", "sd3_mmdit_g_dmd_lsgan")])
def test_two_runs_of_a_transformer_step_are_bit_identical(kind, name):
    """the PixArt-alpha DiT (epsilon prediction, DPM-Solver++ teacher loop, masked T5 keys) and the SD3 MMDiT (flow matching) steps
    with DMD + lsgan GAN through the C++ plans, generator step with every LoRA gradient"""
    from tests.isolate import run_isolated
    run_isolated(__name__, "_dit_twice_body", (kind, name), timeout=600)
