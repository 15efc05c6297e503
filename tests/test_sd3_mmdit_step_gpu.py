"""SURVEY 8a row a18 end to end on the GPU: the flow-matching distillation step (FlashDiffusionSD3: Euler teacher loop with CFG,
x0 = x_t - sigma v, distillation + DMD + lsgan GAN terms, FD3:187-560) with the HIP MMDiT (MiSD3Transformer2DModel) in BOTH
denoiser slots and the HIP PatchGAN head, against fixtures made by the reference's REAL FlashDiffusionSD3 class over its REAL
DiffusersSD3Transformer2DWrapper (tests/golden/sd3_mmdit_*.npz, oracle/make_golden.py sd3_mmdit), every random draw injected.
Tolerances: the bf16 ones of tests/test_flash_gpu.py (outputs 4e-2, losses 6e-2, gradients: global cosine > 0.99)."""
import copy

import pytest
import torch

from oracle.golden_cases import SD3_MMDIT_CASES, build_sd3_mmdit_inputs
from tests.golden_util import load_case, parity_log, rel_err
from tests.isolate import run_isolated

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("name", list(SD3_MMDIT_CASES))
def test_sd3_step_over_the_hip_mmdit_matches_reference_golden(name):
    run_isolated(__name__, "_body", (name,))


def _body(name):
    from flash_diffusion_amd.dit import MiSD3Transformer2DModel
    from flash_diffusion_amd.flash import Draws
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config, FlowMatchEulerDiscreteScheduler
    from oracle.flash_sd3_ref import EmbeddingPipeline
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
    assert type(m.discriminator).__name__ == "MiDiscriminator"
    m.draws = Draws(g["draws"])
    out = m({"image": batch["image"].cuda(), "text": batch["text"]}, step=step)
    assert abs(out["start_timestep"] - g["start_timestep"]) < 1e-3
    errs = {k: rel_err(out[k], g["out"][k]) for k in ("teacher_output", "student_output", "noisy_sample")}
    lerr = []
    for i in (0, 1):
        ref, got = g["loss"][i], float(out["loss"][i])
        lerr.append(abs(got - ref) / abs(ref) if ref != 0 else abs(got))
    parity_log(f"{name}: " + " ".join(f"{k}={v:.3e}" for k, v in errs.items()) + f" loss_rel={lerr[0]:.3e},{lerr[1]:.3e}")
    assert errs["noisy_sample"] < 1e-6 and errs["teacher_output"] < 4e-2 and errs["student_output"] < 4e-2, errs
    assert lerr[0] < 6e-2 and lerr[1] < 6e-2, lerr
    out["loss"][step].backward()
    torch.cuda.synchronize()
    fa, fb = [], []
    for pn, p in m.named_parameters():
        if p.grad is None:
            assert pn not in g["grads"] or float(g["grads"][pn].abs().max()) == 0.0, pn
            continue
        if pn.startswith("student_denoiser.") and ".lora_" not in pn:
            continue
        assert pn in g["grads"], pn
        ref = g["grads"][pn]
        if float(ref.norm()) < 1e-12:
            continue
        fa.append(p.grad.detach().float().cpu().flatten())
        fb.append(ref.float().flatten())
    gc = _cos(torch.cat(fa), torch.cat(fb))
    parity_log(f"{name}: {len(fa)} gradient tensors, global cosine {gc:.4f}")
    assert len(fa) > 0 and gc > 0.99, (len(fa), gc)
