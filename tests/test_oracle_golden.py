"""CPU: the oracle restatement, fed the recorded draws, reproduces the fixtures that the REAL
reference FlashDiffusion produced (tests/golden/*.npz)."""
import pytest
import torch

from oracle.flash_ref import Draws, FlashConfigRef, FlashDiffusionRef, TensorConditioner
from oracle.golden_cases import CASES, SCHEDS, build_models
from tests.golden_util import load_case, rel_err


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_golden(name):
    kw, sched, step, _seed = CASES[name]
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(),
                          discriminator=disc)
    m.draws = Draws(g["draws"])
    batch = {"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * g["z"].shape[0]}
    out = m(batch, step=step)
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output", "noisy_sample"):
        assert rel_err(out[k], g["out"][k]) < 1e-5, k
    for i in (0, 1):
        assert abs(float(out["loss"][i]) - g["loss"][i]) <= 1e-5 * max(1.0, abs(g["loss"][i]))
    out["loss"][step].backward()
    n = 0
    for pn, p in m.named_parameters():
        if p.grad is not None:
            assert rel_err(p.grad, g["grads"][pn]) < 1e-4, pn
            n += 1
    assert n == len(g["grads"])
    # reference invariants (tests/test_flash/test_flash_diffusion.py:146-153)
    if step == 0:
        assert g["loss"][0] > 0 and g["loss"][1] == 0.0
    else:
        assert g["loss"][1] > 0


def test_scheduler_trailing_timesteps():
    from oracle.sched_cpu import DPMSolverMultistepSchedulerRef
    s = DPMSolverMultistepSchedulerRef()
    s.set_timesteps(4)
    assert s.timesteps.tolist() == [999, 749, 499, 249]
    s.set_timesteps(1)
    assert s.timesteps.tolist() == [999]
    assert float(s.sigmas[-1]) == 0.0
    # last step of dpmsolver++ with final sigma 0 returns the x0 prediction
    order, cs, c0, c1 = s.step_coefficients(0, 0)
    assert order == 1 and abs(cs) < 1e-12 and abs(c0 - 1.0) < 1e-6
