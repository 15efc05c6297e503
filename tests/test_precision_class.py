"""CPU: how far the reference's OWN training precision moves its results.  The reference trains with `precision="bf16-mixed"`
(examples/train_flash_sd.py:405); the fixtures under tests/golden/ are fp32 runs of the real class.  Here the pinned oracle
(bit-identical to the real class, tests/test_oracle_vs_reference.py) replays a fixture twice on the CPU -- in fp32 and under
`torch.autocast(bfloat16)`, PyTorch's implementation of that very mode -- and the deviation between the two is what "the
reference's bf16 noise" means for these tiny models: teacher output 1.4-2.3e-2, student output 5.5e-3, total loss 0.2-1.5 %.
The HIP path (bf16 storage, fp32 accumulation) lands at the same distance from the same fp32 fixtures on the GPU
(tests/test_flash_gpu.py: teacher 1.2-2.0e-2, student 5e-3, loss 0.04-0.75 %), i.e. inside the reference's own precision
class; north_star's 1e-3 on the loss is an fp32-vs-fp32 figure that neither bf16 execution of this algorithm meets.
The bounds asserted below are the ones tests/test_flash_gpu.py applies to the HIP path."""
import pytest
import torch

from oracle.flash_ref import Draws, FlashConfigRef, FlashDiffusionRef, TensorConditioner
from oracle.golden_cases import CASES, SCHEDS, build_models
from tests.golden_util import load_case, rel_err


def _run(name, autocast):
    kw, sched, step, _ = CASES[name]
    g = load_case(name)
    teacher, student, disc = build_models()
    m = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=SCHEDS[sched](), conditioner=TensorConditioner(), discriminator=disc)
    m.draws = Draws(g["draws"])
    batch = {"image": g["z"], "crossattn": g["crossattn"], "text": ["a"] * g["z"].shape[0]}
    with torch.no_grad():
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return m(batch, step=step), step
        return m(batch, step=step), step


@pytest.mark.parametrize("name", ["g_dmd_lsgan", "d_hinge"])
def test_reference_bf16_mixed_run_sits_where_the_hip_path_sits(name):
    (a, step), (b, _) = _run(name, False), _run(name, True)
    te = rel_err(b["teacher_output"].float(), a["teacher_output"])
    st = rel_err(b["student_output"].float(), a["student_output"])
    la, lb = float(a["loss"][0]), float(b["loss"][0])       # the generator's total loss (computed on both steps)
    lr = abs(lb - la) / abs(la)
    # the reference's own mixed-precision run is NOT within 1e-3 of its fp32 run ...
    assert te > 5e-3 and st > 2e-3 and lr > 1e-3, (te, st, lr)
    # ... and is inside the bounds the GPU parity tests apply to the HIP path (tests/test_flash_gpu.py)
    assert te < 4e-2 and st < 4e-2 and lr < 6e-2, (te, st, lr)
