"""Size-independent properties checked at BASELINE.json's full C2 sizes (SD1.5 UNet, LoRA r=128, 64x64 latents), where the
CPU oracle is too slow to be the checker: every layer of the UNet is per-sample, so (a) the LoRA gradient of a batch is the sum
of the gradients of its halves, (b) the teacher's batched [cond | uncond] call equals two separate calls, and (c) one full
Flash-Diffusion step leaves the teacher untouched, moves only LoRA parameters and produces finite losses."""
import pytest
import torch

from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu


def _build():
    from flash_diffusion_amd.workloads import SD15, build_flash
    torch.manual_seed(0)
    return build_flash(SD15, lora_rank=128, n_teacher_steps=2, device="cuda", seed=0)


def _cond(B, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return {"cond": {"crossattn": torch.randn(B, 77, 768, generator=g).cuda()}}


def test_student_lora_gradient_is_additive_over_the_batch():
    m = _build()
    st = m.student_denoiser
    g = torch.Generator(device="cpu").manual_seed(1)
    st._reflatten_lora(torch.device("cuda"))  # LoRA parameters / gradients become views of two flat buffers
    for p in st.lora_parameters()[1::2]:  # peft initialises B = 0: give A a gradient too
        p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).cuda())
    B = 4
    x = torch.randn(B, 4, 64, 64, generator=g).cuda()
    w = torch.randn(B, 4, 64, 64, generator=g).cuda()
    t = torch.tensor([999.0, 749.0, 499.0, 249.0]).cuda()
    c = _cond(B, 2)

    def grad(sl):
        st.lora_flat_grad().zero_()
        for p in st.lora_parameters():
            p.grad = None
        cc = {"cond": {"crossattn": c["cond"]["crossattn"][sl].contiguous()}}
        out = st(x[sl].contiguous(), t[sl].contiguous(), cc)
        (out * w[sl]).sum().backward()
        torch.cuda.synchronize()
        return st.lora_flat_grad().clone()

    g_all = grad(slice(0, 4))
    g_a, g_b = grad(slice(0, 2)), grad(slice(2, 4))
    g_again = grad(slice(0, 4))
    assert torch.isfinite(g_all).all() and float(g_all.abs().max()) > 0
    noise = max(rel_err(g_again, g_all), 1e-3)   # float-atomic summation order
    assert rel_err(g_a + g_b, g_all) < 3e-2 + 4 * noise, (rel_err(g_a + g_b, g_all), noise)
    assert rel_err(g_a, g_all) > 0.2             # (each half really is only a part)


def test_teacher_batched_cfg_equals_two_calls():
    m = _build()
    te = m.teacher_denoiser
    g = torch.Generator(device="cpu").manual_seed(3)
    B = 2
    x = torch.randn(B, 4, 64, 64, generator=g).cuda()
    t = torch.full((B,), 749.0).cuda()
    c, u = _cond(B, 4), _cond(B, 5)
    with torch.no_grad():
        e_c, e_u = te(x, t, c).clone(), te(x, t, u).clone()
        e_c2 = te(x, t, c).clone()
        cu = {"cond": {"crossattn": torch.cat([c["cond"]["crossattn"], u["cond"]["crossattn"]], 0)}}
        e = te(torch.cat([x, x], 0), torch.cat([t, t], 0), cu)
    noise = max(rel_err(e_c2, e_c), 1e-3)
    # (the 2B call takes other GEMM tiles / split-K plans than the B calls: equal up to bf16 summation order)
    assert rel_err(e[:B], e_c) < 4 * noise + 1.5e-2 and rel_err(e[B:], e_u) < 4 * noise + 1.5e-2
    assert rel_err(e_c, e_u) > 10 * noise        # the conditioning matters


def test_full_size_step_moves_only_lora():
    from flash_diffusion_amd.trainer import TrainingConfig, TrainingPipeline
    from flash_diffusion_amd.workloads import SD15, synthetic_batch
    m = _build()
    pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-4],
                                              trainable_params=[["student_denoiser"]]))
    pipe.configure_optimizers()
    teacher0 = {k: v.detach().clone() for k, v in m.teacher_denoiser.state_dict().items()}
    base0 = {k: v.detach().clone() for k, v in m.student_denoiser.state_dict().items() if "lora" not in k}
    lora0 = [p.detach().clone() for p in m.student_denoiser.lora_parameters()]
    for i in range(2):
        out = pipe.training_step(synthetic_batch(2, 64, SD15["cross_attention_dim"], seed=10 + i), i)
        assert torch.isfinite(out["loss"]).all() and float(out["loss"]) > 0
    pipe.finish()
    for k, v in m.teacher_denoiser.state_dict().items():
        assert torch.equal(v, teacher0[k]), k
    for k, v in m.student_denoiser.state_dict().items():
        if "lora" not in k:
            assert torch.equal(v, base0[k]), k
    moved = [float((p.detach() - q).abs().max()) for p, q in zip(m.student_denoiser.lora_parameters(), lora0)]
    assert max(moved[1::2]) > 0          # every LoRA B leaves zero, so the second step also moves the A's
    assert max(moved[0::2]) > 0
