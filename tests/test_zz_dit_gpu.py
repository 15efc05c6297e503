"""GPU parity of the transformer denoisers -- PixArt DiT and SD3 MMDiT (row a18): (1) the adaLN-single kernels of csrc/dit.hip and the modulate
path of the LayerNorm kernel against plain PyTorch fp32 references on the same bf16-rounded inputs (tolerances of
tests/test_kernels_gpu.py); (2) flash_diffusion_amd.dit.MiTransformer2DModel / MiSD3Transformer2DModel -- forward, LoRA forward and LoRA gradients --
against fixtures made by the reference's REAL wrapper classes (tests/golden/dit_*.npz, mmdit_*.npz; oracle/make_golden.py dit).
Tolerance for (2): bf16 activations through 2 blocks: outputs 2e-2 relative, LoRA gradients cosine > 0.999 and 6e-2 relative
(what the same composition gives on CPU with bf16 storage mimicked, tests/test_dit_host_logic.py: 0.8e-2 / 2.5e-2).

Every test body runs in its own interpreter (tests/isolate.py): a memory fault or a hang in one kernel costs that one test,
not the pytest process holding the other results.  Plain tests: a failure here is a failure of the suite."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle.golden_cases import DIT_CASES, MMDIT_CASES, build_dit, build_mmdit
from tests.golden_util import load_case, rel_err
from tests.isolate import run_isolated
from tests.test_kernels_gpu import b16, close, rnd

pytestmark = [pytest.mark.gpu]


def _ops():
    from flash_diffusion_amd import ops
    return ops


def _mod_table(B, n, Cc, seed):
    """[B, n, C] bf16 modulation table on the GPU and its fp32 host copy; operands are the strided views table[:, i]"""
    t = b16(rnd(B, n, Cc, seed=seed, scale=0.3))
    return t.cuda(), t.float()


@pytest.mark.parametrize("cfg", [(2, 64, 32), (3, 100, 288), (2, 256, 1152), (1, 40, 2048)])
def test_layernorm_modulate(cfg):
    run_isolated(__name__, "_body_test_layernorm_modulate", (cfg,))


def _body_test_layernorm_modulate(cfg):
    ops = _ops()
    B, T, Cc = cfg
    x = b16(rnd(B * T, Cc, seed=1) * 2 + 0.5)
    tab, tabf = _mod_table(B, 6, Cc, 2)
    xr = x.float().requires_grad_()
    sh, sc = tabf[:, 3].clone().requires_grad_(), tabf[:, 4].clone().requires_grad_()
    ref = F.layer_norm(xr, (Cc,), None, None, 1e-6).view(B, T, Cc) * (1 + sc[:, None]) + sh[:, None]
    y, stats = ops.layernorm_mod_fwd(x.cuda(), tab[:, 3], tab[:, 4], T, 1e-6, need_stats=True)
    close(f"lnmod_fwd{cfg}", y, ref.view(B * T, Cc))
    xf = x.float()
    mean = xf.mean(-1)
    rstd = torch.rsqrt(xf.var(-1, unbiased=False) + 1e-6)
    close(f"lnmod_stats{cfg}", stats, torch.stack([mean, rstd], 1), tol_el=1e-4, tol_fro=1e-4)
    dy = b16(rnd(B * T, Cc, seed=4))
    ref.backward(dy.float().view(B, T, Cc))
    dx = ops.layernorm_mod_bwd(x.cuda(), dy.cuda(), tab[:, 4], T, 1e-6)
    close(f"lnmod_bwd{cfg}", dx, xr.grad, tol_el=2 ** -6, tol_fro=6e-3)
    dscale, dshift = ops.batch_colsum(dy.cuda(), x.cuda(), stats, rows_per_batch=T)
    close(f"lnmod_dscale{cfg}", dscale, sc.grad, tol_el=1e-3, tol_fro=1e-3)
    close(f"lnmod_dshift{cfg}", dshift, sh.grad, tol_el=1e-3, tol_fro=1e-3)


@pytest.mark.parametrize("cfg", [(2, 64, 32), (3, 100, 288), (2, 256, 1152)])
def test_gate_residual_and_gelu(cfg):
    run_isolated(__name__, "_body_test_gate_residual_and_gelu", (cfg,))


def _body_test_gate_residual_and_gelu(cfg):
    ops = _ops()
    B, T, Cc = cfg
    x, res = b16(rnd(B * T, Cc, seed=1)), b16(rnd(B * T, Cc, seed=2))
    tab, tabf = _mod_table(B, 6, Cc, 3)
    g = tabf[:, 2]
    ref = res.float() + (g[:, None] * x.float().view(B, T, Cc)).view(B * T, Cc)
    close(f"gate_res{cfg}", ops.gate_residual(x.cuda(), tab[:, 2], res.cuda(), T), ref)
    close(f"gate_only{cfg}", ops.gate_residual(x.cuda(), tab[:, 2], None, T), ref - res.float())
    dy = b16(rnd(B * T, Cc, seed=4))
    dgate = ops.batch_colsum(dy.cuda(), x.cuda(), None, rows_per_batch=T, want_sum=False)[0]
    close(f"dgate{cfg}", dgate, (dy.float() * x.float()).view(B, T, Cc).sum(1), tol_el=1e-3, tol_fro=1e-3)
    xr = (x.float() * 2).requires_grad_()
    x2 = b16(xr.detach())
    yr = F.gelu(xr, approximate="tanh")
    close(f"gelu{cfg}", ops.gelu_tanh(x2.cuda()), yr)
    yr.backward(dy.float())
    close(f"gelu_bwd{cfg}", ops.gelu_tanh_bwd(x2.cuda(), dy.cuda()), xr.grad, tol_el=2 ** -6, tol_fro=6e-3)


def _cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def _product(cfg, ora, lora_r):
    from flash_diffusion_amd.dit import MiSD3Transformer2DModel, MiTransformer2DModel
    m = (MiSD3Transformer2DModel if "pos_embed_max_size" in cfg else MiTransformer2DModel)(**cfg)
    if lora_r:
        m.add_adapter(lora_r)
    m.load_state_dict({k.replace(".base_layer.", "."): v for k, v in ora.state_dict().items()})
    return m.cuda()


def _to_cuda(cond):
    return {"cond": {k: v.cuda() for k, v in cond["cond"].items()}}


def _build(name, **kw):
    return build_dit(name, **kw) if name in DIT_CASES else build_mmdit(name, **kw)


@pytest.mark.parametrize("name", list(DIT_CASES) + list(MMDIT_CASES))
def test_dit_frozen_forward_matches_reference_golden(name):
    run_isolated(__name__, "_body_test_dit_frozen_forward_matches_reference_golden", (name,))


def _body_test_dit_frozen_forward_matches_reference_golden(name):
    g = load_case(name)
    cfg, ora, (x, t, cond), _ = _build(name)
    m = _product(cfg, ora, 0)
    m.freeze()
    with torch.no_grad():
        out = m(x.cuda(), t.cuda(), _to_cuda(cond))
    assert out.shape == g["out"]["frozen"].shape and out.dtype == torch.float32
    assert rel_err(out, g["out"]["frozen"]) < 2e-2, rel_err(out, g["out"]["frozen"])
    # the same frozen model with a gradient flowing to its input (the GAN generator step through the teacher): the tanh-GELU
    # and the adaLN gate + residual leave the GEMM epilogues and run as their own taped passes -- same function
    out2 = m(x.cuda().requires_grad_(), t.cuda(), _to_cuda(cond))
    assert out2.requires_grad
    assert rel_err(out2, g["out"]["frozen"]) < 2e-2, rel_err(out2, g["out"]["frozen"])
    assert rel_err(out2, out) < 1e-2, rel_err(out2, out)
    assert m.plan_calls == 2, "the forwards must have run through the C++ plan (fdmi_dit_forward)"


@pytest.mark.parametrize("name", list(DIT_CASES) + list(MMDIT_CASES))
def test_dit_lora_step_matches_reference_golden(name):
    run_isolated(__name__, "_body_test_dit_lora_step_matches_reference_golden", (name,))


def _body_test_dit_lora_step_matches_reference_golden(name):
    g = load_case(name)
    cfg, ora, (x, t, cond), w = _build(name, lora_r=8)
    m = _product(cfg, ora, 8)
    out = m(x.cuda(), t.cuda(), _to_cuda(cond))
    assert rel_err(out, g["out"]["lora"]) < 2e-2, rel_err(out, g["out"]["lora"])
    (out * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    n = 0
    for k, p in m.named_parameters():
        if ".lora_" not in k:
            assert p.grad is None, k
            continue
        ref = g["grads"][k]
        assert p.grad is not None and p.grad.shape == ref.shape, k
        assert _cos(p.grad, ref) > 0.999 and rel_err(p.grad, ref) < 6e-2, (k, _cos(p.grad, ref), rel_err(p.grad, ref))
        n += 1
    assert n == len(g["grads"]) and n > 0
    assert m.plan_calls == 1, "the step must have run through the C++ plan (fdmi_dit_forward / fdmi_dit_backward)"


# ---- 256 x 192 variant of the 2-slot ring kernel (gemm4<BN=192>): the tile for N = 1152 / 1536 / 4608 / 6144 -----------------
T192 = (256 << 16) | 192
@pytest.mark.parametrize("shape", [(256, 192, 64), (512, 1152, 1152), (1024, 384, 4608), (768, 1536, 192), (2048, 576, 1536)])
def test_gemm4_bn192_row(shape):
    run_isolated(__name__, "_body_test_gemm4_bn192_row", (shape,))


def _body_test_gemm4_bn192_row(shape):
    ops = _ops()
    M, N, K = shape
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    res = b16(rnd(M, N, seed=5))
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=T192)
    close(f"gemm4_192_row{shape}", out, A.float() @ W.float().t() + bias + res.float())
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), out_f32=True, force_tile=T192)
    close(f"gemm4_192_f32{shape}", out, A.float() @ W.float().t() + bias, tol_el=1e-4, tol_fro=1e-4)
    if K >= 320:
        out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), force_tile=T192, splitk=3)
        close(f"gemm4_192_splitk{shape}", out, A.float() @ W.float().t() + bias)
        acc = torch.zeros(M, N, dtype=torch.float32, device="cuda")
        ops.gemm(A.cuda(), W.cuda(), out=acc, accum_atomic=True, splitk=2, force_tile=T192)
        close(f"gemm4_192_atomic{shape}", acc, A.float() @ W.float().t(), tol_el=1e-4, tol_fro=1e-4)


def test_gemm4_bn192_many_items_and_planner_knob():
    run_isolated(__name__, "_body_test_gemm4_bn192_many_items_and_planner_knob", ())


def _body_test_gemm4_bn192_many_items_and_planner_knob():
    """more (tile, split) items than CUs; and the planner itself picks the tile for a DiT shape"""
    import ctypes as C
    from flash_diffusion_amd import _lib
    ops = _ops()
    M, N, K = 256 * 150, 384, 128
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    out = ops.gemm(A.cuda(), W.cuda(), force_tile=T192)
    close("gemm4_192_many", out, A.float() @ W.float().t())
    assert torch.equal(out, ops.gemm(A.cuda(), W.cuda(), force_tile=T192))
    M, N, K = 8192, 1152, 1152
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    L = _lib.lib()
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.splitk, d.use_glds, d.alpha = M, N, K, K, K, 1, 1, 1.0
    o = [C.c_int32() for _ in range(4)]
    assert L.fdmi_gemm_plan(C.byref(d), *[C.byref(x) for x in o]) == 0 and (o[0].value, o[2].value) == (2, 192)
    close("gemm4_192_planned", ops.gemm(A.cuda(), W.cuda()), A.float() @ W.float().t())


def test_teacher_loop_single_call_matches_the_stepwise_loop():
    run_isolated(__name__, "_body_test_teacher_loop_single_call_matches_the_stepwise_loop", ())


def _body_test_teacher_loop_single_call_matches_the_stepwise_loop():
    """fdmi_teacher_loop (the frozen teacher's CFG loop as ONE C-ABI call: 2B-batched forwards with the context K/V cached,
    guidance folded into the x0 prediction, DPM-Solver++ update from a host coefficient table) against the step-by-step
    loop of flash.py (one forward + one fused scheduler step per iteration).  Tolerance: the tiny UNet's run-to-run
    GroupNorm-atomics noise, as in tests/test_unet_gpu.py::test_ctx_cache_reuse_matches_recompute."""
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from flash_diffusion_amd.unet import MiUNet2DConditionModel
    from flash_diffusion_amd.workloads import TINY
    torch.manual_seed(0)
    net = MiUNet2DConditionModel(**TINY).cuda()
    net.freeze()
    B, hw, L, D, K, si, g = 2, 16, 7, TINY["cross_attention_dim"], 4, 1, 6.0
    ctx2 = torch.randn(2 * B, L, D, device="cuda")
    x = torch.randn(B, 4, hw, hw, device="cuda")

    def stepwise():
        sch = DPMSolverMultistepScheduler()
        sch.set_timesteps(K)
        cur = x
        with torch.no_grad():
            for t in sch.timesteps[si:]:
                tt = torch.full((2 * B,), float(t), device="cuda")
                e = net(torch.cat([cur, cur]), tt, {"cond": {"crossattn": ctx2}})
                e_c, e_u = e.chunk(2)
                cur = sch.fused_cfg_step(e_c.contiguous(), e_u.contiguous(), g, t, cur)
        return cur

    want = stepwise()
    noise = max([rel_err(stepwise(), want) for _ in range(3)] + [5e-3])
    sch = DPMSolverMultistepScheduler()
    sch.set_timesteps(K)
    got = net.teacher_loop(x, [float(t) for t in sch.timesteps[si:]], ctx2, None, sch.loop_coefficients(si, g))
    assert got.shape == x.shape and rel_err(got, want) <= 3 * noise, (rel_err(got, want), noise)
    assert rel_err(got, x) > 10 * noise and rel_err(x, x.clone()) == 0   # the input is left untouched, the output moved


# ---- T2I-adapter residuals: fdmi_unet_set_down_residuals + the threading in FlashDiffusion -----------------
def test_unet_adapter_residuals_match_the_oracle():
    run_isolated(__name__, "_body_test_unet_adapter_residuals_match_the_oracle", ())


def _body_test_unet_adapter_residuals_match_the_oracle():
    from oracle.unet_cpu import TinyT2IAdapter, UNet2DConditionRef, seeded_init_, tiny_config
    from tests.unet_util import mi_from_oracle
    ora = seeded_init_(UNet2DConditionRef(tiny_config()), 1)
    ora.freeze()
    net = mi_from_oracle(ora)
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(2, 4, 32, 32, generator=g), torch.tensor([700.0, 40.0])
    ctx = torch.randn(2, 7, tiny_config().cross_attention_dim, generator=g)
    res = TinyT2IAdapter(tiny_config())(torch.randn(2, 1, 32, 32, generator=g))
    cond = {"cond": {"crossattn": ctx}}
    with torch.no_grad():
        want, plain = ora(x, t, cond, down_intrablock_additional_residuals=res), ora(x, t, cond)
        got = net(x.cuda(), t.cuda(), {"cond": {"crossattn": ctx.cuda()}},
                  down_intrablock_additional_residuals=[r.cuda() for r in res])
        got_plain = net(x.cuda(), t.cuda(), {"cond": {"crossattn": ctx.cuda()}})       # the residuals were consumed
        feat = net(x.cuda(), t.cuda(), {"cond": {"crossattn": ctx.cuda()}}, return_intermediate=True,
                   down_intrablock_additional_residuals=[r.cuda() for r in res])
        want_feat = ora(x, t, cond, down_intrablock_additional_residuals=res, return_intermediate=True)
    assert rel_err(want, plain) > 5e-2                                              # they matter
    assert rel_err(got, want) < 3e-2 and rel_err(got_plain, plain) < 3e-2 and rel_err(feat, want_feat) < 3e-2
    with pytest.raises(AssertionError):
        net(x.cuda(), t.cuda(), {"cond": {"crossattn": ctx.cuda()}}, down_intrablock_additional_residuals=[res[0].cuda()])


def test_step_with_adapter_matches_reference_golden():
    run_isolated(__name__, "_body_test_step_with_adapter_matches_reference_golden", ())


def _body_test_step_with_adapter_matches_reference_golden():
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import ADAPTER_CASES, LORA_RANK, build_models, make_edge
    from oracle.unet_cpu import TinyT2IAdapter, tiny_config
    from tests.unet_util import mi_from_oracle
    import copy
    (name, (kw, sched, step, _)), = ADAPTER_CASES.items()
    g = load_case(name)
    teacher_o, student_o, disc_o = build_models()
    teacher = mi_from_oracle(teacher_o)
    teacher.freeze()
    student = mi_from_oracle(student_o, lora_rank=LORA_RANK)
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(),
                       discriminator=copy.deepcopy(disc_o).cuda(), adapter=TinyT2IAdapter(tiny_config()).cuda()).cuda()
    m.draws = Draws(g["draws"])
    B = g["z"].shape[0]
    out = m({"image": g["z"].cuda(), "crossattn": g["crossattn"].cuda(), "text": ["a"] * B, "edge": make_edge().cuda()},
            step=step, device="cuda")
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output"):
        assert rel_err(out[k], g["out"][k]) < 4e-2, (k, rel_err(out[k], g["out"][k]))
    assert abs(float(out["loss"][0]) - g["loss"][0]) < 6e-2 * abs(g["loss"][0])
    out["loss"][step].backward()
    torch.cuda.synchronize()
    fa, fb = [], []
    for pn, p in m.named_parameters():
        if ".lora_" in pn and p.grad is not None:
            ref = [v for k, v in g["grads"].items() if k.replace(".base_layer.", ".") == pn][0]
            fa.append(p.grad.detach().float().cpu().flatten())
            fb.append(ref.float().flatten())
    gc = _cos(torch.cat(fa), torch.cat(fb)) if fa else float("nan")
    print(f"lpips step: {len(fa)} LoRA grad tensors, global cosine {gc:.4f}, norm ratio "
          f"{float(torch.cat(fa).norm() / torch.cat(fb).norm()) if fa else float('nan'):.3f}", flush=True)
    assert len(fa) > 0 and gc > 0.99, (len(fa), gc)


def test_cfg_halves_prefix_dedupe_matches_the_plain_call():
    run_isolated(__name__, "_body_test_cfg_halves_prefix_dedupe_matches_the_plain_call", ())


def _body_test_cfg_halves_prefix_dedupe_matches_the_plain_call():
    """FDMI_UNET_CFG_HALVES: on a [x | x] batch with [cond | uncond] contexts the layers before the first cross-attention are
    computed once and duplicated -- same result as the plain 2B call up to the tiny UNet's run-to-run GroupNorm-atomics noise;
    also together with the context K/V cache"""
    from flash_diffusion_amd.unet import MiUNet2DConditionModel
    from flash_diffusion_amd.workloads import TINY
    torch.manual_seed(0)
    net = MiUNet2DConditionModel(**TINY).cuda()
    net.freeze()
    B, hw, L, D = 2, 16, 7, TINY["cross_attention_dim"]
    x = torch.randn(B, 4, hw, hw, device="cuda")
    xx, tt = torch.cat([x, x]), torch.full((2 * B,), 600.0, device="cuda")
    ctx = {"cond": {"crossattn": torch.randn(2 * B, L, D, device="cuda")}}
    with torch.no_grad():
        ref = net(xx, tt, ctx).clone()
        noise = max([rel_err(net(xx, tt, ctx), ref) for _ in range(4)] + [5e-3])
        a = net(xx, tt, ctx, cfg_halves=True).clone()
        b = net(xx, tt, ctx, cfg_halves=True, ctx_cache="fill").clone()
        c = net(xx, tt, ctx, cfg_halves=True, ctx_cache="reuse").clone()
    assert rel_err(ref[:B], ref[B:]) > 10 * noise                 # the halves do differ (different contexts)
    for got in (a, b, c):
        assert rel_err(got, ref) <= 3 * noise, (rel_err(got, ref), noise)


def test_rccl_allreduce_entry_points_world_1():
    run_isolated(__name__, "_body_test_rccl_allreduce_entry_points_world_1", ())


def _body_test_rccl_allreduce_entry_points_world_1():
    """fdmi_comm_unique_id / fdmi_allreduce_init / fdmi_allreduce / fdmi_allreduce_destroy on a single-rank communicator:
    the in-place sum over one rank is the identity (f32 and bf16); the multi-rank path is the driver's 8-GPU run"""
    import ctypes as C
    from flash_diffusion_amd import _lib
    from flash_diffusion_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    uid = C.create_string_buffer(128)
    check(L.fdmi_comm_unique_id(uid))
    check(L.fdmi_allreduce_init(0, 1, uid))
    try:
        assert L.fdmi_allreduce_world() == 1
        for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
            x = torch.randn(100003, device="cuda").to(dt)
            y = x.clone()
            check(L.fdmi_allreduce(ptr(y), y.numel(), code, stream_ptr()))
            torch.cuda.synchronize()
            assert torch.equal(x, y)
    finally:
        check(L.fdmi_allreduce_destroy())
    assert L.fdmi_allreduce_world() == 0


# ---- GroupNorm statistics in the producing GEMM's epilogue (fdmi_gemm_gn; the plan's default since round 2) ------------------
# (B, H = W, Cin, Cout, kind): conv3x3 / 1x1-as-row-GEMM problems.  The planner (fdmi_gemm_plan, checked on CPU in
# tests/test_plan_dry.py) sends the first three to the 256 x 320 kernel (M = 65536 / 32768 rows), the next ones to the
# 256 x 160 ring kernel (5 fragments per wave: pairs + the 4-column tail path; group widths 10 and 15 straddle the 8-column
# chunks) and to the 256 x 128 one (group width 8 and 20)
GN_EPI = [(16, 64, 64, 320, "conv"), (16, 64, 320, 320, "row"), (8, 64, 128, 640, "conv"),
          (2, 32, 64, 320, "conv"), (4, 16, 320, 320, "row"), (2, 16, 64, 480, "conv"), (2, 16, 192, 480, "row"),
          (1, 32, 128, 640, "conv"), (2, 32, 128, 256, "row")]


@pytest.mark.parametrize("cfg", GN_EPI)
def test_gemm_epilogue_groupnorm_statistics(cfg):
    run_isolated(__name__, "_body_test_gemm_epilogue_groupnorm_statistics", (cfg,))


def _body_test_gemm_epilogue_groupnorm_statistics(cfg):
    """the GN instantiation of the 256-row kernels: (a) its output is bit-identical to the plain launch, (b) the accumulated
    (sum, sum of squares) equal the sums over the stored bf16 tensor, (c) groupnorm_apply on them equals groupnorm_fwd"""
    ops = _ops()
    B, H, Ci, Co, kind = cfg
    G, HW = 32, H * H
    x = b16(rnd(B, H, H, Ci, seed=1)).cuda()
    bias = rnd(Co, seed=3).cuda()
    res = b16(rnd(B * HW, Co, seed=4)).cuda()
    rowvec = b16(rnd(B, Co, seed=5)).cuda()
    if kind == "conv":
        w = ops.pack_conv_weight(b16(rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5)).float()).cuda()
        conv = dict(Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, KH=3, KW=3, stride=1, pad=1)
        kw = dict(M=B * HW, conv=conv, bias=bias, rowvec=rowvec, rows_per_batch=HW, residual=res)
        assert ops.gemm_gn_ok(B * HW, Co, 9 * Ci, HW, G, conv=conv), "test problem must be eligible"
        A = x
    else:
        w = b16(rnd(Co, Ci, seed=2, scale=Ci ** -0.5)).cuda()
        kw = dict(bias=bias, residual=res)
        assert ops.gemm_gn_ok(B * HW, Co, Ci, HW, G), "test problem must be eligible"
        A = x.view(B * HW, Ci)
    plain = ops.gemm(A, w, **kw)
    stats = torch.zeros(B, G, 2, dtype=torch.float32, device="cuda")
    y = ops.gemm(A, w, gn=(stats, HW), **kw)
    torch.cuda.synchronize()
    assert torch.equal(y, plain), "the GN instantiation must store exactly what the plain kernel stores"
    yf = y.float().view(B, HW, G, Co // G)
    ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], -1)
    close(f"gn_epi_sum{cfg}", stats[..., 0], ref[..., 0], tol_el=1e-3, tol_fro=1e-3)      # sums (cancelling terms) ...
    close(f"gn_epi_sumsq{cfg}", stats[..., 1], ref[..., 1], tol_el=2e-4, tol_fro=1e-4)   # ... and sums of squares, separately
    gamma, beta = (rnd(Co, seed=6) * 0.2 + 1).cuda(), rnd(Co, seed=7).cuda()
    want, st2 = ops.groupnorm_fwd(y.view(B, HW, Co), gamma, beta, G, 1e-5, 1)
    close(f"gn_epi_reduce{cfg}", st2[..., 1], ref[..., 1], tol_el=2e-4, tol_fro=1e-4)   # (the reduce kernel against the same reference)
    got = ops.groupnorm_apply(y.view(B, HW, Co), gamma, beta, stats, 1e-5, 1)
    close(f"gn_epi_apply{cfg}", got, want.float(), tol_el=2 ** -6, tol_fro=2e-3)


def test_gemm_gn_refuses_ineligible_problems():
    run_isolated(__name__, "_body_test_gemm_gn_refuses_ineligible_problems", ())


def _body_test_gemm_gn_refuses_ineligible_problems():
    ops = _ops()
    A, w = b16(rnd(128, 64, seed=1)).cuda(), b16(rnd(64, 64, seed=2)).cuda()     # M < 256: no 256-row kernel
    stats = torch.zeros(1, 32, 2, device="cuda")
    with pytest.raises(RuntimeError, match="gn_stats"):
        ops.gemm(A, w, gn=(stats, 128))


def test_unet_forward_with_epilogue_groupnorm_statistics():
    run_isolated(__name__, "_body_test_unet_forward_with_epilogue_groupnorm_statistics", (), timeout=900)


def _body_test_unet_forward_with_epilogue_groupnorm_statistics():
    """the full-width SD1.5 plan (B = 8, 64x64 latents: 15 of the 61 GroupNorms take their sums from the producing conv /
    linear -- pinned on CPU by the plan's workspace-query walk; smaller batches leave the 256-row kernels to split-K): the
    forward and the input gradient agree with the reduce-kernel path (A/B switch 14 = 1) up to bf16 rounding noise"""
    import ctypes as C
    from flash_diffusion_amd import _lib
    from flash_diffusion_amd.unet import MiUNet2DConditionModel
    from flash_diffusion_amd.workloads import SD15
    L = _lib.lib()
    torch.manual_seed(0)
    net = MiUNet2DConditionModel(**SD15).cuda()
    net.freeze()
    g = torch.Generator(device="cpu").manual_seed(1)
    B = 8
    x = torch.randn(B, 4, 64, 64, generator=g).cuda()
    t = torch.linspace(951.0, 51.0, B).cuda()
    ctx = {"cond": {"crossattn": torch.randn(B, 77, 768, generator=g).cuda()}}
    w = torch.randn(B, 4, 64, 64, generator=g).cuda()

    def run():
        xr = x.clone().requires_grad_()
        out = net(xr, t, ctx)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        tot = C.c_int32()
        return out.detach().clone(), xr.grad.clone(), L.fdmi_unet_last_gn_epilogue(net._plan().handle, C.byref(tot)), tot.value

    L.fdmi_tune_set(14, 1)                          # reference: every GroupNorm runs its own reduction pass
    try:
        ref, gref, n0, tot0 = run()
        again, gagain, _, _ = run()
    finally:
        L.fdmi_tune_set(14, 0)
    noise = max(rel_err(again, ref), 2e-3)          # float-atomic GroupNorm sums differ from run to run
    gnoise = max(rel_err(gagain, gref), 4e-3)
    got, ggot, n1, tot1 = run()
    assert n0 == 0 and tot0 == tot1 == 61 and n1 >= 12, (n0, n1, tot1)
    assert torch.isfinite(got).all() and rel_err(got, ref) <= 4 * noise, (rel_err(got, ref), noise)
    assert rel_err(ggot, gref) <= 4 * gnoise, (rel_err(ggot, gref), gnoise)


# ---- VAE in the loop + LPIPS distillation term (FD:128-133, 383-397; SURVEY 8f row 3: plumbing, the two networks are torch modules) ----
def test_step_with_vae_and_lpips_matches_reference_golden():
    run_isolated(__name__, "_body_test_step_with_vae_and_lpips_matches_reference_golden", ())


def _body_test_step_with_vae_and_lpips_matches_reference_golden():
    """the HIP student / teacher UNets inside a step whose batch is pixels (encoded by the VAE) and whose distillation term is
    LPIPS on the decoded centre crops: outputs, loss and LoRA gradient against the fixture of the real reference (bf16
    tolerances of tests/test_flash_gpu.py); the VAE / LPIPS stand-ins are plain torch modules on the GPU"""
    from flash_diffusion_amd.flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import LORA_RANK, LPIPS_CASES, build_models
    from oracle.unet_cpu import TinyLPIPS, TinyVAE
    from tests.unet_util import mi_from_oracle
    import copy
    (name, (kw, sched, step, _)), = LPIPS_CASES.items()
    g = load_case(name)
    teacher_o, student_o, disc_o = build_models()
    teacher = mi_from_oracle(teacher_o)
    teacher.freeze()
    student = mi_from_oracle(student_o, lora_rank=LORA_RANK)
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=student, teacher_denoiser=teacher,
                       teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(),
                       discriminator=copy.deepcopy(disc_o).cuda(), vae=TinyVAE().cuda(), lpips_model=TinyLPIPS().cuda()).cuda()
    m.draws = Draws(g["draws"])
    B = g["z"].shape[0]
    out = m({"image": g["z"].cuda(), "crossattn": g["crossattn"].cuda(), "text": ["a"] * B}, step=step, device="cuda")
    assert out["start_timestep"] == g["start_timestep"]
    for k in ("teacher_output", "student_output"):
        assert rel_err(out[k], g["out"][k]) < 4e-2, (k, rel_err(out[k], g["out"][k]))
    assert abs(float(out["loss"][0]) - g["loss"][0]) < 6e-2 * abs(g["loss"][0])
    out["loss"][step].backward()
    torch.cuda.synchronize()
    fa, fb = [], []
    for pn, p in m.named_parameters():
        if ".lora_" in pn and p.grad is not None:
            ref = [v for k, v in g["grads"].items() if k.replace(".base_layer.", ".") == pn][0]
            fa.append(p.grad.detach().float().cpu().flatten())
            fb.append(ref.float().flatten())
    gc = _cos(torch.cat(fa), torch.cat(fb)) if fa else float("nan")
    print(f"lpips step: {len(fa)} LoRA grad tensors, global cosine {gc:.4f}, norm ratio "
          f"{float(torch.cat(fa).norm() / torch.cat(fb).norm()) if fa else float('nan'):.3f}", flush=True)
    # bf16 gate.  The perceptual term's gradient is small next to the DMD term's, and the DMD direction is a difference of two
    # nearly equal bf16 denoiser outputs (FD:474-478) -- rounding noise dominates more of the sum than in the l2 fixtures
    # (measured 0.9497 ... 0.961 across boxes and runs: the GroupNorm statistics are fp32 atomics, their order moves the bf16
    # roundings the difference amplifies).  The same step in fp32 validation mode is held to cosine >= 0.9999 per tensor and
    # 1e-3 on every loss term (measured: 3.8e-5 / 1.5e-6): tests/test_fp32_gate_gpu.py::test_step_fixture_at_1e3[g_lpips_dmd_lsgan].
    assert len(fa) > 0 and gc > 0.93, (len(fa), gc)


# ---- GroupNorm reduction / apply passes with four rows in flight per thread ------------------------------------------------------
# row counts around the unroll boundaries: fewer rows per thread than one unrolled trip, exact multiples, ragged tails, C > 2048
GN_UNR = [(2, 64, 32, 32), (2, 256, 320, 32), (1, 100, 960, 32), (2, 16, 2560, 32), (3, 64, 128, 4), (2, 1024, 640, 32),
          (1, 4099, 320, 32), (2, 37, 1280, 32)]


@pytest.mark.parametrize("cfg", GN_UNR)
def test_groupnorm_unrolled_reduction(cfg):
    run_isolated(__name__, "_body_test_groupnorm_unrolled_reduction", (cfg,))


def _body_test_groupnorm_unrolled_reduction(cfg):
    """forward statistics, forward output and input gradient against torch at row counts around the unroll boundaries"""
    ops = _ops()
    B, HW, Cc, G = cfg
    x = b16(rnd(B, HW, Cc, seed=1) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(Cc, seed=2), 0.1 * rnd(Cc, seed=3)
    dy = b16(rnd(B, HW, Cc, seed=4))
    xf = x.float().view(B, HW, G, Cc // G)
    sums = torch.stack([xf.sum((1, 3)), (xf * xf).sum((1, 3))], -1)
    for silu in (0, 1):
        xr = x.float().permute(0, 2, 1).requires_grad_()
        ref = F.group_norm(xr, G, gamma, beta, 1e-5)
        if silu:
            ref = F.silu(ref)
        ref.backward(dy.float().permute(0, 2, 1))
        y1, st1 = ops.groupnorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), G, 1e-5, silu)
        dx1 = ops.groupnorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), st1, G, 1e-5, silu)
        torch.cuda.synchronize()
        close(f"gn_unr_stats{cfg}_{silu}", st1[..., 1], sums[..., 1], tol_el=1e-4, tol_fro=1e-5)
        close(f"gn_unr_fwd{cfg}_{silu}", y1, ref.permute(0, 2, 1))
        close(f"gn_unr_bwd{cfg}_{silu}", dx1, xr.grad.permute(0, 2, 1), tol_el=2 ** -6, tol_fro=6e-3)


# ---- TN weight-gradient kernel (csrc/wgrad.hip): LoRA gradients without transposed operand copies (in-plan use: the LoRA-gradient
# parity tests of tests/test_unet_gpu.py / test_flash_gpu.py against the oracle) ------------------------------------------------------
@pytest.mark.parametrize("shape", [(4096, 320, 128), (65536, 128, 320), (1232, 640, 128), (1232, 128, 768), (16384, 1280, 128),
                                   (130, 72, 40), (64, 64, 128), (100000, 128, 1280), (32768, 1152, 64), (8192, 64, 4608), (333, 200, 136)])
def test_wgrad_tn(shape):
    run_isolated(__name__, "_body_test_wgrad_tn", (shape,))


def _body_test_wgrad_tn(shape):
    """C += X^T Y on row-major bf16 operands (fp32 atomics), against torch in fp64 on the same bf16 values; accumulation into a
    non-zero C and a strided view of X (the gradient of a fused buffer)"""
    ops = _ops()
    M, N1, N2 = shape
    X, Y = b16(rnd(M, N1 + 8, seed=1)), b16(rnd(M, N2, seed=2))
    Xv = X[:, 8:]                                         # leading dimension N1 + 8, 16-byte aligned start
    C0 = rnd(N1, N2, seed=3)
    ref = C0.double() + Xv.double().t() @ Y.double()
    out = C0.clone().cuda()
    ops.wgrad_tn(X.cuda()[:, 8:], Y.cuda(), out)
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale + 1e-4, (err, scale)       # fp32 accumulation of exact bf16 products, split / atomic order


# grouped launches (fdmi_wgrad_tn_group): the (dB, dA) pair of one LoRA linear, the three pairs of a fused q/k/v projection (operands are
# column slices of shared [M, 3C] / [M, 3r] buffers), pairs with different widths, a tiny-M pair, groups whose narrow tiles differ or
# whose widths differ 4x (both fall back to single launches) and a group of one
WGRAD_GROUPS = {"pair_r128": [(16384, 640, 128), (16384, 128, 640)],
                "qkv_r128": "qkv:65536:320:128", "qkv_r64": "qkv:8192:1152:64",
                "pair_ff_r64": [(8192, 4608, 64), (8192, 64, 1152)],     # (36 against 9 column tiles: launched one by one)
                "pair_out_r64": [(8192, 1152, 64), (8192, 64, 1152)],
                "pair_ctx": [(1232, 320, 128), (1232, 128, 768)],
                "ragged": [(130, 72, 40), (333, 40, 200), (64, 64, 128)],
                "mixed_tiles": [(4096, 320, 128), (4096, 64, 320)],
                "single": [(4096, 128, 320)]}


@pytest.mark.parametrize("name", sorted(WGRAD_GROUPS))
def test_wgrad_tn_group(name):
    run_isolated(__name__, "_body_test_wgrad_tn_group", (name,))


def _body_test_wgrad_tn_group(name):
    """every product of a group against torch in fp64 on the same bf16 values (accumulating into non-zero C), and against the same
    products launched one by one (developer switch 47 = 1): the two differ only in the row split, i.e. in fp32 summation order"""
    ops = _ops()
    spec = WGRAD_GROUPS[name]
    probs = []
    if isinstance(spec, str):                     # the fused q/k/v layout of csrc/unet.hip::linear_qkv's backward
        _, M, C, r = spec.split(":")
        M, C, r = int(M), int(C), int(r)
        dy, t3, dt3, x = (b16(rnd(M, 3 * C, seed=1)).cuda(), b16(rnd(M, 3 * r, seed=2)).cuda(), b16(rnd(M, 3 * r, seed=3)).cuda(),
                          b16(rnd(M, C, seed=4)).cuda())
        for s in range(3):
            probs.append((dy[:, s * C:(s + 1) * C], t3[:, s * r:(s + 1) * r]))
            probs.append((dt3[:, s * r:(s + 1) * r], x))
    else:
        for i, (M, N1, N2) in enumerate(spec):
            probs.append((b16(rnd(M, N1 + 8, seed=10 + i)).cuda()[:, 8:], b16(rnd(M, N2, seed=20 + i)).cuda()))
    C0 = [rnd(X.shape[1], Y.shape[1], seed=30 + i) for i, (X, Y) in enumerate(probs)]
    outs = [c.clone().cuda() for c in C0]
    ops.wgrad_tn_group([(X, Y, o) for (X, Y), o in zip(probs, outs)])
    torch.cuda.synchronize()
    from flash_diffusion_amd import _lib
    _lib.lib().fdmi_tune_set(47, 1)
    try:
        singles = [c.clone().cuda() for c in C0]
        ops.wgrad_tn_group([(X, Y, o) for (X, Y), o in zip(probs, singles)])
        torch.cuda.synchronize()
    finally:
        _lib.lib().fdmi_tune_set(47, 0)
    for i, ((X, Y), c0, o, o1) in enumerate(zip(probs, C0, outs, singles)):
        ref = c0.double() + X.double().cpu().t() @ Y.double().cpu()
        scale = ref.abs().max().item()
        err = (o.double().cpu() - ref).abs().max().item()
        assert err <= 2e-5 * scale + 1e-4, (name, i, err, scale)
        err1 = (o.double().cpu() - o1.double().cpu()).abs().max().item()
        assert err1 <= 4e-5 * scale + 2e-4, (name, i, err1, scale)


# ---- the C++ plans against the op-by-op composition of the same module, at a size where the workspace matters ----------------
MID = {"pixart": dict(sample_size=64, num_layers=3, attention_head_dim=32, in_channels=4, out_channels=8, patch_size=2, attention_bias=True,
                      num_attention_heads=4, cross_attention_dim=128, activation_fn="gelu-approximate", num_embeds_ada_norm=1000,
                      norm_type="ada_norm_single", norm_elementwise_affine=False, norm_eps=1e-6, caption_channels=192,
                      projection_class_embeddings_input_dim=64, time_embed_dim=128, timesteps_embedding_num_channels=64,
                      use_concat_vector_conditioning=True, num_vector_conditionings=2),
       "sd3": dict(sample_size=64, patch_size=2, in_channels=16, num_layers=3, attention_head_dim=32, num_attention_heads=4,
                   joint_attention_dim=192, caption_projection_dim=128, pooled_projection_dim=64, out_channels=16, pos_embed_max_size=48)}


@pytest.mark.parametrize("kind,masked", [("pixart", False), ("pixart", True), ("sd3", False)])
def test_plan_matches_the_op_by_op_composition(kind, masked):
    run_isolated(__name__, "_body_test_plan_matches_the_op_by_op_composition", (kind, masked))


def _body_test_plan_matches_the_op_by_op_composition(kind, masked):
    """Same module, same weights (LoRA rank 64 with non-zero B on every target: the folded-operand GEMMs run), B = 8 on 64 x 64
    latents (8192 token rows): forward, input gradient and every LoRA gradient of the ONE-call plan (fdmi_dit_forward / _backward)
    against the Python-issued launches (FDMI_DIT_PLAN=0).  Both are bf16 paths over the same kernels; they differ in fusion
    (q / k / v as one GEMM, LoRA folded into K tiles) and in where roundings fall: 1e-2 on outputs, cosine > 0.999 on gradients.
    Also: a second and third frozen forward on the same slot reproduce the first bit for bit (the workspace is rewound per run)."""
    from flash_diffusion_amd.dit import MiSD3Transformer2DModel, MiTransformer2DModel
    torch.manual_seed(0)
    m = (MiTransformer2DModel if kind == "pixart" else MiSD3Transformer2DModel)(**MID[kind])
    for n, p in m.named_parameters():
        if n.endswith("scale_shift_table"):
            p.data.mul_(0.5)
    B, HW, L = 8, 64, 24
    g = torch.Generator().manual_seed(1)
    Cin = MID[kind]["in_channels"]
    x = torch.randn(B, Cin, HW, HW, generator=g).cuda()
    t = (torch.rand(B, generator=g) * 900 + 50).cuda()
    cond = {"crossattn": torch.randn(B, L, 192, generator=g).cuda(), "vector": torch.randn(B, 128 if kind == "pixart" else 64, generator=g).cuda()}
    if masked:
        lens = [24, 17, 8, 24, 1, 13, 20, 5]
        cond["attention_mask"] = torch.tensor([[1] * n + [0] * (L - n) for n in lens]).cuda()
    cond = {"cond": cond}
    w = torch.randn(B, Cin, HW, HW, generator=g).cuda()
    teacher = m.cuda()
    import copy
    student = copy.deepcopy(teacher)
    teacher.freeze()
    student.add_adapter(64, init_std_b=0.02, generator=torch.Generator().manual_seed(2))

    def run(plan):
        os.environ["FDMI_DIT_PLAN"] = "1" if plan else "0"
        out = {}
        with torch.no_grad():
            out["frozen"] = teacher(x, t, cond).clone()
            if plan:
                assert torch.equal(teacher(x, t, cond), out["frozen"]) and torch.equal(teacher(x, t, cond), out["frozen"])
        xg = x.clone().requires_grad_()
        o = teacher(xg, t, cond)
        (o * w).sum().backward()
        out["frozen_grad_out"], out["dx"] = o.detach().clone(), xg.grad.clone()
        for p in student.parameters():
            p.grad = None
        o = student(x, t, cond)
        (o * w).sum().backward()
        torch.cuda.synchronize()
        out["lora"] = o.detach().clone()
        out["grads"] = {k: p.grad.clone() for k, p in student.named_parameters() if ".lora_" in k}
        return out

    a = run(True)
    assert teacher.plan_calls == 4 and student.plan_calls == 1
    b = run(False)
    assert teacher.plan_calls == 4 and student.plan_calls == 1
    for k in ("frozen", "frozen_grad_out", "lora"):
        assert rel_err(a[k], b[k]) < 1e-2, (k, rel_err(a[k], b[k]))
    assert _cos(a["dx"], b["dx"]) > 0.999 and rel_err(a["dx"], b["dx"]) < 3e-2, (_cos(a["dx"], b["dx"]), rel_err(a["dx"], b["dx"]))
    assert set(a["grads"]) == set(b["grads"]) and len(a["grads"]) > 20
    worst = min((_cos(a["grads"][k], b["grads"][k]), k) for k in a["grads"])
    assert worst[0] > 0.995, worst
    tot = lambda d: torch.cat([v.flatten() for _, v in sorted(d.items())])
    assert rel_err(tot(a["grads"]), tot(b["grads"])) < 3e-2, rel_err(tot(a["grads"]), tot(b["grads"]))


@pytest.mark.parametrize("kind", ["pixart", "sd3"])
def test_dit_teacher_loop_single_call_matches_the_stepwise_loop(kind):
    run_isolated(__name__, "_body_test_dit_teacher_loop_single_call_matches_the_stepwise_loop", (kind,))


def _body_test_dit_teacher_loop_single_call_matches_the_stepwise_loop(kind):
    """fdmi_dit_teacher_loop (the frozen transformer teacher's guidance loop as ONE C-ABI call: 2B-batched plan forwards, the
    guidance and the scheduler update from a host coefficient table) against the step-by-step loops of flash.py (DPM-Solver++,
    PixArt, ragged key mask) and flash_sd3.py (flow-matching Euler, MMDiT): same forwards, the update in two fused launches
    instead of one."""
    import types
    from flash_diffusion_amd.dit import MiSD3Transformer2DModel, MiTransformer2DModel
    torch.manual_seed(0)
    net = (MiTransformer2DModel if kind == "pixart" else MiSD3Transformer2DModel)(**MID[kind]).cuda()
    net.freeze()
    B, HW, L, K, g = 2, 32, 16, 4, 4.5
    gen = torch.Generator().manual_seed(3)
    Cin = MID[kind]["in_channels"]
    x = torch.randn(B, Cin, HW, HW, generator=gen).cuda()
    ctx2 = torch.randn(2 * B, L, 192, generator=gen).cuda()
    vec2 = torch.randn(2 * B, 128 if kind == "pixart" else 64, generator=gen).cuda()
    if kind == "pixart":
        from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
        mask2 = torch.tensor([[1] * n + [0] * (L - n) for n in (16, 9, 3, 16)]).cuda()
        cond2 = {"cond": {"crossattn": ctx2, "vector": vec2, "attention_mask": mask2}}
        sch = DPMSolverMultistepScheduler()
        sch.set_timesteps(K)
        cur = x
        with torch.no_grad():
            for t in sch.timesteps[1:]:
                e_c, e_u = net(torch.cat([cur, cur]), torch.full((2 * B,), float(t), device="cuda"), cond2).chunk(2)
                cur = sch.fused_cfg_step(e_c.contiguous(), e_u.contiguous(), g, t, cur)
        calls = net.plan_calls
        sch = DPMSolverMultistepScheduler()
        sch.set_timesteps(K)
        got = net.teacher_loop(x, [float(t) for t in sch.timesteps[1:]], ctx2, vec2, sch.loop_coefficients(1, g), attention_mask=mask2)
        assert net.plan_calls == calls + K - 1
    else:
        from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlowMatchEulerDiscreteScheduler
        cond = {"cond": {"crossattn": ctx2[:B], "vector": vec2[:B]}}
        uncond = {"cond": {"crossattn": ctx2[B:], "vector": vec2[B:]}}
        me = types.SimpleNamespace(batch_cfg=True)
        outs = []
        for one_call in ("0", "1"):
            os.environ["FDMI_TEACHER_LOOP"] = one_call
            sch = FlowMatchEulerDiscreteScheduler()
            sch.set_timesteps(K)
            with torch.no_grad():
                outs.append(FlashDiffusionSD3._euler_cfg(me, net, sch, sch.timesteps, x, cond, uncond, g))
        cur, got = outs
        assert net.plan_calls == 2 * K
    # (a GEMM over <= 4096 rows may split K with fp32 atomics: run-to-run differences at the level of single bf16 roundings)
    assert got.shape == x.shape and rel_err(got, cur) < 5e-3, rel_err(got, cur)
    assert rel_err(got, x) > 1e-2     # the loop moved the latent (and left the caller's tensor alone)
