"""GPU parity tests of the individual HIP kernels (through the C-ABI) against plain PyTorch fp32
references computed on the host from the same bf16-rounded inputs.

Tolerance (stated): outputs are bf16 with fp32 accumulation, so |out - ref| <= 2^-7 * max|ref|
element-wise (one bf16 ulp at the tensor scale plus accumulation-order noise) and the relative
Frobenius error must be < 4e-3.  fp32 outputs: relative Frobenius error < 1e-4.
Mismatch diagnostics are appended to gpurun_out/kernel_diag.txt."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DIAG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_diag.txt")


def _ops():
    from flash_diffusion_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def b16(x):
    return x.to(torch.bfloat16)


def close(name, out, ref, tol_el=2 ** -7, tol_fro=4e-3):
    out = out.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert out.shape == ref.shape, (name, out.shape, ref.shape)
    err = (out - ref).abs()
    scale = ref.abs().max().item() + 1e-30
    fro = (out - ref).norm().item() / (ref.norm().item() + 1e-30)
    bad = err > tol_el * scale
    ok = (not bad.any().item()) and fro < tol_fro and torch.isfinite(out).all().item()
    if not ok:
        os.makedirs(os.path.dirname(DIAG), exist_ok=True)
        with open(DIAG, "a") as f:
            f.write(f"\n=== {name}: shape {tuple(out.shape)} fro {fro:.3e} maxerr {err.max().item():.3e} "
                    f"scale {scale:.3e} bad {bad.float().mean().item():.4f} nonfinite "
                    f"{(~torch.isfinite(out)).sum().item()}\n")
            o2, r2, b2 = out.reshape(-1, out.shape[-1]), ref.reshape(-1, ref.shape[-1]), bad.reshape(-1, out.shape[-1])
            rows = b2.any(dim=1).nonzero().flatten()
            cols = b2.any(dim=0).nonzero().flatten()
            f.write(f"bad rows ({len(rows)}): {rows[:40].tolist()}\nbad cols ({len(cols)}): {cols[:40].tolist()}\n")
            f.write(f"bad by row%16: {[int(b2[i::16].sum()) for i in range(16)]}\n")
            f.write(f"bad by col%16: {[int(b2[:, i::16].sum()) for i in range(min(16, b2.shape[1]))]}\n")
            for r in rows[:3].tolist():
                f.write(f"row {r} out {o2[r, :12].tolist()}\nrow {r} ref {r2[r, :12].tolist()}\n")
    assert ok, f"{name}: fro {fro:.3e}, max err {err.max().item():.3e} (scale {scale:.3e}), bad {bad.float().mean().item():.4f}"


# ------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(128, 128, 64), (200, 72, 136), (1232, 320, 768), (4096, 320, 320), (64, 640, 2560), (16, 1280, 320)]


@pytest.mark.parametrize("shape", GEMM_SHAPES)
@pytest.mark.parametrize("glds", [True, False])
@pytest.mark.parametrize("tile", [0, (128 << 16) | 128, (128 << 16) | 64, (64 << 16) | 128, (64 << 16) | 64])
def test_gemm_row(shape, glds, tile):
    ops = _ops()
    M, N, K = shape
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    out = ops.gemm(A.cuda(), W.cuda(), use_glds=glds, force_tile=tile)
    close(f"gemm_row{shape}_glds{glds}_tile{tile:x}", out, A.float() @ W.float().t())


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric W catches transposed / permuted fragment layouts."""
    ops = _ops()
    n = 128
    A = b16(torch.eye(n))
    W = b16((torch.arange(n)[:, None] * 0.5 + torch.arange(n)[None, :] * 0.01))
    out = ops.gemm(A.cuda(), W.cuda())
    close("gemm_identity", out, W.float().t(), tol_el=1e-6, tol_fro=1e-6)


@pytest.mark.parametrize("glds", [True, False])
def test_gemm_epilogues(glds):
    ops = _ops()
    M, N, K, rpb = 192, 328, 256, 64
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    rowvec = b16(rnd(M // rpb, N, seed=4))
    res = b16(rnd(M, N, seed=5))
    ref = 0.5 * (A.float() @ W.float().t()) + bias + rowvec.float().repeat_interleave(rpb, 0) + res.float()
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), rowvec=rowvec.cuda(), rows_per_batch=rpb,
                   residual=res.cuda(), alpha=0.5, use_glds=glds)
    close("gemm_epi_full", out, ref)
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=ops.ACT_SILU, use_glds=glds)
    close("gemm_epi_silu", out, F.silu(A.float() @ W.float().t() + bias))
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), out_f32=True, use_glds=glds)
    close("gemm_epi_f32", out, A.float() @ W.float().t() + bias, tol_el=1e-4, tol_fro=1e-4)


@pytest.mark.parametrize("M,N,K", [(192, 328, 256), (512, 1152, 1152), (512, 4608, 1152), (1024, 320, 1280), (512, 640, 640),
                                   (1344, 1536, 1536), (960, 64, 1152)])
def test_gemm_gate_and_gelu_tanh_epilogues(M, N, K):
    """The transformer denoisers' fused epilogues on every tile family (small tiles, 256 x {128,160} ring, 256 x 192 / 320):
    y = (x W^T + b) * gate[sample] + residual (adaLN gate, GemmArgs::rowvec_mul) and tanh-GELU (ACT_GELU_TANH); bf16 and the
    fp32 validation kernel."""
    ops = _ops()
    rpb = 64
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    gate = b16(rnd(M // rpb, N, seed=4))
    res = b16(rnd(M, N, seed=5))
    h = A.float() @ W.float().t() + bias
    ref_gate = h * gate.float().repeat_interleave(rpb, 0) + res.float()
    ref_gelu = F.gelu(h, approximate="tanh")
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), rowvec=gate.cuda(), rows_per_batch=rpb, rowvec_mul=True, residual=res.cuda())
    close(f"gemm_gate_res {M, N, K}", out, ref_gate)
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=ops.ACT_GELU_TANH)
    close(f"gemm_gelu_tanh {M, N, K}", out, ref_gelu)
    # a strided gate view (column block of a [B, 6, N] modulation table, as the denoisers pass it)
    table = b16(rnd(M // rpb, 6 * N, seed=6)).cuda()
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), rowvec=table[:, 2 * N:3 * N], rows_per_batch=rpb, rowvec_mul=True,
                   residual=res.cuda())
    close(f"gemm_gate_view {M, N, K}", out, h * table[:, 2 * N:3 * N].float().cpu().repeat_interleave(rpb, 0) + res.float())
    # the planner's own split-K choice (ops.gemm(splitk=0)): the slab finalize kernel applies the same epilogue
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), rowvec=gate.cuda(), rows_per_batch=rpb, rowvec_mul=True, residual=res.cuda(),
                   splitk=0)
    close(f"gemm_gate_res auto-splitk {M, N, K}", out, ref_gate)
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=ops.ACT_GELU_TANH, splitk=3)
    close(f"gemm_gelu_tanh splitk 3 {M, N, K}", out, ref_gelu)
    if M <= 512 and N <= 1152:
        A32, W32 = A.float().cuda(), W.float().cuda()
        out = ops.gemm(A32, W32, bias=bias.cuda(), rowvec=gate.float().cuda(), rows_per_batch=rpb, rowvec_mul=True,
                       residual=res.float().cuda())
        close(f"gemm32_gate_res {M, N, K}", out, ref_gate, tol_el=1e-5, tol_fro=1e-5)
        out = ops.gemm(A32, W32, bias=bias.cuda(), act=ops.ACT_GELU_TANH)
        close(f"gemm32_gelu_tanh {M, N, K}", out, ref_gelu, tol_el=1e-5, tol_fro=1e-5)


def test_gemm_geglu():
    ops = _ops()
    M, K, Fh = 200, 64, 96
    A = b16(rnd(M, K, seed=1))
    W = b16(rnd(2 * Fh, K, seed=2, scale=K ** -0.5))
    bias = rnd(2 * Fh, seed=3)
    res = b16(rnd(M, Fh, seed=5))
    perm = ops.geglu_perm(Fh)
    h = A.float() @ W.float().t() + bias
    ref = h[:, :Fh] * F.gelu(h[:, Fh:]) + res.float()
    pre = torch.empty(M, 2 * Fh, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm(A.cuda(), W[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(),
                   act=ops.ACT_GEGLU, preact=pre, residual=res.cuda())
    close("gemm_geglu", out, ref)
    close("gemm_geglu_preact", pre, h[:, perm])
    # backward of the gate on the interleaved layout
    dout = b16(rnd(M, Fh, seed=7))
    dpre = ops.geglu_bwd(pre, dout.cuda())
    hp = pre.float().cpu()
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(2 * Fh)
    hh = hp[:, inv]
    val, gate = hh[:, :Fh].requires_grad_(), hh[:, Fh:].requires_grad_()
    (val * F.gelu(gate)).backward(dout.float())
    ref_d = torch.cat([val.grad, gate.grad], 1)[:, perm]
    close("geglu_bwd", dpre, ref_d)


def test_gemm_splitk_and_atomic():
    ops = _ops()
    M, N, K = 96, 200, 4096
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    ref = A.float() @ W.float().t()
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), splitk=4)
    close("gemm_splitk", out, ref + bias)
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), splitk=0)
    close("gemm_splitk_auto", out, ref + bias)
    acc = torch.ones(M, N, dtype=torch.float32, device="cuda")
    ops.gemm(A.cuda(), W.cuda(), out=acc, accum_atomic=True, splitk=8, alpha=2.0)
    close("gemm_atomic", acc, 1.0 + 2.0 * ref, tol_el=1e-4, tol_fro=1e-4)


@pytest.mark.parametrize("N", [320, 328, 324])   # whole 320-wide tiles, a ragged last tile, N % 8 != 0
@pytest.mark.parametrize("sk", [2, 3, 5])
def test_gemm_splitk_finalize_epilogues(N, sk):
    """slab split-K + finalize with every epilogue term: equals the reference and -- the slabs are summed in a fixed order --
    is bit-identical between two runs"""
    ops = _ops()
    M, K, rpb = 512, 2560, 128
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    rowvec = b16(rnd(M // rpb, N, seed=4))
    res = b16(rnd(M, N, seed=5))
    pre = 0.5 * (A.float() @ W.float().t()) + bias + rowvec.float().repeat_interleave(rpb, 0)
    kw = dict(bias=bias.cuda(), rowvec=rowvec.cuda(), rows_per_batch=rpb, residual=res.cuda(), alpha=0.5, splitk=sk)
    out = ops.gemm(A.cuda(), W.cuda(), **kw)
    close(f"splitk_fin_{N}_{sk}", out, pre + res.float())
    assert torch.equal(out, ops.gemm(A.cuda(), W.cuda(), **kw))
    out = ops.gemm(A.cuda(), W.cuda(), act=ops.ACT_SILU, **kw)
    close(f"splitk_fin_silu_{N}_{sk}", out, F.silu(pre + res.float()))
    out = ops.gemm(A.cuda(), W.cuda(), out_f32=True, **kw)
    close(f"splitk_fin_f32_{N}_{sk}", out, pre + res.float(), tol_el=1e-4, tol_fro=1e-4)


CONVS = [
    # (B, H, W, Cin, Cout, k, stride, pad, ups)
    (2, 8, 8, 32, 64, 3, 1, 1, 0),
    (1, 16, 12, 64, 40, 3, 1, 1, 0),
    (2, 16, 16, 32, 32, 3, 2, 1, 0),
    (2, 8, 8, 64, 64, 3, 1, 1, 1),
    (2, 16, 16, 8, 32, 3, 1, 1, 0),
    (2, 8, 8, 128, 16, 4, 2, 1, 0),
    (2, 4, 4, 32, 4, 4, 1, 0, 0),
    (3, 8, 8, 320, 320, 3, 1, 1, 0),
    (2, 8, 8, 96, 64, 1, 1, 0, 0),
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("glds", [True, False])
def test_conv_fwd_and_dgrad(cfg, glds):
    ops = _ops()
    B, H, W, Ci, Co, k, s, p, ups = cfg
    x = b16(rnd(B, Ci, H, W, seed=1))
    w = b16(rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5))
    bias = rnd(Co, seed=3)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    xin.requires_grad_()
    ref = F.conv2d(xin, w.float(), bias, stride=s, padding=p)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = ops.conv2d_nhwc(xn, ops.pack_conv_weight(w.float()).cuda(), KH=k, KW=k, stride=s, pad=p, ups=ups,
                        bias=bias.cuda(), use_glds=glds)
    close(f"conv_fwd{cfg}_glds{glds}", y.permute(0, 3, 1, 2), ref)
    # dgrad: gather form of the transposed conv with the [I][KH][KW][O] operand
    if Co % 8 == 0:
        dy = b16(rnd(*ref.shape, seed=4))
        ref.backward(dy.float())
        dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
        dx = ops.conv2d_nhwc(dyn, ops.pack_conv_weight_dgrad(w.float()).cuda(), KH=k, KW=k, stride=s, pad=p,
                             dgrad=1, out_hw=(H << ups, W << ups), use_glds=glds)
        close(f"conv_dgrad{cfg}_glds{glds}", dx.permute(0, 3, 1, 2), xin.grad)
        if ups:
            dxl = ops.pool2x2_sum(dx.contiguous())
            g_low = xin.grad.reshape(B, Ci, H, 2, W, 2).sum((3, 5))
            close(f"pool2x2{cfg}", dxl.permute(0, 3, 1, 2), g_low)


GNS = [(2, 64, 32, 32), (2, 256, 320, 32), (1, 100, 960, 32), (2, 16, 2560, 32), (3, 64, 128, 4), (2, 1024, 640, 32)]


@pytest.mark.parametrize("cfg", GNS)
@pytest.mark.parametrize("silu", [0, 1])
def test_groupnorm(cfg, silu):
    ops = _ops()
    B, HW, Cc, G = cfg
    x = b16(rnd(B, HW, Cc, seed=1) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(Cc, seed=2), 0.1 * rnd(Cc, seed=3)
    xr = x.float().permute(0, 2, 1).requires_grad_()
    ref = F.group_norm(xr, G, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    y, stats = ops.groupnorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), G, 1e-5, silu)
    close(f"gn_fwd{cfg}_{silu}", y, ref.permute(0, 2, 1))
    dy = b16(rnd(B, HW, Cc, seed=4))
    ref.backward(dy.float().permute(0, 2, 1))
    dx = ops.groupnorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), stats, G, 1e-5, silu)
    close(f"gn_bwd{cfg}_{silu}", dx, xr.grad.permute(0, 2, 1), tol_el=2 ** -6, tol_fro=6e-3)


@pytest.mark.parametrize("cfg", [(200, 320), (64, 1280), (33, 64), (128, 1152)])
def test_layernorm(cfg):
    ops = _ops()
    rows, Cc = cfg
    x = b16(rnd(rows, Cc, seed=1) * 2 + 0.5)
    gamma, beta = 1 + 0.1 * rnd(Cc, seed=2), 0.1 * rnd(Cc, seed=3)
    xr = x.float().requires_grad_()
    ref = F.layer_norm(xr, (Cc,), gamma, beta, 1e-5)
    y = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    close(f"ln_fwd{cfg}", y, ref)
    dy = b16(rnd(rows, Cc, seed=4))
    ref.backward(dy.float())
    dx = ops.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), 1e-5)
    close(f"ln_bwd{cfg}", dx, xr.grad, tol_el=2 ** -6, tol_fro=6e-3)
    prev = b16(rnd(rows, Cc, seed=9))
    dx2 = prev.cuda().clone()
    ops.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), 1e-5, dx=dx2)
    close(f"ln_bwd_acc{cfg}", dx2, xr.grad + prev.float(), tol_el=2 ** -6, tol_fro=6e-3)


ATTN = [
    # (B, H, Sq, Skv, d)
    (2, 2, 64, 64, 16), (1, 2, 128, 128, 32), (2, 8, 256, 256, 40), (1, 3, 100, 100, 64), (2, 8, 256, 77, 40),
    (1, 8, 64, 77, 160), (1, 8, 256, 256, 80), (1, 2, 1024, 1024, 40), (1, 4, 200, 120, 72), (1, 2, 64, 64, 160),
    (1, 2, 130, 70, 8),
    # the software-pipelined forward (d in 33..64): one tile, one ragged tile, odd / even tile counts, ragged last tile
    (1, 2, 64, 64, 40), (1, 2, 96, 50, 64), (1, 4, 128, 192, 40), (1, 2, 70, 150, 64), (1, 2, 128, 320, 40), (1, 2, 200, 448, 56),
    # whole tiles at every (padded head dim, ones-row) instantiation of the forward: the exits of its tile loop (round 2: the
    # compiler placed accumulator copies right behind the last asynchronous MFMAs for head dims 56 and 64)
    (1, 2, 128, 128, 64), (1, 2, 256, 192, 64), (1, 2, 64, 1024, 64), (1, 2, 128, 128, 56), (2, 4, 128, 256, 48), (1, 2, 128, 256, 24),
    (1, 2, 128, 192, 96), (1, 2, 128, 192, 88), (1, 2, 64, 192, 128), (1, 2, 64, 128, 120), (1, 2, 64, 192, 160), (1, 2, 64, 128, 16),
    # the 32x32x16 forward's 80-wide instantiations (d = 72: running maximum in column 72 of the second K sub-tile, ones row 72 of V^T;
    # d = 80: neither): one tile, odd / even whole-tile counts, ragged tails, Sq off the 128-query block
    (1, 2, 64, 64, 72), (2, 3, 300, 448, 72), (1, 2, 128, 150, 72), (1, 16, 1024, 1024, 72), (1, 2, 64, 64, 80), (2, 3, 300, 448, 80),
    (1, 2, 128, 150, 80), (1, 4, 130, 384, 80),
]


def _attn_ref(q, k, v, H, scale):
    B, Sq, Cc = q.shape
    d = Cc // H
    qh = q.view(B, Sq, H, d).transpose(1, 2)
    kh = k.view(B, -1, H, d).transpose(1, 2)
    vh = v.view(B, -1, H, d).transpose(1, 2)
    p = ((qh @ kh.transpose(-1, -2)) * scale).softmax(-1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, Cc)


@pytest.mark.parametrize("cfg", ATTN)
def test_attention_fwd_bwd(cfg):
    ops = _ops()
    B, H, Sq, Skv, d = cfg
    scale = d ** -0.5
    q = b16(rnd(B, Sq, H * d, seed=1))
    k = b16(rnd(B, Skv, H * d, seed=2))
    v = b16(rnd(B, Skv, H * d, seed=3))
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, H, scale)
    o, lse = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, scale, need_lse=True)
    close(f"attn_fwd{cfg}", o, ref)
    if d < 64 or 64 < d <= 80:   # both forward families on every such shape: 32x32x16 (switch 26 = 2 lifts its few-keys rule), 16x16x32 (26 = 1)
        from flash_diffusion_amd import _lib
        for tag, val in (("32x32x16", 2), ("16x16x32", 1)):
            _lib.lib().fdmi_tune_set(26, val)
            try:
                o2, lse2 = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, scale, need_lse=True)
            finally:
                _lib.lib().fdmi_tune_set(26, 0)
            close(f"attn_fwd {tag} {cfg}", o2, ref)
            assert (lse2.float() - lse.float()).abs().max().item() < 2e-2
    do = b16(rnd(B, Sq, H * d, seed=4))
    ref.backward(do.float())
    dq, dk, dv = ops.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o, do.cuda(), lse, H, scale)
    close(f"attn_dq{cfg}", dq, qr.grad, tol_el=2 ** -5, tol_fro=1.2e-2)
    close(f"attn_dk{cfg}", dk, kr.grad, tol_el=2 ** -5, tol_fro=1.2e-2)
    close(f"attn_dv{cfg}", dv, vr.grad, tol_el=2 ** -5, tol_fro=1.2e-2)
    if d <= 80:   # both backward families on every such shape: 32x32x16 (switch 36 = 2 lifts its few-keys rule) and 16x16x32 (35 = 36 = 1)
        from flash_diffusion_amd import _lib
        L = _lib.lib()
        for tag, knobs in (("32x32x16", ((35, 0), (36, 2))), ("16x16x32", ((35, 1), (36, 1)))):
            try:
                for kk, vv in knobs:
                    L.fdmi_tune_set(kk, vv)
                dq, dk, dv = ops.attn_bwd(q.cuda(), k.cuda(), v.cuda(), o, do.cuda(), lse, H, scale)
            finally:
                L.fdmi_tune_set(35, 0)
                L.fdmi_tune_set(36, 0)
            close(f"attn_dq {tag} {cfg}", dq, qr.grad, tol_el=2 ** -5, tol_fro=1.2e-2)
            close(f"attn_dk {tag} {cfg}", dk, kr.grad, tol_el=2 ** -5, tol_fro=1.2e-2)
            close(f"attn_dv {tag} {cfg}", dv, vr.grad, tol_el=2 ** -5, tol_fro=1.2e-2)


@pytest.mark.parametrize("d,factor", [(40, 6.0), (40, 60.0), (64, 6.0), (64, 60.0), (80, 6.0), (72, 6.0), (72, 60.0), (80, 60.0), (56, 60.0),
                                      (48, 60.0)])
def test_attention_spike_forces_rescale(d, factor):
    """One key dominates one query late in the sequence: the online-softmax re-base branch must fire.  Factor 60 drives
    exp2(s - m) of the maximum-free common tile body (head dims <= 64) to +inf before the re-base: the tile is recomputed.
    At that factor the spike key's logit is O(100) for MANY queries and the kernels' bf16 rounding of the pre-scaled Q (2^-9
    relative = several tenths of a logit) legitimately moves near-ties, so only the two planted rows (one-hot by a wide margin)
    are held to the fp32 reference; everything else must agree between the two forward kernels, which round Q identically."""
    ops = _ops()
    from flash_diffusion_amd import _lib
    B, H, S = 1, 2, 448
    q, k, v = rnd(B, S, H * d, seed=1), rnd(B, S, H * d, seed=2), rnd(B, S, H * d, seed=3)
    k[0, 200, :d] = q[0, 17, :d] * factor
    k[0, 330, d:] = q[0, 140, d:] * factor
    q, k, v = b16(q), b16(k), b16(v)
    ref = _attn_ref(q.float(), k.float(), v.float(), H, d ** -0.5)
    o = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, d ** -0.5)
    if factor < 10:
        close(f"attn_spike d={d} x{factor}", o, ref)
        return
    assert torch.isfinite(o.float()).all()
    close(f"attn_spike rows d={d} x{factor}", torch.cat([o[0, 17, :d], o[0, 140, d:]]), torch.cat([ref[0, 17, :d], ref[0, 140, d:]]))
    L = _lib.lib()
    L.fdmi_tune_set(26, 1)
    try:
        o16 = ops.attn_fwd(q.cuda(), k.cuda(), v.cuda(), H, d ** -0.5)
    finally:
        L.fdmi_tune_set(26, 0)
    close(f"attn_spike 32 vs 16 d={d} x{factor}", o, o16.float().cpu(), tol_el=2 ** -5, tol_fro=1.2e-2)


def test_attention_fwd_variants_agree():
    """The 32x32x16 forward (head dims <= 64) against the 16x16x32 one it replaces (developer knob 26), same inputs, incl. a
    ragged key tail and Sq not a multiple of the 128-query block."""
    ops = _ops()
    from flash_diffusion_amd import _lib
    L = _lib.lib()
    for (B, H, Sq, Skv, d) in [(2, 4, 320, 1000, 40), (1, 2, 4096, 4096, 40), (1, 3, 200, 77, 64), (1, 2, 1024, 1024, 64), (1, 2, 130, 640, 32),
                               (1, 4, 1024, 1024, 72), (2, 2, 320, 120, 72), (1, 4, 1024, 1024, 80), (1, 2, 200, 333, 80)]:
        q, k, v = (b16(rnd(B, n, H * d, seed=s)).cuda() for n, s in ((Sq, 1), (Skv, 2), (Skv, 3)))
        o_new, lse_new = ops.attn_fwd(q, k, v, H, d ** -0.5, need_lse=True)
        L.fdmi_tune_set(26, 1)
        try:
            o_old, lse_old = ops.attn_fwd(q, k, v, H, d ** -0.5, need_lse=True)
        finally:
            L.fdmi_tune_set(26, 0)
        ref = _attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), H, d ** -0.5)
        close(f"attn32 {B, H, Sq, Skv, d}", o_new, ref)
        close(f"attn16 {B, H, Sq, Skv, d}", o_old, ref)
        assert (lse_new.float() - lse_old.float()).abs().max().item() < 2e-2


def test_layout_and_misc_kernels():
    ops = _ops()
    x = rnd(2, 4, 8, 8, seed=1)
    y = ops.nchw_to_nhwc(x.cuda(), 8)
    ref = torch.zeros(2, 8, 8, 8)
    ref[..., :4] = x.permute(0, 2, 3, 1)
    close("nchw_to_nhwc", y, b16(ref).float(), tol_el=1e-6, tol_fro=1e-6)
    back = ops.nhwc_to_nchw(y, 4)
    close("nhwc_to_nchw", back, b16(x).float(), tol_el=1e-6, tol_fro=1e-6)
    t = torch.tensor([999.0, 1.0, 500.0])
    emb = ops.timestep_embed(t.cuda(), 320)
    half = 160
    e = t[:, None] * torch.exp(-math.log(10000) * torch.arange(half) / half)[None]
    close("timestep_embed", emb, torch.cat([torch.cos(e), torch.sin(e)], -1), tol_el=2 ** -7, tol_fro=5e-3)
    w = rnd(100, 72, seed=3)
    wb, wtb = ops.cast_transpose(w.cuda())
    close("cast", wb, b16(w).float(), tol_el=1e-6, tol_fro=1e-6)
    close("cast_t", wtb, b16(w).float().t(), tol_el=1e-6, tol_fro=1e-6)
    xb = b16(rnd(130, 200, seed=4))
    close("transpose2d", ops.transpose2d(xb.cuda()), xb.float().t(), tol_el=1e-6, tol_fro=1e-6)
    z, n = rnd(3, 4, 8, 8, seed=5), rnd(3, 4, 8, 8, seed=6)
    sa, sb = torch.tensor([0.9, 0.5, 0.1]), torch.tensor([0.1, 0.7, 0.99])
    close("add_noise", ops.add_noise(z.cuda(), n.cuda(), sa.cuda(), sb.cuda()),
          sa.view(-1, 1, 1, 1) * z + sb.view(-1, 1, 1, 1) * n, tol_el=1e-6, tol_fro=1e-6)
    close("axpby", ops.axpby(z.cuda(), 2.0, n.cuda(), -0.5, z.cuda(), 0.25), 2.25 * z - 0.5 * n, tol_el=1e-6,
          tol_fro=1e-6)


def test_adamw_matches_torch():
    ops = _ops()
    p0, g = rnd(1000, seed=1), rnd(1000, seed=2)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0.01)
    pd, m, v = p0.cuda().clone(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in (1, 2, 3):
        p.grad = g.clone() * step
        opt.step()
        ops.adamw_(pd, (g * step).cuda(), m, v, 1e-3, step=step)
    close("adamw", pd, p.detach(), tol_el=1e-5, tol_fro=1e-5)


# ---- 256-row tile kernel (gemm3: 3-stage LDS-DMA ring, counted vmcnt) ---------------------------------
BIG = [(256 << 16) | 160, (256 << 16) | 128]


@pytest.mark.parametrize("shape", [(1232, 320, 768), (4096, 320, 320), (512, 640, 2560), (300, 1280, 128), (256, 160, 64),
                                   (2048, 200, 192)])
@pytest.mark.parametrize("tile", BIG)
def test_gemm3_row(shape, tile):
    ops = _ops()
    M, N, K = shape
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    res = b16(rnd(M, N, seed=5))
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=tile)
    close(f"gemm3_row{shape}_{tile:x}", out, A.float() @ W.float().t() + bias + res.float())
    if K >= 512:
        out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), force_tile=tile, splitk=3)
        close(f"gemm3_row_splitk{shape}_{tile:x}", out, A.float() @ W.float().t() + bias)


def test_gemm3_geglu_and_auto_plan():
    ops = _ops()
    M, K, Fh = 520, 128, 256
    A = b16(rnd(M, K, seed=1))
    W = b16(rnd(2 * Fh, K, seed=2, scale=K ** -0.5))
    bias = rnd(2 * Fh, seed=3)
    perm = ops.geglu_perm(Fh)
    h = A.float() @ W.float().t() + bias
    ref = h[:, :Fh] * F.gelu(h[:, Fh:])
    pre = torch.empty(M, 2 * Fh, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm(A.cuda(), W[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), act=ops.ACT_GEGLU,
                   preact=pre, force_tile=(256 << 16) | 128)
    close("gemm3_geglu", out, ref)
    close("gemm3_geglu_preact", pre, h[:, perm])
    out = ops.gemm(A.cuda(), W[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), act=ops.ACT_GEGLU)
    close("gemm_auto_geglu", out, ref)
    acc = torch.zeros(M, 2 * Fh, dtype=torch.float32, device="cuda")
    ops.gemm(A.cuda(), W.cuda(), out=acc, accum_atomic=True, splitk=2, force_tile=(256 << 16) | 128)
    close("gemm3_atomic", acc, A.float() @ W.float().t(), tol_el=1e-4, tol_fro=1e-4)


# ---- 256 x 320 tile kernel (gemm4: 2-slot ring, streamed W fragments) ---------------------------------
T4 = (256 << 16) | 320


@pytest.mark.parametrize("shape", [(256, 320, 64), (512, 640, 320), (1024, 320, 1280), (768, 1280, 192), (2048, 960, 2560)])
def test_gemm4_row(shape):
    ops = _ops()
    M, N, K = shape
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    res = b16(rnd(M, N, seed=5))
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=T4)
    close(f"gemm4_row{shape}", out, A.float() @ W.float().t() + bias + res.float())
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=ops.ACT_SILU, force_tile=T4)
    close(f"gemm4_row_silu{shape}", out, F.silu(A.float() @ W.float().t() + bias))
    if K >= 320:
        out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), force_tile=T4, splitk=3)
        close(f"gemm4_row_splitk{shape}", out, A.float() @ W.float().t() + bias)
        acc = torch.zeros(M, N, dtype=torch.float32, device="cuda")
        ops.gemm(A.cuda(), W.cuda(), out=acc, accum_atomic=True, splitk=2, force_tile=T4)
        close(f"gemm4_atomic{shape}", acc, A.float() @ W.float().t(), tol_el=1e-4, tol_fro=1e-4)


def test_gemm4_geglu():
    ops = _ops()
    M, K, Fh = 512, 192, 640
    A = b16(rnd(M, K, seed=1))
    W = b16(rnd(2 * Fh, K, seed=2, scale=K ** -0.5))
    bias = rnd(2 * Fh, seed=3)
    perm = ops.geglu_perm(Fh)
    h = A.float() @ W.float().t() + bias
    ref = h[:, :Fh] * F.gelu(h[:, Fh:])
    pre = torch.empty(M, 2 * Fh, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm(A.cuda(), W[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), act=ops.ACT_GEGLU,
                   preact=pre, force_tile=T4)
    close("gemm4_geglu", out, ref)
    close("gemm4_geglu_preact", pre, h[:, perm])


@pytest.mark.parametrize("Fh", [320, 640])
def test_gemm4_geglu_with_residual(Fh):
    """GEGLU + residual on the 256 x 320 tile (N = 2 Fh = 640 / 1280: NF = 10 fragments per wave, the odd (value,
    gate) pair q = 4 takes the 4-column path) -- every output column carries the residual"""
    ops = _ops()
    M, K = 512, 128
    A = b16(rnd(M, K, seed=1))
    W = b16(rnd(2 * Fh, K, seed=2, scale=K ** -0.5))
    bias = rnd(2 * Fh, seed=3)
    res = b16(rnd(M, Fh, seed=4, scale=3.0))
    perm = ops.geglu_perm(Fh)
    h = A.float() @ W.float().t() + bias
    ref = h[:, :Fh] * F.gelu(h[:, Fh:]) + res.float()
    out = ops.gemm(A.cuda(), W[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), act=ops.ACT_GEGLU,
                   residual=res.cuda(), force_tile=T4)
    close(f"gemm4_geglu_residual{Fh}", out, ref)


def test_gemm4_many_items_per_block():
    """more (tile, split) items than CUs: the persistent loop crosses item boundaries with tiles in flight"""
    ops = _ops()
    M, N, K = 256 * 150, 640, 128
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    out = ops.gemm(A.cuda(), W.cuda(), force_tile=T4)
    close("gemm4_many", out, A.float() @ W.float().t())
    out2 = ops.gemm(A.cuda(), W.cuda(), force_tile=T4)
    assert torch.equal(out, out2)


# ---- two-segment A operand (GemmArgs::A2): [A | A2] W^T without materialising the concatenation ---------------------------------
# ---- 256 x 384 tile (gemm4, round 6): the wide tile for the transformer denoisers' widths -- same K order, same MFMA sequence per output
# element as the 256 x 192 tile, so the results must be bit-identical on every epilogue both serve ---------------------------------------
T384, T192 = (256 << 16) | 384, (256 << 16) | 192


@pytest.mark.parametrize("shape", [(256, 384, 64), (512, 1536, 1536), (768, 4608, 1152), (1024, 1152, 4608), (512, 6144, 1536), (2304, 3456, 1152)])
def test_gemm4_384_tile_matches_the_192_tile(shape):
    """the wide tile has ONE epilogue (the fp32-residual-stream one + a run-time tanh-GELU); the 256 x 192 tile serves the same problems
    through the general / lean epilogues: plain products are bit-identical, the others agree to an fp32 rounding before the bf16 one"""
    ops = _ops()
    M, N, K = shape
    rpb = 64
    A, W = b16(rnd(M, K, seed=1)).cuda(), b16(rnd(N, K, seed=2, scale=K ** -0.5)).cuda()
    bias, res = rnd(N, seed=3).cuda(), b16(rnd(M, N, seed=4)).cuda()
    gate = b16(rnd(M // rpb, N, seed=5)).cuda()
    h = A.float().cpu() @ W.float().cpu().t() + bias.cpu()

    def same(name, x, y, ref):
        close(f"gemm4_384_{name}{shape}", x, ref)
        d = (x.float() - y.float()).abs()
        assert float((d > 0).float().mean()) < 1e-3 and float(d.max()) <= 2.0 ** -6 * float(ref.abs().max()), name   # (a rare bf16 tie)

    assert torch.equal(ops.gemm(A, W, force_tile=T384), ops.gemm(A, W, force_tile=T192))
    same("bias_res", ops.gemm(A, W, bias=bias, residual=res, force_tile=T384), ops.gemm(A, W, bias=bias, residual=res, force_tile=T192),
         h + res.float().cpu())
    kw = dict(bias=bias, rowvec=gate, rows_per_batch=rpb, rowvec_mul=True, residual=res)
    same("gate", ops.gemm(A, W, force_tile=T384, **kw), ops.gemm(A, W, force_tile=T192, **kw),
         h * gate.float().cpu().repeat_interleave(rpb, 0) + res.float().cpu())
    same("gelu_tanh", ops.gemm(A, W, bias=bias, act=ops.ACT_GELU_TANH, force_tile=T384),
         ops.gemm(A, W, bias=bias, act=ops.ACT_GELU_TANH, force_tile=T192), F.gelu(h, approximate="tanh"))
    if K >= 1152:   # slab split-K: the finalize kernel sums the same slabs and applies the same epilogue
        assert torch.equal(ops.gemm(A, W, bias=bias, act=ops.ACT_GELU_TANH, force_tile=T384, splitk=3),
                           ops.gemm(A, W, bias=bias, act=ops.ACT_GELU_TANH, force_tile=T192, splitk=3))
    if K >= 128:    # a two-segment A operand [A1 | A2] (the folded LoRA up-projection) is NOT the wide tile's: the planner keeps 256 x 192
        K1 = K - 64
        A1, A2 = A[:, :K1].contiguous(), A[:, K1:].contiguous()
        assert torch.equal(ops.gemm(A1, W, A2=A2, residual=res), ops.gemm(A, W, residual=res, force_tile=T192))


def test_the_planner_takes_the_384_tile_for_the_transformer_widths_and_switch_51_removes_it():
    ops = _ops()
    from flash_diffusion_amd import _lib
    for (M, N, K) in [(32768, 4608, 1152), (16384, 1536, 6144), (16384, 6144, 1536)]:
        p = ops.gemm_plan(M, N, K)
        assert p[:3] == (2, 256, 384), p
        _lib.lib().fdmi_tune_set(51, 1)
        try:
            q = ops.gemm_plan(M, N, K)
        finally:
            _lib.lib().fdmi_tune_set(51, 0)
        assert q[:3] == (2, 256, 192), q
    assert ops.gemm_plan(32768, 1280, 1280)[2] in (320, 160, 128)   # (384 does not divide the UNets' widths)


# ---- 128 x 320 x 32 tile kernel, two blocks per CU (gemm5, round 6: a measured experiment, reached through force_tile only) ----------
T5 = (128 << 16) | 320


@pytest.mark.parametrize("shape", [(128, 320, 64), (256, 320, 64), (640, 640, 320), (1024, 320, 1280), (8192, 960, 448), (2176, 1280, 192)])
def test_gemm5_row_equals_gemm4_bit_for_bit(shape):
    """same wave tile, same accumulation order along K, same epilogue code: the 128-row kernel's output IS the 256-row kernel's
    (where that one applies: M % 256 == 0); against fp32 otherwise.  Covers one and several items per block, the persistent loop's
    hand-over, bias / residual operand sets"""
    ops = _ops()
    M, N, K = shape
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, res = rnd(N, seed=3), b16(rnd(M, N, seed=5))
    ref = A.float() @ W.float().t()
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=T5)
    close(f"gemm5_row{shape}", out, ref + bias + res.float())
    plain = ops.gemm(A.cuda(), W.cuda(), force_tile=T5)
    close(f"gemm5_plain{shape}", plain, ref)
    if M % 256 == 0:
        assert torch.equal(out, ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=T4))
        assert torch.equal(plain, ops.gemm(A.cuda(), W.cuda(), force_tile=T4))


def test_gemm5_geglu_and_two_segment_a():
    ops = _ops()
    M, K, Fh = 512, 192, 640
    A = b16(rnd(M, K, seed=1))
    W = b16(rnd(2 * Fh, K, seed=2, scale=K ** -0.5))
    bias = rnd(2 * Fh, seed=3)
    perm = ops.geglu_perm(Fh)
    pre = torch.empty(M, 2 * Fh, dtype=torch.bfloat16, device="cuda")
    pre4 = torch.empty_like(pre)
    Wp, bp = W[perm].contiguous().cuda(), bias[perm].contiguous().cuda()
    out = ops.gemm(A.cuda(), Wp, bias=bp, act=ops.ACT_GEGLU, preact=pre, force_tile=T5)
    ref4 = ops.gemm(A.cuda(), Wp, bias=bp, act=ops.ACT_GEGLU, preact=pre4, force_tile=T4)
    h = A.float() @ W.float().t() + bias
    close("gemm5_geglu", out, h[:, :Fh] * F.gelu(h[:, Fh:]))
    assert torch.equal(out, ref4) and torch.equal(pre, pre4)
    # two-segment A (the LoRA up-projection as extra K tiles): the seam at a 64-multiple inside the 32-deep tile sequence
    M, N, K1, K2 = 1024, 320, 320, 128
    A1, A2 = b16(rnd(M, K1, seed=1)), b16(rnd(M, K2, seed=2))
    W = b16(rnd(N, K1 + K2, seed=3, scale=(K1 + K2) ** -0.5))
    res = b16(rnd(M, N, seed=5))
    out = ops.gemm(A1.cuda(), W.cuda(), A2=A2.cuda(), residual=res.cuda(), force_tile=T5)
    close("gemm5_a2", out, torch.cat([A1, A2], 1).float() @ W.float().t() + res.float())
    assert torch.equal(out, ops.gemm(A1.cuda(), W.cuda(), A2=A2.cuda(), residual=res.cuda(), force_tile=T4))


A2_SHAPES = [(512, 640, 320, 128), (1024, 1920, 1280, 640), (768, 320, 64, 64), (2048, 960, 320, 384)]
A2_CASES = [(t, sh) for t in (T4, (256 << 16) | 160, (256 << 16) | 128, (256 << 16) | 192, 0) for sh in A2_SHAPES
            if not t or sh[1] % (t & 0xffff) == 0]          # (a forced tile must divide N)


@pytest.mark.parametrize("tile,shape", A2_CASES)
def test_gemm_two_segment_a(tile, shape):
    """every LDS-DMA kernel reads a two-part A operand: the seam at K1 (a multiple of 64) inside an item, at an item's first tile
    (split-K slices that start behind the seam) and across persistent items; tile 0 = the planner's choice"""
    ops = _ops()
    M, N, K1, K2 = shape
    A1, A2 = b16(rnd(M, K1, seed=1)), b16(rnd(M, K2, seed=2))
    W = b16(rnd(N, K1 + K2, seed=3, scale=(K1 + K2) ** -0.5))
    bias, res = rnd(N, seed=4), b16(rnd(M, N, seed=5))
    ref = torch.cat([A1, A2], 1).float() @ W.float().t() + bias + res.float()
    out = ops.gemm(A1.cuda(), W.cuda(), A2=A2.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=tile)
    close(f"gemm_a2{shape}@{tile:x}", out, ref)
    # the same product from the materialised concatenation, same kernel: bit-identical (same tiles, same order)
    if tile:
        cat = torch.cat([A1, A2], 1).contiguous().cuda()
        assert torch.equal(out, ops.gemm(cat, W.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=tile))
    if tile and (K1 + K2) // 64 >= 4:   # split-K: some slices start behind the seam
        sk = 3 if (K1 + K2) // 64 >= 6 else 2
        out = ops.gemm(A1.cuda(), W.cuda(), A2=A2.cuda(), bias=bias.cuda(), residual=res.cuda(), force_tile=tile, splitk=sk)
        close(f"gemm_a2_splitk{shape}@{tile:x}", out, ref)
    # A2 as a column slice of a wider buffer (the LoRA t of a fused q / k / v projection: row stride 3r)
    wide = b16(rnd(M, 3 * K2, seed=6)).cuda()
    sl = wide[:, K2:2 * K2]
    ref2 = torch.cat([A1, sl.cpu()], 1).float() @ W.float().t()
    close(f"gemm_a2_strided{shape}@{tile:x}", ops.gemm(A1.cuda(), W.cuda(), A2=sl, force_tile=tile), ref2)


def test_gemm_two_segment_a_is_refused_where_no_kernel_reads_it():
    ops = _ops()
    A1, A2 = b16(rnd(128, 64, seed=1)).cuda(), b16(rnd(128, 64, seed=2)).cuda()      # M < 256: only the small tiles could run it
    W = b16(rnd(128, 128, seed=3)).cuda()
    with pytest.raises(RuntimeError, match="second A segment"):
        ops.gemm(A1, W, A2=A2)
    A1, A2 = b16(rnd(512, 96, seed=1)).cuda(), b16(rnd(512, 32, seed=2)).cuda()      # seam not on a 64-deep K tile
    with pytest.raises(RuntimeError, match="second A segment"):
        ops.gemm(A1, W, A2=A2)


@pytest.mark.parametrize("cfg", [(2, 16, 16, 320, 320, 32), (1, 32, 32, 1280, 640, 32), (3, 8, 8, 64, 128, 8), (2, 16, 16, 640, 320, 32)])
@pytest.mark.parametrize("silu", [0, 1])
def test_groupnorm_of_a_concatenation(cfg, silu):
    """GroupNorm(+SiLU) forward and input gradient of [x1 | x2] read from the two tensors (groups straddle the seam when C1 is not
    a multiple of the group width: 1280 + 640 channels in 32 groups of 60) against the same kernels on the materialised tensor"""
    ops = _ops()
    B, H, W_, C1, C2, G = cfg
    x1, x2 = b16(rnd(B, H * W_, C1, seed=1)).cuda(), b16(rnd(B, H * W_, C2, seed=2, scale=2.0)).cuda()
    gamma, beta = (rnd(C1 + C2, seed=3) * 0.2 + 1).cuda(), rnd(C1 + C2, seed=4).cuda()
    cat = torch.cat([x1, x2], 2).contiguous()
    y, st = ops.groupnorm_cat_fwd(x1, x2, gamma, beta, G, 1e-5, silu)
    y0, st0 = ops.groupnorm_fwd(cat, gamma, beta, G, 1e-5, silu)
    close(f"gn_cat_stats{cfg}", st, st0, tol_el=1e-5, tol_fro=1e-5)       # (float atomics: equal up to summation order)
    close(f"gn_cat_fwd{cfg}{silu}", y, y0.float(), tol_el=2 ** -7, tol_fro=1e-3)
    xf = cat.float().cpu().view(B, H * W_, G, -1)
    mean, var = xf.mean((1, 3), keepdim=True), xf.var((1, 3), unbiased=False, keepdim=True)
    ref = ((xf - mean) / (var + 1e-5).sqrt()).view(B, H * W_, C1 + C2) * gamma.cpu() + beta.cpu()
    close(f"gn_cat_ref{cfg}{silu}", y, F.silu(ref) if silu else ref)
    dy = b16(rnd(B, H * W_, C1 + C2, seed=5)).cuda()
    dx = ops.groupnorm_cat_bwd(x1, x2, dy, gamma, beta, st, G, 1e-5, silu)
    dx0 = ops.groupnorm_bwd(cat, dy, gamma, beta, st0, G, 1e-5, silu)
    close(f"gn_cat_bwd{cfg}{silu}", dx, dx0.float(), tol_el=2 ** -7, tol_fro=2e-3)


CONVS4 = [
    (4, 8, 8, 320, 320, 3, 1, 1, 0), (2, 16, 16, 64, 640, 3, 1, 1, 0), (4, 16, 16, 128, 320, 3, 2, 1, 0),
    (4, 8, 8, 64, 320, 3, 1, 1, 1), (2, 16, 16, 192, 320, 1, 1, 0, 0), (1, 32, 32, 640, 320, 3, 1, 1, 0),
]


@pytest.mark.parametrize("cfg", CONVS4)
def test_gemm4_conv_fwd_and_dgrad(cfg):
    ops = _ops()
    B, H, W, Ci, Co, k, s, p, ups = cfg
    x = b16(rnd(B, Ci, H, W, seed=1))
    w = b16(rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5))
    bias = rnd(Co, seed=3)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    xin.requires_grad_()
    ref = F.conv2d(xin, w.float(), bias, stride=s, padding=p)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    Mout = ref.shape[0] * ref.shape[2] * ref.shape[3]
    if Mout % 256 == 0:
        y = ops.conv2d_nhwc(xn, ops.pack_conv_weight(w.float()).cuda(), KH=k, KW=k, stride=s, pad=p, ups=ups,
                            bias=bias.cuda(), force_tile=T4)
        close(f"conv4_fwd{cfg}", y.permute(0, 3, 1, 2), ref)
    if Co % 64 == 0 and Ci % 320 == 0 and (B * (H << ups) * (W << ups)) % 256 == 0:
        dy = b16(rnd(*ref.shape, seed=4))
        ref.backward(dy.float())
        dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
        dx = ops.conv2d_nhwc(dyn, ops.pack_conv_weight_dgrad(w.float()).cuda(), KH=k, KW=k, stride=s, pad=p, dgrad=1,
                             out_hw=(H << ups, W << ups), force_tile=T4)
        close(f"conv4_dgrad{cfg}", dx.permute(0, 3, 1, 2), xin.grad)


CONVS3 = [
    (3, 8, 8, 320, 320, 3, 1, 1, 0), (2, 16, 16, 64, 640, 3, 1, 1, 0), (2, 16, 16, 128, 160, 3, 2, 1, 0),
    (2, 8, 8, 64, 128, 3, 1, 1, 1), (2, 16, 16, 192, 128, 1, 1, 0, 0), (4, 8, 8, 128, 256, 4, 2, 1, 0),
]


@pytest.mark.parametrize("cfg", CONVS3)
@pytest.mark.parametrize("tile", BIG)
def test_gemm3_conv_fwd_and_dgrad(cfg, tile):
    ops = _ops()
    B, H, W, Ci, Co, k, s, p, ups = cfg
    x = b16(rnd(B, Ci, H, W, seed=1))
    w = b16(rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5))
    bias = rnd(Co, seed=3)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    xin.requires_grad_()
    ref = F.conv2d(xin, w.float(), bias, stride=s, padding=p)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    Mout = ref.shape[0] * ref.shape[2] * ref.shape[3]
    if Mout >= 256:
        y = ops.conv2d_nhwc(xn, ops.pack_conv_weight(w.float()).cuda(), KH=k, KW=k, stride=s, pad=p, ups=ups,
                            bias=bias.cuda(), force_tile=tile)
        close(f"conv3_fwd{cfg}_{tile:x}", y.permute(0, 3, 1, 2), ref)
    if Co % 64 == 0 and Ci >= 128 and B * (H << ups) * (W << ups) >= 256:
        dy = b16(rnd(*ref.shape, seed=4))
        ref.backward(dy.float())
        dyn = dy.permute(0, 2, 3, 1).contiguous().cuda()
        dx = ops.conv2d_nhwc(dyn, ops.pack_conv_weight_dgrad(w.float()).cuda(), KH=k, KW=k, stride=s, pad=p, dgrad=1,
                             out_hw=(H << ups, W << ups), force_tile=tile)
        close(f"conv3_dgrad{cfg}_{tile:x}", dx.permute(0, 3, 1, 2), xin.grad)


# ---- round 4: the lean / line-wide epilogues store exactly what the general epilogue stores ------------------------------------------
EPI_CASES = [
    # (kind, tile, M or (B, H), N, K or Cin, bias, residual, rowvec, gn)
    ("row", (256 << 16) | 320, 1024, 320, 320, True, True, False, False),
    ("row", (256 << 16) | 320, 512, 960, 320, True, False, False, False),
    ("row", (256 << 16) | 320, 512, 640, 1280, False, True, False, False),
    ("row", (256 << 16) | 320, 768, 320, 64, False, False, False, False),
    ("row", (256 << 16) | 192, 512, 1152, 1152, True, True, False, False),
    ("row", (256 << 16) | 160, 1232, 320, 768, True, True, False, False),      # ragged M: partial tiles fall back per wave
    ("row", (256 << 16) | 128, 2048, 128, 320, False, True, False, False),
    ("row", (256 << 16) | 128, 512, 384, 128, True, False, False, False),
    ("conv", (256 << 16) | 320, (2, 32), 320, 320, True, False, True, False),
    ("conv", (256 << 16) | 320, (2, 32), 640, 320, True, True, False, False),
    ("conv", (256 << 16) | 160, (2, 16), 320, 640, True, True, True, False),
    ("conv", (256 << 16) | 320, (2, 32), 320, 320, True, True, True, True),   # GroupNorm sums from the epilogue
    ("row", (256 << 16) | 320, 2048, 320, 320, True, True, False, True),
]


@pytest.mark.parametrize("case", EPI_CASES)
def test_lean_and_line_wide_epilogues_store_what_the_general_one_stores(case):
    """developer knob 40: 64 = the general epilogue for every problem, 256 = lean with every column pair single (the MFMA layout's
    16-row x 64-byte accesses), 0 = lean with the line-wide lane exchange.  Same accumulators, same order of the terms: the three
    outputs must be BIT-identical (and the GroupNorm sums equal up to the order of the float atomics)."""
    ops = _ops()
    from flash_diffusion_amd._lib import lib
    kind, tile, Mspec, N, Kc, has_b, has_r, has_v, has_gn = case
    if kind == "row":
        M, B, HW = Mspec, (Mspec // 512 if has_gn else 1), (512 if has_gn else Mspec)
        A = b16(rnd(M, Kc, seed=1)).cuda()
        w = b16(rnd(N, Kc, seed=2, scale=Kc ** -0.5)).cuda()
        kw = {}
    else:
        B, H = Mspec
        HW, M = H * H, B * H * H
        A = b16(rnd(B, H, H, Kc, seed=1)).cuda()
        w = ops.pack_conv_weight(b16(rnd(N, Kc, 3, 3, seed=2, scale=(9 * Kc) ** -0.5)).float()).cuda()
        kw = dict(M=M, conv=dict(Hin=H, Win=H, Cin=Kc, Hout=H, Wout=H, KH=3, KW=3, stride=1, pad=1))
    if has_b:
        kw["bias"] = rnd(N, seed=3).cuda()
    if has_r:
        kw["residual"] = b16(rnd(M, N, seed=4)).cuda()
    if has_v:
        kw["rowvec"], kw["rows_per_batch"] = b16(rnd(B, N, seed=5)).cuda(), HW
    outs, sums = {}, {}
    try:
        for knob in (64, 256, 0):
            lib().fdmi_tune_set(40, knob)
            if has_gn:
                st = torch.zeros(B, 32, 2, dtype=torch.float32, device="cuda")
                outs[knob] = ops.gemm(A, w, gn=(st, HW), **kw)
                sums[knob] = st
            else:
                outs[knob] = ops.gemm(A, w, force_tile=tile, **kw)
            torch.cuda.synchronize()
    finally:
        lib().fdmi_tune_set(40, 0)
    assert torch.isfinite(outs[64].float()).all()
    assert torch.equal(outs[256], outs[64]), f"lean epilogue differs from the general one: {case}"
    assert torch.equal(outs[0], outs[64]), f"line-wide epilogue differs from the general one: {case}"
    if has_gn:
        for knob in (256, 0):
            close(f"epi_gn_sums{case}[{knob}]", sums[knob], sums[64], tol_el=1e-5, tol_fro=1e-5)


# ---- in-launch split-K reduction (round 5; gemm_tile.h::splitk_arrive, launch_gemm's knob 48) -----------------------------------------
def _lib():
    from flash_diffusion_amd._lib import lib
    return lib()


@pytest.mark.parametrize("tile", [(256 << 16) | 320, (256 << 16) | 160, (256 << 16) | 128, (256 << 16) | 192])
@pytest.mark.parametrize("sk", [2, 3, 4])
def test_inlaunch_splitk_reduction_equals_the_finalize_kernel(tile, sk):
    """developer switch 48 = n: the 256-row kernels reduce up to n split-K slabs inside the launch (the block that stores a tile's last
    slab sums them in slab order and runs the epilogue) -- OFF by default, the finalize kernel measured faster on the step
    (profiles/r5_knob_ab_inlaunch_splitk.txt).  The switched path equals the reference with every epilogue term, is bit-identical
    from run to run (the sum does not depend on which block came last) and stays within one bf16 ulp of the finalize-kernel path
    (same slab sums, the finalize kernel spells its epilogue arithmetic differently)"""
    ops, L = _ops(), _lib()
    BN = tile & 0xffff
    M, N, K, rpb = 1024, 3 * BN, 2560, 256
    A, W = b16(rnd(M, K, seed=1)), b16(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, rowvec, res = rnd(N, seed=3), b16(rnd(M // rpb, N, seed=4)), b16(rnd(M, N, seed=5))
    ref = 0.5 * (A.float() @ W.float().t()) + bias + rowvec.float().repeat_interleave(rpb, 0) + res.float()
    kw = dict(bias=bias.cuda(), rowvec=rowvec.cuda(), rows_per_batch=rpb, residual=res.cuda(), alpha=0.5, splitk=sk, force_tile=tile)
    Ad, Wd = A.cuda(), W.cuda()
    try:
        L.fdmi_tune_set(48, 0)
        fin = ops.gemm(Ad, Wd, **kw)
        L.fdmi_tune_set(48, 4)
        out = ops.gemm(Ad, Wd, **kw)
        close(f"inlaunch_sk{sk}_{tile:x}", out, ref)
        for _ in range(5):
            assert torch.equal(out, ops.gemm(Ad, Wd, **kw))
        d = (out.float() - fin.float()).abs()
        assert float((d > 2 ** -7 * fin.float().abs().clamp_min(1e-3)).float().mean()) == 0.0 and float((d > 0).float().mean()) < 0.02
        out = ops.gemm(Ad, Wd, act=ops.ACT_SILU, **kw)
        close(f"inlaunch_sk{sk}_silu_{tile:x}", out, F.silu(ref))
    finally:
        L.fdmi_tune_set(48, 0)


def test_inlaunch_splitk_reduction_conv_and_two_streams():
    """a 3x3 convolution with split-K 2 (the 16x16 level of the C2 teacher: M = 8192, N = 1280, K = 11520 at sk = 2) through the
    in-launch reduction, and two streams launching such GEMMs at the same time -- every stream has its own ticket region"""
    ops, L = _ops(), _lib()
    L.fdmi_tune_set(48, 4)
    try:
        _inlaunch_conv_two_streams(ops)
    finally:
        L.fdmi_tune_set(48, 0)


def _inlaunch_conv_two_streams(ops):
    B, hw, ci, co = 8, 16, 640, 640
    x = b16(rnd(B, hw, hw, ci, seed=1))
    w = b16(rnd(co, 9 * ci, seed=2, scale=(9 * ci) ** -0.5))
    bias = rnd(co, seed=3)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, co)
    conv = dict(Hin=hw, Win=hw, Cin=ci, Hout=hw, Wout=hw, KH=3, KW=3, stride=1, pad=1)
    M = B * hw * hw
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    for tile in ((256 << 16) | 320, (256 << 16) | 160):
        out = ops.gemm(xd, wd, M=M, bias=bd, conv=conv, splitk=2, force_tile=tile)
        close(f"inlaunch_conv_{tile:x}", out, ref)
    A, W = b16(rnd(2048, 1280, seed=7)).cuda(), b16(rnd(640, 1280, seed=8, scale=1280 ** -0.5)).cuda()
    want = ops.gemm(A, W, splitk=4, force_tile=(256 << 16) | 320)
    want_c = ops.gemm(xd, wd, M=M, bias=bd, conv=conv, splitk=2, force_tile=(256 << 16) | 320)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for it in range(20):
        with torch.cuda.stream(s1):
            o1 = ops.gemm(A, W, splitk=4, force_tile=(256 << 16) | 320)
        with torch.cuda.stream(s2):
            o2 = ops.gemm(xd, wd, M=M, bias=bd, conv=conv, splitk=2, force_tile=(256 << 16) | 320)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    for o1, o2 in outs:
        assert torch.equal(o1, want) and torch.equal(o2, want_c)


def test_upload_of_small_host_tensors_keeps_values_and_dtypes():
    """ops.upload (round 6): the step's start timesteps / sigmas / drawn indices reach the device through a fill launch (all elements
    equal) or a pinned non-blocking copy -- never through a pageable copy, which blocks the host until the stream has drained"""
    ops = _ops()
    for t in (torch.full((16,), 999, dtype=torch.long), torch.tensor([10, 250, 500, 750, 10]), torch.tensor([0.25, 0.5]),
              torch.full((4,), 0.7071, dtype=torch.float32), torch.zeros(0), torch.tensor([True, False])):
        d = ops.upload(t, "cuda")
        assert d.is_cuda and d.dtype == t.dtype and d.shape == t.shape and torch.equal(d.cpu(), t)
    on = torch.arange(4, device="cuda")
    assert ops.upload(on, "cuda") is on
