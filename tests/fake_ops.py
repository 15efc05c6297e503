"""TEST INFRASTRUCTURE -- torch-on-CPU stand-ins with the SIGNATURES of flash_diffusion_amd.ops, used only by
tests/test_dit_host_logic.py (monkeypatched in place of the real module) to check the HOST composition of the DiT path --
which launches, in which order, with which operands, and the hand-written backward formulas around them -- on a machine
without a GPU.  bf16 storage is mimicked (fp32 arithmetic, results rounded to bf16).  The product never imports this file:
flash_diffusion_amd.ops has no CPU path and raises without libfdmi.so; kernel numerics are covered by the -m gpu tests."""
import math

import torch

BF16 = torch.bfloat16
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_RELU, ACT_GELU, ACT_GELU_TANH = 0, 1, 2, 3, 4, 5


def _dev(t):
    return t


def upload(t, device):   # ops.upload: a small host tensor on `device` without a blocking copy (on CPU: the tensor itself)
    return t.to(device)


def gemm(A, W, *, bias=None, residual=None, act=ACT_NONE, preact=None, out=None, out_f32=False, alpha=1.0, splitk=1,
         accum_atomic=False, rowvec=None, rows_per_batch=1, rowvec_mul=False, **kw):
    assert A.dtype == BF16 and W.dtype == BF16 and A.shape[1] == W.shape[1] and A.shape[1] % 8 == 0, (A.shape, W.shape)
    assert A.stride(1) == 1 and W.stride(1) == 1 and A.stride(0) % 8 == 0 and W.stride(0) % 8 == 0
    v = alpha * (A.float() @ W.float().t())
    if bias is not None:
        assert bias.dtype == torch.float32
        v = v + bias
    if rowvec is not None:      # per-sample row vector: added, or (rowvec_mul) the adaLN gate multiplying before the residual
        assert rowvec.dtype == BF16 and rowvec.stride(1) == 1 and rowvec.stride(0) % 8 == 0 and v.shape[0] % rows_per_batch == 0
        rv = rowvec.float().repeat_interleave(rows_per_batch, dim=0)
        v = v * rv if rowvec_mul else v + rv
    else:
        assert not rowvec_mul
    if residual is not None:
        assert residual.dtype == BF16 and residual.shape == v.shape
        v = v + residual.float()
    assert preact is None, "the GEMM epilogue saves pre-activations only for GEGLU"
    if act == ACT_SILU:
        v = torch.nn.functional.silu(v)
    elif act == ACT_GELU_TANH:
        v = torch.nn.functional.gelu(v, approximate="tanh")
    elif act == ACT_GELU:
        v = torch.nn.functional.gelu(v)
    else:
        assert act == ACT_NONE
    if out is not None:
        assert out.shape == v.shape and out.stride(1) == 1 and (out.dtype == torch.float32) == bool(out_f32)
        if accum_atomic:
            assert out_f32, "accum_atomic needs f32 C"
            out.add_(v)
        else:
            out.copy_(v)
        return out
    assert not accum_atomic
    return v if out_f32 else v.to(BF16)


TUNE = {}          # developer knobs as the host code sees them (ops.tune)


def tune(key):
    return TUNE.get(key, 0)


def wgrad_tn(X, Y, out):
    """out[N1, N2] (f32) += X[M, N1]^T @ Y[M, N2]"""
    assert X.dtype == BF16 and Y.dtype == BF16 and X.shape[0] == Y.shape[0] and out.dtype == torch.float32
    assert X.stride(1) == 1 and Y.stride(1) == 1 and X.stride(0) % 8 == 0 and Y.stride(0) % 8 == 0 and out.stride(1) == 1
    assert X.shape[1] % 8 == 0 and Y.shape[1] % 8 == 0 and out.shape == (X.shape[1], Y.shape[1])
    out.add_(X.float().t() @ Y.float())
    return out


def cast_transpose(w):
    assert w.dtype == torch.float32 and w.dim() == 2
    return w.to(BF16).contiguous(), w.t().to(BF16).contiguous()


def transpose2d_pad(x, rows_pad):
    rows, cols = x.shape
    out = torch.zeros(cols, rows_pad, dtype=BF16)
    out[:, :rows] = x.t()
    return out


def f32_to_bf16(x):
    assert x.dtype == torch.float32
    return x.to(BF16)


def silu(x):
    return torch.nn.functional.silu(x.float()).to(BF16)


def silu_bwd(x, dy):
    xf = x.float()
    s = torch.sigmoid(xf)
    return (dy.float() * (s * (1 + xf * (1 - s)))).to(BF16)


def timestep_embed(t, dim, flip=True, shift=0.0, dtype=BF16):
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - shift))
    arg = t.float()[:, None] * freq[None]
    sn, cs = torch.sin(arg), torch.cos(arg)
    return (torch.cat([cs, sn], 1) if flip else torch.cat([sn, cs], 1)).to(dtype)


def _heads(x, H):
    B, S, Cc = x.shape
    return x.float().reshape(B, S, H, Cc // H).transpose(1, 2)


def attn_fwd(q, k, v, H, scale, need_lse=False, out=None, lse_out=None):
    for t in (q, k, v):
        assert t.dtype == BF16 and t.dim() == 3 and t.stride(2) == 1
    s = (_heads(q, H) @ _heads(k, H).transpose(-1, -2)) * scale
    lse = torch.logsumexp(s, dim=-1)
    o = (torch.softmax(s, dim=-1) @ _heads(v, H)).transpose(1, 2).reshape(q.shape).to(BF16)
    if out is not None:
        out.copy_(o)
        o = out
    if lse_out is not None:
        lse_out.copy_(lse)
        lse = lse_out
    return (o, lse) if need_lse else o


def attn_bwd(q, k, v, o, do, lse, H, scale, out=None):
    qh, kh, vh, doh, oh = (_heads(t, H) for t in (q, k, v, do, o))
    p = torch.exp((qh @ kh.transpose(-1, -2)) * scale - lse[..., None])
    dv = p.transpose(-1, -2) @ doh
    dp = doh @ vh.transpose(-1, -2)
    ds = p * (dp - (doh * oh).sum(-1, keepdim=True)) * scale
    dq, dk = ds @ kh, ds.transpose(-1, -2) @ qh
    back = lambda t, ref: t.transpose(1, 2).reshape(ref.shape).to(BF16)
    res = back(dq, q), back(dk, k), back(dv, v)
    if out is not None:
        for dst, src in zip(out, res):
            dst.copy_(src)
        return out
    return res


def _mod(v, rows_per_batch):
    assert v.dtype == BF16 and v.dim() == 2 and v.stride(1) == 1 and v.stride(0) % 8 == 0
    return v.float().repeat_interleave(rows_per_batch, dim=0)


def _ln(x, eps):
    xf = x.float()
    mean = xf.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((xf - mean) ** 2).mean(-1, keepdim=True) + eps)
    return (xf - mean) * rstd, mean, rstd


def layernorm_mod_fwd(x, shift, scale, rows_per_batch, eps, need_stats=False):
    assert x.dtype == BF16 and x.dim() == 2 and x.is_contiguous()
    xh, mean, rstd = _ln(x, eps)
    y = xh * (1 + _mod(scale, rows_per_batch)) + _mod(shift, rows_per_batch)
    return y.to(BF16), (torch.cat([mean, rstd], 1) if need_stats else None)


def layernorm_mod_bwd(x, dy, scale, rows_per_batch, eps):
    xh, _, rstd = _ln(x, eps)
    dxh = dy.float() * (1 + _mod(scale, rows_per_batch))
    dx = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
    return dx.to(BF16)


def gate_residual(x, gate, res, rows_per_batch):
    y = _mod(gate, rows_per_batch) * x.float()
    if res is not None:
        y = y + res.float()
    return y.to(BF16)


def gelu_tanh(x):
    return torch.nn.functional.gelu(x.float(), approximate="tanh").to(BF16)


def gelu_tanh_bwd(x, dy):
    xf = x.float().requires_grad_(True)
    with torch.enable_grad():
        y = torch.nn.functional.gelu(xf, approximate="tanh")
    return torch.autograd.grad(y, xf, dy.float())[0].to(BF16)


def batch_colsum(dy, x=None, stats=None, *, rows_per_batch, want_mul=True, want_sum=True):
    rows, Cc = dy.shape
    B = rows // rows_per_batch
    d = dy.float()
    o0 = o1 = None
    if want_mul:
        f = x.float()
        if stats is not None:
            f = (f - stats[:, :1]) * stats[:, 1:]
        o0 = (d * f).view(B, rows_per_batch, Cc).sum(1)
    if want_sum:
        o1 = d.view(B, rows_per_batch, Cc).sum(1)
    return o0, o1


def axpby(x0, c0, x1=None, c1=0.0, x2=None, c2=0.0, x3=None, c3=0.0, out=None):
    assert x0.dtype == torch.float32 and x0.is_contiguous()
    r = x0 * c0
    for x, c in ((x1, c1), (x2, c2), (x3, c3)):
        if x is not None:
            assert x.dtype == torch.float32 and x.is_contiguous() and x.shape == x0.shape
            r = r + x * c
    return r


def add_noise(z, noise, sa, sb):
    shape = (-1,) + (1,) * (z.dim() - 1)
    return sa.view(shape) * z + sb.view(shape) * noise


def adamw_(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=1, grad_scale=1.0):
    """torch.optim.AdamW update (decoupled weight decay, bias correction), in place on fp32 buffers"""
    gg = g * grad_scale
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    p.mul_(1 - lr * weight_decay)
    p.addcdiv_(m / (1 - beta1 ** step), (v / (1 - beta2 ** step)).sqrt() + eps, value=-lr)


# ---- stand-ins for the fused loss launches of flash.py (autograd Functions that call libfdmi.so directly) ---------------------
class FakeDistillLoss:
    """flash._DistillLoss: mean over the batch of the per-sample mean (|s - t|^p), p = 2 (l2) / 1 (l1)  (FD:368-382)"""

    @staticmethod
    def apply(s, t, l1):
        d = (s - t).abs() if l1 else (s - t) ** 2
        return d.reshape(s.shape[0], -1).mean(1).mean()


class FakeDmdLoss:
    """flash._DmdLoss (FD:459-499): x0 = inv_a noisy + ms_a real; w = 1 / (mean|s - x0| + 1e-5) per sample;
    loss = mse(s, (s - w (real - fake) kb).detach())"""

    @staticmethod
    def apply(s, noisy, real, fake, inv_a, ms_a, kb):
        shp = (-1,) + (1,) * (s.dim() - 1)
        x0 = inv_a.view(shp) * noisy + ms_a.view(shp) * real
        w = 1.0 / ((s - x0).abs().mean(list(range(1, s.dim())), keepdim=True) + 1e-5)
        coeff = (real - fake) * kb.view(shp)
        return torch.nn.functional.mse_loss(s, (s - w * coeff).detach(), reduction="mean")
