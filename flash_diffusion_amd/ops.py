"""Thin Python wrappers over the op-level C-ABI (include/fdmi.h).  torch is used only for device
memory and the current stream; every computation below runs in a hand-written HIP kernel."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import GemmDesc, check, lib, ptr, stream_ptr

BF16 = torch.bfloat16
F32 = torch.float32
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_RELU, ACT_GELU, ACT_GELU_TANH = 0, 1, 2, 3, 4, 5
# Every activation op below dispatches on the dtype of its operand: bf16 tensors run the measured MFMA path, fp32 tensors the
# fp32 VALIDATION kernels (csrc/ref32.hip: exact-f32 MFMA, fp32 storage, fp64 statistics) -- the parity gate against the fp32
# oracle.  A model picks one of the two with its `precision` argument; nothing mixes them.


def _f32(t):
    return t.dtype == F32


def _dev(t):
    assert t.is_cuda, "fdmi ops need device tensors (no CPU fallback)"
    return t


# ---- weight packing (one-off, init time) ---------------------------------------------------------
def upload(t, device):
    """A small HOST tensor (start timesteps, sigmas, drawn indices) on `device` WITHOUT making the host wait for the stream: a
    pageable host-to-device copy is stream-ordered AND blocks the caller until it has run -- inside a training step that is a full
    device synchronisation per step (round 6, scripts/host_timeline.py: FlashDiffusion._get_timesteps stood 130 - 150 ms per C2 step
    in `t0.to(device)`, the GPU then idled until the host had issued the next step's first launches).  A tensor whose elements are
    all equal becomes a fill launch (the value travels as a kernel argument); anything else goes through a pinned staging copy
    (torch's caching host allocator keeps the block alive until the copy has run)."""
    device = torch.device(device)
    if t.is_cuda or device.type != "cuda":
        return t.to(device)
    if t.numel() > 0 and t.dtype != torch.bool and bool((t == t.reshape(-1)[0]).all()):
        return torch.full(t.shape, t.reshape(-1)[0].item(), dtype=t.dtype, device=device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def pack_conv_weight(w, dtype=BF16):
    """OIHW f32 -> [O][KH][KW][I] rows (K = KH*KW*I contiguous) in `dtype`; I zero-padded to a multiple of 8."""
    O, I, KH, KW = w.shape
    Ip = (I + 7) // 8 * 8
    p = torch.zeros(O, KH, KW, Ip, dtype=torch.float32, device=w.device)
    p[..., :I] = w.permute(0, 2, 3, 1)
    return p.reshape(O, KH * KW * Ip).to(dtype).contiguous()


def pack_conv_weight_dgrad(w, dtype=BF16):
    """OIHW f32 -> dgrad operand [I][KH][KW][O] in `dtype` (rows = input channels)."""
    O, I, KH, KW = w.shape
    Op = (O + 7) // 8 * 8
    p = torch.zeros(I, KH, KW, Op, dtype=torch.float32, device=w.device)
    p[..., :O] = w.permute(1, 2, 3, 0)
    return p.reshape(I, KH * KW * Op).to(dtype).contiguous()


def geglu_perm(n_half, device=None):
    """Row permutation putting (value, gate) rows of a GEGLU projection into 16-wide interleave."""
    idx = torch.arange(2 * n_half, device=device)
    blk, off = idx // 32, idx % 32
    return torch.where(off < 16, blk * 16 + off, n_half + blk * 16 + off - 16)


# ---- GEMM / conv -------------------------------------------------------------------------------
def gemm(A, W, *, M=None, N=None, K=None, lda=None, bias=None, rowvec=None, rows_per_batch=1, residual=None,
         act=ACT_NONE, preact=None, out=None, out_f32=False, alpha=1.0, splitk=1, ws=None, accum_atomic=False,
         force_tile=0, use_glds=True, conv=None, gn=None, A2=None, rowvec_mul=False):
    """out[M,N] = A[M,K] @ W[N,K]^T (+epilogue).  conv: dict(Hin,Win,Cin,Hout,Wout,KH,KW,stride,pad,ups,dgrad)
    with A the NHWC activation.  splitk: 1 none, > 1 forced, 0 the planner's choice (slab workspace sized to it).  gn=(stats [B,G,2] f32 zeroed, rows_per_sample): the epilogue also accumulates the GroupNorm
    sums of the output (fdmi_gemm_gn; raises when the problem's kernel cannot -- ask gemm_gn_ok first)."""
    _dev(A)
    d = GemmDesc()
    N = W.shape[0] if N is None else N
    K = W.shape[1] if K is None else K
    if conv is None:
        M = A.shape[0] if M is None else M
        d.mode = 0
        d.lda = A.stride(0) if lda is None else lda
    else:
        d.mode = 1
        for k in ("Hin", "Win", "Cin", "Hout", "Wout", "KH", "KW", "stride", "pad", "ups", "dgrad"):
            setattr(d, k, int(conv.get(k, 0)))
        assert M is not None
    d.M, d.N, d.K = M, N, K
    if A2 is not None:   # second segment of the reduction index: the operand is [A | A2] (never materialised)
        assert conv is None and A2.dtype == BF16 and A2.shape[0] == A.shape[0]
        d.A2, d.lda2, d.K1 = ptr(A2), A2.stride(0), A.shape[1]
        assert K == A.shape[1] + A2.shape[1], "W must span both segments"
    d.A, d.W, d.ldw = ptr(A), ptr(W), W.stride(0)
    d.bias = ptr(bias)
    d.rowvec, d.rowvec_ld, d.rows_per_batch = ptr(rowvec), (rowvec.stride(0) if rowvec is not None else 0), rows_per_batch
    d.rowvec_mul = int(bool(rowvec_mul))       # gate: (acc + bias) * rowvec + residual
    d.residual, d.ldr = ptr(residual), (residual.stride(0) if residual is not None else 0)
    d.act = act
    d.preact, d.ldp = ptr(preact), (preact.stride(0) if preact is not None else 0)
    Nout = N // 2 if act == ACT_GEGLU else N
    if _f32(A):   # fp32 validation kernels: every operand float, float output, no split-K workspace
        assert W.dtype == F32 and gn is None and all(t is None or t.dtype == F32 for t in (rowvec, residual, preact, out))
        if out is None:
            out = torch.empty(M, Nout, dtype=F32, device=A.device)
        d.C, d.ldc, d.out_f32, d.alpha, d.splitk, d.accum_atomic = ptr(out), out.stride(0), 1, alpha, 1, int(accum_atomic)
        check(lib().fdmi_gemm_f32(C.byref(d), stream_ptr()))
        return out
    if out is None:
        out = torch.empty(M, Nout, dtype=torch.float32 if out_f32 else BF16, device=A.device)
    d.C, d.ldc, d.out_f32 = ptr(out), out.stride(0), int(out.dtype == torch.float32)
    d.alpha = alpha
    d.accum_atomic, d.force_tile, d.use_glds = int(accum_atomic), force_tile, int(use_glds)
    if splitk <= 0 and not accum_atomic and ws is None:   # auto: ask the launcher's planner (host only), size the slabs to its choice
        d.splitk, d.ws = 0, None
        pk = [C.c_int32() for _ in range(4)]
        check(lib().fdmi_gemm_plan(C.byref(d), *[C.byref(x) for x in pk]))
        splitk = pk[3].value
    if splitk != 1 and not accum_atomic and ws is None:  # fp32 slabs [splitk][M][N]
        ws = torch.empty((splitk if splitk > 1 else 16) * M * N, dtype=torch.float32, device=A.device)
    d.splitk, d.ws = splitk, ptr(ws)
    if gn is not None:
        stats, rows = gn
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape[2] == 2
        check(lib().fdmi_gemm_gn(C.byref(d), ptr(stats), rows, stats.shape[1], stream_ptr()))
    else:
        check(lib().fdmi_gemm(C.byref(d), stream_ptr()))
    return out


def tune(key):
    """current value of a developer knob (fdmi_tune_set / FDMI_TUNE)"""
    return lib().fdmi_tune_value(key)


DETERMINISTIC_KNOB = 50


class deterministic:
    """Context manager / switch of the library's DETERMINISTIC MODE (knob 50, csrc/common.h::fdmi_det): every floating-point
    accumulation whose order the production kernels leave to the hardware (fp32 atomics of the GroupNorm-sum epilogues, of the
    statistics passes, of the TN weight-gradient row splits, of the column sums and of the scalar losses) runs in a fixed order, so
    two runs of the same step are bit-identical.  A test / debugging mode (slower); plans read the switch when they run, so it can
    be flipped between steps.  `with ops.deterministic(): ...` or `ops.deterministic.set(True)`."""

    def __init__(self, on=True):
        self.on, self.prev = bool(on), None

    @staticmethod
    def set(on):
        lib().fdmi_tune_set(DETERMINISTIC_KNOB, 1 if on else 0)

    @staticmethod
    def enabled():
        return lib().fdmi_tune_value(DETERMINISTIC_KNOB) != 0

    def __enter__(self):
        self.prev = deterministic.enabled()
        deterministic.set(self.on)
        return self

    def __exit__(self, *exc):
        deterministic.set(self.prev)
        return False


def wgrad_tn_group(problems):
    """[(X, Y, out), ...] (at most 6, bf16 operands): every out += X^T @ Y as wgrad_tn does, in ONE launch (fdmi_wgrad_tn_group)"""
    from ._lib import WgradProblem
    import ctypes as C
    arr = (WgradProblem * len(problems))()
    for q, (X, Y, out) in zip(arr, problems):
        unit = lambda t: t.shape[1] == 1 or t.stride(1) == 1
        assert Y.shape[0] == X.shape[0] and unit(X) and unit(Y) and unit(out) and out.dtype == torch.float32 and not _f32(X)
        q.X, q.ldx, q.Y, q.ldy, q.M = X.data_ptr(), X.stride(0), Y.data_ptr(), Y.stride(0), X.shape[0]
        q.N1, q.N2, q.C, q.ldc = X.shape[1], Y.shape[1], out.data_ptr(), out.stride(0)
    check(lib().fdmi_wgrad_tn_group(C.cast(arr, C.c_void_p), len(problems), stream_ptr()))
    return [o for _, _, o in problems]


def wgrad_tn(X, Y, out):
    """out[N1, N2] (f32, accumulated) += X[M, N1]^T @ Y[M, N2]: both operands row-major bf16, contraction over the rows"""
    M = X.shape[0]
    # (a size-1 trailing dimension carries an arbitrary stride -- e.g. the [M, 1] logit gradient of a PatchGAN head's last conv)
    unit = lambda t: t.shape[1] == 1 or t.stride(1) == 1
    assert Y.shape[0] == M and unit(X) and unit(Y) and out.dtype == torch.float32 and unit(out)
    if _f32(X):
        check(lib().fdmi_wgrad_tn_f32(ptr(X), X.stride(0), ptr(Y), Y.stride(0), M, X.shape[1], Y.shape[1], ptr(out), out.stride(0),
                                      stream_ptr()))
        return out
    check(lib().fdmi_wgrad_tn(ptr(X), X.stride(0), ptr(Y), Y.stride(0), M, X.shape[1], Y.shape[1], ptr(out), out.stride(0),
                              stream_ptr()))
    return out


def gemm_plan(M, N, K, act=ACT_NONE, conv=None, ws=True, **fields):
    """host-only: (kernel, BM, BN, splitk) the launcher would pick for this problem -- kernel 0: gemm.hip tiles, 1: gemm3
    (256 x BN ring), 2: gemm4 (256 x 320 / 192) (fdmi_gemm_plan)"""
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.use_glds, d.alpha, d.act = M, N, K, K, K, (N // 2 if act == ACT_GEGLU else N), 1, 1.0, act
    d.splitk = 0 if ws else 1          # 0: the planner may split K (a workspace is available)
    if conv is not None:
        d.mode = 1
        for k, v in conv.items():
            setattr(d, k, int(v))
    for k, v in fields.items():
        setattr(d, k, v)
    out = [C.c_int32() for _ in range(4)]
    check(lib().fdmi_gemm_plan(C.byref(d), *[C.byref(x) for x in out]))
    return tuple(x.value for x in out)


def gemm_gn_ok(M, N, K, rows_per_sample, G, conv=None, ldc=None, **fields):
    """host-only: would fdmi_gemm_gn accept this problem (a 256-row kernel, no split-K, full tiles inside one sample)?"""
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.splitk, d.use_glds, d.alpha = M, N, K, K, K, (N if ldc is None else ldc), 1, 1, 1.0
    if conv is not None:
        d.mode = 1
        for k, v in conv.items():
            setattr(d, k, int(v))
    for k, v in fields.items():
        setattr(d, k, v)
    return bool(lib().fdmi_gemm_gn_ok(C.byref(d), rows_per_sample, G))


def conv2d_nhwc(x, w_packed, *, KH, KW, stride=1, pad=0, ups=0, dgrad=0, out_hw=None, **kw):
    """x [B,H,W,C] bf16 NHWC -> [B,Ho,Wo,N] (as [M,N])."""
    B, H, Wd, Cin = x.shape
    if out_hw is None:
        Hv, Wv = H << ups, Wd << ups
        Ho = (Hv + 2 * pad - KH) // stride + 1
        Wo = (Wv + 2 * pad - KW) // stride + 1
    else:
        Ho, Wo = out_hw
    conv = dict(Hin=H, Win=Wd, Cin=Cin, Hout=Ho, Wout=Wo, KH=KH, KW=KW, stride=stride, pad=pad, ups=ups,
                dgrad=dgrad)
    y = gemm(x, w_packed, M=B * Ho * Wo, conv=conv, **kw)
    return y.view(B, Ho, Wo, -1)


# ---- norms -------------------------------------------------------------------------------------
def groupnorm_fwd(x, gamma, beta, G, eps, silu):
    """x [B,HW,C] bf16 -> (y, stats[B,G,2])"""
    B, HW, Cc = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(B, G, 2, dtype=torch.float32, device=x.device)
    if _f32(x):   # stats = (mean, rstd) per (sample, group)
        check(lib().fdmi_groupnorm_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(stats), ptr(y), B, HW, Cc, G, eps, int(silu),
                                           stream_ptr()))
        return y, stats
    check(lib().fdmi_groupnorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(stats), ptr(y), B, HW, Cc, G, eps,
                                   int(silu), stream_ptr()))
    return y, stats


def groupnorm_cat_fwd(x1, x2, gamma, beta, G, eps, silu):
    """GroupNorm of [x1 | x2] along channels without materialising the concatenation: x1 [B,HW,C1], x2 [B,HW,C2] bf16 ->
    (y [B,HW,C1+C2], stats [B,G,2])"""
    B, HW, C1 = x1.shape
    Cc = C1 + x2.shape[2]
    y = torch.empty(B, HW, Cc, dtype=x1.dtype, device=x1.device)
    stats = torch.empty(B, G, 2, dtype=torch.float32, device=x1.device)
    check(lib().fdmi_groupnorm_cat_fwd(ptr(x1), ptr(x2), C1, ptr(gamma), ptr(beta), ptr(stats), ptr(y), B, HW, Cc, G, eps,
                                       int(silu), stream_ptr()))
    return y, stats


def groupnorm_cat_bwd(x1, x2, dy, gamma, beta, stats, G, eps, silu):
    """input gradient of groupnorm_cat_fwd as ONE [B,HW,C1+C2] tensor (the caller splits it)"""
    B, HW, C1 = x1.shape
    Cc = C1 + x2.shape[2]
    dx = torch.empty(B, HW, Cc, dtype=x1.dtype, device=x1.device)
    bstats = torch.zeros(B, G, 2, dtype=torch.float32, device=x1.device)
    check(lib().fdmi_groupnorm_cat_bwd(ptr(x1), ptr(x2), C1, ptr(dy), ptr(gamma), ptr(beta), ptr(stats), ptr(bstats), ptr(dx), B, HW,
                                       Cc, G, eps, int(silu), 0, stream_ptr()))
    return dx


def groupnorm_apply(x, gamma, beta, stats, eps, silu):
    """groupnorm_fwd without its reduction pass: stats [B,G,2] already hold (sum, sum of squares) per (sample, group)"""
    B, HW, Cc = x.shape
    y = torch.empty_like(x)
    check(lib().fdmi_groupnorm_apply(ptr(x), ptr(gamma), ptr(beta), ptr(stats), ptr(y), B, HW, Cc, stats.shape[1], eps,
                                     int(silu), stream_ptr()))
    return y


def groupnorm_bwd(x, dy, gamma, beta, stats, G, eps, silu, dx=None):
    B, HW, Cc = x.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    if _f32(x):
        check(lib().fdmi_groupnorm_bwd_f32(ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(stats), ptr(dx), B, HW, Cc, G, int(silu),
                                           int(acc), stream_ptr()))
        return dx
    bstats = torch.empty(B, G, 2, dtype=torch.float32, device=x.device)
    check(lib().fdmi_groupnorm_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(stats), ptr(bstats), ptr(dx), B,
                                   HW, Cc, G, eps, int(silu), int(acc), stream_ptr()))
    return dx


def layernorm_fwd(x, gamma, beta, eps):
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    if _f32(x):
        check(lib().fdmi_layernorm_fwd_f32(ptr(x), ptr(gamma), ptr(beta), None, None, 0, 1, ptr(y), None, rows, x.shape[-1], eps,
                                           stream_ptr()))
        return y
    check(lib().fdmi_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), rows, x.shape[-1], eps, stream_ptr()))
    return y


def layernorm_bwd(x, dy, gamma, eps, dx=None):
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    if _f32(x):
        check(lib().fdmi_layernorm_bwd_f32(ptr(x), ptr(dy), ptr(gamma), None, 0, 1, ptr(dx), rows, x.shape[-1], eps, int(acc),
                                           stream_ptr()))
        return dx
    check(lib().fdmi_layernorm_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(dx), rows, x.shape[-1], eps, int(acc),
                                   stream_ptr()))
    return dx


# ---- attention ---------------------------------------------------------------------------------
def attn_fwd(q, k, v, H, scale, need_lse=False, out=None, lse_out=None):
    """q [B,Sq,H*d], k/v [B,Skv,H*d] bf16 -> o [B,Sq,H*d] (, lse [B,H,Sq]); `out` / `lse_out`: preallocated results"""
    B, Sq, Cc = q.shape
    Skv = k.shape[1]
    d = Cc // H
    o = torch.empty_like(q) if out is None else out
    if _f32(q):   # materialised scores in scratch; the backward recomputes them, so there is no log-sum-exp to hand on
        n = lib().fdmi_attn_scratch_elems_f32(B, H, Sq, Skv, 0)
        sc = torch.empty(n, dtype=F32, device=q.device)
        check(lib().fdmi_attn_fwd_f32(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), o.stride(1), B, H, Sq,
                                      Skv, d, scale, ptr(sc), n, stream_ptr()))
        return (o, lse_out) if need_lse else o
    vt = torch.empty(lib().fdmi_attn_tr_elems(B, H, Skv, d), dtype=BF16, device=q.device)
    lse = lse_out if lse_out is not None else (
        torch.empty(B, H, Sq, dtype=torch.float32, device=q.device) if need_lse else None)
    check(lib().fdmi_attn_fwd(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), o.stride(1),
                              ptr(vt), ptr(lse), B, H, Sq, Skv, d, scale, stream_ptr()))
    return (o, lse) if need_lse else o


def attn_causal_fwd(q, k, v, H, scale):
    """causal self-attention, forward only (the frozen text encoders): q, k, v [B, S, H*d] -> o.  Scores are materialised by the
    exact-f32 kernels in both precisions (S = 77: the cost is nil); bf16 operands are staged through fp32 copies."""
    B, S, Cc = q.shape
    d = Cc // H
    qf, kf, vf = (t if t.dtype == F32 else t.float() for t in (q, k, v))
    qf, kf, vf = qf.contiguous(), kf.contiguous(), vf.contiguous()
    o = torch.empty_like(qf)
    n = lib().fdmi_attn_scratch_elems_f32(B, H, S, S, 0)
    sc = torch.empty(n, dtype=F32, device=q.device)
    check(lib().fdmi_attn_causal_fwd_f32(ptr(qf), qf.stride(1), ptr(kf), kf.stride(1), ptr(vf), vf.stride(1), ptr(o), o.stride(1), B, H,
                                         S, d, scale, ptr(sc), n, stream_ptr()))
    return o if q.dtype == F32 else f32_to_bf16(o)


def attn_bias_fwd(q, k, v, H, scale, bias=None, kbias=None):
    """softmax(scale q k^T + bias[h] + kbias[b]) v, forward only (the frozen T5 text encoder): q [B, Sq, H*d], k / v [B, Skv, H*d],
    bias [H, Sq, Skv] f32 or None, kbias [B, Skv] f32 or None.  Exact-f32 materialised scores in both precisions."""
    B, Sq, Cc = q.shape
    Skv, d = k.shape[1], Cc // H
    qf, kf, vf = (t.contiguous() if t.dtype == F32 else t.float().contiguous() for t in (q, k, v))
    assert bias is None or (bias.dtype == F32 and bias.is_contiguous() and tuple(bias.shape) == (H, Sq, Skv))
    assert kbias is None or (kbias.dtype == F32 and kbias.is_contiguous() and tuple(kbias.shape) == (B, Skv))
    o = torch.empty_like(qf)
    n = lib().fdmi_attn_scratch_elems_f32(B, H, Sq, Skv, 0)
    sc = torch.empty(n, dtype=F32, device=q.device)
    check(lib().fdmi_attn_bias_fwd_f32(ptr(qf), qf.stride(1), ptr(kf), kf.stride(1), ptr(vf), vf.stride(1), ptr(o), o.stride(1), B, H,
                                       Sq, Skv, d, scale, ptr(bias), ptr(kbias), ptr(sc), n, stream_ptr()))
    return o if q.dtype == F32 else f32_to_bf16(o)


def rmsnorm(x, w, eps):
    """T5LayerNorm: x [rows, C] (bf16 / f32) * rsqrt(mean(x^2) + eps) * w (f32)"""
    _dev(x)
    assert x.dim() == 2 and x.is_contiguous() and w.dtype == F32 and w.numel() == x.shape[1]
    y = torch.empty_like(x)
    check((lib().fdmi_rmsnorm_f32 if _f32(x) else lib().fdmi_rmsnorm)(ptr(x), ptr(w), ptr(y), x.shape[0], x.shape[1], eps, stream_ptr()))
    return y


def mul(a, b):
    _dev(a)
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    y = torch.empty_like(a)
    check((lib().fdmi_mul_f32 if _f32(a) else lib().fdmi_mul)(ptr(a), ptr(b), ptr(y), a.numel(), stream_ptr()))
    return y


def attn_bwd(q, k, v, o, do, lse, H, scale, out=None):
    B, Sq, Cc = q.shape
    Skv = k.shape[1]
    d = Cc // H
    dq, dk, dv = (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)) if out is None else out
    if _f32(q):
        n = lib().fdmi_attn_scratch_elems_f32(B, H, Sq, Skv, 1)
        sc = torch.empty(n, dtype=F32, device=q.device)
        check(lib().fdmi_attn_bwd_f32(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(do), do.stride(1), ptr(dq),
                                      dq.stride(1), ptr(dk), dk.stride(1), ptr(dv), dv.stride(1), B, H, Sq, Skv, d, scale, ptr(sc), n,
                                      stream_ptr()))
        return dq, dk, dv
    ws = torch.empty(lib().fdmi_attn_bwd_ws_bytes(B, H, Sq, Skv, d), dtype=torch.uint8, device=q.device)
    check(lib().fdmi_attn_bwd(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), o.stride(1),
                              ptr(do), do.stride(1), ptr(lse), ptr(dq), dq.stride(1), ptr(dk), dk.stride(1),
                              ptr(dv), dv.stride(1), ptr(ws), B, H, Sq, Skv, d, scale, stream_ptr()))
    return dq, dk, dv


# ---- misc --------------------------------------------------------------------------------------
def nchw_to_nhwc(x, Cpad, dtype=BF16):
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, Cpad, dtype=dtype, device=x.device)
    fn = lib().fdmi_nchw_to_nhwc_f32 if dtype == F32 else lib().fdmi_nchw_to_nhwc
    check(fn(ptr(x.contiguous()), ptr(y), B, Cc, H * W, Cpad, stream_ptr()))
    return y


def nhwc_to_nchw(x, Cc):
    B, H, W, ld = x.shape
    y = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    fn = lib().fdmi_nhwc_to_nchw_f32 if _f32(x) else lib().fdmi_nhwc_to_nchw
    check(fn(ptr(x), ld, ptr(y), B, Cc, H * W, 0, stream_ptr()))
    return y


def timestep_embed(t, dim, flip=True, shift=0.0, dtype=BF16):
    out = torch.empty(t.shape[0], dim, dtype=dtype, device=t.device)
    fn = lib().fdmi_timestep_embed_f32 if dtype == F32 else lib().fdmi_timestep_embed
    check(fn(ptr(t.float().contiguous()), ptr(out), t.shape[0], dim, int(flip), shift, stream_ptr()))
    return out


def geglu_bwd(pre, dout):
    M, F2 = pre.shape
    dpre = torch.empty_like(pre)
    check(lib().fdmi_geglu_bwd(ptr(pre), ptr(dout), ptr(dpre), M, F2 // 2, stream_ptr()))
    return dpre


def pool2x2_sum(dy):
    B, H2, W2, Cc = dy.shape
    dx = torch.empty(B, H2 // 2, W2 // 2, Cc, dtype=BF16, device=dy.device)
    check(lib().fdmi_pool2x2_sum(ptr(dy), ptr(dx), B, H2 // 2, W2 // 2, Cc, 0, stream_ptr()))
    return dx


def cast_transpose(w):
    rows, cols = w.shape
    wb = torch.empty(rows, cols, dtype=BF16, device=w.device)
    wtb = torch.empty(cols, rows, dtype=BF16, device=w.device)
    check(lib().fdmi_cast_transpose(ptr(w), ptr(wb), ptr(wtb), rows, cols, stream_ptr()))
    return wb, wtb


def transpose2d(x):
    rows, cols = x.shape
    out = torch.empty(cols, rows, dtype=BF16, device=x.device)
    check(lib().fdmi_transpose2d(ptr(x), x.stride(0), ptr(out), rows, rows, cols, stream_ptr()))
    return out


def transpose2d_pad(x, rows_pad):
    """[rows, cols] -> [cols, rows_pad] with the columns past `rows` zero (contraction dims must be multiples of 8)"""
    rows, cols = x.shape
    out = torch.empty(cols, rows_pad, dtype=BF16, device=x.device)
    check(lib().fdmi_transpose2d_pad(ptr(x), x.stride(0), ptr(out), rows_pad, rows, cols, rows_pad, stream_ptr()))
    return out


def f32_to_bf16(x):
    """fp32 -> the bf16 storage of the measured path"""
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib().fdmi_f32_to_bf16(ptr(x), ptr(y), x.numel(), stream_ptr()))
    return y


def silu(x):
    y = torch.empty_like(x)
    check((lib().fdmi_silu_f32 if _f32(x) else lib().fdmi_silu)(ptr(x), ptr(y), x.numel(), stream_ptr()))
    return y


def silu_bwd(x, dy):
    dx = torch.empty_like(x)
    check((lib().fdmi_silu_bwd_f32 if _f32(x) else lib().fdmi_silu_bwd)(ptr(x), ptr(dy), ptr(dx), x.numel(), stream_ptr()))
    return dx


def im2col(x, Ho, Wo, KH, KW, stride, pad):
    """x [B,H,W,C] NHWC -> explicit patch matrix [B*Ho*Wo, KH*KW*C] (the discriminator's weight gradient)"""
    B, H, W, Cc = x.shape
    out = torch.empty(B * Ho * Wo, KH * KW * Cc, dtype=x.dtype, device=x.device)
    check((lib().fdmi_im2col_f32 if _f32(x) else lib().fdmi_im2col)(ptr(x), ptr(out), B, H, W, Cc, Ho, Wo, KH, KW, stride, pad,
                                                                    stream_ptr()))
    return out


def colsum(dy, x, stats, out0, out1, rows, Cc, HW, G, eps):
    """out0[c] += sum_r dy ; out1[c] += sum_r dy * xhat (xhat from the GroupNorm stats of the same precision when given)"""
    if _f32(dy):
        check(lib().fdmi_colsum_f32(ptr(dy), ptr(x), ptr(stats), ptr(out0), ptr(out1), rows, Cc, HW, G, stream_ptr()))
    else:
        check(lib().fdmi_colsum(ptr(dy), ptr(x), ptr(stats), ptr(out0), ptr(out1), rows, Cc, HW, G, eps, stream_ptr()))


def pad_cols(x, cols_pad):
    rows, cols = x.shape
    out = torch.empty(rows, cols_pad, dtype=x.dtype, device=x.device)
    check((lib().fdmi_pad_cols_f32 if _f32(x) else lib().fdmi_pad_cols)(ptr(x), cols, ptr(out), cols_pad, rows, stream_ptr()))
    return out


# ---- adaLN-single DiT element-wise ops (csrc/dit.hip) --------------------------------------------------------------
def _mod_ld(v, Cc):
    """per-sample vector operand [B, C] (bf16; fp32 in validation mode), possibly a column-slice view (e.g. mod[:, i] of a
    [B, n, C] table)"""
    assert v.dtype in (BF16, F32) and v.dim() == 2 and v.shape[1] == Cc and v.stride(1) == 1
    return v.stride(0)


def layernorm_mod_fwd(x, shift, scale, rows_per_batch, eps, need_stats=False):
    """y = LayerNorm(x) * (1 + scale[b]) + shift[b]; x [rows, C] bf16 -> (y, stats [rows, 2] = (mean, rstd) or None)"""
    rows, Cc = x.shape
    ld = _mod_ld(scale, Cc)
    assert _mod_ld(shift, Cc) == ld
    y = torch.empty_like(x)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if need_stats else None
    if _f32(x):
        assert shift.dtype == F32 and scale.dtype == F32
        check(lib().fdmi_layernorm_fwd_f32(ptr(x), None, None, ptr(shift), ptr(scale), ld, rows_per_batch, ptr(y), ptr(stats), rows,
                                           Cc, eps, stream_ptr()))
        return y, stats
    check(lib().fdmi_layernorm_mod_fwd(ptr(x), ptr(shift), ptr(scale), ld, rows_per_batch, ptr(y), ptr(stats), rows, Cc,
                                       eps, stream_ptr()))
    return y, stats


def layernorm_mod_bwd(x, dy, scale, rows_per_batch, eps):
    rows, Cc = x.shape
    dx = torch.empty_like(x)
    if _f32(x):
        check(lib().fdmi_layernorm_bwd_f32(ptr(x), ptr(dy), None, ptr(scale), _mod_ld(scale, Cc), rows_per_batch, ptr(dx), rows, Cc,
                                           eps, 0, stream_ptr()))
        return dx
    check(lib().fdmi_layernorm_mod_bwd(ptr(x), ptr(dy), ptr(scale), _mod_ld(scale, Cc), rows_per_batch, ptr(dx), rows, Cc,
                                       eps, 0, stream_ptr()))
    return dx


def gate_residual(x, gate, res, rows_per_batch):
    """res + gate[b] * x (res None: gate[b] * x)"""
    rows, Cc = x.shape
    y = torch.empty_like(x)
    if _f32(x):
        check(lib().fdmi_gate_residual_f32(ptr(x), ptr(gate), _mod_ld(gate, Cc), ptr(res), ptr(y), rows, Cc, rows_per_batch,
                                           stream_ptr()))
        return y
    check(lib().fdmi_gate_residual(ptr(x), ptr(gate), _mod_ld(gate, Cc), ptr(res), ptr(y), rows, Cc, rows_per_batch,
                                   stream_ptr()))
    return y


def gelu_tanh(x):
    y = torch.empty_like(x)
    check((lib().fdmi_gelu_tanh_f32 if _f32(x) else lib().fdmi_gelu_tanh)(ptr(x), ptr(y), x.numel(), stream_ptr()))
    return y


def gelu_tanh_bwd(x, dy):
    dx = torch.empty_like(x)
    check((lib().fdmi_gelu_tanh_bwd_f32 if _f32(x) else lib().fdmi_gelu_tanh_bwd)(ptr(x), ptr(dy), ptr(dx), x.numel(), stream_ptr()))
    return dx


def batch_colsum(dy, x=None, stats=None, *, rows_per_batch, want_mul=True, want_sum=True):
    """per-sample column sums over each sample's rows: (sum_r dy * f(x), sum_r dy), each [B, C] f32 or None;
    f = LayerNorm normalisation with stats [rows, 2] = (mean, rstd), identity when stats is None"""
    rows, Cc = dy.shape
    B = rows // rows_per_batch
    o0 = torch.empty(B, Cc, dtype=torch.float32, device=dy.device) if want_mul else None
    o1 = torch.empty(B, Cc, dtype=torch.float32, device=dy.device) if want_sum else None
    fn = lib().fdmi_batch_colsum_f32 if _f32(dy) else lib().fdmi_batch_colsum
    check(fn(ptr(dy), ptr(x), ptr(stats), ptr(o0), ptr(o1), B, rows_per_batch, Cc, stream_ptr()))
    return o0, o1


def adamw_(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=1, grad_scale=1.0):
    check(lib().fdmi_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                           grad_scale, stream_ptr()))


def add_noise(z, noise, sa, sb):
    out = torch.empty_like(z)
    B = z.shape[0]
    check(lib().fdmi_add_noise(ptr(z), ptr(noise), ptr(sa), ptr(sb), ptr(out), B, z.numel() // B, stream_ptr()))
    return out


def axpby(x0, c0, x1=None, c1=0.0, x2=None, c2=0.0, x3=None, c3=0.0, out=None):
    if out is None:
        out = torch.empty_like(x0)
    check(lib().fdmi_axpby4(ptr(x0), c0, ptr(x1), c1, ptr(x2), c2, ptr(x3), c3, ptr(out), x0.numel(), stream_ptr()))
    return out


def attach_flat_grads(params, flat_grad):
    """param.grad <- views of the flat gradient buffer the plans' backward accumulates into.  A parameter whose .grad is None
    (optimizer.zero_grad(set_to_none=True), or never set) starts from zero: the WHOLE buffer in one launch when that holds for
    every parameter (the normal step), only that parameter's slice otherwise -- gradients other parameters already accumulated
    stay.  A .grad that is some other tensor is copied into its slice first, so accumulation semantics hold."""
    params = list(params)
    none = [p.grad is None for p in params]
    if params and all(none):
        flat_grad.zero_()
    off = 0
    for p, was_none in zip(params, none):
        v = flat_grad[off:off + p.numel()].view(p.shape)
        if was_none:
            if not all(none):
                v.zero_()
            p.grad = v
        elif p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        off += p.numel()
