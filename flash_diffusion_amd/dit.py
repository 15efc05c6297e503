"""MiTransformer2DModel -- drop-in for the reference's PixArt-alpha denoiser ``DiffusersTransformer2DWrapper``
(/root/reference/src/flash/models/transformers/tranformers.py:9-100, with its ``AdaLayerNormSingle``,
transformers/utils.py:8-102): same constructor keywords (examples/train_flash_pixart.py:63-86), same
forward signature / conditioning dict / ``freeze()``, parameters named by their diffusers state_dict keys.

Every operation on token-major activations ([B*T, C] bf16) is a hand-written HIP kernel of libfdmi.so reached through the
op-level C-ABI (include/fdmi.h): bf16 MFMA GEMMs with fused bias / residual / SiLU epilogues, flash attention forward and
backward (head dim 72), LayerNorm with the adaLN modulate fused (``fdmi_layernorm_mod_*``), the gated residual, tanh-GELU and
the per-sample column sums that give the gradients of the modulation vectors (csrc/dit.hip).  torch supplies device
memory, the stream, the autograd graph edges between those launches, the patch (un)folding of the 4-channel latent and
the glue on per-sample vectors ([B, C]: adding the scale-shift tables, chunk / cat).  There is no torch compute path for
the activations and no CPU fallback: without libfdmi.so every call raises.

On the GPU a forward is ONE C-ABI call into the library's plan of the denoiser (include/fdmi.h: fdmi_dit_create / _forward /
_backward -- csrc/dit_plan.h on the tape executor of the UNet plan): patch folding, the embedders, every block, the joint
[latent | text] sequence of the MMDiT and the whole backward run from C++ with activations bump-allocated from one workspace;
torch keeps ONE autograd edge per denoiser call (``_DitFn``).  ``FDMI_DIT_PLAN=0`` keeps the op-by-op composition below on the GPU
(the A/B switch; it is also what the CPU host-logic tests drive through ``tests/fake_ops.py``).

LoRA follows peft (examples/train_flash_pixart.py:237-256): y = W x + B(A x) on every module whose name ends in one of the
target suffixes -- linears and the patch-embedding convolution alike (a k = stride convolution IS a linear map on the
folded patches, so it runs as one)."""
from __future__ import annotations

import math
import weakref
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from . import ops

import ctypes as C
import os
import weakref

BF16 = torch.bfloat16
F32 = torch.float32
# precision: every activation tensor of a model is of ONE dtype -- bf16 (the measured MFMA path) or fp32 (the validation
# kernels of csrc/ref32.hip, `precision="fp32"`: the parity gate against the fp32 oracle).  ops.py dispatches on it.
PIXART_LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2", "proj",
                       "linear", "linear_1", "linear_2")      # examples/train_flash_pixart.py:240-253


# split-K of the projections: the context-stream GEMMs (a few hundred to a few thousand rows against K = 1152 ... 6144) fill a quarter
# of the chip with one tile wave, so for M <= SPLITK_ROWS the launcher's planner decides per problem (ops.gemm(splitk=0)); measured
# (profiles/r3_dit_ab.txt, call 9, same box back to back): splitting for every M: C5 -2.0 % but C4 +0.7 ... +1.2 % (its rank-64 GEMMs over 32768 rows split
# 2-way and pay the finalize); only for M <= 4096: C4 -0.65 %, C5 -1.8 % against never splitting -- the default
SPLITK_ROWS = int(__import__("os").environ.get("FDMI_DIT_SPLITK_ROWS", "4096"))


def _sk(x):
    return 0 if x.shape[0] <= SPLITK_ROWS else 1


def _pad8(n):
    return (n + 7) // 8 * 8


# ---- autograd edges around the HIP launches -------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    """y = x W^T + b [+ (x A^T) B^T] [+ residual]; W frozen, A / B (LoRA, fp32 masters) trainable."""

    @staticmethod
    def forward(ctx, x, residual, lin, out_f32, need_bwd, A, B):
        wb, wtb = lin.base16(x.dtype)
        M = x.shape[0]
        t = None
        if A is None:
            y = ops.gemm(x, wb, bias=lin.bias, residual=residual, out_f32=out_f32, splitk=_sk(x))
        else:
            ab, _, bb, _ = lin.lora16(x.dtype)
            t = ops.gemm(x, ab, splitk=_sk(x))
            y0 = ops.gemm(x, wb, bias=lin.bias, residual=residual, splitk=_sk(x))
            y = ops.gemm(t, bb, residual=y0, out_f32=out_f32)
        lin.count(2.0 * M * lin.out_features * lin.in_features
                  + (2.0 * M * lin.rank * (lin.in_features + lin.out_features) if A is not None else 0.0))
        ctx.lin, ctx.out_f32, ctx.lora = lin, out_f32, A is not None
        ctx.ashape = None if A is None else (A.shape, B.shape)
        if need_bwd:
            ctx.save_for_backward(x, t)
        return y

    @staticmethod
    def backward(ctx, dy):
        lin = ctx.lin
        x, t = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.out_f32 and x.dtype == BF16:
            dy = ops.f32_to_bf16(dy)
        M = x.shape[0]
        dx = dA = dB = None
        _, wtb = lin.base16(x.dtype)
        flops = 0.0
        if ctx.lora:
            _, abt, _, bbt = lin.lora16(x.dtype)
            u = ops.gemm(dy, bbt, splitk=_sk(x))                        # dL/d(A x)            [M, r]
            gv = lin._gviews           # views into the model's flat LoRA gradient (set by _reflatten_lora) or None
            # weight gradients by the TN kernel on the row-major operands (wgrad.hip): no transposed copies
            r = lin.rank
            dB = gv[1] if gv is not None else torch.zeros(lin.out_features, r, dtype=torch.float32, device=dy.device)
            dA = gv[0] if gv is not None else torch.zeros(r, x.shape[1], dtype=torch.float32, device=dy.device)
            ops.wgrad_tn(dy, t, dB)                                     # [N, r] += dy^T t
            ops.wgrad_tn(u, x, dA)                                      # [r, K] += u^T x
            if gv is not None:     # accumulated in place (fp32 atomics) where .grad already points: nothing for autograd to add
                dA = dB = None
            else:                  # (a patch-embedding conv keeps its LoRA tensors 4-D)
                dA, dB = dA.view(ctx.ashape[0]), dB.view(ctx.ashape[1])
            flops += 6.0 * M * lin.rank * lin.out_features + 2.0 * M * lin.rank * lin.in_features
        return _LinearFn._finish(ctx, lin, dy, u if ctx.lora else None, wtb, flops, dA, dB)

    @staticmethod
    def _finish(ctx, lin, dy, u, wtb, flops, dA, dB):
        """input gradient (base + LoRA path) and the tuple autograd expects"""
        dx = None
        M = dy.shape[0]
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy, wtb, splitk=_sk(dy))
            flops += 2.0 * M * lin.out_features * lin.in_features
            if ctx.lora:
                _, abt, _, _ = lin.lora16(dy.dtype)
                dx = ops.gemm(u, abt, residual=dx)
                flops += 2.0 * M * lin.rank * lin.in_features
        lin.count(flops)
        return dx, (dy if ctx.needs_input_grad[1] else None), None, None, None, dA, dB


class _LnModFn(torch.autograd.Function):
    """LayerNorm(x) * (1 + scale[b]) + shift[b] (non-affine LayerNorm: ada_norm_single)"""

    @staticmethod
    def forward(ctx, x, shift, scale, rpb, eps, need_bwd):
        y, stats = ops.layernorm_mod_fwd(x, shift, scale, rpb, eps,
                                         need_stats=need_bwd and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]))
        ctx.rpb, ctx.eps = rpb, eps
        if need_bwd:
            ctx.save_for_backward(x, scale, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, stats = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.layernorm_mod_bwd(x, dy, scale, ctx.rpb, ctx.eps) if ctx.needs_input_grad[0] else None
        dshift = dscale = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dscale, dshift = ops.batch_colsum(dy, x, stats, rows_per_batch=ctx.rpb)
            dscale, dshift = dscale.to(x.dtype), dshift.to(x.dtype)
        return dx, dshift, dscale, None, None, None


class _GateResFn(torch.autograd.Function):
    """res + gate[b] * x"""

    @staticmethod
    def forward(ctx, x, gate, res, rpb, need_bwd):
        ctx.rpb = rpb
        if need_bwd:
            ctx.save_for_backward(x, gate)
        return ops.gate_residual(x, gate, res, rpb)

    @staticmethod
    def backward(ctx, dy):
        x, gate = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gate_residual(dy, gate, None, ctx.rpb) if ctx.needs_input_grad[0] else None
        dgate = None
        if ctx.needs_input_grad[1]:
            dgate = ops.batch_colsum(dy, x, None, rows_per_batch=ctx.rpb, want_sum=False)[0].to(x.dtype)
        return dx, dgate, (dy if ctx.needs_input_grad[2] else None), None, None


class _SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, need_bwd):
        if need_bwd:
            ctx.save_for_backward(x)
        return ops.silu(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.silu_bwd(ctx.saved_tensors[0], dy.contiguous()), None


def _silu(x):
    x = x.contiguous()
    return _SiluFn.apply(x, torch.is_grad_enabled() and x.requires_grad)


def _linear_gelu(lin, x):
    """FeedForward's first projection + tanh-GELU (diffusers GELU(approximate="tanh"))"""
    if lin.fusable(x):
        return lin(x, act=ops.ACT_GELU_TANH)
    f = lin(x)
    return _GeluTanhFn.apply(f, torch.is_grad_enabled() and f.requires_grad)


def _linear_gate_res(lin, x, gate, res, rpb):
    """res + gate[sample] * lin(x): the adaLN gate + residual after an attention / feed-forward output projection"""
    if lin.fusable(x, gate, res):
        return lin(x, residual=res, gate=(gate, rpb))
    y = lin(x)
    return _GateResFn.apply(y, gate, res, rpb, torch.is_grad_enabled() and (y.requires_grad or gate.requires_grad or res.requires_grad))


class _GeluTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, need_bwd):
        if need_bwd:
            ctx.save_for_backward(x)
        return ops.gelu_tanh(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.gelu_tanh_bwd(ctx.saved_tensors[0], dy.contiguous()), None


class _AttnFn(torch.autograd.Function):
    """softmax(scale Q K^T) V per head; `lens` (host ints or None): per-sample number of valid keys (prefix mask)"""

    @staticmethod
    def forward(ctx, q, k, v, H, scale, lens, need_bwd, owner):
        B, Sq, Cc = q.shape
        Skv = k.shape[1]
        if lens is None:
            r = ops.attn_fwd(q, k, v, H, scale, need_lse=need_bwd)
            o, lse = r if need_bwd else (r, None)
            owner.count(4.0 * B * Sq * Skv * Cc)
        else:   # masked keys (T5 padding): one launch per sample on its valid prefix
            o = torch.empty_like(q)
            lse = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device) if need_bwd else None
            for b, n in enumerate(lens):
                ops.attn_fwd(q[b:b + 1], k[b:b + 1, :n], v[b:b + 1, :n], H, scale, need_lse=need_bwd, out=o[b:b + 1],
                             lse_out=None if lse is None else lse[b:b + 1])
                owner.count(4.0 * Sq * n * Cc)
        ctx.H, ctx.scale, ctx.lens, ctx.owner = H, scale, lens, owner
        if need_bwd:
            ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = do.contiguous()
        B, Sq, Cc = q.shape
        if ctx.lens is None:
            dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, ctx.H, ctx.scale)
            ctx.owner.count(8.0 * B * Sq * k.shape[1] * Cc)
        else:
            dq, dk, dv = torch.empty_like(q), torch.zeros_like(k), torch.zeros_like(v)
            for b, n in enumerate(ctx.lens):
                ops.attn_bwd(q[b:b + 1], k[b:b + 1, :n], v[b:b + 1, :n], o[b:b + 1], do[b:b + 1], lse[b:b + 1], ctx.H,
                             ctx.scale, out=(dq[b:b + 1], dk[b:b + 1, :n], dv[b:b + 1, :n]))
                ctx.owner.count(8.0 * Sq * n * Cc)
        return dq, dk, dv, None, None, None, None, None


# ---- modules ------------------------------------------------------------------------------------------------------------------
class MiLinear(nn.Module):
    """A frozen-or-plain linear map held as fp32 master weights + cached bf16 operand copies (W and W^T), with an optional
    peft-style LoRA pair.  ``weight`` may be 4-D (the patch-embedding convolution): it is used as [out, in*kh*kw]."""

    def __init__(self, out_features, in_features, bias=True, wshape=None):
        super().__init__()
        self.out_features, self.in_features = out_features, in_features
        self.weight = nn.Parameter(torch.randn(wshape or (out_features, in_features)) * in_features ** -0.5)
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        self.rank = 0
        self._b16 = None
        self._l16 = None
        self._gviews = None
        self._owner = [None]     # the model (list: not a sub-module) -- flop counter and LoRA epoch

    def count(self, flops):
        if self._owner[0] is not None:
            self._owner[0].count(flops)

    @staticmethod
    def _pair(w2d, dtype):
        """(W, W^T) GEMM operands: bf16 copies by the cast + transpose kernel; in fp32 validation mode the master itself and
        a transposed copy (pure data movement)"""
        if dtype == F32:
            return w2d.contiguous(), w2d.t().contiguous()
        return tuple(ops.cast_transpose(w2d))

    def base16(self, dtype=BF16):
        w = self.weight
        key = (w.data_ptr(), w._version, dtype)
        if self._b16 is None or self._b16[0] != key:
            assert ops._dev(w).dtype == torch.float32, "parameters must be fp32 on the GPU"
            self._b16 = (key,) + self._pair(w.detach().reshape(self.out_features, self.in_features), dtype)
        return self._b16[1], self._b16[2]

    def lora16(self, dtype=BF16):
        """(A, A^T, B, B^T) operands; re-made once per model forward (in-place optimizer kernels do not bump ._version)"""
        epoch = self._owner[0]._lora_epoch if self._owner[0] is not None else -1
        A, B = self.lora_A.default.weight, self.lora_B.default.weight
        key = (epoch, A.data_ptr(), B.data_ptr(), A._version, B._version, dtype)
        if self._l16 is None or self._l16[0] != key:
            ab, abt = self._pair(A.detach().reshape(self.rank, self.in_features), dtype)
            bb, bbt = self._pair(B.detach().reshape(self.out_features, self.rank), dtype)
            self._l16 = (key, ab, abt, bb, bbt)
        return self._l16[1:]

    def add_lora(self, r, init_std_b=0.0, generator=None):
        dev = self.weight.device
        ashape = (r,) + tuple(self.weight.shape[1:])
        bshape = (self.out_features, r) + (1,) * (self.weight.dim() - 2)
        a = torch.randn(ashape, generator=generator).to(dev) / r                     # peft "gaussian" init: N(0, 1/r)
        b = (torch.randn(bshape, generator=generator) * init_std_b).to(dev) if init_std_b else torch.zeros(bshape, device=dev)
        self.lora_A = nn.ModuleDict({"default": _Leaf(a)})
        self.lora_B = nn.ModuleDict({"default": _Leaf(b)})
        self.rank = r

    def fusable(self, *ts):
        """True when this call needs no autograd edge and carries no LoRA pair: activation / gate may ride in the GEMM epilogue"""
        return self.rank == 0 and not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts))

    def forward(self, x, residual=None, out_f32=False, act=0, gate=None):
        """act / gate (= (vector [B, out], rows per sample): y = (x W^T + b) * gate[sample] + residual) only where fusable():
        one launch instead of GEMM + element-wise pass (the frozen teacher's whole forward, the student's sampler)"""
        assert x.dtype in (BF16, F32) and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == self.in_features
        if act or gate is not None:
            assert self.fusable(x, residual, gate[0] if gate is not None else None)
            wb, _ = self.base16(x.dtype)
            self.count(2.0 * x.shape[0] * self.out_features * self.in_features)
            return ops.gemm(x, wb, bias=self.bias, residual=residual, out_f32=out_f32, act=act,
                            rowvec=gate[0] if gate is not None else None, rows_per_batch=gate[1] if gate is not None else 1,
                            rowvec_mul=gate is not None, splitk=_sk(x))
        need_bwd = torch.is_grad_enabled() and (x.requires_grad or self.rank > 0
                                                or (residual is not None and residual.requires_grad))
        A = self.lora_A.default.weight if self.rank else None
        B = self.lora_B.default.weight if self.rank else None
        return _LinearFn.apply(x, residual, self, out_f32, need_bwd, A, B)


class _Leaf(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.weight = nn.Parameter(w)


class _Node(nn.Module):
    """plain container so parameters get their dotted diffusers names"""


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = MiLinear(dim, in_dim)
        self.linear_2 = MiLinear(dim, dim)

    def forward(self, x):
        return self.linear_2(_silu(self.linear_1(x)))


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim, bias):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        kv = cross_dim if cross_dim is not None else dim
        self.to_q = MiLinear(inner, dim, bias)
        self.to_k = MiLinear(inner, kv, bias)
        self.to_v = MiLinear(inner, kv, bias)
        self.to_out = nn.ModuleList([MiLinear(dim, inner, True)])


class _FF(nn.Module):
    def __init__(self, dim):
        super().__init__()
        n0 = _Node()
        n0.proj = MiLinear(4 * dim, dim)
        self.net = nn.ModuleList([n0, nn.Identity(), MiLinear(dim, 4 * dim)])


class _Block(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim, bias):
        super().__init__()
        self.attn1 = _Attention(dim, heads, dim_head, None, bias)
        self.attn2 = _Attention(dim, heads, dim_head, cross_dim, bias)
        self.ff = _FF(dim)
        self.scale_shift_table = nn.Parameter(torch.randn(6, dim) / dim ** 0.5)


def _sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed(dim, gh, gw, base_size, interpolation_scale):
    """diffusers get_2d_sincos_pos_embed: first half of the channels from the column coordinate, second from the row"""
    ys = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    xs = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(xs, ys), axis=0).reshape([2, 1, gw, gh])
    return np.concatenate([_sincos_1d(dim // 2, grid[0]), _sincos_1d(dim // 2, grid[1])], axis=1)


class _DenoiserBase(nn.Module):
    """bookkeeping shared by the transformer denoisers: flop counter, LoRA (peft semantics), freeze, deepcopy"""
    per_sample = True   # no layer mixes samples: callers may batch [cond | uncond] into one call (flash_sd3._euler_cfg)

    def _init_base(self):
        self.lora_r = 0
        self._lora_flat = self._lora_grad = None
        self._lora_epoch = 0
        self._pos_cache = {}
        self.last_flops = 0.0
        self.step_flops = 0.0   # running sum of algorithmic MFMA flops (bench.py resets it)
        self.plan_calls = 0     # forwards that ran as ONE fdmi_dit_forward (the tests assert the plan is the path that ran)
        for m in self.modules():
            if isinstance(m, MiLinear):
                m._owner[0] = self

    def count(self, flops):
        self.last_flops += flops
        self.step_flops += flops

    def __deepcopy__(self, memo):      # student = deepcopy(teacher) (examples/train_flash_pixart.py:174): re-point owners
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_pos_cache" else copy.deepcopy(v, memo)
        for m in new.modules():
            if isinstance(m, MiLinear):
                m._owner[0] = new
                m._b16 = m._l16 = m._gviews = None
        new._lora_flat = new._lora_grad = None
        return new

    def freeze(self):                                                                               # TW:94-100
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def add_adapter(self, r: int, target_modules: Sequence[str] = PIXART_LORA_TARGETS, init_std_b: float = 0.0,
                    generator: Optional[torch.Generator] = None):
        """peft ``get_peft_model(LoraConfig(r, target_modules))`` semantics: every base parameter frozen; A ~ N(0, 1/r),
        B = 0 (init_std_b > 0 only for tests) on each module whose dotted name ends in a target suffix."""
        assert self.lora_r == 0, "adapter already added"
        for p in self.parameters():
            p.requires_grad = False
        n = 0
        for name, m in self.named_modules():
            if isinstance(m, MiLinear) and any(name == t or name.endswith("." + t) for t in target_modules):
                m.add_lora(r, init_std_b, generator)
                n += 1
        assert n > 0, "no module matched target_modules"
        self.lora_r = r
        return self

    def lora_parameters(self):
        return [p for n, p in self.named_parameters() if ".lora_" in n]

    # ---- flat LoRA storage: ONE fused AdamW launch and ONE all-reduce per step (trainer.py) -----------------------
    @property
    def lora_rank(self):
        return self.lora_r

    def _reflatten_lora(self, device, attach=True):
        """(Re)establish the invariant that every LoRA tensor is a view into one flat fp32 buffer and every LoRA ``.grad`` a
        view into one flat gradient buffer which the backward GEMMs accumulate into directly (after .to(device) / deepcopy /
        load_state_dict).  Returns True when nothing had to move.  attach=False (the plan path, at every forward) only makes
        sure the two flat buffers exist: ``.grad`` is attached by ``_attach_lora_grads`` at backward time."""
        named = [(n, p) for n, p in self.named_parameters() if ".lora_" in n]
        total = sum(p.numel() for _, p in named)
        flat = self._lora_flat
        ok = flat is not None and flat.device == device and flat.numel() == total
        if ok:
            off = 0
            for _, p in named:
                if p.data_ptr() != flat.data_ptr() + off * 4:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.empty(total, dtype=torch.float32, device=device)
            off = 0
            for _, p in named:
                flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            self._lora_flat = flat
            self._lora_grad = None
        g = self._lora_grad
        if g is None or g.device != device:
            g = self._lora_grad = torch.zeros(total, dtype=torch.float32, device=device)
        if not attach:
            return ok
        off = 0
        views = {}
        for n, p in named:
            v = g[off:off + p.numel()]
            p.grad = v.view(p.shape)
            mod, which = n.split(".lora_")[0], n.split(".lora_")[1][0]
            lin = self.get_submodule(mod)
            views.setdefault(mod, [None, None])[0 if which == "A" else 1] = \
                v.view(lin.rank, lin.in_features) if which == "A" else v.view(lin.out_features, lin.rank)
            off += p.numel()
        for mod, (ga, gb) in views.items():
            self.get_submodule(mod)._gviews = (ga, gb)
        return ok

    def lora_flat(self):
        return self._lora_flat

    def lora_flat_grad(self):
        return self._lora_grad

    # ---- the C++ plan of this denoiser (include/fdmi.h fdmi_dit_*; same slot / workspace protocol as unet.py's _Plan) --------
    _PLANS: Dict[int, object] = {}

    def _use_plan(self, sample):
        return sample.is_cuda and os.environ.get("FDMI_DIT_PLAN", "1") == "1"

    def _plan(self):
        p = _DenoiserBase._PLANS.get(id(self))
        if p is None:
            from .unet import _Plan
            p = _Plan(self._dit_cfg(), create="fdmi_dit_create")
            _DenoiserBase._PLANS[id(self)] = p
            weakref.finalize(self, _DenoiserBase._drop_plan, id(self))
        return p

    @staticmethod
    def _drop_plan(key):
        p = _DenoiserBase._PLANS.pop(key, None)
        if p is not None:
            p.close()

    def invalidate_plan(self):
        """Call after changing frozen base weights in place (load_state_dict does it for you)."""
        p = _DenoiserBase._PLANS.get(id(self))
        if p is not None:
            p.packed = False

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_plan()
        return r

    def _lora_modules(self):
        return [(n, m) for n, m in self.named_modules() if isinstance(m, MiLinear) and m.rank]

    def _ensure_packed(self, device):
        from ._lib import check, i64, lib, ptr, stream_ptr
        plan = self._plan()
        L = lib()
        if not plan.packed:
            n = L.fdmi_unet_num_params(plan.handle)
            expected = {}
            buf = C.create_string_buffer(512)
            ne = i64()
            for i in range(n):
                check(L.fdmi_unet_param_name(plan.handle, i, buf, 512, C.byref(ne)))
                expected[buf.value.decode()] = ne.value
            mine = {k: v for k, v in self.named_parameters() if ".lora_" not in k}
            assert set(expected) == set(mine), (set(expected) ^ set(mine))
            for name, p in mine.items():
                assert p.is_cuda and p.dtype == torch.float32, f"{name}: parameters must be fp32 on the GPU"
                t = p.detach().contiguous()
                check(L.fdmi_unet_set_param(plan.handle, name.encode(), ptr(t), t.numel(), stream_ptr()))
            torch.cuda.current_stream().synchronize()
            check(L.fdmi_unet_ready(plan.handle))
            plan.packed = True
        if self.lora_r:
            same = self._reflatten_lora(device, attach=False)
            key = (self._lora_flat.data_ptr(), self._lora_grad.data_ptr())
            if not (same and plan.lora_bound == key):
                g, off = self._lora_grad, 0
                grads = {}
                for pn, p in self.named_parameters():      # (the flat buffers hold the LoRA tensors in named_parameters order)
                    if ".lora_" in pn:
                        grads[pn] = g[off:off + p.numel()]
                        off += p.numel()
                for name, m in self._lora_modules():
                    a, b = m.lora_A.default.weight, m.lora_B.default.weight
                    check(L.fdmi_unet_set_lora(plan.handle, name.encode(), ptr(a), ptr(b), ptr(grads[name + ".lora_A.default.weight"]),
                                               ptr(grads[name + ".lora_B.default.weight"]), m.rank))
                plan.lora_bound = key
        return plan

    def _attach_lora_grads(self):
        """param.grad <- views of the flat gradient buffer the plan's backward accumulates into (ops.attach_flat_grads: zeroed
        where the grads were None: optimizer.zero_grad(set_to_none=True))"""
        ops.attach_flat_grads([p for n, p in self.named_parameters() if ".lora_" in n], self._lora_grad)

    def _plan_call(self, sample, timestep, ctx, vector, pos, lens, keep):
        """one denoiser call through the plan: sample [B, C, H, W], timestep [B], ctx [B, L, .], vector [B, .] or None, pos
        [T, D] fp32, lens = per-sample valid keys (host ints) or None -> [B, keep, H, W] fp32"""
        B = sample.shape[0]
        dev = sample.device
        if not torch.is_tensor(timestep):
            timestep = torch.full((B,), float(timestep), device=dev)
        t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        lora = self.lora_parameters() if self.lora_r else []
        need_grad = torch.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in lora))
        x = sample.float().contiguous()
        cx = ctx.float().contiguous()
        vec = vector.float().contiguous() if vector is not None else None
        args = (x, t, cx, vec, pos, tuple(lens) if lens is not None else None, keep)
        if need_grad:
            return _DitFn.apply(self, args, *([x] + lora))
        out, _ = self._run_plan(args, 0)
        return out

    def _run_plan(self, args, flags):
        from ._lib import check, lib, ptr, stream_ptr
        from .unet import FDMI_UNET_INPUT_GRAD, FDMI_UNET_SAVE
        x, t, cx, vec, pos, lens, keep = args
        plan = self._ensure_packed(x.device)
        L = lib()
        B, _, H, W = x.shape
        Lc = cx.shape[1]
        save = bool(flags & FDMI_UNET_SAVE)
        if save:
            free = [s for s in range(1, 8) if s not in plan.busy]
            if not free:
                raise RuntimeError("fdmi: seven denoiser calls with saved activations are outstanding on this model "
                                   "(call backward or release_saved())")
            slot = free[0]
            plan.busy.add(slot)
            plan.gen[slot] = plan.gen.get(slot, 0) + 1
            qflags = flags | (FDMI_UNET_INPUT_GRAD if x.requires_grad else 0)
        else:
            slot, qflags = 0, flags
        need = L.fdmi_dit_workspace_bytes(plan.handle, B, H, W, Lc, int(lens is not None), qflags)
        if need < 0:
            raise RuntimeError("fdmi: " + L.fdmi_last_error().decode())
        ws = plan.workspaces.get(slot)
        if ws is None or ws.numel() < need or ws.device != x.device:
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            plan.workspaces[slot] = ws
        out = torch.empty(B, keep, H, W, dtype=torch.float32, device=x.device)
        kl = (C.c_int32 * B)(*lens) if lens is not None else None
        plan.enter(slot)
        try:
            check(L.fdmi_dit_forward(plan.handle, slot, ptr(x), ptr(t), ptr(cx), ptr(vec), ptr(pos), kl, ptr(out), B, H, W, Lc, keep,
                                     ptr(ws), ws.numel(), flags, stream_ptr()))
        finally:
            plan.leave(slot)
        self.last_flops = L.fdmi_unet_last_flops(plan.handle)
        self.step_flops += self.last_flops
        self.plan_calls += 1
        return out, slot

    def _run_plan_backward(self, slot, grad_out, needs_x, xshape, gen):
        from ._lib import check, lib, ptr, stream_ptr
        plan = self._plan()
        L = lib()
        if self.lora_r:
            self._attach_lora_grads()
        g = grad_out.float().contiguous()
        gx = torch.empty(xshape, dtype=torch.float32, device=g.device) if needs_x else None
        plan.enter(slot)
        try:
            check(L.fdmi_dit_backward(plan.handle, slot, ptr(g), ptr(gx), stream_ptr()))
        finally:
            plan.leave(slot)
            plan.release(slot, gen)
        self.last_flops = L.fdmi_unet_last_flops(plan.handle)
        self.step_flops += self.last_flops
        return gx

    def release_saved(self):
        """Drop saved-for-backward runs that will never be back-propagated (e.g. after an exception)."""
        self._plan().busy.clear()

    teacher_loop_takes_mask = True

    @torch.no_grad()
    def teacher_loop(self, x, timesteps, crossattn2, vector2, coeffs, attention_mask=None, attention_mask_lens=None):
        """The frozen teacher's whole guidance loop in ONE C-ABI call (include/fdmi.h: fdmi_dit_teacher_loop): x [B, C, H, W] is
        advanced through the steps `timesteps` (host floats) with the host coefficient rows `coeffs` ([n][6]: x0 = a0 x + a1 e_c
        + a2 e_u; x = a3 x + a4 x0 + a5 x0_prev); crossattn2 [2B, L, .] / vector2 [2B, .] / attention_mask [2B, L] hold the
        conditional rows first, then the unconditional ones.  Returns the new latent."""
        from ._lib import check, lib, ptr, stream_ptr
        assert x.is_cuda and not self.lora_r, "teacher_loop is for a frozen denoiser on the GPU"
        c = self.config_dict
        B, Cx, H, W = x.shape
        assert Cx == c["in_channels"] and crossattn2.shape[0] == 2 * B and len(coeffs) == len(timesteps)
        plan = self._ensure_packed(x.device)
        L = lib()
        n, Lc, p = len(timesteps), crossattn2.shape[1], c["patch_size"]
        lens = self._key_lens(attention_mask, Lc, attention_mask_lens) if attention_mask is not None else None
        x = x.float().contiguous().clone()
        enc = crossattn2.float().contiguous()
        vec = vector2.float().contiguous() if vector2 is not None else None
        pos = self._pos32(H // p, W // p, x.device)
        need = L.fdmi_dit_workspace_bytes(plan.handle, 2 * B, H, W, Lc, int(lens is not None), 0)
        sneed = L.fdmi_dit_teacher_loop_scratch_bytes(plan.handle, B, H, W)
        if need < 0 or sneed < 0:
            raise RuntimeError("fdmi: " + L.fdmi_last_error().decode())
        ws = plan.workspaces.get(0)
        if ws is None or ws.numel() < need or ws.device != x.device:
            ws = plan.workspaces[0] = torch.empty(need, dtype=torch.uint8, device=x.device)
        sc = plan.workspaces.get("teacher_loop")
        if sc is None or sc.numel() < sneed or sc.device != x.device:
            sc = plan.workspaces["teacher_loop"] = torch.empty(sneed, dtype=torch.uint8, device=x.device)
        ts = (C.c_float * n)(*[float(t) for t in timesteps])
        cf = (C.c_float * (6 * n))(*[float(v) for r in coeffs for v in r])
        kl = (C.c_int32 * (2 * B))(*lens) if lens is not None else None
        plan.enter(0)
        try:
            check(L.fdmi_dit_teacher_loop(plan.handle, 0, ptr(x), ts, n, ptr(enc), ptr(vec), ptr(pos), kl, cf, B, H, W, Lc, ptr(ws),
                                          ws.numel(), ptr(sc), sc.numel(), stream_ptr()))
        finally:
            plan.leave(0)
        self.last_flops = L.fdmi_unet_last_flops(plan.handle)
        self.step_flops += self.last_flops
        self.plan_calls += n
        return x


class _DitFn(torch.autograd.Function):
    """the ONE autograd edge of a denoiser call through the plan (as unet._UNetFn): x = the fp32 sample, *lora = the student's
    LoRA tensors (their gradients are accumulated in place by the plan's backward: nothing is returned for them)"""

    @staticmethod
    def forward(ctx, mod, args, x, *lora):
        from .unet import FDMI_UNET_SAVE
        out, slot = mod._run_plan(args, FDMI_UNET_SAVE)
        plan = mod._plan()
        ctx.mod, ctx.slot, ctx.gen = mod, slot, plan.gen[slot]
        weakref.finalize(ctx, plan.release, slot, ctx.gen)   # a graph dropped without backward gives its slot back
        ctx.needs_x = x.requires_grad
        ctx.nlora = len(lora)
        ctx.xshape = x.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.mod._plan().gen.get(ctx.slot) != ctx.gen:
            raise RuntimeError("fdmi: backward of a denoiser call whose saved activations were released (slot reused)")
        gx = ctx.mod._run_plan_backward(ctx.slot, grad_out, ctx.needs_x, ctx.xshape, ctx.gen)
        return (None, None, gx) + (None,) * ctx.nlora


class MiTransformer2DModel(_DenoiserBase):
    """See module docstring.  diffusers keywords outside the PixArt-alpha configuration raise."""

    def __init__(self, time_embed_dim=256, timesteps_embedding_num_channels=256, projection_class_embeddings_input_dim=None,
                 use_concat_vector_conditioning=False, num_vector_conditionings=None, num_attention_heads=16,
                 attention_head_dim=88, in_channels=None, out_channels=None, num_layers=1, cross_attention_dim=None,
                 attention_bias=False, sample_size=None, patch_size=None, activation_fn="geglu", num_embeds_ada_norm=None,
                 norm_type="layer_norm", norm_elementwise_affine=True, norm_eps=1e-5, caption_channels=None,
                 interpolation_scale=None, precision="bf16", **unused):
        super().__init__()
        assert precision in ("bf16", "fp32")
        self.dt = F32 if precision == "fp32" else BF16
        if not (patch_size is not None and norm_type == "ada_norm_single" and activation_fn == "gelu-approximate"
                and not norm_elementwise_affine):
            raise NotImplementedError("only the PixArt-alpha configuration (patched input, ada_norm_single, gelu-approximate, "
                                      "non-affine norms: examples/train_flash_pixart.py:63-86) is built")
        for k, v in unused.items():
            if v not in (None, False, 0, 0.0, "default"):
                raise NotImplementedError(f"{k}={v!r} is outside the reference's configurations")
        D = num_attention_heads * attention_head_dim
        assert time_embed_dim == D, "adaLN-single feeds the blocks: time_embed_dim must equal heads * head_dim"
        assert attention_head_dim % 8 == 0 and attention_head_dim <= 160 and D % 8 == 0
        out_channels = in_channels if out_channels is None else out_channels
        self.config_dict = dict(sample_size=sample_size, patch_size=patch_size, in_channels=in_channels,
                                out_channels=out_channels, num_layers=num_layers, heads=num_attention_heads,
                                head_dim=attention_head_dim, inner_dim=D, cross_attention_dim=cross_attention_dim,
                                caption_channels=caption_channels, norm_eps=norm_eps, tdim=timesteps_embedding_num_channels,
                                attention_bias=bool(attention_bias),
                                interpolation_scale=interpolation_scale if interpolation_scale is not None
                                else max(sample_size // 64, 1))
        p = patch_size
        self.pos_embed = _Node()
        self.pos_embed.proj = MiLinear(D, in_channels * p * p, True, wshape=(D, in_channels, p, p))
        ada = _Node()
        ada.timestep_embedder = _TimestepEmbedding(timesteps_embedding_num_channels, time_embed_dim)
        self.vdim, self.n_vec = projection_class_embeddings_input_dim, num_vector_conditionings
        if self.vdim is not None:                                                                 # TU:51-70
            if not use_concat_vector_conditioning:
                ada.add_embedding = _TimestepEmbedding(self.vdim, time_embed_dim)
            else:
                assert num_vector_conditionings is not None, \
                    "num_vector_conditionings must be provided if use_concat_conditioning is True"
                ada.add_embedding = nn.ModuleList([_TimestepEmbedding(self.vdim, time_embed_dim // num_vector_conditionings)
                                                   for _ in range(num_vector_conditionings)])
        ada.linear = MiLinear(6 * time_embed_dim, time_embed_dim)
        self.adaln_single = ada
        if caption_channels is not None:
            self.caption_projection = _Node()
            self.caption_projection.linear_1 = MiLinear(D, caption_channels)
            self.caption_projection.linear_2 = MiLinear(D, D)
        else:
            self.caption_projection = None
        self.transformer_blocks = nn.ModuleList([_Block(D, num_attention_heads, attention_head_dim, cross_attention_dim,
                                                        attention_bias) for _ in range(num_layers)])
        self.scale_shift_table = nn.Parameter(torch.randn(2, D) / D ** 0.5)
        self.proj_out = MiLinear(p * p * out_channels, D)
        self._mask_cache = None
        self._init_base()

    def _dit_cfg(self):
        from ._lib import DitCfg
        c, s = self.config_dict, DitCfg()
        s.kind, s.in_channels, s.out_channels, s.patch_size = 1, c["in_channels"], c["out_channels"], c["patch_size"]
        s.num_layers, s.heads, s.head_dim = c["num_layers"], c["heads"], c["head_dim"]
        s.cross_dim, s.caption_channels, s.tdim = c["cross_attention_dim"], c["caption_channels"] or 0, c["tdim"]
        s.vec_dim = self.vdim or 0
        s.n_vec = self.n_vec if (self.vdim is not None and isinstance(self.adaln_single.add_embedding, nn.ModuleList)) else 0
        s.attention_bias, s.norm_eps, s.precision = int(c["attention_bias"]), c["norm_eps"], int(self.dt == F32)
        return s

    def _pos32(self, h, w, device):
        key = ("f32", h, w, str(device))
        if key not in self._pos_cache:
            c = self.config_dict
            pe = sincos_pos_embed(c["inner_dim"], h, w, c["sample_size"] // c["patch_size"], c["interpolation_scale"])
            self._pos_cache[key] = torch.from_numpy(pe).float().to(device).contiguous()
        return self._pos_cache[key]

    # ---- forward ------------------------------------------------------------------------------------------------------
    def _pos(self, h, w, B, device):
        key = (h, w, B, str(device))
        if key not in self._pos_cache:
            c = self.config_dict
            base = c["sample_size"] // c["patch_size"]
            pe = sincos_pos_embed(c["inner_dim"], h, w, base, c["interpolation_scale"])
            pe = torch.from_numpy(pe).float().to(device).to(self.dt)
            self._pos_cache[key] = pe.unsqueeze(0).expand(B, -1, -1).reshape(B * h * w, -1).contiguous()
        return self._pos_cache[key]

    def _key_lens(self, mask, L, host_lens=None) -> Optional[List[int]]:
        """per-sample key prefix lengths of a padding mask (TW:75-77), or None when nothing is masked.  `host_lens` (the optional
        conditioning entry "attention_mask_lens": host integers the conditioner already knows from its tokenizer) avoids the
        device -> host read.  Otherwise the mask is read on EVERY call except when it is the very same tensor object, unmodified,
        as in the previous call (held by weak reference; ADVICE r4: an address is not an identity -- the caching allocator hands
        a freed block to the next batch's mask -- so a (data_ptr, version, shape) key could reuse another batch's lengths)."""
        if mask is None:
            return None
        if host_lens is not None:
            lens = [int(n) for n in host_lens]
            if len(lens) != mask.shape[0] or any(n < 1 or n > L for n in lens):
                raise ValueError(f"attention_mask_lens {lens} does not describe a [{mask.shape[0]}, {L}] prefix mask")
            return None if all(n == L for n in lens) else lens
        c = self._mask_cache
        if c is not None and c[0]() is mask and c[1] == mask._version:
            return c[2]
        m = (mask.detach().to("cpu") != 0)
        lens = m.sum(dim=1).tolist()
        for b, n in enumerate(lens):
            if n == 0 or not bool(m[b, :n].all()):
                raise NotImplementedError("attention_mask must keep a non-empty prefix of the keys (tokenizer padding)")
        lens = None if all(n == L for n in lens) else lens
        self._mask_cache = (weakref.ref(mask), mask._version, lens)
        return lens

    def _attn(self, a: _Attention, xq, xkv, B, Sq, Skv, lens, residual=None, gate=None):
        q = a.to_q(xq).view(B, Sq, -1)
        k = a.to_k(xkv).view(B, Skv, -1)
        v = a.to_v(xkv).view(B, Skv, -1)
        nb = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
        o = _AttnFn.apply(q, k, v, a.heads, a.scale, lens, nb, self).view(B * Sq, -1)
        if gate is not None:
            return _linear_gate_res(a.to_out[0], o, gate, residual, Sq)
        return a.to_out[0](o, residual=residual)

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                conditioning: Dict[str, Dict[str, torch.Tensor]], hidden_states_masks: Optional[torch.Tensor] = None,
                *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"                 # TW:69
        cnd = conditioning["cond"]
        vector, crossattn, concat = cnd.get("vector", None), cnd.get("crossattn", None), cnd.get("concat", None)
        mask, mask_lens = cnd.get("attention_mask", None), cnd.get("attention_mask_lens", None)
        ops._dev(sample)      # device tensors only: there is no CPU fallback
        c = self.config_dict
        C_in = sample.shape[1]
        if concat is not None:                                                                     # TW:79-80
            sample = torch.cat([sample, concat], dim=1)
        assert sample.shape[1] == c["in_channels"]
        self.last_flops = 0.0
        if self.lora_r:
            self._lora_epoch += 1
        grad = torch.is_grad_enabled()
        B, _, Hh, Ww = sample.shape
        p, D = c["patch_size"], c["inner_dim"]
        h, w = Hh // p, Ww // p
        T = h * w
        dev = sample.device
        if self._use_plan(sample):      # ONE call into the library's plan of this denoiser (csrc/dit_plan.h)
            assert C_in <= c["out_channels"]
            return self._plan_call(sample, timestep, crossattn, vector, self._pos32(h, w, dev),
                                   self._key_lens(mask, crossattn.shape[1], mask_lens), C_in)

        # adaLN-single (TU:75-102): per-sample vectors [B, .]
        if not torch.is_tensor(timestep):
            timestep = torch.full((B,), float(timestep), device=dev)
        ada = self.adaln_single
        emb = ada.timestep_embedder(ops.timestep_embed(timestep.reshape(-1).to(dev), c["tdim"], True, 0.0, self.dt))
        if self.vdim is not None:
            vb = vector.to(self.dt)
            if isinstance(ada.add_embedding, nn.ModuleList):
                chunks = torch.chunk(vb, self.n_vec, dim=1)
                emb = emb + torch.cat([ada.add_embedding[i](chunks[i].contiguous())
                                       for i in range(len(ada.add_embedding))], dim=1)
            else:
                emb = emb + ada.add_embedding(vb.contiguous())
        mod6 = ada.linear(_silu(emb)).float().view(B, 6, D)

        # patch embedding: fold p x p patches (c, py, px order = the convolution weight's), one GEMM + positions
        patches = sample.float().reshape(B, c["in_channels"], h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B * T, -1)
        hid = self.pos_embed.proj(patches.to(self.dt).contiguous(), residual=self._pos(h, w, B, dev))

        # caption projection
        L = crossattn.shape[1]
        ctx = crossattn.to(self.dt).reshape(B * L, -1).contiguous()
        if self.caption_projection is not None:
            ctx = self.caption_projection.linear_2(_linear_gelu(self.caption_projection.linear_1, ctx))
        lens = self._key_lens(mask, L, mask_lens)

        eps = c["norm_eps"]
        for blk in self.transformer_blocks:
            mod = (blk.scale_shift_table[None] + mod6).to(self.dt)     # shift_msa, scale_msa, gate_msa, shift_mlp, ...
            nb = grad and (hid.requires_grad or mod.requires_grad)
            n1 = _LnModFn.apply(hid, mod[:, 0], mod[:, 1], T, eps, nb)
            hid = self._attn(blk.attn1, n1, n1, B, T, T, None, residual=hid, gate=mod[:, 2])
            hid = self._attn(blk.attn2, hid, ctx, B, T, L, lens, residual=hid)
            nb = grad and (hid.requires_grad or mod.requires_grad)
            n2 = _LnModFn.apply(hid, mod[:, 3], mod[:, 4], T, eps, nb)
            hid = _linear_gate_res(blk.ff.net[2], _linear_gelu(blk.ff.net[0].proj, n2), mod[:, 5], hid, T)

        fin = (self.scale_shift_table[None] + emb.float()[:, None]).to(self.dt)                    # shift, scale
        n = _LnModFn.apply(hid, fin[:, 0], fin[:, 1], T, 1e-6, grad and (hid.requires_grad or fin.requires_grad))
        y = self.proj_out(n, out_f32=True)                                                       # [B*T, p*p*out]
        oc = c["out_channels"]
        y = torch.einsum("nhwpqc->nchpwq", y.view(B, h, w, p, p, oc)).reshape(B, oc, h * p, w * p)
        return y[:, :C_in]                                                                         # TW:91


# ======================================================================================================================
# SD3 MMDiT: drop-in for DiffusersSD3Transformer2DWrapper (tranformers.py:103-163)
# ======================================================================================================================
class _AdaNorm(nn.Module):
    """AdaLayerNormZero (6 vectors) / AdaLayerNormContinuous (2 vectors: scale, shift): only the projection has weights"""

    def __init__(self, dim, n):
        super().__init__()
        self.linear = MiLinear(n * dim, dim)
        self.n = n

    def forward(self, silu_temb):
        B = silu_temb.shape[0]
        return self.linear(silu_temb).view(B, self.n, -1)


class _JointAttention(nn.Module):
    def __init__(self, dim, heads, context_pre_only):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.to_q, self.to_k, self.to_v = MiLinear(dim, dim), MiLinear(dim, dim), MiLinear(dim, dim)
        self.add_k_proj, self.add_v_proj, self.add_q_proj = MiLinear(dim, dim), MiLinear(dim, dim), MiLinear(dim, dim)
        self.to_out = nn.ModuleList([MiLinear(dim, dim)])
        if not context_pre_only:
            self.to_add_out = MiLinear(dim, dim)


class _JointBlock(nn.Module):
    def __init__(self, dim, heads, context_pre_only):
        super().__init__()
        self.context_pre_only = context_pre_only
        self.norm1 = _AdaNorm(dim, 6)
        self.norm1_context = _AdaNorm(dim, 2 if context_pre_only else 6)
        self.attn = _JointAttention(dim, heads, context_pre_only)
        self.ff = _FF(dim)
        if not context_pre_only:
            self.ff_context = _FF(dim)


class _TextProjection(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = MiLinear(dim, in_dim)
        self.linear_2 = MiLinear(dim, dim)

    def forward(self, x):
        return self.linear_2(_silu(self.linear_1(x)))


class MiSD3Transformer2DModel(_DenoiserBase):
    """Drop-in for the reference's SD3 denoiser ``DiffusersSD3Transformer2DWrapper`` (tranformers.py:103-163; constructor
    keywords of examples/train_flash_sd3.py:65-77, SD3-medium: no qk-norm), parameters and the ``pos_embed.pos_embed``
    buffer under diffusers' state_dict keys (``load_state_dict(pipe.transformer.state_dict(), strict=True)`` works).
    Same construction as MiTransformer2DModel: token-major activations only ever touched by HIP launches (GEMM, flash
    attention over the concatenated [latent | text] tokens, LayerNorm+modulate, gated residual, tanh-GELU); torch holds
    the autograd edges, concatenates / splits the joint sequence and does the per-sample-vector glue."""

    def __init__(self, sample_size=128, patch_size=2, in_channels=16, num_layers=18, attention_head_dim=64,
                 num_attention_heads=18, joint_attention_dim=4096, caption_projection_dim=1152, pooled_projection_dim=2048,
                 out_channels=16, pos_embed_max_size=96, precision="bf16", **unused):
        super().__init__()
        assert precision in ("bf16", "fp32")
        self.dt = F32 if precision == "fp32" else BF16
        for k, v in unused.items():
            if v not in (None, False, 0, 0.0, "default"):
                raise NotImplementedError(f"{k}={v!r} is outside the reference's configurations")
        D = num_attention_heads * attention_head_dim
        assert caption_projection_dim == D, "joint attention needs caption_projection_dim == heads * head_dim"
        assert attention_head_dim % 8 == 0 and attention_head_dim <= 160
        p = patch_size
        self.config_dict = dict(sample_size=sample_size, patch_size=p, in_channels=in_channels, out_channels=out_channels,
                                num_layers=num_layers, heads=num_attention_heads, head_dim=attention_head_dim, inner_dim=D,
                                joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
                                pos_embed_max_size=pos_embed_max_size)
        self.pos_embed = _Node()
        self.pos_embed.proj = MiLinear(D, in_channels * p * p, True, wshape=(D, in_channels, p, p))
        pe = sincos_pos_embed(D, pos_embed_max_size, pos_embed_max_size, sample_size // p, 1)
        self.pos_embed.register_buffer("pos_embed", torch.from_numpy(pe).float().unsqueeze(0), persistent=True)
        tte = _Node()
        tte.timestep_embedder = _TimestepEmbedding(256, D)
        tte.text_embedder = _TextProjection(pooled_projection_dim, D)
        self.time_text_embed = tte
        self.context_embedder = MiLinear(D, joint_attention_dim)
        self.transformer_blocks = nn.ModuleList([_JointBlock(D, num_attention_heads, i == num_layers - 1)
                                                 for i in range(num_layers)])
        self.norm_out = _AdaNorm(D, 2)
        self.proj_out = MiLinear(p * p * out_channels, D)
        self._init_base()

    def _dit_cfg(self):
        from ._lib import DitCfg
        c, s = self.config_dict, DitCfg()
        s.kind, s.in_channels, s.out_channels, s.patch_size = 2, c["in_channels"], c["out_channels"], c["patch_size"]
        s.num_layers, s.heads, s.head_dim = c["num_layers"], c["heads"], c["head_dim"]
        s.cross_dim, s.caption_channels, s.tdim = c["inner_dim"], c["joint_attention_dim"], 256
        s.vec_dim, s.n_vec, s.attention_bias, s.norm_eps, s.precision = c["pooled_projection_dim"], 0, 1, 1e-6, int(self.dt == F32)
        return s

    def _pos32(self, h, w, device):
        buf = self.pos_embed.pos_embed
        key = ("f32", h, w, str(device), buf.data_ptr(), buf._version)
        if key not in self._pos_cache:
            m = self.config_dict["pos_embed_max_size"]
            assert h <= m and w <= m, "input larger than pos_embed_max_size"
            top, left = (m - h) // 2, (m - w) // 2
            pe = buf.reshape(m, m, -1)[top:top + h, left:left + w, :].reshape(h * w, -1)
            self._pos_cache[key] = pe.to(device).float().contiguous()
        return self._pos_cache[key]

    def _pos(self, h, w, B, device):
        buf = self.pos_embed.pos_embed
        key = (h, w, B, str(device), buf.data_ptr(), buf._version)
        if key not in self._pos_cache:
            self._pos_cache.clear()
            m = self.config_dict["pos_embed_max_size"]
            assert h <= m and w <= m, "input larger than pos_embed_max_size"
            top, left = (m - h) // 2, (m - w) // 2
            pe = buf.reshape(1, m, m, -1)[:, top:top + h, left:left + w, :].reshape(1, h * w, -1)
            self._pos_cache[key] = pe.to(device).to(self.dt).expand(B, -1, -1).reshape(B * h * w, -1).contiguous()
        return self._pos_cache[key]


    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                conditioning: Dict[str, Dict[str, torch.Tensor]], hidden_states_masks: Optional[torch.Tensor] = None,
                *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"                 # TW:131
        cnd = conditioning["cond"]
        vector, crossattn, concat = cnd.get("vector", None), cnd.get("crossattn", None), cnd.get("concat", None)
        ops._dev(sample)      # device tensors only: there is no CPU fallback
        c = self.config_dict
        C_in = sample.shape[1]
        if concat is not None:                                                                     # TW:141-142
            sample = torch.cat([sample, concat], dim=1)
        assert sample.shape[1] == c["in_channels"]
        self.last_flops = 0.0
        if self.lora_r:
            self._lora_epoch += 1
        grad = torch.is_grad_enabled()
        B, _, Hh, Ww = sample.shape
        p, D, H = c["patch_size"], c["inner_dim"], c["heads"]
        h, w = Hh // p, Ww // p
        T, L = h * w, crossattn.shape[1]
        dev = sample.device
        eps = 1e-6
        if self._use_plan(sample):      # ONE call into the library's plan of this denoiser (csrc/dit_plan.h)
            assert C_in <= c["out_channels"]
            return self._plan_call(sample, timestep, crossattn, vector, self._pos32(h, w, dev), None, C_in)

        # CombinedTimestepTextProjEmbeddings: per-sample vectors [B, D]
        if not torch.is_tensor(timestep):
            timestep = torch.full((B,), float(timestep), device=dev)
        tte = self.time_text_embed
        temb = tte.timestep_embedder(ops.timestep_embed(timestep.reshape(-1).to(dev), 256, True, 0.0, self.dt)) \
            + tte.text_embedder(vector.to(self.dt).contiguous())
        st = _silu(temb)

        patches = sample.float().reshape(B, c["in_channels"], h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B * T, -1)
        x = self.pos_embed.proj(patches.to(self.dt).contiguous(), residual=self._pos(h, w, B, dev))
        ctx = self.context_embedder(crossattn.to(self.dt).reshape(B * L, -1).contiguous())

        def need(*ts):
            return grad and any(t.requires_grad for t in ts)

        for blk in self.transformer_blocks:
            a = blk.attn
            m = blk.norm1(st)                                         # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
            n = _LnModFn.apply(x, m[:, 0], m[:, 1], T, eps, need(x, m))
            mc = blk.norm1_context(st)
            if blk.context_pre_only:                                   # AdaLayerNormContinuous: (scale, shift)
                nc = _LnModFn.apply(ctx, mc[:, 1], mc[:, 0], L, eps, need(ctx, mc))
            else:
                nc = _LnModFn.apply(ctx, mc[:, 0], mc[:, 1], L, eps, need(ctx, mc))
            q = torch.cat([a.to_q(n).view(B, T, D), a.add_q_proj(nc).view(B, L, D)], dim=1)
            k = torch.cat([a.to_k(n).view(B, T, D), a.add_k_proj(nc).view(B, L, D)], dim=1)
            v = torch.cat([a.to_v(n).view(B, T, D), a.add_v_proj(nc).view(B, L, D)], dim=1)
            o = _AttnFn.apply(q, k, v, H, a.scale, None, need(q, k, v), self)
            x = _linear_gate_res(a.to_out[0], o[:, :T].reshape(B * T, D), m[:, 2], x, T)
            n2 = _LnModFn.apply(x, m[:, 3], m[:, 4], T, eps, need(x, m))
            x = _linear_gate_res(blk.ff.net[2], _linear_gelu(blk.ff.net[0].proj, n2), m[:, 5], x, T)
            if not blk.context_pre_only:
                ctx = _linear_gate_res(a.to_add_out, o[:, T:].reshape(B * L, D), mc[:, 2], ctx, L)
                n2c = _LnModFn.apply(ctx, mc[:, 3], mc[:, 4], L, eps, need(ctx, mc))
                ctx = _linear_gate_res(blk.ff_context.net[2], _linear_gelu(blk.ff_context.net[0].proj, n2c), mc[:, 5], ctx, L)

        mo = self.norm_out(st)                                        # (scale, shift)
        n = _LnModFn.apply(x, mo[:, 1], mo[:, 0], T, eps, need(x, mo))
        y = self.proj_out(n, out_f32=True)
        oc_ = c["out_channels"]
        y = torch.einsum("nhwpqc->nchpwq", y.view(B, h, w, p, p, oc_)).reshape(B, oc_, h * p, w * p)
        return y[:, :C_in]                                                                         # TW:154
