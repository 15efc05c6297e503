"""MiDiscriminator -- the reference's PatchGAN heads (nn.Sequential of Conv2d / SiLU / GroupNorm / Flatten,
/root/reference/examples/train_flash_sd.py:225-240, train_flash_sdxl.py:242-267,
tests/test_flash/test_flash_diffusion.py:100-113) executed on the hand-written HIP kernels:
forward, input gradient (for the generator step through the frozen backbone) and parameter gradients
(the head is trainable).  Subclasses nn.Sequential, so parameters keep the reference's names
("discriminator.0.weight", ...) and `convert()` wraps an existing nn.Sequential in place."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import ops
from ._lib import check, lib, ptr, stream_ptr

BF16 = torch.bfloat16


def _conv_fwd(x, w, bias, stride, pad):
    """x [B,H,W,Cp] NHWC (Cp = Cin padded to 8; bf16, or fp32 in validation mode) ; w OIHW f32"""
    O, I, KH, KW = w.shape
    return ops.conv2d_nhwc(x, ops.pack_conv_weight(w.detach(), x.dtype), KH=KH, KW=KW, stride=stride, pad=pad,
                           bias=bias.detach().float().contiguous() if bias is not None else None)


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        L = lib()
        B, Cin, H, W = x.shape
        Cp = (Cin + 7) // 8 * 8
        DT = torch.float32 if getattr(mod, "precision", "bf16") == "fp32" else BF16
        h = ops.nchw_to_nhwc(x.float().contiguous(), Cp, DT)       # [B,H,W,Cp] bf16 (fp32: validation mode)
        saved = []
        pi = 0
        for layer in mod:
            if isinstance(layer, nn.Conv2d):
                w = params[pi]; pi += 1
                b = None
                if layer.bias is not None:
                    b = params[pi]; pi += 1
                y = _conv_fwd(h, w, b, layer.stride[0], layer.padding[0])
                saved.append(("conv", h, w, b is not None, layer.stride[0], layer.padding[0]))
                h = y
            elif isinstance(layer, nn.SiLU):
                y = ops.silu(h)
                saved.append(("silu", h))
                h = y
            elif isinstance(layer, nn.GroupNorm):
                g, bt = params[pi], params[pi + 1]; pi += 2
                Bn, Hh, Ww, Cc = h.shape
                y, stats = ops.groupnorm_fwd(h.view(Bn, Hh * Ww, Cc), g.detach().float().contiguous(),
                                             bt.detach().float().contiguous(), layer.num_groups, layer.eps, 0)
                saved.append(("gn", h, g, stats, layer.num_groups, layer.eps))
                h = y.view(Bn, Hh, Ww, Cc)
            elif isinstance(layer, nn.Flatten):
                saved.append(("flatten",))
            else:
                raise NotImplementedError(f"MiDiscriminator: unsupported layer {type(layer).__name__}")
        ctx.saved, ctx.mod, ctx.xshape, ctx.cp, ctx.dt = saved, mod, x.shape, Cp, DT
        ctx.needs_x = x.requires_grad
        Bn, Hh, Ww, Cc = h.shape
        out = h.float().permute(0, 3, 1, 2).reshape(Bn, -1)        # Flatten of NCHW (tiny: B x k logits)
        ctx.out_shape = (Bn, Hh, Ww, Cc)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = lib()
        Bn, Hh, Ww, Cc = ctx.out_shape
        DT = ctx.dt
        dy = gout.reshape(Bn, Cc, Hh, Ww).permute(0, 2, 3, 1).contiguous().to(DT)    # NHWC, the forward's precision
        grads = []
        for rec in reversed(ctx.saved):
            kind = rec[0]
            if kind == "flatten":
                continue
            if kind == "silu":
                dy = ops.silu_bwd(rec[1], dy.contiguous())
            elif kind == "gn":
                _, x, g, stats, G, eps = rec
                Bq, Hq, Wq, Cq = x.shape
                dyc = dy.contiguous()
                dgam = torch.zeros(Cq, dtype=torch.float32, device=x.device)
                dbet = torch.zeros(Cq, dtype=torch.float32, device=x.device)
                ops.colsum(dyc, x, stats, dbet, dgam, Bq * Hq * Wq, Cq, Hq * Wq, G, eps)
                dx = ops.groupnorm_bwd(x.view(Bq, Hq * Wq, Cq), dyc.view(Bq, Hq * Wq, Cq), g.detach().float().contiguous(),
                                       torch.zeros(Cq, device=x.device), stats, G, eps, 0)
                grads.append(dbet)
                grads.append(dgam)
                dy = dx.view(Bq, Hq, Wq, Cq)
            elif kind == "conv":
                _, x, w, has_b, stride, pad = rec
                O, I, KH, KW = w.shape
                Bq, Hq, Wq, Cp = x.shape
                _, Ho, Wo, _ = dy.shape
                M = Bq * Ho * Wo
                dyc = dy.contiguous().view(M, O)
                if has_b:
                    db = torch.zeros(O, dtype=torch.float32, device=x.device)
                    ops.colsum(dyc, None, None, db, None, M, O, 1, 1, 0.0)
                # ---- weight gradient: dW[O][KH*KW*Cp] += dY^T [O][M] * im2col(X)^T [K][M] ----
                K = KH * KW * Cp
                xcol = ops.im2col(x, Ho, Wo, KH, KW, stride, pad)
                dwp = torch.zeros(O, K, dtype=torch.float32, device=x.device)
                if x.dtype == torch.float32:          # validation mode: the TN product straight from the row-major operands
                    ops.wgrad_tn(dyc, xcol, dwp)
                else:
                    Mp = (M + 7) // 8 * 8
                    xcolT = torch.empty(K, Mp, dtype=BF16, device=x.device)
                    check(L.fdmi_transpose2d_pad(ptr(xcol), K, ptr(xcolT), Mp, M, K, Mp, stream_ptr()))
                    Op8 = (O + 7) // 8 * 8
                    dyT = torch.zeros(Op8, Mp, dtype=BF16, device=x.device)
                    check(L.fdmi_transpose2d_pad(ptr(dyc), O, ptr(dyT), Mp, M, O, Mp, stream_ptr()))
                    ops.gemm(dyT, xcolT, M=O, N=K, K=Mp, out=dwp, accum_atomic=True)
                dw = dwp.view(O, KH, KW, Cp)[..., :I].permute(0, 3, 1, 2).contiguous()
                # ---- input gradient (gather form of the transposed conv) ----
                Op = (O + 7) // 8 * 8
                dyp = dyc if Op == O else ops.pad_cols(dyc, Op)
                dx = ops.conv2d_nhwc(dyp.view(Bq, Ho, Wo, Op), ops.pack_conv_weight_dgrad(w.detach(), x.dtype), KH=KH, KW=KW,
                                     stride=stride, pad=pad, dgrad=1, out_hw=(Hq, Wq))
                if dx.shape[-1] != Cp:   # N = true Cin; re-pad channels for the next dgrad
                    t = torch.zeros(Bq, Hq, Wq, Cp, dtype=x.dtype, device=x.device)
                    t[..., :dx.shape[-1]] = dx
                    dx = t
                if has_b:
                    grads.append(db)
                grads.append(dw)
                dy = dx
        gx = None
        if ctx.needs_x:
            gx = ops.nhwc_to_nchw(dy.contiguous(), ctx.xshape[1])
        return (None, gx) + tuple(reversed(grads))


class MiDiscriminator(nn.Sequential):
    precision = "bf16"   # "fp32": the validation kernels (csrc/ref32.hip), for the parity gate against the fp32 oracle

    def forward(self, x):
        assert x.is_cuda, "MiDiscriminator runs on the GPU only (no CPU fallback)"
        params = []
        for layer in self:
            if isinstance(layer, nn.Conv2d):
                params.append(layer.weight)
                if layer.bias is not None:
                    params.append(layer.bias)
            elif isinstance(layer, nn.GroupNorm):
                params += [layer.weight, layer.bias]
        return _DiscFn.apply(self, x, *params)

    @classmethod
    def convert(cls, seq: nn.Sequential) -> "MiDiscriminator":
        return cls(*list(seq.children()))
