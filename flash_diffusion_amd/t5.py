"""T5 text encoder on the HIP path: ``MiT5EncoderModel`` is a drop-in for transformers' ``T5EncoderModel`` as the
reference's conditioner calls it (/root/reference/src/flash/models/embedders/t5/t5_embedder_model.py:11-104:
``transformer(input_ids=tokens, attention_mask=attention_mask, output_hidden_states=...)`` -> ``last_hidden_state`` /
``hidden_states[layer_idx]``), with transformers' state_dict names (``shared.weight``, ``encoder.embed_tokens.weight``,
``encoder.block.{i}.layer.0.SelfAttention.{q,k,v,o}.weight``, ``encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight``,
``encoder.block.{i}.layer.{0,1}.layer_norm.weight``, ``encoder.block.{i}.layer.1.DenseReluDense.{wi_0,wi_1,wo | wi,wo}.weight``,
``encoder.final_layer_norm.weight``), and ``MiT5TextEmbedder`` mirrors ``T5TextEmbedder.forward`` (layer selection, attention mask
output, ``force_zero_embedding``).  Frozen, forward only: the PixArt / SD3 conditioners run under the step's no-grad conditioning calls
(FD:188-205, examples/train_flash_pixart.py: T5-XXL, 120 tokens).

The arithmetic is upstream T5 (v1.0 ``relu`` and v1.1 / flan ``gated-gelu``): T5LayerNorm (RMS norm, no bias), un-scaled dot-product
attention whose scores receive the bucketed relative-position bias of block 0 (shared by all blocks) and the additive key mask,
``h += o(attn)``, gated feed-forward ``wo(gelu_new(wi_0 x) * wi_1 x)``, final T5LayerNorm.  Every token-major operation is a launch of
libfdmi.so's op-level C-ABI: ``fdmi_rmsnorm``, bf16 MFMA GEMMs (tanh-GELU and the residual in the epilogues), ``fdmi_mul``, and the
biased attention through the exact-f32 materialised-score kernels (``fdmi_attn_bias_fwd_f32``; 120 keys: the cost is nil).  torch does
the embedding lookup and builds the [H, S, S] bias table from the bucket embedding (integer bucketing of S x S positions, cached per
S).  GPU only; there is no CPU fallback."""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .conditioners import BaseConditioner

BF16, F32 = torch.bfloat16, torch.float32


class _W(nn.Module):
    """bias-free linear map (transformers nn.Linear(bias=False) names: ``.weight``) with a cached GEMM operand"""

    def __init__(self, out_f, in_f):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_f, in_f) * in_f ** -0.5)
        self._cache = None

    def operand(self, dt):
        w = self.weight
        key = (w.data_ptr(), w._version, dt)
        if self._cache is None or self._cache[0] != key:
            self._cache = (key, w.detach().contiguous() if dt == F32 else ops.f32_to_bf16(w.detach().contiguous()))
        return self._cache[1]


class _Norm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))


class _SelfAttention(nn.Module):
    def __init__(self, d, inner, heads, buckets):
        super().__init__()
        self.q, self.k, self.v, self.o = _W(inner, d), _W(inner, d), _W(inner, d), _W(d, inner)
        if buckets:
            self.relative_attention_bias = nn.Embedding(buckets, heads)


class _Dense(nn.Module):
    def __init__(self, d, ff, gated):
        super().__init__()
        if gated:
            self.wi_0, self.wi_1 = _W(ff, d), _W(ff, d)
        else:
            self.wi = _W(ff, d)
        self.wo = _W(d, ff)


class _LayerSA(nn.Module):
    def __init__(self, d, inner, heads, buckets):
        super().__init__()
        self.SelfAttention, self.layer_norm = _SelfAttention(d, inner, heads, buckets), _Norm(d)


class _LayerFF(nn.Module):
    def __init__(self, d, ff, gated):
        super().__init__()
        self.DenseReluDense, self.layer_norm = _Dense(d, ff, gated), _Norm(d)


class _Block(nn.Module):
    def __init__(self, d, inner, heads, ff, gated, buckets):
        super().__init__()
        self.layer = nn.ModuleList([_LayerSA(d, inner, heads, buckets), _LayerFF(d, ff, gated)])


class _Stack(nn.Module):
    def __init__(self, vocab, d, inner, heads, ff, gated, n, buckets):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, d)
        self.block = nn.ModuleList([_Block(d, inner, heads, ff, gated, buckets if i == 0 else 0) for i in range(n)])
        self.final_layer_norm = _Norm(d)


def relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """bidirectional T5 bucketing of memory_position - query_position (transformers T5Attention._relative_position_bucket)"""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


class MiT5EncoderModel(nn.Module):
    def __init__(self, vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
                 relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                 feed_forward_proj="gated-gelu", precision="bf16", **unused):
        super().__init__()
        assert precision in ("bf16", "fp32") and feed_forward_proj in ("gated-gelu", "relu"), (precision, feed_forward_proj)
        inner = d_kv * num_heads
        assert d_model % 8 == 0 and inner % 8 == 0 and d_ff % 8 == 0
        self.dt = F32 if precision == "fp32" else BF16
        self.config = SimpleNamespace(vocab_size=vocab_size, d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=num_layers,
                                      num_heads=num_heads, relative_attention_num_buckets=relative_attention_num_buckets,
                                      relative_attention_max_distance=relative_attention_max_distance,
                                      layer_norm_epsilon=layer_norm_epsilon, feed_forward_proj=feed_forward_proj)
        self.shared = nn.Embedding(vocab_size, d_model)
        self.encoder = _Stack(vocab_size, d_model, inner, num_heads, d_ff, feed_forward_proj == "gated-gelu", num_layers,
                              relative_attention_num_buckets)
        self.encoder.embed_tokens.weight = self.shared.weight          # tied, as in transformers
        self._bias_cache = None

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _act(self, x):
        return x if x.dtype == self.dt else (x.float() if self.dt == F32 else ops.f32_to_bf16(x.contiguous()))

    def _position_bias(self, S, device):
        w = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight
        key = (S, w.data_ptr(), w._version, str(device))
        if self._bias_cache is None or self._bias_cache[0] != key:
            c = self.config
            pos = torch.arange(S, dtype=torch.long, device=device)
            bucket = relative_position_bucket(pos[None, :] - pos[:, None], c.relative_attention_num_buckets,
                                              c.relative_attention_max_distance)
            self._bias_cache = (key, w.detach().float()[bucket].permute(2, 0, 1).contiguous())                 # [H, S, S]
        return self._bias_cache[1]

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, output_hidden_states=False, **kw):
        c = self.config
        assert input_ids.is_cuda, "MiT5EncoderModel runs on the GPU only (no CPU fallback)"
        B, S = input_ids.shape
        D, H, dt, eps = c.d_model, c.num_heads, self.dt, c.layer_norm_epsilon
        inner = c.d_kv * H
        h32 = self.shared.weight[input_ids]                                                                   # [B, S, D] f32
        h = self._act(h32.reshape(B * S, D).contiguous())
        hidden: List[torch.Tensor] = []
        bias = self._position_bias(S, input_ids.device)
        kbias = None
        if attention_mask is not None:    # transformers: (1 - mask) * finfo(float32).min added to the scores
            kbias = ((1.0 - attention_mask.to(device=input_ids.device, dtype=F32)) * torch.finfo(F32).min).contiguous()
        for blk in self.encoder.block:
            if output_hidden_states:
                hidden.append(h.float().view(B, S, D))
            sa, ff = blk.layer[0], blk.layer[1]
            a = sa.SelfAttention
            n1 = ops.rmsnorm(h, sa.layer_norm.weight, eps)
            q = ops.gemm(n1, a.q.operand(dt), out_f32=True).view(B, S, inner)
            k = ops.gemm(n1, a.k.operand(dt), out_f32=True).view(B, S, inner)
            v = ops.gemm(n1, a.v.operand(dt), out_f32=True).view(B, S, inner)
            o = ops.attn_bias_fwd(q, k, v, H, 1.0, bias, kbias).view(B * S, inner)     # T5 does not scale the scores
            h = ops.gemm(self._act(o), a.o.operand(dt), residual=h)
            n2 = ops.rmsnorm(h, ff.layer_norm.weight, eps)
            dn = ff.DenseReluDense
            if c.feed_forward_proj == "gated-gelu":
                f = ops.mul(ops.gemm(n2, dn.wi_0.operand(dt), act=ops.ACT_GELU_TANH), ops.gemm(n2, dn.wi_1.operand(dt)))
            else:
                f = ops.gemm(n2, dn.wi.operand(dt), act=ops.ACT_RELU)
            h = ops.gemm(f, dn.wo.operand(dt), residual=h)
        last = ops.rmsnorm(h, self.encoder.final_layer_norm.weight, eps).float().view(B, S, D)
        if output_hidden_states:
            hidden.append(last)
        return SimpleNamespace(last_hidden_state=last, hidden_states=tuple(hidden) if output_hidden_states else None)


class MiT5TextEmbedder(BaseConditioner):
    """``T5TextEmbedder`` (t5_embedder_model.py:11-104) over MiT5EncoderModel.  ``tokenizer``: any callable
    ``(list of str) -> (input_ids LongTensor [B, max_length], attention_mask [B, max_length])`` (transformers' T5Tokenizer with
    padding="max_length" needs its sentencepiece model file)."""

    def __init__(self, transformer: MiT5EncoderModel, tokenizer, layer="last", layer_idx=None, returns_attention_mask=False,
                 input_key="text", unconditional_conditioning_rate=0.0):
        super().__init__(input_key, unconditional_conditioning_rate)
        assert layer in ("last", "hidden")
        assert layer != "hidden" or layer_idx is not None, "Layer index is required for hidden layer"
        self.transformer, self.tokenizer = transformer, tokenizer
        self.layer, self.layer_idx, self.returns_attention_mask = layer, layer_idx, returns_attention_mask

    def freeze(self):
        self.transformer.freeze()

    def forward(self, batch: Dict[str, Any], force_zero_embedding: bool = False, device="cuda", *args, **kwargs):
        tokens, attention_mask = self.tokenizer(batch[self.input_key])
        tokens, attention_mask = tokens.to(device), attention_mask.to(device)
        outputs = self.transformer(input_ids=tokens, attention_mask=attention_mask, output_hidden_states=self.layer == "hidden")
        z = outputs.last_hidden_state if self.layer == "last" else outputs.hidden_states[self.layer_idx]
        if force_zero_embedding:
            z = 0 * z
            attention_mask = 0 * attention_mask
        if self.returns_attention_mask:
            return {self.dim2outputkey[z.dim()]: z, "attention_mask": attention_mask}
        return {self.dim2outputkey[z.dim()]: z}
