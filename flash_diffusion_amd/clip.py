"""CLIP text encoder on the HIP path: ``MiCLIPTextModel`` is a drop-in for transformers' ``CLIPTextModel`` /
``CLIPTextModelWithProjection`` as the reference's conditioners call it
(/root/reference/src/flash/models/embedders/clip/clip_embedder_model.py:10-104 and :107-201: ``transformer(input_ids=tokens,
output_hidden_states=...)`` -> ``last_hidden_state`` / ``pooler_output`` / ``hidden_states[layer_idx]`` / ``text_embeds``), with
transformers' state_dict names (``text_model.embeddings.token_embedding.weight`` ... ``text_model.final_layer_norm.bias``,
``text_projection.weight``), and ``MiClipEmbedder`` mirrors ``ClipEmbedder.forward`` (layer selection, pooled output,
``force_zero_embedding``).  Frozen, forward only (the conditioners run under the step's no-grad conditioning calls, FD:188-205).

Every token-major operation is a launch of libfdmi.so's op-level C-ABI: LayerNorm (affine), bf16 MFMA GEMMs with fused bias /
residual / activation epilogues (quick-GELU as SiLU(1.702 x) / 1.702 folded into the two MLP GEMMs' alpha, exact GELU as its own
epilogue), causal self-attention through the exact-f32 materialised-score kernels (77 tokens: the cost is nil).  torch does the
embedding lookup (a gather on parameters) and the pooled-row selection.  GPU only."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .conditioners import BaseConditioner

BF16, F32 = torch.bfloat16, torch.float32


class _Lin(nn.Module):
    def __init__(self, out_f, in_f, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_f, in_f) * in_f ** -0.5)
        self.bias = nn.Parameter(torch.zeros(out_f)) if bias else None
        self._cache = None

    def operands(self, dt, bias_scale=1.0):
        w, b = self.weight, self.bias
        key = (w.data_ptr(), w._version, None if b is None else b._version, dt, bias_scale)
        if self._cache is None or self._cache[0] != key:
            wb = w.detach().contiguous() if dt == F32 else ops.f32_to_bf16(w.detach().contiguous())
            bb = None if b is None else (b.detach() * bias_scale).float().contiguous()
            self._cache = (key, wb, bb)
        return self._cache[1], self._cache[2]


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(d)), nn.Parameter(torch.zeros(d))


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = _Lin(d, d), _Lin(d, d), _Lin(d, d), _Lin(d, d)


class _MLP(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.fc1, self.fc2 = _Lin(inter, d), _Lin(d, inter)


class _Layer(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.self_attn, self.layer_norm1, self.mlp, self.layer_norm2 = _Attn(d), _LN(d), _MLP(d, inter), _LN(d)


class _Embeddings(nn.Module):
    def __init__(self, vocab, d, maxpos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, d)
        self.position_embedding = nn.Embedding(maxpos, d)
        self.register_buffer("position_ids", torch.arange(maxpos).unsqueeze(0), persistent=False)


class _Encoder(nn.Module):
    def __init__(self, n, d, inter):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(d, inter) for _ in range(n)])


class _TextTransformer(nn.Module):
    def __init__(self, vocab, d, inter, n, maxpos):
        super().__init__()
        self.embeddings = _Embeddings(vocab, d, maxpos)
        self.encoder = _Encoder(n, d, inter)
        self.final_layer_norm = _LN(d)


class MiCLIPTextModel(nn.Module):
    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=None, eos_token_id=2,
                 precision="bf16", **unused):
        super().__init__()
        assert precision in ("bf16", "fp32") and hidden_act in ("quick_gelu", "gelu"), (precision, hidden_act)
        assert hidden_size % num_attention_heads == 0 and hidden_size % 8 == 0 and intermediate_size % 8 == 0
        self.dt = F32 if precision == "fp32" else BF16
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      max_position_embeddings=max_position_embeddings, hidden_act=hidden_act,
                                      layer_norm_eps=layer_norm_eps, projection_dim=projection_dim, eos_token_id=eos_token_id)
        self.text_model = _TextTransformer(vocab_size, hidden_size, intermediate_size, num_hidden_layers, max_position_embeddings)
        self.text_projection = _Lin(projection_dim, hidden_size, bias=False) if projection_dim else None   # ...WithProjection

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _act(self, x):
        return x if x.dtype == self.dt else (x.float() if self.dt == F32 else ops.f32_to_bf16(x.contiguous()))

    @torch.no_grad()
    def forward(self, input_ids, output_hidden_states=False, **kw):
        c = self.config
        tm = self.text_model
        assert input_ids.is_cuda, "MiCLIPTextModel runs on the GPU only (no CPU fallback)"
        B, S = input_ids.shape
        D, H = c.hidden_size, c.num_attention_heads
        emb = tm.embeddings
        h32 = emb.token_embedding.weight[input_ids] + emb.position_embedding.weight[emb.position_ids[0, :S]][None]   # [B, S, D] f32
        h = self._act(h32.reshape(B * S, D).contiguous())
        hidden: List[torch.Tensor] = [h32.reshape(B, S, D)] if output_hidden_states else []
        eps, dt = c.layer_norm_eps, self.dt
        for lyr in tm.encoder.layers:
            a = lyr.self_attn
            n1 = ops.layernorm_fwd(h, lyr.layer_norm1.weight, lyr.layer_norm1.bias, eps)
            q = ops.gemm(n1, *a.q_proj.operands(dt)[:1], bias=a.q_proj.operands(dt)[1], out_f32=True).view(B, S, D)
            k = ops.gemm(n1, *a.k_proj.operands(dt)[:1], bias=a.k_proj.operands(dt)[1], out_f32=True).view(B, S, D)
            v = ops.gemm(n1, *a.v_proj.operands(dt)[:1], bias=a.v_proj.operands(dt)[1], out_f32=True).view(B, S, D)
            o = ops.attn_causal_fwd(q, k, v, H, (D // H) ** -0.5).view(B * S, D)          # fp32 in, fp32 out
            h = ops.gemm(self._act(o), a.out_proj.operands(dt)[0], bias=a.out_proj.operands(dt)[1], residual=h)
            n2 = ops.layernorm_fwd(h, lyr.layer_norm2.weight, lyr.layer_norm2.bias, eps)
            if c.hidden_act == "quick_gelu":    # x sigmoid(1.702 x) = SiLU(1.702 x) / 1.702: alpha / scaled bias in fc1, 1 / 1.702 in fc2
                w1, b1 = lyr.mlp.fc1.operands(dt, bias_scale=1.702)
                f = ops.gemm(n2, w1, bias=b1, alpha=1.702, act=ops.ACT_SILU)
                w2, b2 = lyr.mlp.fc2.operands(dt)
                h = ops.gemm(f, w2, bias=b2, alpha=1.0 / 1.702, residual=h)
            else:
                w1, b1 = lyr.mlp.fc1.operands(dt)
                f = ops.gemm(n2, w1, bias=b1, act=ops.ACT_GELU)
                w2, b2 = lyr.mlp.fc2.operands(dt)
                h = ops.gemm(f, w2, bias=b2, residual=h)
            if output_hidden_states:
                hidden.append(h.float().view(B, S, D))
        last = ops.layernorm_fwd(h, tm.final_layer_norm.weight, tm.final_layer_norm.bias, eps).float().view(B, S, D)
        # pooled row (transformers: argmax of the ids for the legacy eos id 2, else the first eos position)
        if c.eos_token_id == 2:
            idx = input_ids.to(torch.int).argmax(dim=-1)
        else:
            idx = (input_ids.to(torch.int) == c.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=last.device), idx]
        out = SimpleNamespace(last_hidden_state=last, pooler_output=pooled, hidden_states=tuple(hidden) if output_hidden_states else None)
        if self.text_projection is not None:
            wp, _ = self.text_projection.operands(dt)
            out.text_embeds = ops.gemm(self._act(pooled.contiguous()), wp, out_f32=True)
        return out


class MiClipEmbedder(BaseConditioner):
    """``ClipEmbedder`` / ``ClipEmbedderWithProjection`` (clip_embedder_model.py:10-201) over MiCLIPTextModel.  ``tokenizer``:
    any callable ``(list of str) -> LongTensor [B, max_length]`` (transformers' CLIPTokenizer needs its vocabulary files)."""

    def __init__(self, transformer: MiCLIPTextModel, tokenizer, layer="last", layer_idx=None, always_return_pooled=False,
                 input_key="text", unconditional_conditioning_rate=0.0):
        super().__init__(input_key, unconditional_conditioning_rate)
        assert layer in ("last", "pooled", "hidden")
        assert layer != "hidden" or layer_idx is not None
        self.transformer, self.tokenizer = transformer, tokenizer
        self.layer, self.layer_idx, self.always_return_pooled = layer, layer_idx, always_return_pooled

    def freeze(self):
        self.transformer.freeze()

    def forward(self, batch: Dict[str, Any], force_zero_embedding: bool = False, device="cuda", *args, **kwargs):
        tokens = self.tokenizer(batch[self.input_key]).to(device)
        outputs = self.transformer(input_ids=tokens, output_hidden_states=self.layer == "hidden")
        proj = self.transformer.text_projection is not None
        pooled = outputs.text_embeds if proj else outputs.pooler_output
        if self.layer == "last":
            z = outputs.last_hidden_state
        elif self.layer == "pooled":
            z = pooled[:, None, :]
        else:
            z = outputs.hidden_states[self.layer_idx]
        if force_zero_embedding:
            z = 0 * z
        output = {self.dim2outputkey[z.dim()]: z}
        if self.always_return_pooled:
            output.update({self.dim2outputkey[2]: 0 * pooled if force_zero_embedding else pooled})
        return output
