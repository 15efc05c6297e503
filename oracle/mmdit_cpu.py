"""TEST INFRASTRUCTURE (oracle) -- PyTorch fp32 restatement of diffusers' ``SD3Transformer2DModel`` (MMDiT: joint blocks over
the latent tokens and the text tokens), the base class of the reference wrapper ``DiffusersSD3Transformer2DWrapper``
(/root/reference/src/flash/models/transformers/tranformers.py:103-163, "TW"), plus a restatement of that wrapper.

The base-class arithmetic lives in third-party diffusers (un-vendored fork branch, requirements.txt:1; absent from this
container, no network) -- PARITY UNPINNED by the reference for that part: this file restates the published upstream
algorithm (SD3-medium: no qk-norm, no dual attention) for the hyper-parameters the reference pins in-tree
(examples/train_flash_sd3.py:65-77).  The wrapper IS in-tree: tests/test_oracle_vs_reference.py runs the reference's real
wrapper class on top of ``SD3Transformer2DModelRef`` (registered as the ``diffusers`` stub base class by
oracle/shim_import.py) and checks that ``SD3TransformerRef`` below reproduces it bit for bit.

Module / parameter / buffer names equal diffusers' state_dict keys (examples/train_flash_sd3.py:79 loads them strict=True)
so the same weights load into this oracle and into the HIP path (flash_diffusion_amd/dit.py: MiSD3Transformer2DModel).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .dit_cpu import FeedForward, TimestepEmbedding, Timesteps, sincos_2d


class PatchEmbedCropped(nn.Module):
    """PatchEmbed with ``pos_embed_max_size``: a persistent sin-cos table on a max x max grid, centre-cropped to the input"""

    def __init__(self, height, width, patch_size, in_channels, embed_dim, pos_embed_max_size):
        super().__init__()
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)
        self.patch_size, self.pos_embed_max_size = patch_size, pos_embed_max_size
        pe = sincos_2d(embed_dim, pos_embed_max_size, base_size=height // patch_size, interpolation_scale=1)
        self.register_buffer("pos_embed", torch.from_numpy(pe).float().unsqueeze(0), persistent=True)

    def forward(self, latent):
        h, w = latent.shape[-2] // self.patch_size, latent.shape[-1] // self.patch_size
        m = self.pos_embed_max_size
        assert h <= m and w <= m
        top, left = (m - h) // 2, (m - w) // 2
        pe = self.pos_embed.reshape(1, m, m, -1)[:, top:top + h, left:left + w, :]
        pe = pe.reshape(1, -1, pe.shape[-1])
        latent = self.proj(latent).flatten(2).transpose(1, 2)
        return (latent + pe).to(latent.dtype)


class TextProjection(nn.Module):
    """PixArtAlphaTextProjection(act_fn="silu") as used by CombinedTimestepTextProjEmbeddings"""

    def __init__(self, in_features, hidden_size):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.time_proj = Timesteps(256, True, 0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = TextProjection(pooled_projection_dim, embedding_dim)

    def forward(self, timestep, pooled_projection):
        t = self.timestep_embedder(self.time_proj(timestep).to(pooled_projection.dtype))
        return t + self.text_embedder(pooled_projection)


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.linear(F.silu(emb)).chunk(6, dim=1)
        return self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None], gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormContinuous(nn.Module):
    """NOTE the chunk order: (scale, shift)"""

    def __init__(self, dim, cond_dim):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, 1e-6, elementwise_affine=False)

    def forward(self, x, cond):
        scale, shift = torch.chunk(self.linear(F.silu(cond).to(x.dtype)), 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class JointAttention(nn.Module):
    """Attention(added_kv_proj_dim=dim, context_pre_only=...) with JointAttnProcessor2_0: latent tokens first, text tokens
    second, one softmax over the concatenated keys"""

    def __init__(self, dim, heads, context_pre_only):
        super().__init__()
        self.heads, self.context_pre_only = heads, context_pre_only
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.add_k_proj, self.add_v_proj, self.add_q_proj = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        if not context_pre_only:
            self.to_add_out = nn.Linear(dim, dim)

    def forward(self, x, c):
        B, T, D = x.shape
        H = self.heads
        q = torch.cat([self.to_q(x), self.add_q_proj(c)], dim=1)
        k = torch.cat([self.to_k(x), self.add_k_proj(c)], dim=1)
        v = torch.cat([self.to_v(x), self.add_v_proj(c)], dim=1)
        q, k, v = (t.view(B, -1, H, D // H).transpose(1, 2) for t in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, D)
        ox, oc = o[:, :T], o[:, T:]
        ox = self.to_out[0](ox)
        if not self.context_pre_only:
            oc = self.to_add_out(oc)
        return ox, oc


class JointTransformerBlock(nn.Module):
    def __init__(self, dim, heads, context_pre_only=False):
        super().__init__()
        self.context_pre_only = context_pre_only
        self.norm1 = AdaLayerNormZero(dim)
        self.norm1_context = AdaLayerNormContinuous(dim, dim) if context_pre_only else AdaLayerNormZero(dim)
        self.attn = JointAttention(dim, heads, context_pre_only)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim)
        if not context_pre_only:
            self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
            self.ff_context = FeedForward(dim)

    def forward(self, x, c, temb):
        n, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(x, emb=temb)
        if self.context_pre_only:
            nc = self.norm1_context(c, temb)
        else:
            nc, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(c, emb=temb)
        ax, ac = self.attn(n, nc)
        x = x + gate_msa.unsqueeze(1) * ax
        n = self.norm2(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        x = x + gate_mlp.unsqueeze(1) * self.ff(n)
        if self.context_pre_only:
            return None, x
        c = c + c_gate_msa.unsqueeze(1) * ac
        nc = self.norm2_context(c) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        c = c + c_gate_mlp.unsqueeze(1) * self.ff_context(nc)
        return c, x


class SD3Transformer2DModelRef(nn.Module):
    def __init__(self, sample_size=128, patch_size=2, in_channels=16, num_layers=18, attention_head_dim=64,
                 num_attention_heads=18, joint_attention_dim=4096, caption_projection_dim=1152, pooled_projection_dim=2048,
                 out_channels=16, pos_embed_max_size=96):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        assert caption_projection_dim == inner, "joint attention concatenates latent and text tokens of one width"
        self.config = SimpleNamespace(sample_size=sample_size, patch_size=patch_size, in_channels=in_channels,
                                      num_layers=num_layers, attention_head_dim=attention_head_dim,
                                      num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                                      caption_projection_dim=caption_projection_dim,
                                      pooled_projection_dim=pooled_projection_dim, out_channels=out_channels,
                                      pos_embed_max_size=pos_embed_max_size)
        self.out_channels, self.inner_dim = out_channels, inner
        self.pos_embed = PatchEmbedCropped(sample_size, sample_size, patch_size, in_channels, inner, pos_embed_max_size)
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(inner, pooled_projection_dim)
        self.context_embedder = nn.Linear(joint_attention_dim, caption_projection_dim)
        self.transformer_blocks = nn.ModuleList([JointTransformerBlock(inner, num_attention_heads, i == num_layers - 1)
                                                 for i in range(num_layers)])
        self.norm_out = AdaLayerNormContinuous(inner, inner)
        self.proj_out = nn.Linear(inner, patch_size * patch_size * out_channels, bias=True)

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, **unused):
        height, width = hidden_states.shape[-2:]
        x = self.pos_embed(hidden_states)
        temb = self.time_text_embed(timestep, pooled_projections)
        c = self.context_embedder(encoder_hidden_states)
        for blk in self.transformer_blocks:
            c, x = blk(x, c, temb)
        x = self.proj_out(self.norm_out(x, temb))
        p = self.config.patch_size
        height, width = height // p, width // p
        x = x.reshape(x.shape[0], height, width, p, p, self.out_channels)
        x = torch.einsum("nhwpqc->nchpwq", x)
        return SimpleNamespace(sample=x.reshape(x.shape[0], self.out_channels, height * p, width * p))


class SD3TransformerRef(SD3Transformer2DModelRef):
    """TW:103-163: denoiser call contract over the MMDiT (``pooled_projections`` = the ``vector`` conditioning)"""

    def forward(self, sample, timestep, conditioning: Dict[str, torch.Tensor], hidden_states_masks=None, *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"                      # TW:131
        c = conditioning["cond"]
        vector, crossattn, concat = c.get("vector"), c.get("crossattn"), c.get("concat")                 # TW:133-136
        ch = sample.shape[1]
        if concat is not None:                                                                           # TW:141-142
            sample = torch.cat([sample, concat], dim=1)
        out = super().forward(hidden_states=sample, timestep=timestep, encoder_hidden_states=crossattn,
                              pooled_projections=vector)                                                  # TW:146-153
        return out.sample[:, :ch]                                                                        # TW:154

    def freeze(self):                                                                                    # TW:157-163
        self.eval()
        for p in self.parameters():
            p.requires_grad = False


SD3_MEDIUM = dict(sample_size=128, patch_size=2, in_channels=16, num_layers=24, attention_head_dim=64,
                  num_attention_heads=24, joint_attention_dim=4096, caption_projection_dim=1536, pooled_projection_dim=2048,
                  out_channels=16, pos_embed_max_size=192)                   # examples/train_flash_sd3.py:65-77

TINY_MMDIT = dict(sample_size=16, patch_size=2, in_channels=16, num_layers=2, attention_head_dim=8, num_attention_heads=4,
                  joint_attention_dim=48, caption_projection_dim=32, pooled_projection_dim=24, out_channels=16,
                  pos_embed_max_size=12)
