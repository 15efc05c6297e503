"""TEST INFRASTRUCTURE (oracle) -- PyTorch fp32 restatement of diffusers' ``Transformer2DModel`` on its PixArt-alpha path
(patched input, ``norm_type="ada_norm_single"``, caption projection), the base class of the reference wrapper
``DiffusersTransformer2DWrapper`` (/root/reference/src/flash/models/transformers/tranformers.py:9-100, "TW"), plus a
restatement of that wrapper and of the reference's own ``AdaLayerNormSingle`` (transformers/utils.py:8-102, "TU").

The base-class arithmetic lives in third-party diffusers (un-vendored fork branch, requirements.txt:1; absent from this
container, no network) -- PARITY UNPINNED by the reference for that part: this file restates the published upstream
algorithm for the hyper-parameters the reference pins in-tree (examples/train_flash_pixart.py:63-86).  The wrapper and
``AdaLayerNormSingle`` ARE in-tree: tests/test_oracle_vs_reference.py runs the reference's real wrapper class on top of
``Transformer2DModelRef`` (registered as the ``diffusers`` stub base class by oracle/shim_import.py) and checks that
``PixartTransformerRef`` below reproduces it bit for bit.

Module / parameter names equal diffusers' state_dict keys (after the remap of examples/train_flash_pixart.py:90-172) so the
same weights load into this oracle and into the HIP path (flash_diffusion_amd/dit.py).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_cpu import timestep_embedding


# ---- diffusers.models.embeddings (upstream semantics) ------------------------------------------------------------------
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(F.silu(self.linear_1(sample)))


def sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim, grid_size, base_size=16, interpolation_scale=1.0):
    """get_2d_sincos_pos_embed: the first half of the channels encodes the column (w) coordinate, the second the row"""
    gh, gw = (grid_size, grid_size) if isinstance(grid_size, int) else grid_size
    grid_h = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    grid_w = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, gw, gh])
    return np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)


class PatchEmbed(nn.Module):
    def __init__(self, height, width, patch_size, in_channels, embed_dim, interpolation_scale=1):
        super().__init__()
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)
        self.patch_size = patch_size
        self.height, self.width = height // patch_size, width // patch_size
        self.base_size = height // patch_size
        self.interpolation_scale = interpolation_scale
        pe = sincos_2d(embed_dim, int((self.height * self.width) ** 0.5), self.base_size, interpolation_scale)
        self.register_buffer("pos_embed", torch.from_numpy(pe).float().unsqueeze(0), persistent=False)

    def forward(self, latent):
        h, w = latent.shape[-2] // self.patch_size, latent.shape[-1] // self.patch_size
        latent = self.proj(latent).flatten(2).transpose(1, 2)
        if (h, w) != (self.height, self.width):
            pe = sincos_2d(self.pos_embed.shape[-1], (h, w), self.base_size, self.interpolation_scale)
            pe = torch.from_numpy(pe).float().unsqueeze(0).to(latent.device)
        else:
            pe = self.pos_embed
        return (latent + pe).to(latent.dtype)


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)

    def forward(self, caption):
        return self.linear_2(F.gelu(self.linear_1(caption), approximate="tanh"))


# ---- diffusers.models.attention (upstream semantics, ada_norm_single only) ------------------------------------------------
FUSED_ATTENTION = False   # set by the generator of the full-width step fixtures only (see Attention.forward)


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_dim=None, bias=True):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        kv = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])

    def forward(self, x, ctx=None, mask=None):
        ctx = x if ctx is None else ctx
        B, S, _ = x.shape
        H = self.heads
        q = self.to_q(x).view(B, S, H, -1).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], H, -1).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], H, -1).transpose(1, 2)
        if FUSED_ATTENTION and mask is None:   # host memory only (the full-width step fixtures, oracle/make_golden.py fullstep)
            o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False, scale=self.scale)
            return self.to_out[0](o.transpose(1, 2).reshape(B, S, -1))
        s = (q @ k.transpose(-1, -2)) * self.scale
        if mask is not None:            # additive bias [B, 1, L] -> every head and query
            s = s + mask[:, None]
        o = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, S, -1)
        return self.to_out[0](o)


class GELUProj(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GELUProj(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class AdaSingleBlock(nn.Module):
    """BasicTransformerBlock with norm_type="ada_norm_single": no norm before the cross-attention, ``norm2`` (non-affine)
    in front of the feed-forward, six modulation vectors = scale_shift_table + the shared adaLN-single projection."""

    def __init__(self, dim, heads, dim_head, cross_dim, bias, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=eps)
        self.attn1 = Attention(dim, heads, dim_head, None, bias)
        self.norm2 = nn.LayerNorm(dim, eps, elementwise_affine=False)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim, bias)
        self.ff = FeedForward(dim)
        self.scale_shift_table = nn.Parameter(torch.randn(6, dim) / dim ** 0.5)

    def forward(self, h, ctx, ctx_mask, timestep):
        B = h.shape[0]
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + timestep.reshape(B, 6, -1)).chunk(6, dim=1)
        n = self.norm1(h) * (1 + scale_msa) + shift_msa
        h = gate_msa * self.attn1(n) + h
        h = self.attn2(h, ctx, ctx_mask) + h
        n = self.norm2(h) * (1 + scale_mlp) + shift_mlp
        return gate_mlp * self.ff(n) + h


class Transformer2DModelRef(nn.Module):
    """diffusers' Transformer2DModel restricted to what examples/train_flash_pixart.py:63-86 instantiates.  Keyword names are
    diffusers'; unsupported values raise."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None, num_layers=1,
                 cross_attention_dim=None, attention_bias=False, sample_size=None, patch_size=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, norm_type="layer_norm", norm_elementwise_affine=True,
                 norm_eps=1e-5, caption_channels=None, interpolation_scale=None, **unused):
        super().__init__()
        assert patch_size is not None and norm_type == "ada_norm_single" and activation_fn == "gelu-approximate" \
            and not norm_elementwise_affine, "only the PixArt-alpha configuration is restated"
        for k, v in unused.items():
            assert v in (None, False, 0, 0.0, "default"), f"unsupported Transformer2DModel argument {k}={v!r}"
        inner = num_attention_heads * attention_head_dim
        out_channels = in_channels if out_channels is None else out_channels
        self.config = SimpleNamespace(norm_type=norm_type, patch_size=patch_size, sample_size=sample_size,
                                      in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                                      num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                                      cross_attention_dim=cross_attention_dim, caption_channels=caption_channels)
        self.patch_size, self.out_channels, self.inner_dim = patch_size, out_channels, inner
        scale = interpolation_scale if interpolation_scale is not None else max(sample_size // 64, 1)
        self.pos_embed = PatchEmbed(sample_size, sample_size, patch_size, in_channels, inner, scale)
        self.transformer_blocks = nn.ModuleList(
            [AdaSingleBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim, attention_bias, norm_eps)
             for _ in range(num_layers)])
        self.norm_out = nn.LayerNorm(inner, elementwise_affine=False, eps=1e-6)
        self.scale_shift_table = nn.Parameter(torch.randn(2, inner) / inner ** 0.5)
        self.proj_out = nn.Linear(inner, patch_size * patch_size * out_channels)
        self.adaln_single = None          # the reference wrapper installs its own AdaLayerNormSingle (TW:40-47)
        self.caption_projection = PixArtAlphaTextProjection(caption_channels, inner) if caption_channels is not None else None

    def forward(self, hidden_states, timestep=None, encoder_hidden_states=None, encoder_attention_mask=None,
                added_cond_kwargs=None, **unused):
        if encoder_attention_mask is not None and encoder_attention_mask.ndim == 2:
            encoder_attention_mask = ((1 - encoder_attention_mask.to(hidden_states.dtype)) * -10000.0).unsqueeze(1)
        p = self.patch_size
        height, width = hidden_states.shape[-2] // p, hidden_states.shape[-1] // p
        h = self.pos_embed(hidden_states)
        B = h.shape[0]
        timestep, embedded_timestep = self.adaln_single(timestep, added_cond_kwargs, batch_size=B, hidden_dtype=h.dtype)
        if self.caption_projection is not None:
            encoder_hidden_states = self.caption_projection(encoder_hidden_states).view(B, -1, h.shape[-1])
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states, encoder_attention_mask, timestep)
        shift, scale = (self.scale_shift_table[None] + embedded_timestep[:, None]).chunk(2, dim=1)
        h = self.norm_out(h) * (1 + scale) + shift
        h = self.proj_out(h)
        h = h.reshape(-1, height, width, p, p, self.out_channels)
        h = torch.einsum("nhwpqc->nchpwq", h)
        return SimpleNamespace(sample=h.reshape(-1, self.out_channels, height * p, width * p))


# ---- the reference's own classes, restated (pinned bit-identically against the real ones) -----------------------------------
class AdaLayerNormSingleRef(nn.Module):
    """TU:8-102: timestep embedding (+ vector conditioning: one embedder, or one per chunk concatenated) -> SiLU -> Linear 6D"""

    def __init__(self, time_embed_dim, timesteps_embedding_num_channels=256, projection_class_embeddings_input_dim=None,
                 use_concat_conditioning=False, num_vector_conditionings=None):
        super().__init__()
        self.time_proj = Timesteps(timesteps_embedding_num_channels, True, 0)                           # TU:37-41
        self.timestep_embedder = TimestepEmbedding(timesteps_embedding_num_channels, time_embed_dim)     # TU:42-44
        self.vdim = projection_class_embeddings_input_dim
        self.n_vec = num_vector_conditionings
        if self.vdim is not None:                                                                        # TU:51-70
            if not use_concat_conditioning:
                self.add_embedding = TimestepEmbedding(self.vdim, time_embed_dim)
            else:
                assert num_vector_conditionings is not None
                self.add_embedding = nn.ModuleList([TimestepEmbedding(self.vdim, time_embed_dim // num_vector_conditionings)
                                                    for _ in range(num_vector_conditionings)])
        self.linear = nn.Linear(time_embed_dim, 6 * time_embed_dim, bias=True)                           # TU:73

    def forward(self, timestep, added_cond_kwargs=None, *args, **kwargs):
        emb = self.timestep_embedder(self.time_proj(timestep.reshape(-1)).to(timestep.device))           # TU:82-85
        if self.vdim is not None:
            vec = added_cond_kwargs.get("vector_conditioning", None)
            if isinstance(self.add_embedding, nn.ModuleList):                                            # TU:89-100
                chunks = torch.chunk(vec, self.n_vec, dim=1)
                emb = emb + torch.cat([self.add_embedding[i](chunks[i]) for i in range(len(self.add_embedding))], dim=1)
            else:
                emb = emb + self.add_embedding(vec)
        return self.linear(F.silu(emb)), emb                                                             # TU:102


class PixartTransformerRef(Transformer2DModelRef):
    """TW:9-100: the denoiser call contract (conditioning dict, optional concat conditioning, output sliced to the input
    channels) over the PixArt transformer with the reference's AdaLayerNormSingle."""

    def __init__(self, time_embed_dim=256, timesteps_embedding_num_channels=256, projection_class_embeddings_input_dim=None,
                 use_concat_vector_conditioning=False, num_vector_conditionings=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.adaln_single = AdaLayerNormSingleRef(time_embed_dim, timesteps_embedding_num_channels,          # TW:40-47
                                                  projection_class_embeddings_input_dim,
                                                  use_concat_vector_conditioning, num_vector_conditionings)

    def forward(self, sample, timestep, conditioning: Dict[str, torch.Tensor], hidden_states_masks=None, *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"                      # TW:69
        c = conditioning["cond"]
        vector, crossattn, concat, mask = c.get("vector"), c.get("crossattn"), c.get("concat"), c.get("attention_mask")
        ch = sample.shape[1]
        if concat is not None:                                                                           # TW:79-80
            sample = torch.cat([sample, concat], dim=1)
        out = super().forward(hidden_states=sample, timestep=timestep, encoder_hidden_states=crossattn,
                              encoder_attention_mask=mask, added_cond_kwargs={"vector_conditioning": vector})
        return out.sample[:, :ch]                                                                        # TW:91

    def freeze(self):                                                                                    # TW:94-100
        self.eval()
        for p in self.parameters():
            p.requires_grad = False


PIXART_XL_2 = dict(sample_size=128, num_layers=28, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2,
                   attention_bias=True, num_attention_heads=16, cross_attention_dim=1152,
                   activation_fn="gelu-approximate", num_embeds_ada_norm=1000, norm_type="ada_norm_single",
                   norm_elementwise_affine=False, norm_eps=1e-6, caption_channels=4096,
                   projection_class_embeddings_input_dim=256, time_embed_dim=1152, timesteps_embedding_num_channels=256,
                   use_concat_vector_conditioning=True, num_vector_conditionings=3)   # examples/train_flash_pixart.py:63-86

TINY_DIT = dict(sample_size=16, num_layers=2, attention_head_dim=8, in_channels=4, out_channels=8, patch_size=2,
                attention_bias=True, num_attention_heads=4, cross_attention_dim=32, activation_fn="gelu-approximate",
                num_embeds_ada_norm=1000, norm_type="ada_norm_single", norm_elementwise_affine=False, norm_eps=1e-6,
                caption_channels=48, projection_class_embeddings_input_dim=16, time_embed_dim=32,
                timesteps_embedding_num_channels=16, use_concat_vector_conditioning=True, num_vector_conditionings=2)


def seeded_init_(module: nn.Module, seed: int, std_scale: float = 1.0):
    """deterministic, non-degenerate weights (biases and tables non-zero so every term is exercised)"""
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        if p.dim() > 1 and "scale_shift_table" not in name:
            fan_in = int(math.prod(p.shape[1:]))
            p.data.copy_(torch.randn(p.shape, generator=g) * std_scale * fan_in ** -0.5)
        else:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return module


# ---- peft-0.9 style LoRA on the oracle (examples/train_flash_pixart.py:237-256) ---------------------------------------------
class LoraConv2d(nn.Module):
    """peft's Conv2d LoRA: A = Conv2d(in, r, k, stride, padding) without bias, B = Conv2d(r, out, 1); scale alpha/r = 1"""

    def __init__(self, base: nn.Conv2d, r: int):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.ModuleDict({"default": nn.Conv2d(base.in_channels, r, base.kernel_size, base.stride, base.padding,
                                                          bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Conv2d(r, base.out_channels, 1, 1, bias=False)})

    def forward(self, x):
        return self.base_layer(x) + self.lora_B["default"](self.lora_A["default"](x))


PIXART_LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2", "proj",
                       "linear", "linear_1", "linear_2")      # examples/train_flash_pixart.py:240-253


def add_lora_(model: nn.Module, r: int, targets=PIXART_LORA_TARGETS, seed: int = 0, b_std: float = 0.02):
    """get_peft_model semantics: freeze everything, wrap every Linear / Conv2d whose dotted name ends in a target suffix;
    A ~ N(0, 1/r) ("gaussian" init), B ~ N(0, b_std) (peft: 0 -- non-zero here so the LoRA gradients are exercised)."""
    from .unet_cpu import LoraLinear
    for p in model.parameters():
        p.requires_grad = False
    g = torch.Generator().manual_seed(seed)
    names = [n for n, m in model.named_modules() if isinstance(m, (nn.Linear, nn.Conv2d))
             and any(n == t or n.endswith("." + t) for t in targets)]
    for n in names:
        parent = model
        parts = n.split(".")
        for q in parts[:-1]:
            parent = getattr(parent, q) if not q.isdigit() else parent[int(q)]
        leaf = parts[-1]
        base = parent[int(leaf)] if leaf.isdigit() else getattr(parent, leaf)
        wrapped = LoraLinear(base, r) if isinstance(base, nn.Linear) else LoraConv2d(base, r)
        a, b = wrapped.lora_A["default"].weight, wrapped.lora_B["default"].weight
        a.data.copy_(torch.randn(a.shape, generator=g) / r)
        b.data.copy_(torch.randn(b.shape, generator=g) * b_std)
        if leaf.isdigit():
            parent[int(leaf)] = wrapped
        else:
            setattr(parent, leaf, wrapped)
    return names
