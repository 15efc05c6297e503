"""TEST INFRASTRUCTURE (oracle) -- PyTorch fp32 restatement of diffusers' UNet2DConditionModel
as used behind the reference wrapper DiffusersUNet2DCondWrapper
(/root/reference/src/flash/models/unets/unet.py:47-127, "UW").

The arithmetic lives in third-party diffusers (un-vendored fork branch,
requirements.txt:1; absent from this container, no network) -- PARITY UNPINNED by
the reference.  This file restates the published upstream algorithm for the
hyper-parameters the reference pins in-tree:
  SD1.5: examples/train_flash_sd.py:56-114 (+ state_dict key names 119-152)
  SDXL : examples/train_flash_sdxl.py:66-118
Module / parameter names equal diffusers' state_dict keys so the same weights load
into this oracle and into the HIP path.

Fork-only ``return_intermediate`` (UW:116): assumed to return the mid-block output
(SURVEY.md section 7 "Hard parts": the discriminators of train_flash_sd.py:225-239 and
train_flash_sdxl.py:242-266 map exactly 1280 x H/8 (resp. H/4) features to 1 logit).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Sequence[int] = (320, 640, 1280, 1280)
    down_block_types: Sequence[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                       "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Sequence[str] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                                     "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    transformer_layers_per_block: Union[int, Sequence[int]] = 1
    attention_head_dim: Union[int, Sequence[int]] = 8  # = NUMBER OF HEADS (diffusers quirk)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    class_embed_type: Optional[str] = None            # None | "projection"
    projection_class_embeddings_input_dim: Optional[int] = None
    flip_sin_to_cos: bool = True
    freq_shift: int = 0

    def heads(self):
        a = self.attention_head_dim
        return [a] * len(self.block_out_channels) if isinstance(a, int) else list(a)

    def tlayers(self):
        t = self.transformer_layers_per_block
        return [t] * len(self.block_out_channels) if isinstance(t, int) else list(t)


def sd15_config():
    return UNetConfig()


def sdxl_config():
    return UNetConfig(block_out_channels=(320, 640, 1280),
                      down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                      up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                      cross_attention_dim=2048, transformer_layers_per_block=(1, 2, 10),
                      attention_head_dim=(5, 10, 20), class_embed_type="projection",
                      projection_class_embeddings_input_dim=2816)


def tiny_config():
    """Small SD1.5-shaped UNet used for golden fixtures / fast parity tests."""
    return UNetConfig(block_out_channels=(32, 64, 64, 64), cross_attention_dim=64,
                      attention_head_dim=2)


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class LoraLinear(nn.Module):
    """peft-0.9 style LoRA wrapper (setup.py:37; injected at examples/train_flash_sd.py:191-200):
    y = base(x) + (alpha/r) * B(A(x)), alpha = r -> scale 1."""

    def __init__(self, base: nn.Linear, r: int):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        self.scaling = 1.0

    @property
    def weight(self):
        return self.base_layer.weight

    def forward(self, x):
        return self.base_layer(x) + self.scaling * self.lora_B["default"](self.lora_A["default"](x))


FUSED_ATTENTION = False   # set by the generator of the large-batch fixture only (see Attention.forward)


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        kv = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])

    def forward(self, x, ctx=None, mask=None):
        ctx = x if ctx is None else ctx
        B, S, _ = x.shape
        H = self.heads
        q = self.to_q(x).view(B, S, H, -1).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], H, -1).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], H, -1).transpose(1, 2)
        if FUSED_ATTENTION and mask is None:
            # memory only: the B = 16 fixture (oracle/make_golden.py c2 16) cannot keep 16 x 5 materialised [8, 4096, 4096]
            # probability tensors for the backward in 62 GB of host memory; PyTorch's fp32 CPU kernel keeps the log-sum-exp
            o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False, scale=self.scale)
            return self.to_out[0](o.transpose(1, 2).reshape(B, S, -1))
        s = (q @ k.transpose(-1, -2)) * self.scale
        if mask is not None:
            s = s + mask
        p = s.softmax(dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, S, -1)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, h, ctx):
        h = h + self.attn1(self.norm1(h))
        h = h + self.attn2(self.norm2(h), ctx)
        h = h + self.ff(self.norm3(h))
        return h


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, num_layers, cross_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_dim) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        h = self.proj_out(h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + res


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        sc = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return sc + h


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, n, groups, eps, add_down, attn, heads, tl, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb_ch, groups, eps) for i in range(n)])
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, cout // heads, cout, tl, cross_dim, groups) for _ in range(n)])
        self.has_attn = attn
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.add_down = add_down

    def forward(self, h, temb, ctx, additional_residuals=None):
        """`additional_residuals` (T2I adapter): CrossAttnDownBlock2D adds it after its last (resnet, attention) pair, before
        the state is recorded as a skip and before the downsampler (upstream diffusers)"""
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, ctx)
                if i == len(self.resnets) - 1 and additional_residuals is not None:
                    h = h + additional_residuals
            outs.append(h)
        if self.add_down:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb_ch, groups, eps, heads, tl, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_ch, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, c // heads, c, tl, cross_dim, groups)])

    def forward(self, h, temb, ctx):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, ctx)
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb_ch, n, groups, eps, add_up, attn, heads, tl, cross_dim):
        super().__init__()
        rs = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb_ch, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, cout // heads, cout, tl, cross_dim, groups) for _ in range(n)])
        self.has_attn = attn
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.add_up = add_up

    def forward(self, h, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            h = torch.cat([h, skips.pop()], dim=1)
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, ctx)
        if self.add_up:
            h = self.upsamplers[0](h)
        return h


class UNet2DConditionRef(nn.Module):
    """Obeys the wrapper call contract UW:66-119."""

    def __init__(self, cfg: UNetConfig = None):
        super().__init__()
        cfg = cfg or UNetConfig()
        self.cfg = cfg
        boc = list(cfg.block_out_channels)
        temb_ch = boc[0] * 4
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        heads, tls = cfg.heads(), cfg.tlayers()
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_ch)
        if cfg.class_embed_type == "projection":
            self.class_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb_ch)
        else:
            self.class_embedding = None
        downs = []
        out_ch = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            downs.append(DownBlock(in_ch, out_ch, temb_ch, cfg.layers_per_block, g, eps,
                                   i != len(boc) - 1, t.startswith("CrossAttn"), heads[i], tls[i],
                                   cfg.cross_attention_dim))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb_ch, g, eps, heads[-1], tls[-1], cfg.cross_attention_dim)
        rboc, rheads, rtls = boc[::-1], heads[::-1], tls[::-1]
        ups = []
        out_ch = rboc[0]
        for i, t in enumerate(cfg.up_block_types):
            prev = out_ch
            out_ch = rboc[i]
            in_ch = rboc[min(i + 1, len(boc) - 1)]
            ups.append(UpBlock(in_ch, out_ch, prev, temb_ch, cfg.layers_per_block + 1, g, eps,
                               i != len(boc) - 1, t.startswith("CrossAttn"), rheads[i], rtls[i],
                               cfg.cross_attention_dim))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    # -- wrapper contract (UW:66-119) ---------------------------------------------------
    def forward(self, sample, timestep, conditioning, down_intrablock_additional_residuals=None,
                return_intermediate=False, *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"
        class_labels = conditioning["cond"].get("vector", None)
        ctx = conditioning["cond"].get("crossattn", None)
        concat = conditioning["cond"].get("concat", None)
        if concat is not None:
            sample = torch.cat([sample, concat], dim=1)
        res = None
        if down_intrablock_additional_residuals is not None:       # UW:100-106: cloned, "since unet will modify it"
            res = [r.clone() for r in down_intrablock_additional_residuals]
        return self.unet_forward(sample, timestep, ctx, class_labels, return_intermediate, res)

    def unet_forward(self, sample, timestep, ctx, class_labels=None, return_intermediate=False, down_residuals=None):
        cfg = self.cfg
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.float32 if isinstance(timestep, float)
                                    else torch.int64, device=sample.device)
        elif timestep.ndim == 0:
            timestep = timestep[None]
        timestep = timestep.to(sample.device).expand(sample.shape[0])
        t_emb = timestep_embedding(timestep, cfg.block_out_channels[0], cfg.flip_sin_to_cos,
                                   cfg.freq_shift).to(sample.dtype)
        emb = self.time_embedding(t_emb)
        if self.class_embedding is not None:
            emb = emb + self.class_embedding(class_labels.to(sample.dtype))
        h = self.conv_in(sample)
        skips = [h]
        # T2I-adapter residuals (upstream UNet2DConditionModel.forward, is_adapter branch): one per down block, popped in
        # order; a block without attention gets it added IN PLACE to its output -- the same tensor object as the block's
        # last skip, which therefore carries it too
        down_residuals = list(down_residuals) if down_residuals is not None else []
        for blk in self.down_blocks:
            if blk.has_attn:
                extra = down_residuals.pop(0) if down_residuals else None
                h, outs = blk(h, emb, ctx, extra)
            else:
                h, outs = blk(h, emb, ctx)
                if down_residuals:
                    h += down_residuals.pop(0)
            skips.extend(outs)
        h = self.mid_block(h, emb, ctx)
        if return_intermediate:
            return h
        for blk in self.up_blocks:
            h = blk(h, skips, emb, ctx)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return h

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    # -- LoRA (peft semantics; examples/train_flash_sd.py:191-200) -----------------------
    def add_adapter(self, r, target_modules=("to_k", "to_q", "to_v", "to_out.0")):
        for name, mod in list(self.named_modules()):
            if not isinstance(mod, Attention):
                continue
            for t in target_modules:
                if t == "to_out.0":
                    if not isinstance(mod.to_out[0], LoraLinear):
                        mod.to_out[0] = LoraLinear(mod.to_out[0], r)
                elif not isinstance(getattr(mod, t), LoraLinear):
                    setattr(mod, t, LoraLinear(getattr(mod, t), r))
        for n, p in self.named_parameters():
            p.requires_grad = "lora_" in n
        return self


def seeded_init_(module: nn.Module, seed: int, lora_b_std: float = 0.02):
    """Deterministic, platform-stable init (numpy legacy RandomState, frozen stream):
    weights ~ N(0, 1/fan_in) (x0.7 so deep nets keep O(1) activations), biases ~ N(0, 0.02),
    norm gamma = 1 + 0.1 n, beta = 0.05 n, LoRA A ~ N(0, 1/r) (peft 'gaussian'),
    LoRA B ~ N(0, lora_b_std) (non-zero so LoRA grads are exercised)."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            shape = tuple(p.shape)
            n = rs.standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape)
            if "lora_A" in name:
                v = n / shape[0]
            elif "lora_B" in name:
                v = n * lora_b_std
            elif "norm" in name and name.endswith("weight"):
                v = 1.0 + 0.1 * n
            elif "norm" in name and name.endswith("bias"):
                v = 0.05 * n
            elif name.endswith("bias"):
                v = 0.02 * n
            else:
                fan_in = int(np.prod(shape[1:]))
                v = n * (0.7 / math.sqrt(fan_in))
            p.copy_(torch.from_numpy(v))
    return module


def make_discriminator(kind="sd15", color_dim=1280, feat=64, last_k=4):
    """Example discriminators: examples/train_flash_sd.py:225-240 ("sd15"),
    tests/test_flash/test_flash_diffusion.py:100-113 ("test")."""
    if kind == "sd15":
        return nn.Sequential(
            nn.Conv2d(color_dim, feat, 3, 1, 1), nn.SiLU(True),
            nn.Conv2d(feat, feat * 2, 4, 2, 1, bias=False), nn.SiLU(True),
            nn.GroupNorm(4, feat * 2),
            nn.Conv2d(feat * 2, 1, last_k, 1, 0, bias=False), nn.Flatten())
    if kind == "test":
        return nn.Sequential(
            nn.Conv2d(color_dim, feat, 4, 2, 1, bias=False), nn.SiLU(True),
            nn.Conv2d(feat, 1, 4, 1, 0, bias=False), nn.Flatten())
    raise ValueError(kind)


class TinyT2IAdapter(nn.Module):
    """Test stand-in for DiffusersT2IAdapterWrapper (adapters/t2i_adapter.py:7-27; diffusers' T2IAdapter is absent): a
    frozen conv pyramid mapping a control image [B, c, H, W] (H, W = latent size) to one residual per UNet down block, each
    with the shape of that block's output at the point diffusers adds it -- level i at H >> i (every earlier block halves
    the resolution; the last block has no downsampler)."""

    def __init__(self, unet_cfg, in_channels=1, seed=7):
        super().__init__()
        n = len(unet_cfg.block_out_channels)
        self.convs = nn.ModuleList([nn.Conv2d(in_channels, c, 3, 1, 1) for c in unet_cfg.block_out_channels])
        self.n = n
        g = torch.Generator().manual_seed(seed)
        for p in self.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.2)
            p.requires_grad = False

    def forward(self, cond):
        out = []
        for i, c in enumerate(self.convs):
            x = F.avg_pool2d(cond, 1 << i) if i else cond
            out.append(c(x))
        return out


class TinyVAE(nn.Module):
    """Test stand-in for flash.models.vae.AutoencoderKLDiffusers (vae/autoencoderKL.py:11-128; diffusers' AutoencoderKL and its
    pretrained weights are absent): exactly the surface FlashDiffusion touches -- `config.input_key`, `latent_channels`,
    `downsampling_factor`, `encode(images) -> latents * scaling_factor` (deterministic here: the posterior mean),
    `decode(latents) -> images` with the division by the scaling factor (autoencoderKL.py:78-79) -- around a frozen
    two-conv encoder / decoder."""

    def __init__(self, latent_channels=4, downsampling_factor=2, input_key="image", scaling_factor=0.18215, seed=21):
        super().__init__()
        from types import SimpleNamespace
        self.config = SimpleNamespace(input_key=input_key)
        self.latent_channels, self.downsampling_factor, self.scaling_factor = latent_channels, downsampling_factor, scaling_factor
        f = downsampling_factor
        self.enc = nn.Conv2d(3, latent_channels, f, f)
        self.dec1 = nn.Conv2d(latent_channels, 12, 3, 1, 1)
        self.dec2 = nn.Conv2d(12, 3, 3, 1, 1)
        g = torch.Generator().manual_seed(seed)
        for p in self.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1, p[0].numel()) ** 0.5 if p.dim() > 1 else 0.05))
            p.requires_grad = False

    def encode(self, x):
        return self.enc(x) * self.scaling_factor

    def decode(self, z):
        h = F.interpolate(z / self.scaling_factor, scale_factor=float(self.downsampling_factor), mode="nearest")
        return 0.6 * self.dec2(F.silu(self.dec1(h)))   # (for unit-variance latents ~10 % of the pixels leave [-1, 1]: the clamp of FD:394 acts)


class TinyLPIPS(nn.Module):
    """Test stand-in for lpips.LPIPS(net="vgg") (lpips==0.1.4, setup.py:40; package and VGG16 weights absent), same structure
    at toy width: fixed input shift / scale, a ReLU conv feature pyramid with 2x2 max-pools between its slices, per-layer
    channel-unit-normalisation, squared difference, a non-negative 1x1 `lin` head, spatial mean, sum over layers ->
    [B, 1, 1, 1].  Frozen (the reference never trains it)."""

    def __init__(self, widths=(8, 12, 16), seed=22):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        chans = (3,) + tuple(widths)
        self.slices = nn.ModuleList([nn.Conv2d(chans[i], chans[i + 1], 3, 1, 1) for i in range(len(widths))])
        self.lins = nn.ModuleList([nn.Conv2d(c, 1, 1, bias=False) for c in widths])
        g = torch.Generator().manual_seed(seed)
        for p in self.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(1, p[0].numel()) ** 0.5 if p.dim() > 1 else 0.05))
            p.requires_grad = False
        for l in self.lins:
            l.weight.data.abs_()

    def forward(self, in0, in1):
        f0, f1 = (in0 - self.shift) / self.scale, (in1 - self.shift) / self.scale
        total = 0
        for i, (conv, lin) in enumerate(zip(self.slices, self.lins)):
            if i:
                f0, f1 = F.max_pool2d(f0, 2), F.max_pool2d(f1, 2)
            f0, f1 = F.relu(conv(f0)), F.relu(conv(f1))
            n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + 1e-10)
            total = total + lin((n0 - n1) ** 2).mean([2, 3], keepdim=True)
        return total
