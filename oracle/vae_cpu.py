"""TEST INFRASTRUCTURE (oracle) -- CPU fp32 restatement of the networks the LPIPS distillation loss runs
(/root/reference/src/flash/models/flash/flash_diffusion_model.py:383-397) and of the T2I adapter
(/root/reference/src/flash/models/adapters/t2i_adapter.py:7-26):

* ``AutoencoderKLDecoderRef``: diffusers ``AutoencoderKL.decode`` (post_quant_conv + ``Decoder``: conv_in, UNetMidBlock2D with one
  single-head attention, UpDecoderBlock2D x n with layers_per_block + 1 ResnetBlock2D(temb_channels=None, eps=1e-6) and a
  nearest-2x Upsample2D conv, GroupNorm + SiLU + conv_out) -- what vae/autoencoderKL.py:126 calls;
* ``LPIPSRef``: lpips.LPIPS(net="vgg") of lpips==0.1.4 (setup.py:40): ScalingLayer, torchvision VGG16 features sliced at
  relu1_2 / 2_2 / 3_3 / 4_3 / 5_3, normalize_tensor (eps 1e-10), squared difference, NetLinLayer (1x1, no bias), spatial mean;
* ``T2IAdapterRef``: diffusers ``T2IAdapter`` full_adapter / full_adapter_xl (examples/train_flash_canny_adapter.py:182-188).

diffusers, lpips and torchvision are absent from this container and the reference's tests hold no golden vectors for
them: PARITY UNPINNED for this file -- it restates the published upstream semantics; parameter names are the upstream state_dict
keys so that real checkpoints map one to one."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2DRef(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class VaeAttentionRef(nn.Module):
    """diffusers Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups, bias=True, eps=1e-6)"""

    def __init__(self, C, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, C, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(C, C), nn.Linear(C, C), nn.Linear(C, C)
        self.to_out = nn.ModuleList([nn.Linear(C, C), nn.Identity()])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
        o = self.to_out[0](a @ v).transpose(1, 2).reshape(B, C, H, W)
        return o + x


class _Mid(nn.Module):
    def __init__(self, C, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2DRef(C, C, groups, eps), ResnetBlock2DRef(C, C, groups, eps)])
        self.attentions = nn.ModuleList([VaeAttentionRef(C, groups, eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Up(nn.Module):
    def __init__(self, cin, cout, n, add_up, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2DRef(cin if j == 0 else cout, cout, groups, eps) for j in range(n)])
        if add_up:
            up = nn.Module()
            up.conv = nn.Conv2d(cout, cout, 3, 1, 1)
            self.upsamplers = nn.ModuleList([up])
        else:
            self.upsamplers = None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class _Decoder(nn.Module):
    def __init__(self, latent, out_ch, boc, layers_per_block, groups, eps):
        super().__init__()
        top = boc[-1]
        self.conv_in = nn.Conv2d(latent, top, 3, 1, 1)
        self.mid_block = _Mid(top, groups, eps)
        ups, prev = [], top
        for i, co in enumerate(reversed(boc)):
            ups.append(_Up(prev, co, layers_per_block + 1, i != len(boc) - 1, groups, eps))
            prev = co
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, prev, eps=eps)
        self.conv_out = nn.Conv2d(prev, out_ch, 3, 1, 1)

    def forward(self, z):
        h = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            h = u(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class AutoencoderKLDecoderRef(nn.Module):
    def __init__(self, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                 norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _Decoder(latent_channels, out_channels, list(block_out_channels), layers_per_block, norm_num_groups, 1e-6)

    def decode_raw(self, z):
        """AutoencoderKL.decode(z).sample"""
        return self.decoder(self.post_quant_conv(z))

    def decode(self, z):
        """the reference wrapper's decode (vae/autoencoderKL.py:63-128, no latents_mean / std): z / scaling_factor first"""
        return self.decode_raw(z / self.scaling_factor)


# ---- LPIPS --------------------------------------------------------------------------------------------------------------------
_VGG_CFG = [(1, 0, 64, 3), (1, 2, 64, 64), (2, 5, 128, 64), (2, 7, 128, 128), (3, 10, 256, 128), (3, 12, 256, 256), (3, 14, 256, 256),
            (4, 17, 512, 256), (4, 19, 512, 512), (4, 21, 512, 512), (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]


class _Vgg16Slices(nn.Module):
    """lpips.pretrained_networks.vgg16: torchvision vgg16().features split into five nn.Sequential slices (ReLU and MaxPool modules
    keep their feature indices, so the convolutions are net.slice{k}.{features index})"""

    def __init__(self):
        super().__init__()
        ends = {1: 4, 2: 9, 3: 16, 4: 23, 5: 30}
        pools = {4, 9, 16, 23}
        convs = {idx: (co, ci) for _, idx, co, ci in _VGG_CFG}
        start = 0
        for k in range(1, 6):
            seq = nn.Sequential()
            for i in range(start, ends[k]):
                if i in convs:
                    seq.add_module(str(i), nn.Conv2d(convs[i][1], convs[i][0], 3, 1, 1))
                elif i in pools:
                    seq.add_module(str(i), nn.MaxPool2d(2, 2))
                else:
                    seq.add_module(str(i), nn.ReLU(inplace=False))
            setattr(self, f"slice{k}", seq)
            start = ends[k]

    def forward(self, x):
        outs = []
        for k in range(1, 6):
            x = getattr(self, f"slice{k}")(x)
            outs.append(x)
        return outs


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class _NetLin(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(c, 1, 1, 1, 0, bias=False))


class LPIPSRef(nn.Module):
    def __init__(self):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.net = _Vgg16Slices()
        for l, c in enumerate((64, 128, 256, 512, 512)):
            setattr(self, f"lin{l}", _NetLin(c))
        self.eval()

    @staticmethod
    def _unit(x, eps=1e-10):
        return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)

    def forward(self, in0, in1):
        f0, f1 = self.net(self.scaling_layer(in0)), self.net(self.scaling_layer(in1))
        val = 0
        for l in range(5):
            d = (self._unit(f0[l]) - self._unit(f1[l])) ** 2
            val = val + getattr(self, f"lin{l}").model(d).mean([2, 3], keepdim=True)
        return val


# ---- T2I adapter ----------------------------------------------------------------------------------------------------------------
class _AdapterResnet(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block1 = nn.Conv2d(c, c, 3, padding=1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(c, c, 1)

    def forward(self, x):
        return self.block2(self.act(self.block1(x))) + x


class _AdapterBlock(nn.Module):
    def __init__(self, cin, cout, n, down):
        super().__init__()
        self.downsample = nn.AvgPool2d(2, 2, ceil_mode=True) if down else None
        self.in_conv = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.resnets = nn.Sequential(*[_AdapterResnet(cout) for _ in range(n)])

    def forward(self, x):
        if self.downsample is not None:
            x = self.downsample(x)
        if self.in_conv is not None:
            x = self.in_conv(x)
        return self.resnets(x)


class _FullAdapter(nn.Module):
    def __init__(self, in_channels, channels, num_res_blocks, downscale_factor, xl):
        super().__init__()
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.conv_in = nn.Conv2d(in_channels * downscale_factor ** 2, channels[0], 3, padding=1)
        body = []
        for i, co in enumerate(channels):
            ci = channels[0] if i == 0 else channels[i - 1]
            body.append(_AdapterBlock(ci, co, num_res_blocks, (i == 2) if xl else (i > 0)))
        self.body = nn.ModuleList(body)

    def forward(self, x):
        x = self.conv_in(self.unshuffle(x))
        feats = []
        for b in self.body:
            x = b(x)
            feats.append(x)
        return feats


class T2IAdapterRef(nn.Module):
    def __init__(self, in_channels=3, channels=(320, 640, 1280, 1280), num_res_blocks=2, downscale_factor=8,
                 adapter_type="full_adapter"):
        super().__init__()
        self.adapter = _FullAdapter(in_channels, list(channels), num_res_blocks, downscale_factor, adapter_type == "full_adapter_xl")

    def forward(self, x):
        return self.adapter(x)


def seeded_net_init_(module: nn.Module, seed: int):
    """deterministic weights that keep activations O(1): He-style conv / linear weights, small biases, GroupNorm gains around
    1, non-negative LPIPS lin weights (upstream clamps them at 0)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            n = torch.randn(p.shape, generator=g)
            if name.startswith("lin") and ".model." in name:
                p.copy_(n.abs() * 0.1)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * n)
            elif "norm" in name:
                p.copy_(0.05 * n)
            elif p.dim() == 1:
                p.copy_(0.05 * n)
            else:
                fan_in = int(math.prod(p.shape[1:]))
                p.copy_(n * math.sqrt(2.0 / fan_in) if "net.slice" in name or "adapter" in name else n * (0.8 / math.sqrt(fan_in)))
    return module
