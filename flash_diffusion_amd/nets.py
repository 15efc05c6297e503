"""The frozen convolutional networks beside the denoiser on the HIP path:

* ``MiAutoencoderKL`` / ``MiAutoencoderKLDiffusers`` -- drop-in for the reference's VAE wrapper ``AutoencoderKLDiffusers``
  (/root/reference/src/flash/models/vae/autoencoderKL.py:9-128) as the distillation step uses it: ``decode(z)`` of both
  outputs' centre crops inside ``_distill_loss`` (flash_diffusion_model.py:383-397), with the input gradient of the student's
  decode.  Parameters carry diffusers' ``AutoencoderKL`` state_dict names (``post_quant_conv.*``, ``decoder.*``).
* ``MiLPIPS`` -- drop-in for ``lpips.LPIPS(net="vgg")`` (flash_diffusion_model.py:102-103; lpips==0.1.4, setup.py:40):
  ``forward(in0, in1) -> [N, 1, 1, 1]`` with the gradient with respect to ``in0``; state_dict names of lpips
  (``net.slice1.0.weight`` ... ``lin4.model.1.weight``, buffers ``scaling_layer.shift / scale``).
* ``MiT2IAdapter`` -- drop-in for ``DiffusersT2IAdapterWrapper`` (adapters/t2i_adapter.py:7-26; diffusers ``T2IAdapter`` of type
  ``full_adapter`` / ``full_adapter_xl``, examples/train_flash_canny_adapter.py:182-196): frozen, forward only, returns the list
  of feature maps that ``FlashDiffusion`` threads into every denoiser call (flash_diffusion_model.py:207-218).

Each is a plan of libfdmi.so's executor (include/fdmi.h: ``fdmi_net_*``): implicit-GEMM MFMA convolutions, GroupNorm, max /
average pooling, the LPIPS distance kernels, a C++ tape for the input gradient.  torch supplies parameters, device memory, the
stream and the autograd edge around the one forward / backward call.  GPU only: there is no fallback path."""
from __future__ import annotations

import ctypes as C
import math
import weakref
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from ._lib import NetCfg, check, ptr, stream_ptr

FDMI_NET_VAE_DECODER, FDMI_NET_VGG_LPIPS, FDMI_NET_T2I_ADAPTER = 1, 3, 4
FDMI_UNET_SAVE = 1


class _Node(nn.Module):
    """plain container so parameters get their dotted upstream names"""


def _ensure_path(root: nn.Module, parts: Sequence[str]) -> nn.Module:
    m = root
    for p in parts:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    return m


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, x2):
        out, slot = mod._run(x, x2, save=True)
        ctx.mod, ctx.slot, ctx.xshape = mod, slot, x.shape
        # a ctx that outlives its backward (a kept loss, a delayed GC) must not free the slot's NEXT tenant: release by generation
        weakref.finalize(ctx, mod._release, slot, mod._gen.get(slot, 0))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        gx = ctx.mod._run_backward(ctx.slot, grad_out, ctx.xshape)
        return None, gx, None


class _NetBase(nn.Module):
    """parameters by upstream name -> one plan handle (packed lazily, re-packed after load_state_dict), workspace per run slot"""
    precision = "bf16"

    def _init_net(self, shapes, cfg: NetCfg):
        self._shapes = shapes
        self._cfg = cfg
        for name, shape in shapes:
            parts = name.split(".")
            leaf = _ensure_path(self, parts[:-1])
            fan_in = 1
            for s in shape[1:]:
                fan_in *= int(s)
            if "norm" in name and parts[-1] == "weight" and len(shape) == 1:
                v = torch.ones(shape)
            elif len(shape) == 1:
                v = torch.zeros(shape)
            else:
                v = torch.randn(shape) * (fan_in ** -0.5)
            leaf.register_parameter(parts[-1], nn.Parameter(v))
        self._handle = None
        self._packed = False
        self._ws: Dict[int, torch.Tensor] = {}
        self._busy = set()
        self._gen = {}          # per run slot: acquisition counter (a late finalizer only frees its own acquisition)
        self.last_flops = 0.0

    # ---- plan ----
    def _plan(self):
        if self._handle is None:
            h = _lib.lib().fdmi_net_create(C.byref(self._cfg))
            if not h:
                raise RuntimeError("fdmi: " + _lib.lib().fdmi_last_error().decode())
            self._handle = h
            weakref.finalize(self, _lib.lib().fdmi_unet_destroy, h)
        return self._handle

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_handle", "_packed", "_ws", "_busy", "_gen"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._handle, new._packed, new._ws, new._busy, new._gen = None, False, {}, set(), {}
        return new

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = False
        return r

    def invalidate_plan(self):
        """call after changing weights in place"""
        self._packed = False

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _ensure_packed(self):
        h = self._plan()
        L = _lib.lib()
        if not self._packed:
            n = L.fdmi_unet_num_params(h)
            expected = {}
            buf = C.create_string_buffer(512)
            ne = C.c_int64()
            for i in range(n):
                check(L.fdmi_unet_param_name(h, i, buf, 512, C.byref(ne)))
                expected[buf.value.decode()] = ne.value
            mine = dict(self.named_parameters())
            assert set(expected) == set(mine), (set(expected) ^ set(mine))
            for name, p in mine.items():
                assert p.is_cuda and p.dtype == torch.float32, f"{name}: parameters must be fp32 on the GPU"
                assert p.numel() == expected[name], (name, tuple(p.shape), expected[name])
                t = p.detach().contiguous()
                check(L.fdmi_unet_set_param(h, name.encode(), ptr(t), t.numel(), stream_ptr()))
            torch.cuda.current_stream().synchronize()
            check(L.fdmi_unet_ready(h))
            self._packed = True
        return h

    def _release(self, slot, gen=None):
        """free a tape slot; with `gen` only if the slot still belongs to that acquisition (unet._Plan.release's rule)"""
        if gen is None or self._gen.get(slot, 0) == gen:
            self._busy.discard(slot)

    def _workspace(self, slot, need, device):
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws[slot] = ws
        return ws

    def _out_like(self, x):
        raise NotImplementedError

    def _run(self, x, x2, save):
        assert x.is_cuda, f"{type(self).__name__} runs on the GPU only (no CPU fallback)"
        h = self._ensure_packed()
        L = _lib.lib()
        x = x.detach().float().contiguous()
        x2 = None if x2 is None else x2.detach().float().contiguous()
        B, _, H, W = x.shape
        slot = 0
        if save:
            slot = next((s for s in range(1, 8) if s not in self._busy), None)
            if slot is None:
                raise RuntimeError(f"{type(self).__name__}: 7 taped forwards are outstanding (every run slot holds a tape whose "
                                   "backward has not run); call backward() or drop the graphs before taping another forward")
            self._busy.add(slot)
            self._gen[slot] = self._gen.get(slot, 0) + 1
        flags = FDMI_UNET_SAVE if save else 0
        need = L.fdmi_net_workspace_bytes(h, B, H, W, flags)
        if need < 0:
            raise RuntimeError("fdmi: " + L.fdmi_last_error().decode())
        ws = self._workspace(slot, need, x.device)
        out = self._out_like(x)
        check(L.fdmi_net_forward(h, slot, ptr(x), ptr(x2), ptr(out), B, H, W, ptr(ws), ws.numel(), flags, stream_ptr()))
        self.last_flops = L.fdmi_unet_last_flops(h)
        return out, slot

    def _run_backward(self, slot, grad_out, xshape):
        L = _lib.lib()
        g = grad_out.detach().float().contiguous()
        gx = torch.empty(xshape, dtype=torch.float32, device=g.device)
        try:
            check(L.fdmi_net_backward(self._handle, slot, ptr(g), ptr(gx), stream_ptr()))
        finally:
            self._busy.discard(slot)
        return gx

    def _call_net(self, x, x2=None):
        if torch.is_grad_enabled() and x.requires_grad:
            return _NetFn.apply(self, x, x2)
        out, _ = self._run(x, x2, save=False)
        return out


# ====================================================================================================================
# AutoencoderKL (decoder)
# ====================================================================================================================
def vae_decoder_shapes(in_ch, out_ch, boc, layers_per_block):
    """(name, shape) of diffusers AutoencoderKL's post_quant_conv + decoder parameters"""
    out = []

    def conv(n, co, ci, k):
        out.extend([(n + ".weight", (co, ci, k, k)), (n + ".bias", (co,))])

    def norm(n, c):
        out.extend([(n + ".weight", (c,)), (n + ".bias", (c,))])

    def lin(n, co, ci):
        out.extend([(n + ".weight", (co, ci)), (n + ".bias", (co,))])

    def resnet(n, ci, co):
        norm(n + ".norm1", ci)
        conv(n + ".conv1", co, ci, 3)
        norm(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    top = boc[-1]
    conv("post_quant_conv", in_ch, in_ch, 1)
    conv("decoder.conv_in", top, in_ch, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    norm("decoder.mid_block.attentions.0.group_norm", top)
    for t in ("to_q", "to_k", "to_v", "to_out.0"):
        lin("decoder.mid_block.attentions.0." + t, top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    prev = top
    n = len(boc)
    for i in range(n):
        co = boc[n - 1 - i]
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i != n - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("decoder.conv_norm_out", prev)
    conv("decoder.conv_out", out_ch, prev, 3)
    return out


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class MiAutoencoderKL(_NetBase):
    """diffusers ``AutoencoderKL`` as the step uses it: ``decode(z).sample`` (post_quant_conv + Decoder; the mid block's
    attention has one head over all positions).  Keywords follow diffusers' config (SD: block_out_channels (128, 256, 512, 512),
    layers_per_block 2, latent_channels 4, norm_num_groups 32, scaling_factor 0.18215)."""

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, precision="bf16", **unused):
        super().__init__()
        assert precision in ("bf16", "fp32")
        latents_mean, latents_std = unused.pop("latents_mean", None), unused.pop("latents_std", None)
        for k, v in unused.items():
            if k in ("down_block_types", "up_block_types", "act_fn", "sample_size", "force_upcast", "use_quant_conv", "shift_factor"):
                continue          # encoder-side / bookkeeping keys: no effect on decode()
            if (k, v) in (("use_post_quant_conv", True), ("mid_block_add_attention", True)):
                continue          # the defaults: what the plan builds
            # (use_post_quant_conv=False -- the SD3 VAE -- and mid_block_add_attention=False change the decoder's arithmetic:
            # the plan always applies post_quant_conv and the mid-block attention, so these must not be dropped silently)
            raise NotImplementedError(f"{k}={v!r} is outside the VAE decoder plan (post_quant_conv + mid-block attention are always built)")
        boc = list(block_out_channels)
        assert 1 <= len(boc) <= 4
        self.precision = precision
        self.config = type("Cfg", (), dict(scaling_factor=scaling_factor, latent_channels=latent_channels, in_channels=in_channels,
                                           out_channels=out_channels, block_out_channels=boc, layers_per_block=layers_per_block,
                                           norm_num_groups=norm_num_groups, latents_mean=latents_mean, latents_std=latents_std))()
        cfg = NetCfg()
        cfg.kind, cfg.in_channels, cfg.out_channels, cfg.n_levels = FDMI_NET_VAE_DECODER, latent_channels, out_channels, len(boc)
        for i, v in enumerate(boc):
            cfg.block_out[i] = v
        cfg.layers_per_block, cfg.groups, cfg.eps, cfg.precision = layers_per_block, norm_num_groups, 1e-6, int(precision == "fp32")
        self._init_net(vae_decoder_shapes(latent_channels, out_channels, boc, layers_per_block), cfg)
        self._up = 2 ** (len(boc) - 1)

    def _out_like(self, x):
        B, _, H, W = x.shape
        return torch.empty(B, self.config.out_channels, H * self._up, W * self._up, dtype=torch.float32, device=x.device)

    def decode(self, z, return_dict=True):
        out = self._call_net(z)
        return _DecoderOutput(out) if return_dict else (out,)

    def encode(self, x):
        raise NotImplementedError("only the decoder is on the HIP path (the step encodes under no_grad before the hot path, "
                                  "flash_diffusion_model.py:128-133); pass latents, or a torch encoder to MiAutoencoderKLDiffusers")


class MiAutoencoderKLDiffusers(nn.Module):
    """Drop-in for the reference wrapper ``AutoencoderKLDiffusers`` (vae/autoencoderKL.py:9-128): ``decode(z)`` divides by the
    scaling factor (or applies latents_mean / latents_std) and runs the decoder; latents larger than ``tiling_size`` are decoded
    in overlapping tiles merged with the reference's gaussian weights (autoencoderKL.py:80-123, models/utils.py:12-257) -- here
    ALL tiles of ALL samples go through the decoder plan as batches and are merged on the device (the reference decodes tile
    by tile and merges on the host; ``decode`` returns a device tensor).  ``encoder``: an optional torch module with the wrapper's
    ``encode(x) -> latents`` contract for recipes that feed pixels (the encode runs under no_grad, outside the path)."""

    def __init__(self, vae_model: MiAutoencoderKL, input_key="image", tiling_size=(64, 64), tiling_overlap=(16, 16),
                 encoder: Optional[nn.Module] = None, tile_batch: int = 8):
        super().__init__()
        self.vae_model = vae_model
        self.config = type("Cfg", (), dict(input_key=input_key, tiling_size=tuple(tiling_size), tiling_overlap=tuple(tiling_overlap)))()
        self.tiling_size = tuple(tiling_size)
        self.tiling_overlap = tuple(tiling_overlap)
        self.tile_batch = int(tile_batch)           # tiles per decoder launch (bounds the plan's workspace)
        self.encoder = encoder
        self.downsampling_factor = vae_model._up
        self.latent_channels = vae_model.config.latent_channels

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def encode(self, x, batch_size: int = 8):
        if self.encoder is None:
            raise NotImplementedError("MiAutoencoderKLDiffusers was built without an encoder")
        return self.encoder.encode(x)

    def decode(self, z):
        c = self.vae_model.config
        if c.latents_mean is not None and c.latents_std is not None:
            mean = torch.tensor(c.latents_mean).view(1, -1, 1, 1).to(z.device, z.dtype)
            std = torch.tensor(c.latents_std).view(1, -1, 1, 1).to(z.device, z.dtype)
            z = z * std / c.scaling_factor + mean
        else:
            z = z / c.scaling_factor
        if z.shape[2] > self.tiling_size[0] or z.shape[3] > self.tiling_size[1]:
            return self._decode_tiled(z)
        return self.vae_model.decode(z).sample

    @staticmethod
    def _gaussian_weights(tile_w, tile_h, device):
        """models/utils.py:155-201 (fp64, as numpy computes them there); the x midpoint is (w - 1) / 2, the y midpoint h / 2"""
        var = 0.01
        x = torch.arange(tile_w, dtype=torch.float64)
        y = torch.arange(tile_h, dtype=torch.float64)
        mx, my = (tile_w - 1) / 2, tile_h / 2
        xp = torch.exp(-(x - mx) * (x - mx) / (tile_w * tile_w) / (2 * var)) / math.sqrt(2 * math.pi * var)
        yp = torch.exp(-(y - my) * (y - my) / (tile_h * tile_h) / (2 * var)) / math.sqrt(2 * math.pi * var)
        return torch.outer(yp, xp).to(device)

    def _decode_tiled(self, z, decode_fn=None):
        """autoencoderKL.py:86-123: tiles of `tiling_size` every `tiling_size - overlap` latents (trailing ones smaller: zero-padded
        for the decoder, cropped afterwards), gaussian-weighted merge.  Same arithmetic as the reference (fp32 accumulators, fp64
        weights), tiles batched through the decoder."""
        decode_fn = decode_fn or (lambda t: self.vae_model.decode(t).sample)
        B, C, H, W = z.shape
        th, tw = self.tiling_size
        ov_h = self.tiling_overlap[0] if H > th else 0
        ov_w = self.tiling_overlap[1] if W > tw else 0
        assert ov_h < th and ov_w < tw, "tiling_overlap must be smaller than tiling_size (the reference's tile step is their difference)"
        f = self.downsampling_factor
        origins = [(i, j) for i in range(0, H, th - ov_h) for j in range(0, W, tw - ov_w)]
        tiles = torch.zeros(B * len(origins), C, th, tw, dtype=z.dtype, device=z.device)
        shapes = []
        for b in range(B):
            for k, (i, j) in enumerate(origins):
                t = z[b, :, i:i + th, j:j + tw]
                tiles[b * len(origins) + k, :, :t.shape[1], :t.shape[2]] = t
                if b == 0:
                    shapes.append((t.shape[1], t.shape[2]))
        dec = torch.cat([decode_fn(tiles[s:s + self.tile_batch]) for s in range(0, tiles.shape[0], self.tile_batch)], 0)
        oc = dec.shape[1]
        out = torch.zeros(B, oc, H * f, W * f, dtype=torch.float32, device=z.device)
        wsum = torch.zeros(1, 1, H * f, W * f, dtype=torch.float32, device=z.device)
        for k, (i, j) in enumerate(origins):
            hh, ww = shapes[k][0] * f, shapes[k][1] * f
            w = self._gaussian_weights(ww, hh, z.device)
            sl = (slice(None), slice(None), slice(i * f, i * f + hh), slice(j * f, j * f + ww))
            out[sl] += dec[k::len(origins), :, :hh, :ww] * w
            wsum[sl] += w
        return out / wsum


# ====================================================================================================================
# LPIPS (VGG16)
# ====================================================================================================================
_VGG = [(1, 0, 64, 3), (1, 2, 64, 64), (2, 5, 128, 64), (2, 7, 128, 128), (3, 10, 256, 128), (3, 12, 256, 256), (3, 14, 256, 256),
        (4, 17, 512, 256), (4, 19, 512, 512), (4, 21, 512, 512), (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]
_LIN = (64, 128, 256, 512, 512)
LPIPS_SHIFT, LPIPS_SCALE = (-0.030, -0.088, -0.188), (0.458, 0.448, 0.450)


class MiLPIPS(_NetBase):
    """``lpips.LPIPS(net="vgg")`` (lpips=True, spatial=False, eval): forward(in0, in1[, normalize]) -> [N, 1, 1, 1]"""

    def __init__(self, net="vgg", precision="bf16", **unused):
        super().__init__()
        assert net == "vgg" and precision in ("bf16", "fp32"), "the reference builds lpips.LPIPS(net='vgg')"
        self.precision = precision
        shapes = []
        for sl, idx, co, ci in _VGG:
            shapes += [(f"net.slice{sl}.{idx}.weight", (co, ci, 3, 3)), (f"net.slice{sl}.{idx}.bias", (co,))]
        for l, c in enumerate(_LIN):
            shapes.append((f"lin{l}.model.1.weight", (1, c, 1, 1)))
        cfg = NetCfg()
        cfg.kind, cfg.in_channels, cfg.precision, cfg.groups = FDMI_NET_VGG_LPIPS, 3, int(precision == "fp32"), 32
        for i in range(3):
            cfg.lpips_shift[i], cfg.lpips_scale[i] = LPIPS_SHIFT[i], LPIPS_SCALE[i]
        self._init_net(shapes, cfg)
        sl = _ensure_path(self, ["scaling_layer"])
        sl.register_buffer("shift", torch.tensor(LPIPS_SHIFT)[None, :, None, None])
        sl.register_buffer("scale", torch.tensor(LPIPS_SCALE)[None, :, None, None])

    def _out_like(self, x):
        return torch.empty(x.shape[0], dtype=torch.float32, device=x.device)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """accepts ``lpips.LPIPS(net="vgg").state_dict()`` as it is: lpips 0.1.4 also registers its five linear layers in a
        ModuleList (`lins.{k}.model.1.weight`, the same tensors as `lin{k}.model.1.weight`) -- those duplicates are dropped"""
        sd = {k: v for k, v in state_dict.items() if not k.startswith("lins.")}
        return super().load_state_dict(sd, strict=strict, **kw)

    def forward(self, in0, in1, retPerLayer=False, normalize=False):
        assert not retPerLayer
        if normalize:   # lpips: [0, 1] -> [-1, 1]
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        if torch.is_grad_enabled() and in0.requires_grad:
            out = _NetFn.apply(self, in0, in1.detach())
        else:
            out, _ = self._run(in0, in1, save=False)
        return out.view(-1, 1, 1, 1)


# ====================================================================================================================
# T2I adapter
# ====================================================================================================================
class MiT2IAdapter(_NetBase):
    """diffusers ``T2IAdapter(adapter_type="full_adapter" | "full_adapter_xl")``: forward(cond [B, c, H, W]) -> list of feature
    maps, frozen (examples/train_flash_canny_adapter.py:182-200)."""

    def __init__(self, in_channels=3, channels=(320, 640, 1280, 1280), num_res_blocks=2, downscale_factor=8,
                 adapter_type="full_adapter", precision="bf16"):
        super().__init__()
        assert adapter_type in ("full_adapter", "full_adapter_xl") and precision in ("bf16", "fp32")
        ch = list(channels)
        assert 1 <= len(ch) <= 4
        self.precision = precision
        f = downscale_factor
        shapes = [("adapter.conv_in.weight", (ch[0], in_channels * f * f, 3, 3)), ("adapter.conv_in.bias", (ch[0],))]
        for i, co in enumerate(ch):
            ci = ch[0] if i == 0 else ch[i - 1]
            if ci != co:
                shapes += [(f"adapter.body.{i}.in_conv.weight", (co, ci, 1, 1)), (f"adapter.body.{i}.in_conv.bias", (co,))]
            for j in range(num_res_blocks):
                shapes += [(f"adapter.body.{i}.resnets.{j}.block1.weight", (co, co, 3, 3)), (f"adapter.body.{i}.resnets.{j}.block1.bias", (co,)),
                           (f"adapter.body.{i}.resnets.{j}.block2.weight", (co, co, 1, 1)), (f"adapter.body.{i}.resnets.{j}.block2.bias", (co,))]
        cfg = NetCfg()
        cfg.kind, cfg.in_channels, cfg.n_levels, cfg.layers_per_block = FDMI_NET_T2I_ADAPTER, in_channels, len(ch), num_res_blocks
        for i, v in enumerate(ch):
            cfg.block_out[i] = v
        cfg.precision, cfg.groups, cfg.adapter_downscale, cfg.adapter_xl = int(precision == "fp32"), 32, f, int(adapter_type == "full_adapter_xl")
        self._init_net(shapes, cfg)
        self.total_downscale_factor = f * 2 ** (1 if cfg.adapter_xl else len(ch) - 1)

    @torch.no_grad()
    def forward(self, t2i_adapter_cond: torch.Tensor) -> List[torch.Tensor]:
        x = t2i_adapter_cond
        assert x.is_cuda, "MiT2IAdapter runs on the GPU only (no CPU fallback)"
        h = self._ensure_packed()
        L = _lib.lib()
        x = x.detach().float().contiguous()
        B, _, H, W = x.shape
        n = self._cfg.n_levels
        outs = []
        cc, hh, ww = C.c_int32(), C.c_int32(), C.c_int32()
        for i in range(n):
            check(L.fdmi_adapter_out_shape(h, i, H, W, C.byref(cc), C.byref(hh), C.byref(ww)))
            outs.append(torch.empty(B, cc.value, hh.value, ww.value, dtype=torch.float32, device=x.device))
        need = L.fdmi_net_workspace_bytes(h, B, H, W, 0)
        if need < 0:
            raise RuntimeError("fdmi: " + L.fdmi_last_error().decode())
        ws = self._workspace(0, need, x.device)
        arr = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        check(L.fdmi_adapter_forward(h, 0, ptr(x), arr, n, B, H, W, ptr(ws), ws.numel(), stream_ptr()))
        self.last_flops = L.fdmi_unet_last_flops(h)
        return outs
