"""Conditioner plumbing either side of the hot path: drop-ins for the reference's ``ConditionerWrapper``
(/root/reference/src/flash/models/embedders/conditioners_wrapper.py:16-106), ``BaseConditioner`` (base/base_conditioner.py:13-58),
``TimestepsEmbedder`` (timesteps/timesteps_embedding.py:6-45 -- the SDXL size / crop embedder) and ``TorchNNEmbedder``
(torch_nn/embedders.py:10-56), plus ``TensorEmbedder`` for batches that already carry an embedding.

Same semantics as the reference: every conditioner returns ``{"vector" | "crossattn" | "concat": tensor}`` keyed by the
tensor's rank (2 / 3 / 4); the wrapper concatenates same-key outputs (vector: dim 1, crossattn: dim 2, concat: dim 1);
classifier-free-guidance dropout: a conditioner whose ``input_key`` is in ``ucg_keys`` is zeroed, otherwise it is zeroed
with probability ``ucg_rate`` (one ``torch.rand(1)`` per conditioner with a positive rate, on the host, in list order --
the reference's draw order) unless ``set_ucg_rate_zero``.  The sinusoidal embedding runs in the HIP kernel
``fdmi_timestep_embed``.  The text encoders themselves (CLIP / T5) are ``clip.py`` / ``t5.py`` (same C-ABI op layer)."""
from __future__ import annotations

import importlib
from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops

KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}
DIM2CONDITIONING = {2: "vector", 3: "crossattn", 4: "concat"}


class BaseConditioner(nn.Module):
    def __init__(self, input_key: str = "text", unconditional_conditioning_rate: float = 0.0):
        super().__init__()
        assert 0.0 <= unconditional_conditioning_rate <= 1.0, "Unconditional conditioning rate should be between 0 and 1"
        self.input_key = input_key
        self.dim2outputkey = DIM2CONDITIONING
        self.ucg_rate = unconditional_conditioning_rate

    def forward(self, batch: Dict[str, Any], force_zero_embedding: bool = False, *args, **kwargs):
        raise NotImplementedError("Forward pass must be implemented in child class")


class TimestepsEmbedder(BaseConditioner):
    """sinusoidal embedding of every scalar of ``batch[input_key]`` [B, k] -> "vector" [B, k * num_channels]"""

    def __init__(self, num_channels: int = 256, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0,
                 input_key: str = "timesteps", unconditional_conditioning_rate: float = 0.0):
        super().__init__(input_key, unconditional_conditioning_rate)
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, batch, force_zero_embedding: bool = False, *args, **kwargs):
        x = batch[self.input_key]
        e = ops.timestep_embed(x.flatten().float().contiguous(), self.num_channels, self.flip_sin_to_cos,
                               float(self.downscale_freq_shift))
        e = e.float().reshape(x.shape[0], -1)
        if force_zero_embedding:
            e = 0 * e
        return {self.dim2outputkey[e.dim()]: e}


class TensorEmbedder(BaseConditioner):
    """``batch[input_key]`` already IS the embedding (pre-computed text features): passed through, zeroed when dropped --
    what ``force_zero_embedding`` yields for the CLIP / T5 embedders (clip_embedder_model.py:93-94)"""

    def forward(self, batch, force_zero_embedding: bool = False, *args, **kwargs):
        x = batch[self.input_key]
        if force_zero_embedding:
            x = 0 * x
        return {self.dim2outputkey[x.dim()]: x}


class TorchNNEmbedder(BaseConditioner):
    """a chain of torch.nn modules named by import path (torch_nn/embedders.py:23-33), applied to ``batch[input_key]``"""

    def __init__(self, nn_modules: Optional[Sequence[str]] = None, nn_modules_kwargs: Optional[Sequence[Dict[str, Any]]] = None,
                 flatten_output: bool = False, input_key: str = "image", unconditional_conditioning_rate: float = 0.0):
        super().__init__(input_key, unconditional_conditioning_rate)
        nn_modules, nn_modules_kwargs = list(nn_modules or []), list(nn_modules_kwargs or [])
        assert len(nn_modules) == len(nn_modules_kwargs), "Number of modules and kwargs should be same"
        self.flatten_output = flatten_output
        chain = []
        for path, kw in zip(nn_modules, nn_modules_kwargs):
            mod, cls = path.rsplit(".", 1)
            chain.append(getattr(importlib.import_module(mod), cls)(**kw))
        self.nn_modules = nn.Sequential(*chain)

    def forward(self, batch, force_zero_embedding: bool = False, *args, **kwargs):
        x = self.nn_modules(batch[self.input_key])
        if force_zero_embedding:
            x = 0 * x
        if self.flatten_output:
            x = x.view(x.size(0), -1)
        return {self.dim2outputkey[x.dim()]: x}


class ConditionerWrapper(nn.Module):
    def __init__(self, conditioners: Optional[List[BaseConditioner]] = None):
        super().__init__()
        self.conditioners = nn.ModuleList(conditioners)

    def forward(self, batch: Dict[str, Any], ucg_keys: Optional[List[str]] = None, set_ucg_rate_zero: bool = False,
                *args, **kwargs):
        if ucg_keys is None:
            ucg_keys = []
        out: Dict[str, Dict[str, torch.Tensor]] = dict(cond={})
        for c in self.conditioners:
            if c.input_key in ucg_keys:
                zero = True
            elif c.ucg_rate > 0 and not set_ucg_rate_zero:
                zero = bool(torch.rand(1) < c.ucg_rate)
            else:
                zero = False
            for key, val in c.forward(batch, force_zero_embedding=zero, *args, **kwargs).items():
                if key in out["cond"]:
                    out["cond"][key] = torch.cat([out["cond"][key], val], KEY2CATDIM[key])
                else:
                    out["cond"][key] = val
        return out

    def to(self, device):
        for c in self.conditioners:
            c.to(device)
        self.device = device
        return self
