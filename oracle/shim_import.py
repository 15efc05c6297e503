"""TEST INFRASTRUCTURE (oracle) -- import the reference's OWN FlashDiffusion unmodified.

``flash.models.flash.FlashDiffusion`` (/root/reference/src/flash/models/flash/
flash_diffusion_model.py:38) imports third-party modules that are absent from this
container (lpips, diffusers*).  They are only used as base
classes / type hints on the hot path, so tiny stubs registered in ``sys.modules``
*before* the import let the reference class import and run on CPU unchanged.

Only usable where /root/reference exists (this build container).  Nothing that runs on
the GPU box may call this; it is used by ``oracle/make_golden.py`` (fixture generation)
and by ``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "flash"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns (FlashDiffusion, FlashDiffusionConfig) -- the reference's own classes."""
    if not reference_available():
        raise RuntimeError("reference sources not present at " + REFERENCE_SRC)
    import torch.nn as nn

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    class _LPIPS(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, a, b):
            raise NotImplementedError("lpips needs pretrained VGG weights (unavailable offline)")

    if "lpips" not in sys.modules:
        _stub("lpips", LPIPS=_LPIPS)
    if "diffusers" not in sys.modules:
        d = _stub("diffusers", T2IAdapter=_Dummy, DiffusionPipeline=_Dummy)
        d.schedulers = _stub("diffusers.schedulers", DDPMScheduler=_Dummy, LCMScheduler=_Dummy)
        d.models = _stub("diffusers.models", UNet2DConditionModel=_Dummy, UNet2DModel=_Dummy,
                         AutoencoderKL=_Dummy)
        # The DiT wrapper (transformers/tranformers.py:9) SUBCLASSES diffusers' Transformer2DModel and the reference's
        # AdaLayerNormSingle (transformers/utils.py:8) builds diffusers' Timesteps / TimestepEmbedding: the stubs for those
        # (and SD3Transformer2DModel, the base of the SD3 wrapper at tranformers.py:103) are the oracle's restatements, so the reference's real wrapper runs on top of them (import_reference_dit).
        from . import dit_cpu, mmdit_cpu
        d.models.transformers = _stub("diffusers.models.transformers",
                                      SD3Transformer2DModel=mmdit_cpu.SD3Transformer2DModelRef,
                                      Transformer2DModel=dit_cpu.Transformer2DModelRef)
        d.models.embeddings = _stub("diffusers.models.embeddings", TimestepEmbedding=dit_cpu.TimestepEmbedding,
                                    Timesteps=dit_cpu.Timesteps)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    from flash.models.flash import FlashDiffusion, FlashDiffusionConfig  # noqa: E402
    return FlashDiffusion, FlashDiffusionConfig


def import_reference_sd3():
    """Returns (FlashDiffusionSD3, FlashDiffusionSD3Config) -- the reference's own flow-matching (SD3) model classes
    (/root/reference/src/flash/models/flash_sd3/flash_diffusion_model.py:42), same stubs as above."""
    import_reference()
    from flash.models.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config  # noqa: E402
    return FlashDiffusionSD3, FlashDiffusionSD3Config


def import_reference_dit():
    """Returns (DiffusersTransformer2DWrapper, AdaLayerNormSingle) -- the reference's own PixArt wrapper
    (/root/reference/src/flash/models/transformers/tranformers.py:9) and adaLN-single module (transformers/utils.py:8),
    unmodified, with the oracle's restated ``Transformer2DModel`` / ``Timesteps`` / ``TimestepEmbedding`` standing in for the
    absent diffusers classes they derive from."""
    import_reference()
    from flash.models.transformers import DiffusersTransformer2DWrapper  # noqa: E402
    from flash.models.transformers.utils import AdaLayerNormSingle  # noqa: E402
    return DiffusersTransformer2DWrapper, AdaLayerNormSingle


def import_reference_sd3_wrapper():
    """Returns DiffusersSD3Transformer2DWrapper (/root/reference/src/flash/models/transformers/tranformers.py:103),
    unmodified, on top of the oracle's restated ``SD3Transformer2DModel`` (oracle/mmdit_cpu.py)."""
    import_reference()
    from flash.models.transformers import DiffusersSD3Transformer2DWrapper  # noqa: E402
    return DiffusersSD3Transformer2DWrapper
