"""Shared helpers for the GPU parity tests: move oracle weights into the HIP modules."""
import copy

import torch

from oracle.unet_cpu import UNet2DConditionRef, seeded_init_


def mi_kwargs(cfg):
    return dict(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                down_block_types=tuple(cfg.down_block_types), up_block_types=tuple(cfg.up_block_types),
                block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
                cross_attention_dim=cfg.cross_attention_dim,
                transformer_layers_per_block=cfg.tlayers(), attention_head_dim=cfg.heads(),
                norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps, class_embed_type=cfg.class_embed_type,
                projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim)


def mi_from_oracle(oracle_unet, lora_rank=0, device="cuda", precision="bf16"):
    """Build a MiUNet2DConditionModel holding exactly the oracle module's weights (precision "fp32": the validation plan)."""
    from flash_diffusion_amd.unet import MiUNet2DConditionModel
    m = MiUNet2DConditionModel(**mi_kwargs(oracle_unet.cfg), precision=precision)
    sd = {k.replace(".base_layer.", "."): v for k, v in oracle_unet.state_dict().items()}
    base = {k: v for k, v in sd.items() if ".lora_" not in k}
    m.load_state_dict(base, strict=True)
    m = m.to(device)
    if lora_rank:
        m.add_adapter(lora_rank)
        lora = {k: v for k, v in sd.items() if ".lora_" in k}
        missing = m.load_state_dict({**base, **lora}, strict=True)
    else:
        for p in m.parameters():
            p.requires_grad = False
    return m
