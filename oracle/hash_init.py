"""TEST INFRASTRUCTURE (oracle) -- device-independent deterministic weights for the FULL-SIZE parity fixtures.

The seeded generators of unet_cpu.seeded_init_ / dit_cpu.seeded_init_ draw serially on the host: 2 minutes for the 2.6 B
parameters of the SDXL UNet -- minutes of idle GPU when a test on the GPU box has to rebuild the weights a fixture was made
with.  `hash_init_` fills every parameter with a counter-based integer hash of (seed, parameter index, element index): pure
int64 tensor arithmetic that never leaves 63 bits, identical bit for bit on the CPU (where oracle/make_golden.py makes the
fixtures) and on the GPU (where the -m gpu tests rebuild the same weights in a second), then mapped to a centred uniform of the
same variance rules the seeded generators use (1/fan_in weights x 0.7, small biases, norm gains around 1, LoRA A ~ 1/r)."""
from __future__ import annotations

import math

import torch

_M31 = 0x7FFFFFFF


def _mix(x):
    """x: int64 tensor with values in [0, 2^31).  Three multiply-xorshift rounds; every product stays below 2^62."""
    x = (x * 1103515245 + 12345) & _M31
    x = x ^ (x >> 15)
    x = (x * 69069 + 907633385) & _M31
    x = x ^ (x >> 13)
    x = (x * 1664525 + 1013904223) & _M31
    x = x ^ (x >> 16)
    return x


def hash_uniform(numel: int, seed: int, stream: int, device) -> torch.Tensor:
    """float64 values in [-sqrt(3), sqrt(3)) (unit variance), a pure function of (seed, stream, element index)"""
    idx = torch.arange(numel, dtype=torch.int64, device=device)
    key = ((seed * 1000003 + stream * 7919 + 12345) & _M31)
    lo = _mix((idx & _M31) ^ key)
    hi = _mix(((idx >> 31) + stream * 2654435 + seed) & _M31)
    x = _mix(lo ^ hi)
    y = _mix((x + 0x2545F491) & _M31)                  # second word: 31 + 22 bits -> a 53-bit mantissa
    u = (x.double() * float(1 << 22) + (y >> 9).double()) / float(1 << 53)      # [0, 1), exactly representable
    return (u * 2.0 - 1.0) * math.sqrt(3.0)


def scale_rule(name: str, shape, lora_b_std: float):
    """(offset, scale) of a parameter by its name -- the variance rules of unet_cpu.seeded_init_ / dit_cpu.seeded_init_"""
    if "lora_A" in name:
        return 0.0, 1.0 / shape[0]
    if "lora_B" in name:
        return 0.0, lora_b_std
    if "scale_shift_table" in name:
        return 0.0, 0.1
    if "norm" in name and name.endswith("weight") and len(shape) == 1:
        return 1.0, 0.1
    if "norm" in name and name.endswith("bias"):
        return 0.0, 0.05
    if len(shape) == 1:
        return 0.0, 0.02
    fan_in = 1
    for s in shape[1:]:
        fan_in *= int(s)
    return 0.0, 0.7 / math.sqrt(fan_in)


@torch.no_grad()
def hash_init_(module: torch.nn.Module, seed: int, lora_b_std: float = 0.02, rename=lambda n: n):
    """Fill every parameter of `module` (on whatever device it lives).  The stream of a parameter is the position of its
    (renamed) name in the sorted name list, so two modules holding the same named tensors -- the oracle restatement on the
    host, the HIP module on the GPU (rename strips peft's `.base_layer`) -- receive identical values."""
    names = sorted(rename(n) for n, _ in module.named_parameters())
    pos = {n: i for i, n in enumerate(names)}
    assert len(pos) == len(names), "parameter names collide after renaming"
    for n, p in module.named_parameters():
        key = rename(n)
        off, sc = scale_rule(key, tuple(p.shape), lora_b_std)
        v = hash_uniform(p.numel(), seed, pos[key], p.device) * sc + off
        p.copy_(v.to(torch.float32).reshape(p.shape))
    return module
