"""CPU tests of round-5 host-side fixes:
  * the key-length cache of the transformer denoisers is keyed on tensor IDENTITY (weak reference + version), not on an address the
    caching allocator recycles for the next batch's mask; host lengths passed beside the mask avoid the device read;
  * attaching the flat LoRA gradient buffer keeps what other parameters already accumulated when only some grads are None."""
import types

import pytest
import torch

from flash_diffusion_amd import ops
from flash_diffusion_amd.dit import MiTransformer2DModel


def _mask(lens, L=8):
    m = torch.zeros(len(lens), L, dtype=torch.long)
    for i, n in enumerate(lens):
        m[i, :n] = 1
    return m


def test_key_lens_follow_the_mask_content_not_its_address():
    host = types.SimpleNamespace(_mask_cache=None)
    f = MiTransformer2DModel._key_lens
    m1 = _mask([3, 5])
    assert f(host, m1, 8) == [3, 5]
    addr = m1.data_ptr()
    del m1                               # the allocator may hand the same block to the next mask
    seen_same_address = False
    for _ in range(8):
        m2 = _mask([2, 5])
        seen_same_address |= m2.data_ptr() == addr
        assert f(host, m2, 8) == [2, 5]
        del m2
    m3 = _mask([4, 4])
    assert f(host, m3, 8) == [4, 4]
    calls = []
    orig = torch.Tensor.to
    try:                                 # the SAME tensor object, unmodified: served from the cache (no second device read)
        torch.Tensor.to = lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1]
        assert f(host, m3, 8) == [4, 4] and calls == []
        m3[0, 3] = 0                     # modified in place: the version moved, the lengths are read again
        assert f(host, m3, 8) == [3, 4] and calls
    finally:
        torch.Tensor.to = orig
    assert f(host, _mask([8, 8]), 8) is None                       # nothing masked
    with pytest.raises(NotImplementedError):
        f(host, torch.tensor([[1, 0, 1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 1]]), 8)


def test_host_lengths_beside_the_mask_are_used_and_checked():
    host = types.SimpleNamespace(_mask_cache=None)
    f = MiTransformer2DModel._key_lens
    m = _mask([3, 5])
    assert f(host, m, 8, [3, 5]) == [3, 5] and host._mask_cache is None      # no read, no cache entry
    assert f(host, m, 8, [8, 8]) is None
    for bad in ([3], [0, 5], [3, 9]):
        with pytest.raises(ValueError):
            f(host, m, 8, bad)


def test_attach_flat_grads_zeroes_only_what_was_none():
    flat = torch.full((10,), 7.0)
    p = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(2, 3))]
    ops.attach_flat_grads(p, flat)                                  # every grad None: the whole buffer starts from zero
    assert torch.equal(flat, torch.zeros(10)) and p[0].grad.data_ptr() == flat.data_ptr()
    flat[:4] = 1.0                                                  # p[0] accumulated something
    flat[4:] = 2.0
    p[1].grad = None                                                # only p[1] was reset (user code, a partial param group)
    ops.attach_flat_grads(p, flat)
    assert torch.equal(flat[:4], torch.ones(4)) and torch.equal(flat[4:], torch.zeros(6))
    p[0].grad = torch.full((4,), 5.0)                               # a foreign .grad tensor: its content moves into the slice
    ops.attach_flat_grads(p, flat)
    assert torch.equal(flat[:4], torch.full((4,), 5.0)) and p[0].grad.data_ptr() == flat.data_ptr()


def test_milpips_takes_the_lpips_package_state_dict_with_its_duplicate_lins():
    """flash.py (FD:102-103): `MiLPIPS.load_state_dict(lpips.LPIPS(net="vgg").state_dict())`.  lpips 0.1.4 registers its five linear
    layers twice -- as attributes lin0 .. lin4 AND in the ModuleList `lins` -- so its state_dict carries every
    `lin{k}.model.1.weight` a second time as `lins.{k}.model.1.weight`.  The package is absent from the image (VERDICT r4 missing
    5): a state_dict FABRICATED with that key set (the oracle's restatement of the architecture + the duplicates) must load with
    strict=True, land in the right tensors, and a key that is really foreign must still be refused."""
    from flash_diffusion_amd.nets import MiLPIPS
    from oracle.vae_cpu import LPIPSRef, seeded_net_init_
    ref = seeded_net_init_(LPIPSRef(), 5)
    sd = dict(ref.state_dict())
    for k in range(5):
        sd[f"lins.{k}.model.1.weight"] = sd[f"lin{k}.model.1.weight"]          # the same tensors, as lpips exposes them
    assert sum(k.startswith("lins.") for k in sd) == 5
    m = MiLPIPS(precision="fp32")
    res = m.load_state_dict(sd)                                                # strict
    assert not res.missing_keys and not res.unexpected_keys
    own = m.state_dict()
    assert set(own) == {k for k in sd if not k.startswith("lins.")}
    for k, v in own.items():
        assert torch.equal(v.cpu().float().reshape(-1), sd[k].float().reshape(-1)), k
    with pytest.raises(RuntimeError):
        m.load_state_dict(dict(sd, **{"net.slice9.0.weight": torch.zeros(1)}))
