"""GPU parity of MiDiscriminator (HIP conv / SiLU / GroupNorm forward, input gradient and parameter
gradients) against the same nn.Sequential evaluated by PyTorch in fp32 on the CPU.
Tolerance: logits rel. Frobenius < 2e-2; gradients cosine > 0.995 and norm within 5 %."""
import copy

import pytest
import torch

from oracle.unet_cpu import make_discriminator, seeded_init_

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("kind,cdim,feat,hw,last_k", [("sd15", 64, 16, 8, 4), ("sd15", 1280, 64, 8, 4), ("test", 64, 16, 8, 4),
                                                       ("sd15", 64, 16, 4, 2)])
def test_discriminator_fwd_bwd(kind, cdim, feat, hw, last_k):
    from flash_diffusion_amd.discriminator import MiDiscriminator
    ref = seeded_init_(make_discriminator(kind, color_dim=cdim, feat=feat, last_k=last_k), 3)
    mi = MiDiscriminator.convert(copy.deepcopy(ref)).cuda()
    assert [n for n, _ in mi.named_parameters()] == [n for n, _ in ref.named_parameters()]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, cdim, hw, hw, generator=g)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    G = torch.randn(yr.shape, generator=g)
    (yr * G).sum().backward()
    xm = x.cuda().requires_grad_()
    ym = mi(xm)
    assert ym.shape == yr.shape
    (ym * G.cuda()).sum().backward()
    err = float((ym.detach().cpu() - yr.detach()).norm() / yr.detach().norm())
    assert err < 2e-2, err
    assert _cos(xm.grad, xr.grad) > 0.995
    for (n, p), (_, q) in zip(mi.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        c = _cos(p.grad, q.grad)
        r = float(p.grad.float().norm().cpu() / q.grad.norm())
        assert c > 0.995 and abs(r - 1) < 0.05, (n, c, r)


def test_fused_losses_match_torch():
    import torch.nn.functional as F
    from flash_diffusion_amd.flash import _DistillLoss, _DmdLoss
    g = torch.Generator().manual_seed(1)
    B = 3
    s, t = torch.randn(B, 4, 16, 16, generator=g), torch.randn(B, 4, 16, 16, generator=g)
    for l1 in (False, True):
        sr = s.clone().requires_grad_()
        ref = (torch.abs(sr - t) if l1 else (sr - t) ** 2).reshape(B, -1).mean(1).mean()
        (ref * 0.7).backward()
        sm = s.cuda().requires_grad_()
        out = _DistillLoss.apply(sm, t.cuda(), l1)
        (out * 0.7).backward()
        assert abs(float(out) - float(ref)) < 1e-5 * max(1, abs(float(ref)))
        assert torch.allclose(sm.grad.cpu(), sr.grad, atol=1e-6)
    noisy, real, fake = (torch.randn(B, 4, 16, 16, generator=g) for _ in range(3))
    ia, ma, kb = torch.rand(B, generator=g) + 1, -torch.rand(B, generator=g), torch.rand(B, generator=g) * 3
    sr = s.clone().requires_grad_()
    x0 = ia.view(-1, 1, 1, 1) * noisy + ma.view(-1, 1, 1, 1) * real
    w = 1.0 / ((sr - x0).abs().mean([1, 2, 3], keepdim=True) + 1e-5).detach()
    coeff = (real - fake) * kb.view(-1, 1, 1, 1)
    ref = F.mse_loss(sr, (sr - w * coeff).detach())
    ref.backward()
    sm = s.cuda().requires_grad_()
    out = _DmdLoss.apply(sm, noisy.cuda(), real.cuda(), fake.cuda(), ia.cuda(), ma.cuda(), kb.cuda())
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-4 * max(1, abs(float(ref)))
    assert torch.allclose(sm.grad.cpu(), sr.grad, atol=1e-5, rtol=1e-4)
