"""One tiny invocation of the hot path on cuda:0, checked against the CPU oracle (test
infrastructure: this is the only product-side file allowed to import `oracle`, and only as the
checker -- __graft_entry__.smoke())."""
import torch


def run():
    from . import ops
    torch.manual_seed(0)
    A = torch.randn(256, 192).to(torch.bfloat16)
    W = (torch.randn(320, 192) / 14).to(torch.bfloat16)
    out = ops.gemm(A.cuda(), W.cuda()).float().cpu()
    ref = A.float() @ W.float().t()
    err = (out - ref).norm() / ref.norm()
    assert err < 4e-3, f"gemm smoke mismatch {err}"
    try:
        from . import flash_smoke
    except ImportError:
        flash_smoke = None
    if flash_smoke is not None:
        flash_smoke.run()
    print("smoke ok: gemm rel err %.2e" % float(err))
