"""Planner calibration: time every GEMM / conv shape of the SD1.5 UNet (B=16 student, 2B=32 teacher) with each
large-tile kernel forced and with the automatic plan (dev tool; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops
from kbench import timeit, BF

T160, T128, T320 = (256 << 16) | 160, (256 << 16) | 128, (256 << 16) | 320


def lin(m, n, k, ft, geglu=False):
    x = torch.randn(m, k, device="cuda").to(BF)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(BF)
    out = torch.empty(m, n // 2 if geglu else n, dtype=BF, device="cuda")
    ws = torch.empty(16 * m * n, dtype=torch.float32, device="cuda") if ft == 0 else None
    kw = dict(act=ops.ACT_GEGLU) if geglu else {}
    try:
        return timeit(lambda: ops.gemm(x, w, out=out, ws=ws, splitk=0 if ft == 0 else 1, force_tile=ft, **kw), 10)
    except Exception as e:
        return float("nan")


def conv(B, hw, ci, co, ft, ups=0):
    x = torch.randn(B, hw, hw, ci, device="cuda").to(BF)
    w = (torch.randn(co, 9 * ci, device="cuda") * (9 * ci) ** -0.5).to(BF)
    ho = hw << ups
    M = B * ho * ho
    out = torch.empty(M, co, dtype=BF, device="cuda")
    ws = torch.empty(16 * M * co, dtype=torch.float32, device="cuda") if ft == 0 else None
    cv = dict(Hin=hw, Win=hw, Cin=ci, Hout=ho, Wout=ho, KH=3, KW=3, stride=1, pad=1, ups=ups)
    try:
        return timeit(lambda: ops.gemm(x, w, M=M, out=out, conv=cv, ws=ws, splitk=0 if ft == 0 else 1, force_tile=ft), 10)
    except Exception as e:
        return float("nan")


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    rows = []
    for B in (16, 32):
        for (hw, C) in [(64, 320), (32, 640), (16, 1280), (8, 1280)]:
            M = B * hw * hw
            for (n, k, geglu) in [(C, C, False), (8 * C, C, True), (C, 4 * C, False)]:
                rows.append((f"lin M={M} N={n} K={k}{' geglu' if geglu else ''}", lambda ft, M=M, n=n, k=k, g=geglu: lin(M, n, k, ft, g), 2.0 * M * n * k))
            rows.append((f"lin M={B * 77} N={C} K=768", lambda ft, M=B * 77, n=C: lin(M, n, 768, ft), 2.0 * B * 77 * C * 768))
        for (hw, ci, co) in [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 320, 640), (32, 640, 640), (32, 960, 640), (32, 1280, 640),
                             (32, 1920, 640), (16, 640, 1280), (16, 1280, 1280), (16, 1920, 1280), (16, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280)]:
            rows.append((f"conv B={B} {hw}x{hw} {ci}->{co}", lambda ft, B=B, hw=hw, ci=ci, co=co: conv(B, hw, ci, co, ft), 2.0 * B * hw * hw * co * 9 * ci))
    print(f"{'shape':40s} {'auto':>8s} {'t160':>8s} {'t128':>8s} {'t320':>8s}  auto TF/s")
    tot = [0, 0]
    for name, fn, fl in rows:
        if only and only not in name:
            continue
        ts = [fn(ft) for ft in (0, T160, T128, T320)]
        best = min(t for t in ts[1:] if t == t)
        tot[0] += ts[0]; tot[1] += min(best, ts[0])
        flag = "" if ts[0] <= best * 1.08 else "  <-- plan"
        print(f"{name:40s} {ts[0]:8.1f} {ts[1]:8.1f} {ts[2]:8.1f} {ts[3]:8.1f}  {fl / ts[0] / 1e6:7.0f}{flag}", flush=True)
    print("sum auto", tot[0], "sum best", tot[1])


main()
