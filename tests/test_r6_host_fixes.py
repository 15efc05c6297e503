"""CPU tests of round-6 host-side changes:
  * a discriminator head that stays a torch module (unsupported layer, or a conversion error) says so ONCE through `logging`, for both
    step classes -- it used to be swallowed by `except Exception: pass`;
  * the SD3 recipe builds the HIP twin of lpips.LPIPS (nets.MiLPIPS) like the UNet recipe (FD3:130-131) and fails with the same
    message without the `lpips` package;
  * bench.py prices a set of launches against the roof its arithmetic intensity puts it under (HBM below the 2.5 PFLOP/s : 8 TB/s
    ridge) and parses the MFMA-rate micro-benchmark's output;
  * the deterministic-mode switch is documented in the C header next to fdmi_tune_set and exposed by ops.deterministic."""
import logging
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flash_with(disc):
    from flash_diffusion_amd.flash import FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from flash_diffusion_amd.schedulers import DPMSolverMultistepScheduler
    from oracle.golden_cases import build_models
    teacher, student, _ = build_models()
    return FlashDiffusion(FlashDiffusionConfig(K=[4], num_iterations_per_K=[10]), student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=DPMSolverMultistepScheduler(), conditioner=TensorConditioner(), discriminator=disc)


def test_a_discriminator_left_on_torch_is_logged_once(caplog, monkeypatch):
    from flash_diffusion_amd import flash
    monkeypatch.setattr(flash, "_torch_disc_warned", False)
    exotic = torch.nn.Sequential(torch.nn.Conv2d(8, 8, 4, 2, 1), torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 4, 2, 1))
    with caplog.at_level(logging.WARNING, logger="flash_diffusion_amd.flash"):
        m = _flash_with(exotic)
        assert isinstance(m.discriminator, torch.nn.Sequential) and type(m.discriminator).__name__ == "Sequential"
        msgs = [r.getMessage() for r in caplog.records if "discriminator head is NOT on the HIP path" in r.getMessage()]
        assert len(msgs) == 1 and "ReLU" in msgs[0], msgs
        _flash_with(torch.nn.Sequential(torch.nn.Conv2d(8, 8, 4, 2, 1), torch.nn.Tanh()))
        msgs = [r.getMessage() for r in caplog.records if "discriminator head is NOT on the HIP path" in r.getMessage()]
        assert len(msgs) == 1                      # once per process


def test_a_failed_conversion_is_logged_not_swallowed(caplog, monkeypatch):
    from flash_diffusion_amd import discriminator, flash
    monkeypatch.setattr(flash, "_torch_disc_warned", False)

    def boom(seq):
        raise ValueError("kernel 5 stride 3 is not a PatchGAN stage")
    monkeypatch.setattr(discriminator.MiDiscriminator, "convert", staticmethod(boom))
    head = torch.nn.Sequential(torch.nn.Conv2d(8, 8, 4, 2, 1), torch.nn.SiLU(), torch.nn.Conv2d(8, 1, 4, 2, 1))
    with caplog.at_level(logging.WARNING, logger="flash_diffusion_amd.flash"):
        m = _flash_with(head)
    assert m.discriminator is head
    assert any("convert failed" in r.getMessage() and "PatchGAN" in r.getMessage() for r in caplog.records)


def test_sd3_recipe_builds_the_hip_lpips_twin_or_fails_like_the_unet_recipe(monkeypatch):
    """FD3:130-131 builds lpips.LPIPS(net="vgg"); the SD3 step class now wraps its pretrained weights in nets.MiLPIPS as flash.py
    does (it used to keep the torch module: VERDICT r5 weak 10).  Without the package both raise an ImportError that names MiLPIPS."""
    import inspect
    from flash_diffusion_amd import flash_sd3
    src = inspect.getsource(flash_sd3.FlashDiffusionSD3.__init__)
    assert "MiLPIPS(" in src and "lpips_model.load_state_dict(ref_lpips.state_dict())" in src
    assert "lpips_model = lpips.LPIPS" not in src            # (the torch module is no longer what the recipe trains against)
    try:
        import lpips  # noqa: F401
        pytest.skip("the lpips package is installed: the ImportError branch is not reachable here")
    except ImportError:
        pass
    from flash_diffusion_amd.flash_sd3 import FlashDiffusionSD3, FlashDiffusionSD3Config
    with pytest.raises(ImportError, match="MiLPIPS"):
        FlashDiffusionSD3(FlashDiffusionSD3Config(distill_loss_type="lpips"), student_denoiser=torch.nn.Identity(),
                          teacher_denoiser=torch.nn.Identity(), teacher_noise_scheduler=None, vae=torch.nn.Identity())


def _bench():
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("bench")


def test_bench_ridge_and_bucket_tables():
    b = _bench()
    assert abs(b.RIDGE - 312.5) < 1e-9 and b.PEAK_HBM == 8.0e12 and b.PEAK_BF16 == 2.5e15
    # the library's bucket enumeration (csrc/common.h) and bench.py's names must stay in step: 24 buckets
    hdr = open(os.path.join(ROOT, "flash_diffusion_amd", "csrc", "common.h")).read()
    m = re.search(r"PROF_NBUCKETS = (\d+)", hdr)
    assert m and int(m.group(1)) == len(b.BUCKETS) == 24
    assert int(re.search(r"PROF_GEMM4_ROW_HBM = (\d+)", hdr).group(1)) == b.BUCKETS.index("gemm4_kernel<256x320,row|hbm-side>")
    assert int(re.search(r"PROF_GEMM3_ROW_HBM = (\d+)", hdr).group(1)) == b.BUCKETS.index("gemm3_kernel<256xBN,row|hbm-side>")
    assert int(re.search(r"PROF_GEMM5 = (\d+)", hdr).group(1)) == b.BUCKETS.index("gemm5_kernel<128x320,row>")
    assert int(re.search(r"PROF_GEMM5_HBM = (\d+)", hdr).group(1)) == b.BUCKETS.index("gemm5_kernel<128x320,row|hbm-side>")
    assert set(b.SUBSET_BUCKETS) == {n for n in b.BUCKETS if n.endswith("|hbm-side>")}
    # which side of the ridge the step's row GEMMs sit on (2 M N K flop over 2 (M K + N K + M N (1 + residual)) bytes)
    def ai(M, N, K, res, nout=None):
        nout = N if nout is None else nout
        return 2.0 * M * N * K / (2.0 * (M * K + N * K + M * nout * (2 if res else 1)))
    assert ai(131072, 320, 320, True) < b.RIDGE and ai(131072, 320, 1280, True) < b.RIDGE and ai(131072, 960, 320, False) < b.RIDGE
    assert ai(131072, 2560, 320, False, nout=1280) > b.RIDGE and ai(32768, 640, 2560, True) > b.RIDGE


def test_bench_parses_the_mfma_rate_output(monkeypatch, tmp_path):
    b = _bench()
    fake = tmp_path / "scripts" / "ubench"
    fake.mkdir(parents=True)
    exe = fake / "mfma_rate"
    exe.write_text("#!/bin/sh\ncat <<'EOF'\n"
                   "16x16x32 2 waves/SIMD random rep 0: 6.768 ms  1586.4 TFLOP/s\n"
                   "32x32x16 2 waves/SIMD random rep 0: 5.884 ms  1824.8 TFLOP/s\n"
                   "32x32x16 1 wave/SIMD  random rep 2: 5.873 ms  1828.2 TFLOP/s\n"
                   "32x32x16 2 waves/SIMD zeros  rep 1: 4.319 ms  2486.1 TFLOP/s\n"
                   "32x32x16 1 wave/SIMD  zeros  rep 2: 4.326 ms  2481.9 TFLOP/s\nEOF\n")
    exe.chmod(0o755)
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    r = b.sustained_mfma()
    assert r is not None and abs(r["random_operands"] - 1.8282) < 1e-6 and abs(r["zero_operands"] - 2.4861) < 1e-6
    assert r["unit"] == "PFLOP/s" and r["clock_ghz"]["random_operands"] < r["clock_ghz"]["zero_operands"]
    monkeypatch.setattr(b, "ROOT", str(tmp_path / "nothing"))
    assert b.sustained_mfma() is None


def test_deterministic_switch_is_documented_and_exposed():
    hdr = open(os.path.join(ROOT, "include", "fdmi.h")).read()
    assert "key 50 = 1: DETERMINISTIC MODE" in hdr and "fdmi_prof_collect2" in hdr
    from flash_diffusion_amd import ops
    assert ops.DETERMINISTIC_KNOB == 50 and hasattr(ops.deterministic, "set") and hasattr(ops.deterministic, "enabled")
    common = open(os.path.join(ROOT, "flash_diffusion_amd", "csrc", "common.h")).read()
    assert "fdmi_tune_get(50)" in common
    # every fp32 atomic of the bf16 path has its ordered twin behind the switch: the launchers that issue atomics consult it
    for f, needle in (("norm.hip", "gn_reduce_det_kernel"), ("wgrad.hip", "fdmi_det()"), ("elem.hip", "fdmi_det()"), ("dit.hip", "fdmi_det()"),
                      ("netops.hip", "lpips_level_fwd_det_kernel"), ("gemm.hip", "fdmi_det()"), ("unet.hip", "!fdmi_det()")):
        assert needle in open(os.path.join(ROOT, "flash_diffusion_amd", "csrc", f)).read(), (f, needle)
