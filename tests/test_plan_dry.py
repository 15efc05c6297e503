"""CPU: the C++ UNet plan executor (csrc/unet.hip) walked in its workspace-query mode -- every op of the forward and of the
taped backward is visited, buffers are bump-allocated and the algorithmic FLOPs counted, but no kernel is launched -- so the
plan's topology, its flag handling and its FLOP accounting are pinned without a GPU against SURVEY.md appendix C
(SD1.5 0.8033 TFLOP/image, SDXL 6.7612 at 128x128 latents, down+mid only 0.2614 / 2.933; backward = dgrad + 2x attention)."""
import pytest
import torch

from flash_diffusion_amd import _lib
from flash_diffusion_amd.unet import (FDMI_UNET_CFG_HALVES, FDMI_UNET_CTX_FILL, FDMI_UNET_CTX_REUSE, FDMI_UNET_INPUT_GRAD,
                                      FDMI_UNET_INTERMEDIATE, FDMI_UNET_SAVE, MiUNet2DConditionModel)
from flash_diffusion_amd.workloads import SD15, SDXL, TINY


def _plan(arch):
    with torch.device("meta"):
        m = MiUNet2DConditionModel(**arch)
    return m, m._plan()


def _query(plan, B, HW, L, flags):
    lib = _lib.lib()
    need = lib.fdmi_unet_workspace_bytes(plan.handle, B, HW, HW, L, flags)
    assert need > 0, lib.fdmi_last_error()
    return need, lib.fdmi_unet_last_flops(plan.handle)


@pytest.mark.parametrize("name,arch,B,HW,per_image,down_mid", [("sd15", SD15, 16, 64, 0.8033, 0.2614),
                                                              ("sdxl", SDXL, 8, 128, 6.7612, 2.933)])
def test_forward_flops_match_the_analytical_model(name, arch, B, HW, per_image, down_mid):
    m, plan = _plan(arch)
    _, fl = _query(plan, B, HW, 77, 0)
    assert abs(fl / 1e12 / B - per_image) < 2e-4 * per_image + 1e-4, fl
    _, fl2 = _query(plan, 2 * B, HW, 77, 0)
    assert abs(fl2 - 2 * fl) < 1e-6 * fl                       # linear in the batch
    _, fdm = _query(plan, B, HW, 77, FDMI_UNET_INTERMEDIATE)   # the GAN backbone: down + mid blocks only
    assert abs(fdm / 1e12 / B - down_mid) < 1e-3 * down_mid + 1e-4, fdm


def test_backward_flops_and_workspace_sd15():
    """frozen base weights: backward = dgrad of every GEMM / conv + 2x the attention forward = 14.87 TFLOP at B=16
    (+ the first conv's input gradient); a saved forward needs more workspace than an inference forward, and asking for the
    input gradient costs no extra buffer"""
    m, plan = _plan(SD15)
    ws0, f = _query(plan, 16, 64, 77, 0)
    ws1, b = _query(plan, 16, 64, 77, FDMI_UNET_SAVE)
    ws2, b2 = _query(plan, 16, 64, 77, FDMI_UNET_SAVE | FDMI_UNET_INPUT_GRAD)
    A = 16 * 0.1261e12
    assert abs(b - (f + A)) < 0.02 * f, (b, f + A)
    assert b2 == b and ws1 > ws0 and ws2 == ws1
    assert ws1 < 64 * 2 ** 30      # far inside one MI355X's 288 GB together with the teacher's workspace


def test_context_cache_flags_keep_the_topology():
    """FDMI_UNET_CTX_FILL / _REUSE move the cross-attention K/V into plan-owned buffers: same algorithmic FLOPs (the plan counts
    the projections it skips as not executed only on the real run), slightly less workspace"""
    m, plan = _plan(SD15)
    ws0, f0 = _query(plan, 32, 64, 77, 0)
    ws1, f1 = _query(plan, 32, 64, 77, FDMI_UNET_CTX_FILL)
    ws2, f2 = _query(plan, 32, 64, 77, FDMI_UNET_CTX_REUSE)
    assert f0 == f1 == f2 and ws1 <= ws0 and ws2 == ws1


@pytest.mark.parametrize("arch,B,HW", [(SD15, 32, 64), (TINY, 4, 32)])
def test_cfg_halves_removes_exactly_the_shared_prefix(arch, B, HW):
    """FDMI_UNET_CFG_HALVES on a [x | x] batch: conv_in, the first ResNet block, and the first transformer's GroupNorm / proj_in /
    LayerNorm / q,k,v / self-attention / to_out run at B/2 -- the saving is half of those ops' FLOPs, computed here from the
    architecture alone"""
    m, plan = _plan(arch)
    _, full = _query(plan, B, HW, 77, FDMI_UNET_CTX_FILL)
    _, half = _query(plan, B, HW, 77, FDMI_UNET_CTX_FILL | FDMI_UNET_CFG_HALVES)
    c = m.config_dict
    C0, cin, heads = c["block_out_channels"][0], c["in_channels"], c["attention_head_dim"][0]
    cin_pad = (cin + 7) // 8 * 8
    M = B * HW * HW
    conv_in = 2 * M * C0 * 9 * cin_pad
    resnet = 2 * (2 * M * C0 * 9 * C0)
    lin = 2 * M * C0 * C0
    attn = 4 * B * heads * (HW * HW) ** 2 * (C0 // heads)
    prefix = conv_in + resnet + 5 * lin + attn          # proj_in, q, k, v, to_out
    assert abs((full - half) - prefix / 2) < 1e-6 * full, (full - half, prefix / 2)
    # a plan whose first down block has no attention ignores the flag
    m2, plan2 = _plan(SDXL)
    assert _query(plan2, 4, 64, 77, FDMI_UNET_CFG_HALVES)[1] == _query(plan2, 4, 64, 77, 0)[1]


def test_odd_batch_ignores_cfg_halves_and_bad_sizes_fail():
    m, plan = _plan(TINY)
    assert _query(plan, 3, 32, 77, FDMI_UNET_CFG_HALVES)[1] == _query(plan, 3, 32, 77, 0)[1]


# ---- GroupNorm statistics in the producing GEMM's epilogue (default; A/B switch 14 turns it off; GPU numerics in tests/test_zz_dit_gpu.py) ----
def _gn_plan(B, H, Ci, Co, kind):
    import ctypes as C
    from flash_diffusion_amd import ops
    HW = H * H
    d = _lib.GemmDesc()
    d.M, d.N, d.K = B * HW, Co, (9 * Ci if kind == "conv" else Ci)
    conv = None
    if kind == "conv":
        conv = dict(Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, KH=3, KW=3, stride=1, pad=1)
        d.mode = 1
        for k, v in conv.items():
            setattr(d, k, v)
    d.lda = d.ldw = d.K
    d.splitk, d.use_glds, d.alpha, d.ldc = 1, 1, 1.0, Co
    out = [C.c_int32() for _ in range(4)]
    assert _lib.lib().fdmi_gemm_plan(C.byref(d), *[C.byref(x) for x in out]) == 0
    return ops.gemm_gn_ok(B * HW, Co, d.K, HW, 32, conv=conv), out[0].value, out[2].value


def test_gn_epilogue_test_problems_cover_all_three_kernels():
    """the op-level GPU test's problems are eligible and reach the 256x320, 256x160 and 256x128 instantiations"""
    from tests.test_zz_dit_gpu import GN_EPI
    seen = set()
    for cfg in GN_EPI:
        ok, kernel, bn = _gn_plan(*cfg)
        assert ok, cfg
        seen.add((kernel, bn))
    assert seen == {(2, 320), (1, 160), (1, 128)}, seen


def test_gn_epilogue_eligibility():
    from flash_diffusion_amd import ops
    cv = lambda H, C: dict(Hin=H, Win=H, Cin=C, Hout=H, Wout=H, KH=3, KW=3, stride=1, pad=1)
    assert ops.gemm_gn_ok(16 * 4096, 320, 2880, 4096, 32, conv=cv(64, 320))         # SD1.5 level 0 conv
    assert ops.gemm_gn_ok(65536, 320, 320, 4096, 32)                                # proj_out (+ residual) at level 0
    assert not ops.gemm_gn_ok(16 * 64, 1280, 11520, 64, 32, conv=cv(8, 1280))       # 8x8 level: a 256-row tile spans 4 samples
    assert not ops.gemm_gn_ok(65536, 128, 320, 4096, 32)                            # 4 channels per group < the 8-column chunk
    assert not ops.gemm_gn_ok(65536, 320, 320, 4096, 32, out_f32=1)                 # sums are defined on the stored bf16 values
    assert not ops.gemm_gn_ok(65536, 320, 320, 4096, 32, splitk=4)                  # split-K slabs: no single owner of a value
    assert not ops.gemm_gn_ok(65536 + 128, 320, 320, 4096, 32)                      # ragged M
    assert not ops.gemm_gn_ok(65536, 320, 320, 4096, 32, ldc=324)                   # output rows not 16-byte aligned
    assert not ops.gemm_gn_ok(200, 320, 320, 100, 32)                               # too small for a 256-row kernel


def test_plan_takes_groupnorm_sums_from_the_epilogue():
    """workspace-query walk of the SD1.5 plan (A/B switch 14 = 1 turns the feature off): the GroupNorms fed by an eligible conv / linear (64x64 and 32x32
    levels; the up path's concatenated inputs and the small levels keep the reduce kernel) skip their reduction; FLOPs and
    workspace are unchanged; a forward carrying T2I-adapter residuals (added in place after production) never uses it"""
    import ctypes as C
    lib = _lib.lib()
    m, plan = _plan(SD15)

    def counts(B, flags):
        ws, fl = _query(plan, B, 64, 77, flags)
        tot = C.c_int32()
        n = lib.fdmi_unet_last_gn_epilogue(plan.handle, C.byref(tot))
        return ws, fl, n, tot.value

    lib.fdmi_tune_set(14, 1)
    try:
        base = counts(16, 0)
    finally:
        lib.fdmi_tune_set(14, 0)
    assert base[2] == 0 and base[3] == 61
    if True:
        on = counts(16, 0)
        assert on[:2] == base[:2] and on[3] == 61 and 20 <= on[2] <= 25, on
        on2 = counts(32, FDMI_UNET_CTX_FILL | FDMI_UNET_CFG_HALVES)
        assert on2[2] >= on[2]                     # the 2B teacher batch fills the 256x320 grid at the 32x32 level too
        sv = counts(16, FDMI_UNET_SAVE)
        assert sv[2] == on[2]                      # same forward when the tape is recorded
        assert counts(8, FDMI_UNET_SAVE | FDMI_UNET_INPUT_GRAD)[2] == 15     # the configuration of the GPU test (test_zz_dit_gpu.py)
        null = (C.c_void_p * 4)()
        assert lib.fdmi_unet_set_down_residuals(plan.handle, null, 4, C.c_float(1.0)) == 0
        assert counts(16, 0)[2] == 0
        assert lib.fdmi_unet_set_down_residuals(plan.handle, None, 0, C.c_float(1.0)) == 0
        mt, pt = _plan(TINY)
        _query(pt, 2, 32, 77, 0)
        assert lib.fdmi_unet_last_gn_epilogue(pt.handle, None) == 0     # nothing in the tiny plan reaches a 256-row kernel


def test_hbm_byte_counters_of_the_memory_bound_families():
    """fdmi_unet_last_hbm_bytes (scripts/hbm_table.py prices the measured kernel times against these): linear in the batch, the
    GroupNorm apply pass moves twice the bytes of the reduce pass over the same tensors, the backward adds its share, the epilogue
    statistics remove the reduce bytes of the tensors whose sums come from a GEMM epilogue"""
    lib = _lib.lib()
    m, plan = _plan(SD15)

    def fam(B, flags):
        _query(plan, B, 64, 77, flags)
        return [lib.fdmi_unet_last_hbm_bytes(plan.handle, i) for i in range(9)]

    on = fam(16, 0)
    lib.fdmi_tune_set(14, 1)       # the byte model below is the one of the plain reduce + apply passes
    f16, f32 = fam(16, 0), fam(32, 0)
    assert all(abs(b - 2 * a) <= 1e-9 * max(b, 1.0) for a, b in zip(f16[:8], f32[:8]))   # (split-K choices depend on the row count)
    assert f16[0] > 0 and abs(f16[1] - 2 * f16[0]) < 1e-9 * f16[1]            # forward: reduce reads x, apply reads x + writes y
    assert f16[2] > 0 and f16[4] > 0 and f16[6] == 0 and f16[7] == 0   # (GEGLU backward / pooling: backward only)
    # round 3: the up path's [h | skip] concatenations are never materialised (two-part operands of GroupNorm and of the 1x1
    # shortcut GEMM): a forward moves no copy2d bytes at all; A/B switch 30 = 1 brings the copies back
    assert f16[5] == 0
    lib.fdmi_tune_set(30, 1)
    try:
        assert fam(16, 0)[5] > 0
    finally:
        lib.fdmi_tune_set(30, 0)
    assert lib.fdmi_unet_last_hbm_bytes(plan.handle, 9) == -1.0 and f16[8] > 0      # (the deep levels run split-K)
    sv = fam(16, FDMI_UNET_SAVE)
    lib.fdmi_tune_set(14, 0)
    assert all(s >= f for s, f in zip(sv, f16)) and sv[6] > 0 and sv[7] > 0 and sv[0] > 2.5 * f16[0]
    assert on[1] == f16[1] and 0.2 * f16[0] < on[0] < 0.6 * f16[0], (on[0], f16[0])   # the up path's concatenated inputs remain


def test_plan_report_lists_the_forward_work_list():
    """scripts/plan_report.py (FDMI_PLAN_LOG=1 on a workspace-query walk): the GEMM / conv FLOPs it lists plus the attention FLOPs
    are the plan's own total; the C2 teacher forward runs 70 % of its contraction FLOPs on the 256x320 kernel without split-K"""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "plan_report.py"), "sd15", "32", "64"], capture_output=True,
                         text=True, check=True).stdout
    head = re.match(r"sd15 B=32 64x64 flags=8: (\d+) GEMM/conv launches, ([\d.]+) TFLOP", out)
    assert head and int(head.group(1)) > 200
    m, plan = _plan(SD15)
    _, total = _query(plan, 32, 64, 77, FDMI_UNET_CTX_FILL)
    attention = 32 * 0.1261e12
    assert abs(float(head.group(2)) * 1e12 + attention - total) < 5e-3 * total
    share = float(re.search(r"gemm4\s+BN=320\s+(\d+\.\d)%", out).group(1))
    assert 60 < share < 80


# ---- the frozen networks beside the denoiser (round 3): the same executor, walked without a GPU -------------------------------------
def _net_cfg(kind, cin, cout, levels, lpb, down=8, xl=0):
    c = _lib.NetCfg()
    c.kind, c.in_channels, c.out_channels, c.n_levels, c.layers_per_block = kind, cin, cout, len(levels), lpb
    for i, v in enumerate(levels):
        c.block_out[i] = v
    c.groups, c.eps, c.precision, c.adapter_downscale, c.adapter_xl = 32, 1e-6, 0, down, xl
    for i in range(3):
        c.lpips_shift[i], c.lpips_scale[i] = 0.0, 1.0
    return c


def _torch_flops(fn):
    from torch.utils.flop_counter import FlopCounterMode
    with FlopCounterMode(display=False) as fc:
        fn()
    return float(fc.get_total_flops())


def test_frozen_net_plans_count_the_flops_of_the_upstream_modules():
    """VAE decoder, LPIPS-VGG16 and T2I adapter plans (fdmi_net_create, csrc/unet.hip NetVae / NetVgg / NetAdapter) in workspace-query
    mode against torch's FLOP counter over the oracle restatements of the upstream modules (oracle/vae_cpu.py, meta tensors): the plan
    visits every convolution / attention of diffusers' AutoencoderKL decoder, lpips.LPIPS(net="vgg") and the full T2I adapter."""
    import ctypes as C
    from oracle.vae_cpu import AutoencoderKLDecoderRef, LPIPSRef, T2IAdapterRef
    lib = _lib.lib()

    def plan_flops(cfg, B, H, W):
        h = lib.fdmi_net_create(C.byref(cfg))
        assert h, lib.fdmi_last_error()
        need = lib.fdmi_net_workspace_bytes(h, B, H, W, 0)
        assert need > 0, lib.fdmi_last_error()
        assert lib.fdmi_net_workspace_bytes(h, B, H, W, FDMI_UNET_SAVE) >= need        # the tape keeps activations
        lib.fdmi_net_workspace_bytes(h, B, H, W, 0)
        fl = lib.fdmi_unet_last_flops(h)
        lib.fdmi_unet_destroy(h)
        return fl

    with torch.device("meta"):
        dec, lp, ad = AutoencoderKLDecoderRef(), LPIPSRef(), T2IAdapterRef()
        ref_vae = _torch_flops(lambda: dec.decode_raw(torch.empty(1, 4, 64, 64)))
        ref_lp = _torch_flops(lambda: lp(torch.empty(2, 3, 256, 256), torch.empty(2, 3, 256, 256)))
        ref_ad = _torch_flops(lambda: ad(torch.empty(1, 3, 512, 512)))
    vae = plan_flops(_net_cfg(1, 4, 3, [128, 256, 512, 512], 2), 1, 64, 64)
    assert abs(vae - ref_vae) < 1e-3 * ref_vae, (vae, ref_vae)              # SD VAE decoder: 2.51 TFLOP per 512-px image
    assert 2.4e12 < vae < 2.6e12
    lpv = plan_flops(_net_cfg(3, 3, 0, [], 0), 2, 256, 256)                 # kinds: 1 VAE decoder, 3 LPIPS-VGG16, 4 T2I adapter (fdmi.h)
    # (the first VGG convolution runs on the 3 -> 8 zero-padded channels of the NHWC layout: + 64 * 256^2 * 9 * 5 * 2 flop per image)
    pad = 4 * 64 * 256 * 256 * 9 * 5 * 2
    assert abs(lpv - pad - ref_lp) < 2e-3 * ref_lp, (lpv, ref_lp, pad)
    adv = plan_flops(_net_cfg(4, 3, 0, [320, 640, 1280, 1280], 2), 1, 512, 512)
    assert abs(adv - ref_ad) < 1e-6 * ref_ad, (adv, ref_ad)


# ---- the transformer denoisers' plans (csrc/dit_plan.h), walked the same way ------------------------------------------------
def _dit_plan(kind, arch, lora):
    from flash_diffusion_amd import dit
    lib = _lib.lib()
    with torch.device("meta"):
        m = (dit.MiTransformer2DModel if kind == "pixart" else dit.MiSD3Transformer2DModel)(**arch)
        if lora:
            m.add_adapter(lora)
    p = m._plan()
    for n, mod in m._lora_modules():      # (no GPU here: declare the adapters instead of binding them)
        assert lib.fdmi_unet_declare_lora(p.handle, n.encode(), mod.rank) == 0, lib.fdmi_last_error()
    return m, p


def _dit_query(p, B, HW, L, masked, flags):
    lib = _lib.lib()
    need = lib.fdmi_dit_workspace_bytes(p.handle, B, HW, HW, L, masked, flags)
    assert need > 0, lib.fdmi_last_error()
    return need, lib.fdmi_unet_last_flops(p.handle)


@pytest.mark.parametrize("kind,arch_name", [("pixart", "TINY_PIXART"), ("pixart", "PIXART"), ("sd3", "TINY_SD3"), ("sd3", "SD3")])
def test_dit_plan_registers_every_parameter_of_the_module(kind, arch_name):
    import ctypes as C
    from flash_diffusion_amd import workloads
    m, p = _dit_plan(kind, getattr(workloads, arch_name), 0)
    lib = _lib.lib()
    buf, ne, names = C.create_string_buffer(512), C.c_int64(), set()
    for i in range(lib.fdmi_unet_num_params(p.handle)):
        assert lib.fdmi_unet_param_name(p.handle, i, buf, 512, C.byref(ne)) == 0
        names.add((buf.value.decode(), ne.value))
    assert names == {(k, v.numel()) for k, v in m.named_parameters()}


@pytest.mark.parametrize("kind,arch_name,B,HW,L,r", [("pixart", "PIXART", 8, 128, 120, 64), ("sd3", "SD3", 4, 128, 333, 64)])
def test_dit_plan_flops_and_workspace_at_the_benchmarked_shapes(kind, arch_name, B, HW, L, r):
    """forward FLOPs against the architecture's closed form (projections + attention + feed-forward per block; the embedders and
    per-sample vectors are noise at this size); a frozen forward needs two blocks' worth of workspace whatever the depth; a
    saved run's backward recycles the forward's buffers (about the forward footprint, not twice it); LoRA adds its rank-r GEMMs"""
    from flash_diffusion_amd import workloads
    arch = getattr(workloads, arch_name)
    m, p = _dit_plan(kind, arch, 0)
    D, nl = arch["num_attention_heads"] * arch["attention_head_dim"], arch["num_layers"]
    T = (HW // 2) ** 2
    ws0, f0 = _dit_query(p, B, HW, L, 0, 0)
    if kind == "pixart":
        per = 2 * T * D * D * (4 + 2 + 8) + 4 * T * T * D + 4 * T * L * D + 2 * L * D * D * 2      # self, cross (q, out | k, v), ff
        blocks = nl * per
    else:
        S = T + L
        full = 2 * T * D * D * (4 + 8) + 2 * L * D * D * (4 + 8) + 4 * S * S * D
        last = 2 * T * D * D * (4 + 8) + 2 * L * D * D * 3 + 4 * S * S * D
        blocks = (nl - 1) * full + last
    assert abs(f0 / B - blocks) < 0.01 * blocks, (f0 / B / 1e12, blocks / 1e12)
    ws1, f1 = _dit_query(p, 2 * B, HW, L, 0, 0)
    assert abs(f1 - 2 * f0) < 1e-6 * f0 and ws0 < 8 * 2 ** 30
    wsS, _ = _dit_query(p, B, HW, L, 0, FDMI_UNET_SAVE)
    wsG, _ = _dit_query(p, B, HW, L, 0, FDMI_UNET_SAVE | FDMI_UNET_INPUT_GRAD)
    assert wsS > 4 * ws0 and wsG >= wsS and wsS < 80 * 2 ** 30
    ml, pl = _dit_plan(kind, arch, r)
    _, fl = _dit_query(pl, B, HW, L, 0, 0)
    assert 1.01 * f0 < fl < 1.12 * f0
    if kind == "pixart":
        wsM, fM = _dit_query(p, B, HW, L, 1, 0)            # key lengths: per-sample cross-attention launches, same work
        assert fM == f0 and abs(wsM - ws0) < 0.05 * ws0


@pytest.mark.parametrize("kind,arch_name", [("pixart", "TINY_PIXART"), ("sd3", "TINY_SD3")])
def test_dit_plan_walks_every_mode_at_toy_size(kind, arch_name):
    from flash_diffusion_amd import workloads
    arch = getattr(workloads, arch_name)
    for lora in (0, 8):
        for precision in ("bf16", "fp32"):
            m, p = _dit_plan(kind, dict(arch, precision=precision), lora)
            for masked in ((0, 1) if kind == "pixart" else (0,)):
                for flags in (0, FDMI_UNET_SAVE, FDMI_UNET_SAVE | FDMI_UNET_INPUT_GRAD):
                    ws, fl = _dit_query(p, 2, 16, 12, masked, flags)
                    assert ws > 0 and fl > 0
    assert _lib.lib().fdmi_dit_workspace_bytes(p.handle, 2, 15, 16, 12, 0, 0) < 0      # H not a multiple of the patch size
