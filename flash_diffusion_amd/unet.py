"""MiUNet2DConditionModel -- drop-in for the reference's DiffusersUNet2DCondWrapper
(/root/reference/src/flash/models/unets/unet.py:47-127): same constructor keywords for the
hyper-parameters the reference pins (examples/train_flash_sd.py:56-114, train_flash_sdxl.py:66-118),
same forward signature / conditioning dict / freeze(), parameters named by their diffusers
state_dict keys -- but forward and backward run in libfdmi.so's hand-written HIP kernels through the
C-ABI plan API (include/fdmi.h: fdmi_unet_*).  torch supplies device memory, the stream and the
autograd graph edge only."""
from __future__ import annotations

import ctypes as C
import math
import weakref
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, f32, i32, i64, ptr, stream_ptr, vp

FDMI_UNET_SAVE, FDMI_UNET_INTERMEDIATE, FDMI_UNET_INPUT_GRAD = 1, 2, 4
FDMI_UNET_CTX_FILL, FDMI_UNET_CTX_REUSE, FDMI_UNET_CFG_HALVES = 8, 16, 32


from ._lib import UNetCfg  # noqa: E402  (C struct fdmi_unet_config)


class _Node(nn.Module):
    """Plain container so parameters get their dotted diffusers names."""


def _ensure_path(root: nn.Module, parts: Sequence[str]) -> nn.Module:
    m = root
    for p in parts:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    return m


def _listify(v, n):
    return [v] * n if isinstance(v, int) else list(v)


def unet_param_shapes(boc, down_types, up_types, layers_per_block, cross_dim, tlayers, heads, in_ch, out_ch,
                      class_embed_dim):
    """(name, shape) of every parameter, in diffusers state_dict naming."""
    out = []
    temb = boc[0] * 4

    def conv(n, co, ci, k):
        out.append((n + ".weight", (co, ci, k, k)))
        out.append((n + ".bias", (co,)))

    def lin(n, o, i, bias=True):
        out.append((n + ".weight", (o, i)))
        if bias:
            out.append((n + ".bias", (o,)))

    def norm(n, c):
        out.append((n + ".weight", (c,)))
        out.append((n + ".bias", (c,)))

    def resnet(n, ci, co):
        norm(n + ".norm1", ci)
        conv(n + ".conv1", co, ci, 3)
        lin(n + ".time_emb_proj", co, temb)
        norm(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    def transformer(n, c, layers):
        norm(n + ".norm", c)
        lin(n + ".proj_in", c, c)
        for k in range(layers):
            b = f"{n}.transformer_blocks.{k}"
            for i in (1, 2, 3):
                norm(f"{b}.norm{i}", c)
            for a, kv in (("attn1", c), ("attn2", cross_dim)):
                lin(f"{b}.{a}.to_q", c, c, False)
                lin(f"{b}.{a}.to_k", c, kv, False)
                lin(f"{b}.{a}.to_v", c, kv, False)
                lin(f"{b}.{a}.to_out.0", c, c)
            lin(f"{b}.ff.net.0.proj", 8 * c, c)
            lin(f"{b}.ff.net.2", c, 4 * c)
        lin(n + ".proj_out", c, c)

    nl = len(boc)
    conv("conv_in", boc[0], in_ch, 3)
    lin("time_embedding.linear_1", temb, boc[0])
    lin("time_embedding.linear_2", temb, temb)
    if class_embed_dim:
        lin("class_embedding.linear_1", temb, class_embed_dim)
        lin("class_embedding.linear_2", temb, temb)
    oc = boc[0]
    for i in range(nl):
        ic, oc = oc, boc[i]
        for j in range(layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
            if down_types[i].startswith("CrossAttn"):
                transformer(f"down_blocks.{i}.attentions.{j}", oc, tlayers[i])
        if i != nl - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", oc, oc, 3)
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1], tlayers[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    rboc, rtl = boc[::-1], tlayers[::-1]
    oc = rboc[0]
    for i in range(nl):
        prev, oc = oc, rboc[i]
        ic = rboc[min(i + 1, nl - 1)]
        n = layers_per_block + 1
        for j in range(n):
            skip = ic if j == n - 1 else oc
            rin = prev if j == 0 else oc
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, oc)
            if up_types[i].startswith("CrossAttn"):
                transformer(f"up_blocks.{i}.attentions.{j}", oc, rtl[i])
        if i != nl - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", oc, oc, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", out_ch, boc[0], 3)
    return out


class _Plan:
    def __init__(self, cfg, create="fdmi_unet_create"):
        self.handle = getattr(_lib.lib(), create)(C.byref(cfg))
        if not self.handle:
            raise RuntimeError("fdmi: " + _lib.lib().fdmi_last_error().decode())
        self.packed = False
        self.lora_bound = None
        self.workspaces: Dict[int, torch.Tensor] = {}
        self.next_slot = 1
        self.busy = set()
        self.gen: Dict[int, int] = {}   # per-slot generation: a stale finalizer / backward must not free a re-acquired slot
        self.last_use: Dict[int, tuple] = {}   # slot -> (stream, event after its last launch): cross-stream reuse waits on it

    def enter(self, slot):
        """A run slot (workspace + tape + statistics pool) is used from one stream at a time.  The teacher loop runs slot 0 on a
        side stream while no-grad calls of the DMD / GAN branches use the same slot from the caller's stream after the join:
        instead of relying on every caller to have joined, a use from a DIFFERENT stream than the previous one first waits for
        an event recorded behind that previous use (same-stream reuse is ordered by the stream itself: no event wait)."""
        cur = torch.cuda.current_stream()
        last = self.last_use.get(slot)
        if last is not None and last[0] != cur:
            cur.wait_event(last[1])

    def leave(self, slot):
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        self.last_use[slot] = (cur, ev)

    def release(self, slot, gen):
        if self.gen.get(slot) == gen:
            self.busy.discard(slot)

    def close(self):
        if self.handle:
            _lib.lib().fdmi_unet_destroy(self.handle)
            self.handle = None


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, sample, t, enc, vec, flags, *params):
        out, slot = mod._run_forward(sample, t, enc, vec, flags)
        plan = mod._plan()
        ctx.mod, ctx.slot, ctx.gen = mod, slot, plan.gen[slot]
        # a saved run whose graph is dropped without backward (e.g. the detached D-step branch,
        # flash_diffusion_model.py:583) must give its slot back -- but only while the slot still holds THIS run (the
        # graph of step n is usually dropped after step n+1 re-acquired the slot its backward had freed)
        weakref.finalize(ctx, plan.release, slot, ctx.gen)
        ctx.needs_x = sample.requires_grad
        ctx.nparams = len(params)
        ctx.xshape = sample.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.mod._plan().gen.get(ctx.slot) != ctx.gen:
            raise RuntimeError("fdmi: backward of a denoiser call whose saved activations were released (slot reused)")
        gx = ctx.mod._run_backward(ctx.slot, grad_out, ctx.needs_x, ctx.xshape, ctx.gen)
        return (None, gx, None, None, None, None) + (None,) * ctx.nparams


class MiUNet2DConditionModel(nn.Module):
    """See module docstring.  Unknown diffusers kwargs are accepted and ignored when they carry the
    reference's values (None / defaults); unsupported non-default values raise."""
    supports_ctx_cache = True  # forward(..., ctx_cache="fill"|"reuse"): see FDMI_UNET_CTX_* in include/fdmi.h
    supports_cfg_halves = True  # forward(..., cfg_halves=True): [x | x] batch, see FDMI_UNET_CFG_HALVES

    def __init__(self, in_channels=4, out_channels=4, down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, cross_attention_dim=768, transformer_layers_per_block=1, attention_head_dim=8,
                 norm_num_groups=32, norm_eps=1e-5, class_embed_type=None, projection_class_embeddings_input_dim=None,
                 flip_sin_to_cos=True, freq_shift=0, use_linear_projection=True, precision="bf16", **unused):
        """precision: "bf16" = the measured path (bf16 MFMA, fp32 accumulation: the reference's bf16-mixed); "fp32" = the
        VALIDATION plan (fp32 storage, exact-f32 MFMA, fp64 norm statistics -- csrc/ref32.hip), the parity gate against the
        fp32 CPU oracle at north_star's 1e-3."""
        super().__init__()
        assert precision in ("bf16", "fp32"), precision
        assert use_linear_projection, "only use_linear_projection=True (as in every reference example)"
        for k, allowed in (("mid_block_type", ("UNetMidBlock2DCrossAttn",)), ("act_fn", ("silu",)),
                           ("resnet_time_scale_shift", ("default",)), ("time_embedding_type", ("positional",)),
                           ("conv_in_kernel", (3,)), ("conv_out_kernel", (3,)), ("addition_embed_type", (None,)),
                           ("dual_cross_attention", (False,)), ("only_cross_attention", (False,))):
            if k in unused and unused[k] not in allowed:
                raise NotImplementedError(f"{k}={unused[k]!r} is outside the reference's configurations")
        boc = list(block_out_channels)
        nl = len(boc)
        assert 1 <= nl <= 4
        self.config_dict = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                                down_block_types=list(down_block_types), up_block_types=list(up_block_types),
                                layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                                transformer_layers_per_block=_listify(transformer_layers_per_block, nl),
                                attention_head_dim=_listify(attention_head_dim, nl), norm_num_groups=norm_num_groups,
                                norm_eps=norm_eps,
                                class_embed_dim=(projection_class_embeddings_input_dim or 0)
                                if class_embed_type == "projection" else 0,
                                flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift, precision=precision)
        c = self.config_dict
        self._shapes = unet_param_shapes(boc, c["down_block_types"], c["up_block_types"], layers_per_block,
                                         cross_attention_dim, c["transformer_layers_per_block"],
                                         c["attention_head_dim"], in_channels, out_channels, c["class_embed_dim"])
        for name, shape in self._shapes:
            parts = name.split(".")
            leaf = _ensure_path(self, parts[:-1])
            fan_in = int(math.prod(shape[1:])) if len(shape) > 1 else 1
            if "norm" in parts[-2] and parts[-1] == "weight":
                v = torch.ones(shape)
            elif len(shape) == 1:
                v = torch.zeros(shape)
            else:
                v = torch.randn(shape) * (fan_in ** -0.5)
            leaf.register_parameter(parts[-1], nn.Parameter(v))
        self.lora_rank = 0
        self._lora_targets: List[str] = []
        self.last_flops = 0.0
        self.step_flops = 0.0   # running sum of algorithmic MFMA flops (bench.py resets it)

    # ---- plan management ---------------------------------------------------------------------------
    _PLANS: Dict[int, _Plan] = {}

    def _cfg_struct(self):
        c = self.config_dict
        s = UNetCfg()
        s.in_channels, s.out_channels, s.n_levels = c["in_channels"], c["out_channels"], len(c["block_out_channels"])
        rheads = c["attention_head_dim"]
        for i, v in enumerate(c["block_out_channels"]):
            s.block_out[i] = v
            s.down_attn[i] = int(c["down_block_types"][i].startswith("CrossAttn"))
            s.up_attn[i] = int(c["up_block_types"][i].startswith("CrossAttn"))
            s.tlayers[i] = c["transformer_layers_per_block"][i]
            s.heads[i] = rheads[i]
        s.layers_per_block, s.cross_dim = c["layers_per_block"], c["cross_attention_dim"]
        s.groups, s.eps = c["norm_num_groups"], c["norm_eps"]
        s.class_embed_dim = c["class_embed_dim"]
        s.flip_sin_to_cos, s.freq_shift = int(c["flip_sin_to_cos"]), float(c["freq_shift"])
        s.precision = 1 if c.get("precision", "bf16") == "fp32" else 0
        return s

    def _plan(self) -> _Plan:
        p = MiUNet2DConditionModel._PLANS.get(id(self))
        if p is None:
            p = _Plan(self._cfg_struct())
            MiUNet2DConditionModel._PLANS[id(self)] = p
            weakref.finalize(self, MiUNet2DConditionModel._drop_plan, id(self))
        return p

    @staticmethod
    def _drop_plan(key):
        p = MiUNet2DConditionModel._PLANS.pop(key, None)
        if p is not None:
            p.close()

    def invalidate_plan(self):
        """Call after changing frozen base weights in place (load_state_dict does it for you)."""
        p = MiUNet2DConditionModel._PLANS.get(id(self))
        if p is not None:
            p.packed = False

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_plan()
        return r

    def _base_params(self):
        return [(n, p) for n, p in self.named_parameters() if ".lora_" not in n]

    def _ensure_packed(self, device):
        plan = self._plan()
        L = _lib.lib()
        if not plan.packed:
            n = L.fdmi_unet_num_params(plan.handle)
            expected = {}
            buf = C.create_string_buffer(512)
            ne = i64()
            for i in range(n):
                check(L.fdmi_unet_param_name(plan.handle, i, buf, 512, C.byref(ne)))
                expected[buf.value.decode()] = ne.value
            mine = dict(self._base_params())
            assert set(expected) == set(mine), (set(expected) ^ set(mine))
            for name, p in mine.items():
                assert p.is_cuda and p.dtype == torch.float32, f"{name}: parameters must be fp32 on the GPU"
                t = p.detach().contiguous()
                check(L.fdmi_unet_set_param(plan.handle, name.encode(), ptr(t), t.numel(), stream_ptr()))
            torch.cuda.current_stream().synchronize()
            check(L.fdmi_unet_ready(plan.handle))
            plan.packed = True
        if self.lora_rank:
            self._bind_lora(plan)
        return plan

    # ---- LoRA (peft semantics: examples/train_flash_sd.py:191-200) ----------------------------------
    def add_adapter(self, r: int, target_modules=("to_k", "to_q", "to_v", "to_out.0"), init_std_b: float = 0.0,
                    generator: Optional[torch.Generator] = None):
        """y = W x + B(A x); A ~ N(0, 1/r) ('gaussian' init), B = 0 (init_std_b > 0 only for tests).
        All base parameters are frozen (peft behaviour); LoRA tensors live in ONE flat fp32 buffer
        (and one flat grad buffer) so the data-parallel all-reduce is a single collective."""
        assert self.lora_rank == 0, "adapter already added"
        self.lora_rank = r
        targets = []
        for name, shape in self._shapes:
            if not name.endswith(".weight") or ".attn" not in name:
                continue
            base = name[:-len(".weight")]
            if any(base.endswith("." + t) for t in target_modules):
                targets.append((base, shape))
        dev = next(self.parameters()).device
        total = sum(r * (s[0] + s[1]) for _, s in targets)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.parameters():
            p.requires_grad = False
        for base, (o, i) in targets:
            a = flat[off:off + r * i].view(r, i)
            a.copy_(torch.randn(r, i, generator=generator).to(dev) / r)
            off += r * i
            b = flat[off:off + o * r].view(o, r)
            if init_std_b:
                b.copy_(torch.randn(o, r, generator=generator).to(dev) * init_std_b)
            off += o * r
            leaf = _ensure_path(self, base.split("."))
            la, lb = _ensure_path(leaf, ["lora_A", "default"]), _ensure_path(leaf, ["lora_B", "default"])
            la.register_parameter("weight", nn.Parameter(a))
            lb.register_parameter("weight", nn.Parameter(b))
        self._lora_targets = [b for b, _ in targets]
        self._lora_flat = flat
        self._lora_grad = None
        return self

    def lora_parameters(self):
        out = []
        for base in self._lora_targets:
            leaf = _ensure_path(self, base.split("."))
            out += [leaf.lora_A.default.weight, leaf.lora_B.default.weight]
        return out

    def _reflatten_lora(self, device):
        """(Re)establish the flat-buffer invariant after .to(device) / deepcopy / load_state_dict."""
        params = self.lora_parameters()
        total = sum(p.numel() for p in params)
        flat = getattr(self, "_lora_flat", None)
        ok = flat is not None and flat.device == device and flat.numel() == total
        if ok:
            off = 0
            for p in params:
                if p.data_ptr() != flat.data_ptr() + off * 4:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.empty(total, dtype=torch.float32, device=device)
            off = 0
            for p in params:
                flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            self._lora_flat = flat
            self._lora_grad = None
        g = getattr(self, "_lora_grad", None)
        if g is None or g.device != device:
            self._lora_grad = torch.zeros(total, dtype=torch.float32, device=device)
        return ok

    def lora_flat(self):
        return self._lora_flat

    def lora_flat_grad(self):
        return self._lora_grad

    def _bind_lora(self, plan: _Plan):
        dev = next(self.parameters()).device
        same = self._reflatten_lora(dev)
        key = (self._lora_flat.data_ptr(), self._lora_grad.data_ptr())
        if same and plan.lora_bound == key:
            return
        L = _lib.lib()
        off = 0
        r = self.lora_rank
        params = self.lora_parameters()
        for idx, base in enumerate(self._lora_targets):
            a, b = params[2 * idx], params[2 * idx + 1]
            ga = self._lora_grad[off:off + a.numel()]
            off += a.numel()
            gb = self._lora_grad[off:off + b.numel()]
            off += b.numel()
            check(L.fdmi_unet_set_lora(plan.handle, base.encode(), ptr(a), ptr(b), ptr(ga), ptr(gb), r))
        plan.lora_bound = key

    def _attach_lora_grads(self):
        """param.grad <- views of the flat grad buffer (ops.attach_flat_grads: zeroed where the grads were None)."""
        ops.attach_flat_grads(self.lora_parameters(), self._lora_grad)

    # ---- reference wrapper contract (unet.py:66-127) -------------------------------------------------
    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                conditioning: Dict[str, torch.Tensor], down_intrablock_additional_residuals=None,
                return_intermediate: bool = False, *args, **kwargs):
        assert isinstance(conditioning, dict), "conditionings must be a dictionary"
        class_labels = conditioning["cond"].get("vector", None)
        crossattn = conditioning["cond"].get("crossattn", None)
        concat = conditioning["cond"].get("concat", None)
        if concat is not None:
            sample = torch.cat([sample, concat], dim=1)
        assert crossattn is not None, "crossattn conditioning is required by UNet2DConditionModel"
        B = sample.shape[0]
        # T2I-adapter residuals (UW:100-106): constants of this call (the adapter is frozen), handed to the plan right
        # before the forward (include/fdmi.h: fdmi_unet_set_down_residuals)
        self._pending_res = None
        if down_intrablock_additional_residuals is not None:
            want = self._down_residual_shapes(B, sample.shape[2], sample.shape[3])
            res = [r.detach().float().contiguous() for r in down_intrablock_additional_residuals]
            assert [tuple(r.shape) for r in res] == want, f"adapter residual shapes {[tuple(r.shape) for r in res]} != {want}"
            self._pending_res = res
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([float(timestep)], device=sample.device)
        t = timestep.to(device=sample.device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        flags = FDMI_UNET_INTERMEDIATE if return_intermediate else 0
        # frozen-teacher loops: the caller promises the context equals the one of the last ctx_cache="fill" call, so
        # the cross-attention K/V projections are read back instead of recomputed (include/fdmi.h FDMI_UNET_CTX_*)
        ctx_cache = kwargs.pop("ctx_cache", None)
        if ctx_cache is not None and not self.lora_rank:
            flags |= FDMI_UNET_CTX_FILL if ctx_cache == "fill" else FDMI_UNET_CTX_REUSE
        # classifier-free-guidance batch [x | x] (same sample and timestep in both halves, the caller's promise): the layers
        # before the first cross-attention are computed once (FDMI_UNET_CFG_HALVES)
        if kwargs.pop("cfg_halves", False) and not self.lora_rank and down_intrablock_additional_residuals is None:
            flags |= FDMI_UNET_CFG_HALVES
        lora = self.lora_parameters() if self.lora_rank else []
        need_grad = torch.is_grad_enabled() and (sample.requires_grad or any(p.requires_grad for p in lora))
        sample = sample.float().contiguous()
        enc = crossattn.float().contiguous()
        vec = class_labels.float().contiguous() if class_labels is not None else None
        if need_grad:
            return _UNetFn.apply(self, sample, t, enc, vec, flags | FDMI_UNET_SAVE, *lora)
        out, _ = self._run_forward(sample, t, enc, vec, flags)
        return out

    def _down_residual_shapes(self, B, H, W):
        """shape of each down block's output where diffusers adds the adapter residual: a cross-attention block before
        its downsampler, an attention-free block after it"""
        c = self.config_dict
        nl = len(c["block_out_channels"])
        out = []
        for i, ch in enumerate(c["block_out_channels"]):
            sh = i + (1 if (not c["down_block_types"][i].startswith("CrossAttn") and i != nl - 1) else 0)
            out.append((B, ch, H >> sh, W >> sh))
        return out

    # ---- C-ABI calls ---------------------------------------------------------------------------------
    def _run_forward(self, sample, t, enc, vec, flags):
        assert sample.is_cuda, "MiUNet2DConditionModel runs on the GPU only (no CPU fallback)"
        plan = self._ensure_packed(sample.device)
        L = _lib.lib()
        B, _, H, W = sample.shape
        Lc = enc.shape[1]
        save = bool(flags & FDMI_UNET_SAVE)
        if save:
            slot = next(s for s in range(1, 8) if s not in plan.busy)
            plan.busy.add(slot)
            plan.gen[slot] = plan.gen.get(slot, 0) + 1
            qflags = flags | (FDMI_UNET_INPUT_GRAD if sample.requires_grad else 0)
        else:
            slot, qflags = 0, flags
        need = L.fdmi_unet_workspace_bytes(plan.handle, B, H, W, Lc, qflags)
        if need < 0:
            raise RuntimeError("fdmi: " + L.fdmi_last_error().decode())
        ws = plan.workspaces.get(slot)
        if ws is None or ws.numel() < need or ws.device != sample.device:
            ws = torch.empty(need, dtype=torch.uint8, device=sample.device)
            plan.workspaces[slot] = ws
        cfg = self.config_dict
        nl = len(cfg["block_out_channels"])
        if flags & FDMI_UNET_INTERMEDIATE:
            f = 1 << (nl - 1)
            out = torch.empty(B, cfg["block_out_channels"][-1], H // f, W // f, dtype=torch.float32, device=sample.device)
        else:
            out = torch.empty(B, cfg["out_channels"], H, W, dtype=torch.float32, device=sample.device)
        res = getattr(self, "_pending_res", None)
        self._pending_res = None
        if res is not None:
            arr = (C.c_void_p * len(res))(*[r.data_ptr() for r in res])
            check(L.fdmi_unet_set_down_residuals(plan.handle, arr, len(res), 1.0))
        plan.enter(slot)
        try:
            check(L.fdmi_unet_forward(plan.handle, slot, ptr(sample), ptr(t), ptr(enc), ptr(vec), ptr(out), B, H, W, Lc,
                                      ptr(ws), ws.numel(), flags, stream_ptr()))
        finally:
            plan.leave(slot)
        self.last_flops = L.fdmi_unet_last_flops(plan.handle)
        self.step_flops += self.last_flops
        return out, slot

    @torch.no_grad()
    def teacher_loop(self, x, timesteps, crossattn2, vector2, coeffs):
        """The frozen teacher's whole CFG loop in ONE C-ABI call (include/fdmi.h: fdmi_teacher_loop): x [B,C,H,W] f32 is
        advanced through the steps `timesteps` (host floats) with the host coefficient rows `coeffs` ([n][6], see
        schedulers.DPMSolverMultistepScheduler.loop_coefficients); crossattn2 [2B,L,D] / vector2 [2B,V] hold the conditional
        rows first, then the unconditional ones.  Returns the new latent."""
        assert x.is_cuda and not self.lora_rank, "teacher_loop is for a frozen denoiser on the GPU"
        plan = self._ensure_packed(x.device)
        L = _lib.lib()
        B, _, H, W = x.shape
        n = len(timesteps)
        assert crossattn2.shape[0] == 2 * B and len(coeffs) == n and all(len(r) == 6 for r in coeffs)
        Lc = crossattn2.shape[1]
        x = x.float().contiguous().clone()
        enc = crossattn2.float().contiguous()
        vec = vector2.float().contiguous() if vector2 is not None else None
        # (the loop runs its forwards with CTX_FILL / CTX_REUSE and, by default, the [x | x] prefix dedupe)
        need = max(L.fdmi_unet_workspace_bytes(plan.handle, 2 * B, H, W, Lc, FDMI_UNET_CTX_FILL),
                   L.fdmi_unet_workspace_bytes(plan.handle, 2 * B, H, W, Lc, FDMI_UNET_CTX_FILL | FDMI_UNET_CFG_HALVES))
        sneed = L.fdmi_teacher_loop_scratch_bytes(plan.handle, B, H, W)
        if need < 0 or sneed < 0:
            raise RuntimeError("fdmi: " + L.fdmi_last_error().decode())
        ws = plan.workspaces.get(0)
        if ws is None or ws.numel() < need or ws.device != x.device:
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            plan.workspaces[0] = ws
        sc = plan.workspaces.get("teacher_loop")
        if sc is None or sc.numel() < sneed or sc.device != x.device:
            sc = torch.empty(sneed, dtype=torch.uint8, device=x.device)
            plan.workspaces["teacher_loop"] = sc
        ts = (C.c_float * n)(*[float(t) for t in timesteps])
        cf = (C.c_float * (6 * n))(*[float(v) for r in coeffs for v in r])
        plan.enter(0)
        try:
            check(L.fdmi_teacher_loop(plan.handle, 0, ptr(x), ts, n, ptr(enc), ptr(vec), cf, B, H, W, Lc, ptr(ws), ws.numel(),
                                      ptr(sc), sc.numel(), stream_ptr()))
        finally:
            plan.leave(0)
        self.last_flops = L.fdmi_unet_last_flops(plan.handle)
        self.step_flops += self.last_flops
        return x

    def _run_backward(self, slot, grad_out, needs_x, xshape, gen=None):
        plan = self._plan()
        L = _lib.lib()
        if self.lora_rank:
            self._attach_lora_grads()
        g = grad_out.float().contiguous()
        gx = torch.empty(xshape, dtype=torch.float32, device=g.device) if needs_x else None
        plan.enter(slot)
        try:
            check(L.fdmi_unet_backward(plan.handle, slot, ptr(g), ptr(gx), stream_ptr()))
        finally:
            plan.leave(slot)
            if gen is None:
                plan.busy.discard(slot)
            else:
                plan.release(slot, gen)
        self.last_flops = L.fdmi_unet_last_flops(plan.handle)
        self.step_flops += self.last_flops
        return gx

    def release_saved(self):
        """Drop saved-for-backward runs that will never be back-propagated (e.g. after an exception)."""
        self._plan().busy.clear()
