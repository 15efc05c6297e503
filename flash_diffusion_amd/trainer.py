"""TrainingPipeline -- Lightning-free twin of the reference trainer
(/root/reference/src/flash/trainer/trainer.py:16-251, "TR") for the distillation hot path:
  * configure_optimizers: regex-selected parameter groups per optimizer, everything unmatched frozen
    (TR:76-139);
  * training_step: one optimizer -> single backward; several optimizers -> the manual loop that runs
    the FULL model forward once per optimizer with step=i (TR:187-218);
  * data parallel: one process per GPU, gradients of the trainable (LoRA / discriminator) tensors are
    summed with ONE RCCL all-reduce per optimizer on a flat buffer (the reference gets bucketed NCCL
    all-reduces from Lightning's DDP strategy, examples/train_flash_sd.py:386).  Because the teacher is
    frozen, the all-reduce + AdamW of iteration i run on a side stream concurrently with the teacher
    loop of iteration i+1; the student forward waits on that stream (FlashDiffusion.before_student);
  * deferred backward (GPU only; A/B switch FDMI_DEFER_BACKWARD=0): the backward + all-reduce + AdamW of a forward are
    ISSUED from the before_student hook of the NEXT forward (the next iteration's, or -- with several optimizers -- the
    next optimizer's forward of the same batch), i.e. after that forward's teacher loop has been handed to its side
    stream and before its student call.  The teacher is frozen and reads nothing the backward writes, so its 2B-row
    kernels run beside the backward's many small launches (rank-r LoRA products, split-K tails, deep UNet levels)
    instead of after them; the order of every read and write of the trainable tensors is the reference's (backward i,
    step i, forward i+1).  With a discriminator the generator loss also back-propagates through the frozen teacher's
    plan (GAN backbone) while the next teacher loop runs on it: the two runs live in different slots of the plan (own
    workspace, own statistics pool; a SAVE run never touches the plan-owned context K/V cache), so they share only
    read-only weights.  finish() -- and every reader of the parameters on this class -- drains the outstanding backward.
AdamW itself is the fused HIP kernel fdmi_adamw (torch.optim.AdamW semantics)."""
from __future__ import annotations

import logging
import os
import re
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops


@dataclass
class TrainingConfig:
    """Mirror of the fields of the reference TrainingConfig that the hot path reads
    (trainer/training_config.py:10-136)."""
    experiment_id: Optional[str] = None
    optimizers_name: List[str] = field(default_factory=lambda: ["AdamW"])
    optimizers_kwargs: List[dict] = field(default_factory=lambda: [{}])
    learning_rates: List[float] = field(default_factory=lambda: [1e-3])
    lr_schedulers_name: List[Optional[str]] = field(default_factory=lambda: [None])
    lr_schedulers_kwargs: List[dict] = field(default_factory=lambda: [{}])
    lr_schedulers_interval: List[Optional[str]] = field(default_factory=lambda: ["step"])
    lr_schedulers_frequency: List[Optional[int]] = field(default_factory=lambda: [1])
    trainable_params: List[List[str]] = field(default_factory=lambda: [[".*"]])
    log_keys: Any = "txt"
    log_samples_model_kwargs: dict = field(default_factory=dict)

    def __post_init__(self):
        if self.optimizers_kwargs == [{}]:
            self.optimizers_kwargs = [{} for _ in self.optimizers_name]
        assert len(self.optimizers_name) == len(self.optimizers_kwargs)
        if self.trainable_params == [[".*"]]:
            self.trainable_params = [[".*"] for _ in self.optimizers_name]
        assert len(self.optimizers_name) == len(self.trainable_params)
        assert len(self.optimizers_name) == len(self.learning_rates)
        # training_config.py:108-131: the scheduler lists are sized by len(lr_schedulers_name) -- NOT by the optimizer count
        # (a 2-optimizer config with lr_schedulers_name=["StepLR"] is valid upstream: only optimizer 0 gets a scheduler);
        # defaults are recognised by VALUE as upstream does (`!= [{}]`, `!= [1]`, `!= ["step"]`) and broadcast to that length
        n = len(self.lr_schedulers_name)
        if self.lr_schedulers_kwargs != [{}]:
            assert n == len(self.lr_schedulers_kwargs), (
                f"The length of lr_schedulers_name ({n}) must be equal to the length of lr_schedulers_kwargs "
                f"({len(self.lr_schedulers_kwargs)})")
            if self.lr_schedulers_frequency != [1]:
                assert n == len(self.lr_schedulers_frequency), (
                    f"The length of lr_schedulers_name ({n}) must be equal to the length of lr_schedulers_frequency "
                    f"({len(self.lr_schedulers_frequency)})")
            else:
                self.lr_schedulers_frequency = [1 for _ in range(n)]
            if self.lr_schedulers_interval != ["step"]:
                assert n == len(self.lr_schedulers_interval), (
                    f"The length of lr_schedulers_name ({n}) must be equal to the length of lr_schedulers_interval "
                    f"({len(self.lr_schedulers_interval)})")
            else:
                self.lr_schedulers_interval = ["step" for _ in range(n)]
        else:
            self.lr_schedulers_kwargs = [{} for _ in range(n)]
            # (upstream leaves frequency / interval at their one-entry defaults on this branch and would raise IndexError for
            # a second named scheduler without kwargs; broadcasting them is the one deliberate superset here)
            if len(self.lr_schedulers_frequency) < n:
                self.lr_schedulers_frequency = list(self.lr_schedulers_frequency) + [1] * (n - len(self.lr_schedulers_frequency))
            if len(self.lr_schedulers_interval) < n:
                self.lr_schedulers_interval = list(self.lr_schedulers_interval) + ["step"] * (n - len(self.lr_schedulers_interval))
        assert n <= len(self.optimizers_name), "more lr schedulers than optimizers (TR:140-166 indexes self.optims[i])"


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay 1e-2, betas (0.9, 0.999), eps 1e-8), one
    fdmi_adamw launch per contiguous fp32 buffer.  Parameters that are views into one flat buffer
    (the LoRA tensors) are updated with a single launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flat=None, flat_grad=None,
                 amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None):
        # torch.optim.AdamW's remaining keywords are accepted at the values that leave its update rule unchanged; anything
        # that would change the arithmetic this kernel implements is refused, not ignored
        if amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("FusedAdamW implements plain AdamW: amsgrad / maximize / capturable / differentiable "
                                      "must be False")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat, self.flat_grad = flat, flat_grad
        self.grad_scale = 1.0
        self._flat_state = None
        self.step_count = 0

    def state_dict(self):
        sd = super().state_dict()   # (TrainingPipeline's pre-hook has drained a deferred step by now)
        # the moments of the flat (LoRA) buffer live outside torch's per-parameter state
        if self._flat_state is not None:
            sd["flat_state"] = {"m": self._flat_state[0], "v": self._flat_state[1], "step_count": self.step_count}
        else:
            sd["flat_state"] = {"step_count": self.step_count}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        fs = sd.pop("flat_state", None)
        super().load_state_dict(sd)
        if fs is not None:
            self.step_count = int(fs.get("step_count", 0))
            if "m" in fs and self.flat is not None:
                self._flat_state = (fs["m"].to(self.flat).clone(), fs["v"].to(self.flat).clone())

    @torch.no_grad()
    def step(self, closure=None):
        self.step_count += 1
        for grp in self.param_groups:
            b1, b2 = grp["betas"]
            if self.flat is not None:
                if self._flat_state is None:
                    self._flat_state = (torch.zeros_like(self.flat), torch.zeros_like(self.flat))
                m, v = self._flat_state
                ops.adamw_(self.flat, self.flat_grad, m, v, grp["lr"], b1, b2, grp["eps"], grp["weight_decay"],
                           self.step_count, self.grad_scale)
                continue
            for p in grp["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                assert p.is_contiguous() and p.grad.is_contiguous()
                ops.adamw_(p.data, p.grad, st["m"], st["v"], grp["lr"], b1, b2, grp["eps"], grp["weight_decay"],
                           self.step_count, self.grad_scale)


class TrainingPipeline(nn.Module):
    def __init__(self, model: nn.Module, pipeline_config: TrainingConfig, verbose: bool = False, overlap: bool = True,
                 share_start_idx: Optional[bool] = None, **kwargs):
        super().__init__()
        self.model = model
        self.pipeline_config = pipeline_config
        self.log_samples_model_kwargs = pipeline_config.log_samples_model_kwargs
        self.verbose = verbose
        self.automatic_optimization = True
        self.optims: List[torch.optim.Optimizer] = []
        self.overlap = overlap
        self.reduced_grad_hook = None   # callable(optimizer, grad_scale): see _reduce_and_step
        self._comm_stream = None
        # comm_timing = True (bench.py): HIP events around every gradient exchange on the stream it runs on, and around the
        # main stream's wait for the exchange + optimizer step (comm_report() reads them after a device synchronisation)
        self.comm_timing = False
        self._comm_events: List[Any] = []     # (start, stop) of each all-reduce
        self._wait_events: List[Any] = []     # (before, after) of each main-stream wait on `_pending`
        self.comm_payload_bytes = 0
        self._pending = None      # event recorded on the comm stream after the deferred optimizer step
        self._deferred = None     # (loss, optimizer index) whose backward + step the next before_student hook issues
        self.global_rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        # a process group of ONE rank still runs the collective (a no-op exchange): the RCCL path of a single-GPU box is
        # then the same code the 8-GPU job runs
        self.distributed = torch.distributed.is_initialized()
        self.timer = None
        # Data-parallel training: all ranks run the SAME number of teacher steps per iteration -- the model's start
        # index comes from a host generator seeded identically everywhere (rank 0's seed, broadcast once here; nothing is
        # exchanged inside the step).  Default: on whenever more than one rank trains; share_start_idx=False restores the
        # reference's per-rank draw (FD:167), where every step waits for the rank with the longest teacher loop.
        self.share_start_idx = (self.world > 1) if share_start_idx is None else bool(share_start_idx)
        if self.share_start_idx and hasattr(model, "share_start_idx"):
            seed = [int(torch.randint(0, 2 ** 31 - 1, (1,)).item())]
            if self.distributed and self.world > 1:
                torch.distributed.broadcast_object_list(seed, src=0)
            model.share_start_idx(seed[0])
        self.global_step = 0      # training_step calls: the counter the shared start-index draw is derived from (flash.share_start_idx)

    @property
    def device(self):
        return next(self.model.parameters()).device

    # ---- TR:76-139 -------------------------------------------------------------------------------
    def configure_optimizers(self):
        cfg = self.pipeline_config
        optimizers = []
        student = getattr(self.model, "student_denoiser", None)
        for i, name in enumerate(cfg.optimizers_name):
            params, n_params, names = [], 0, []
            for pname, p in self.model.named_parameters():
                if any(re.match(re.compile(rx), pname) for rx in cfg.trainable_params[i]) and p.requires_grad:
                    params.append(p)
                    names.append(pname)
                    n_params += p.numel()
            logging.info(f"Number of trainable parameters for optimizer {i}: {n_params}")
            kw = dict(cfg.optimizers_kwargs[i])
            if name == "AdamW" and params and all(p.is_cuda for p in params):
                flat = flat_grad = None
                lora = student.lora_parameters() if (student is not None and getattr(student, "lora_rank", 0)) else []
                if lora and len(lora) == len(params) and all(a is b for a, b in zip(lora, params)):
                    student._reflatten_lora(params[0].device)
                    flat, flat_grad = student.lora_flat(), student.lora_flat_grad()
                opt = FusedAdamW(params, lr=cfg.learning_rates[i], flat=flat, flat_grad=flat_grad, **kw)
            else:
                opt = getattr(torch.optim, name)(params, lr=cfg.learning_rates[i], **kw)
            optimizers.append(opt)
        if len(optimizers) > 1:
            self.automatic_optimization = False
        self.optims = optimizers
        for opt in optimizers:   # optimizer.state_dict() (checkpoint callbacks) first drains a deferred backward / step
            opt.register_state_dict_pre_hook(lambda optimizer: self.drain())
        self.lr_schedulers = self.configure_lr_schedulers()
        for pname, p in self.model.named_parameters():
            keep = any(re.match(re.compile(rx), pname) for rxs in cfg.trainable_params for rx in rxs) and p.requires_grad
            if not keep:
                p.requires_grad = False
        logging.info("Number of trainable parameters: %d",
                     sum(p.numel() for p in self.model.parameters() if p.requires_grad))
        if self.overlap and torch.cuda.is_available():
            self._comm_stream = torch.cuda.Stream()
            self.model.before_student = self._before_student
        # readers that bypass this class (pipe.model.state_dict(), EMA / checkpoint callbacks walking the model) must not
        # see parameters one optimizer step stale: drain before the model serialises itself
        if not getattr(self, "_sd_hook", None):
            self._sd_hook = self.model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.drain())
        if any(sc is not None for sc in self.lr_schedulers):
            return optimizers, [sc for sc in self.lr_schedulers]
        return optimizers

    def configure_lr_schedulers(self):
        """TR:140-166: torch.optim.lr_scheduler classes by name on the matching optimizer (FusedAdamW reads its group's lr
        at every step, so any torch scheduler drives it).  Without Lightning the stepping is this class's job: interval
        "step" schedulers advance every `frequency` optimizer steps in the one-optimizer (automatic) mode; in the manual
        several-optimizer loop Lightning leaves scheduler stepping to the user and the reference never does it (TR:194-217),
        so they stay untouched there; interval "epoch" ones advance in on_train_epoch_end()."""
        import importlib
        cfg = self.pipeline_config
        out = []
        for i, name in enumerate(cfg.lr_schedulers_name):
            if name is None:
                out.append(None)
                continue
            cls = getattr(importlib.import_module("torch.optim.lr_scheduler"), name)
            out.append({"scheduler": cls(self.optims[i], **cfg.lr_schedulers_kwargs[i]),
                        "interval": cfg.lr_schedulers_interval[i], "monitor": "val_loss",
                        "frequency": cfg.lr_schedulers_frequency[i] or 1})
        return out

    def _lr_step(self, i, interval):
        sc = self.lr_schedulers[i] if i < len(getattr(self, "lr_schedulers", [])) else None
        if sc is None or sc["interval"] != interval:
            return
        sc["_n"] = sc.get("_n", 0) + 1
        if sc["_n"] % sc["frequency"] == 0:
            sc["scheduler"].step()

    def on_train_epoch_end(self):
        # the last training_step's backward + optimizer step may still be deferred: it must run with THIS epoch's lr, as in
        # the immediate schedule and in the reference (Lightning steps epoch-interval schedulers after the last optimizer step)
        self.drain()
        for i in range(len(self.optims)):
            self._lr_step(i, "epoch")

    def optimizers(self):
        """Lightning's accessor.  A caller that reads optimizer state (checkpointing) must see the step of the last
        training_step: drain the deferred backward first."""
        self.drain()
        return self.optims

    def drain(self):
        """Issue the deferred backward + optimizer step (if any) and make the current stream wait for it -- without a device
        synchronisation.  Every reader of trainable parameters / optimizer state outside training_step goes through here."""
        self._run_deferred()
        self._wait_pending()

    # ---- gradient exchange + optimizer step ---------------------------------------------------------
    def _wait_pending(self):
        if self._pending is not None and torch.cuda.is_available():
            cur = torch.cuda.current_stream()
            if self.comm_timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_event(self._pending)
                e1.record(cur)
                self._wait_events.append((e0, e1))
            else:
                cur.wait_event(self._pending)
            self._pending = None

    def comm_report(self, reset=True):
        """{"allreduce_ms", "exposed_ms", "exchanges", "payload_bytes"}: mean duration of a gradient exchange on its stream and the
        mean time the main stream stood still waiting for exchange + optimizer step (what the step pays for data parallelism when
        the overlap with the next teacher loop fails), over the steps since the last reset.  Call after a device synchronisation."""
        ar = [a.elapsed_time(b) for a, b in self._comm_events]
        ex = [a.elapsed_time(b) for a, b in self._wait_events]
        rec = {"exchanges": len(ar), "allreduce_ms": sum(ar) / len(ar) if ar else None,
               "exposed_ms": sum(ex) / len(ex) if ex else None, "waits": len(ex), "payload_bytes": self.comm_payload_bytes}
        if reset:
            self._comm_events, self._wait_events = [], []
        return rec

    def _defer_ok(self):
        """the backward of a forward may wait for the next forward's hook: GPU, overlap on, a model that calls the hook
        right before its student call"""
        return (self._comm_stream is not None and getattr(self.model, "calls_before_student", False)
                and os.environ.get("FDMI_DEFER_BACKWARD", "1") == "1")

    def _backward_and_step(self, loss, i, manual=False):
        opt = self.optims[i]
        if manual:   # TR:199-216: only optimizer i's parameters accumulate during its backward
            self._toggle(i)
        # zero_grad may only run once the deferred step has consumed the previous gradients
        self._zero_grad(opt)
        if torch.is_tensor(loss) and loss.requires_grad:
            loss.backward()
            self._reduce_and_step(opt)
        if manual:
            self._untoggle()
        else:
            self._lr_step(i, "step")

    def _run_deferred(self):
        if self._deferred is not None:
            (loss, i, manual), self._deferred = self._deferred, None
            self._backward_and_step(loss, i, manual)

    def _before_student(self):
        self._run_deferred()
        self._wait_pending()

    def _reduce_and_step(self, opt):
        """All-reduce (sum) the optimizer's gradients over the data-parallel ranks, then step.  Runs on
        the comm stream when overlap is on; the 1/world mean is folded into the fused AdamW."""
        side = self._comm_stream if (self.overlap and self._comm_stream is not None) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(side) if side is not None else _null()
        with ctx:
            if self.distributed:
                ev = None
                if self.comm_timing and torch.cuda.is_available():
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(torch.cuda.current_stream())
                if isinstance(opt, FusedAdamW) and opt.flat_grad is not None:
                    torch.distributed.all_reduce(opt.flat_grad)
                    self.comm_payload_bytes = opt.flat_grad.numel() * opt.flat_grad.element_size()
                else:
                    grads = [p.grad for g in opt.param_groups for p in g["params"] if p.grad is not None]
                    if grads:
                        flat = torch._utils._flatten_dense_tensors(grads)
                        torch.distributed.all_reduce(flat)
                        for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
                            g.copy_(f)
                        self.comm_payload_bytes = flat.numel() * flat.element_size()
                if ev is not None:   # (the synchronous all_reduce made the current stream wait for the collective's stream)
                    ev[1].record(torch.cuda.current_stream())
                    self._comm_events.append(ev)
                if isinstance(opt, FusedAdamW):
                    opt.grad_scale = 1.0 / self.world
                else:
                    for g in opt.param_groups:
                        for p in g["params"]:
                            if p.grad is not None:
                                p.grad.div_(self.world)
            if self.reduced_grad_hook is not None:
                # test hook: called after the exchange, before the step, with the factor the step applies to the gradient
                # (FusedAdamW folds the 1 / world mean into its update; the torch optimizers' grads are already divided)
                self.reduced_grad_hook(opt, opt.grad_scale if isinstance(opt, FusedAdamW) else 1.0)
            opt.step()
            if side is not None:
                self._pending = torch.cuda.Event()
                self._pending.record(side)

    def _zero_grad(self, opt):
        if isinstance(opt, FusedAdamW) and opt.flat_grad is not None:
            self._wait_pending()  # the deferred step of the previous iteration still reads the grads
            opt.flat_grad.zero_()
        else:
            self._wait_pending()
            opt.zero_grad()

    # ---- TR:169-218 ------------------------------------------------------------------------------------
    def training_step(self, train_batch: Dict[str, Any], batch_idx: int = 0) -> dict:
        if not self.optims:
            self.configure_optimizers()
        if self.share_start_idx:
            self.model.shared_start_step = self.global_step   # (a rank-local forward outside training_step does not advance it)
        self.global_step += 1
        if not getattr(self.model, "calls_before_student", False):
            # a model whose forward never calls the before_student hook would read parameters the deferred step is
            # still writing: wait here instead (FlashDiffusion / FlashDiffusionSD3 call the hook after their teacher loop)
            self._before_student()
        if self.automatic_optimization:
            out = self.model(train_batch, device=self.device)   # its before_student hook issued the previous backward
            self._run_deferred()                                # (a forward that never reached the hook)
            loss = out["loss"][0] if isinstance(out["loss"], (list, tuple)) else out["loss"]
            if self._defer_ok():
                self._deferred = (loss, 0, False)
            else:
                self._backward_and_step(loss, 0)
            return {"loss": loss.detach(), "batch_idx": batch_idx, "start_timestep": out.get("start_timestep")}
        outputs = {"batch_idx": batch_idx}
        for i, opt in enumerate(self.optims):
            # (the forward's before_student hook issues the previous forward's backward: optimizer i-1's, or the last
            # optimizer's of the previous batch -- before anything of this forward reads a trainable parameter)
            model_output = self.model(train_batch, device=self.device, step=i, batch_idx=batch_idx)
            self._run_deferred()                                # (a forward that never reached the hook)
            loss = model_output["loss"]
            if "start_timestep" in model_output:
                outputs["start_timestep"] = model_output["start_timestep"]
            outputs[f"loss_optimizer_{i}"] = loss[i].detach() if torch.is_tensor(loss[i]) else loss[i]
            if self._defer_ok() and torch.is_tensor(loss[i]) and loss[i].requires_grad:
                self._deferred = (loss[i], i, True)
            else:
                self._backward_and_step(loss[i], i, manual=True)
        return outputs

    def _toggle(self, i):
        """Lightning's toggle_optimizer: only optimizer i's parameters require grad during its backward."""
        mine = {id(p) for g in self.optims[i].param_groups for p in g["params"]}
        self._toggled = []
        for j, o in enumerate(self.optims):
            if j == i:
                continue
            for g in o.param_groups:
                for p in g["params"]:
                    if id(p) not in mine and p.requires_grad:
                        p.requires_grad = False
                        self._toggled.append(p)

    def _untoggle(self):
        for p in self._toggled:
            p.requires_grad = True
        self._toggled = []

    def finish(self):
        """Drain the deferred backward and optimizer step (end of training / before reading parameters)."""
        self._run_deferred()
        self._wait_pending()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    # every reader of the trainable parameters other than the training step itself first drains the deferred
    # all-reduce + AdamW (it runs on a side stream while the next teacher loop is in flight)
    def state_dict(self, *a, **k):
        self.finish()
        return super().state_dict(*a, **k)

    def log_samples(self, batch: Dict[str, Any]):
        """TR:227-251"""
        self.finish()
        logs = self.model.log_samples(batch, device=self.device, **self.log_samples_model_kwargs)
        N = min(logs[k].shape[0] for k in logs) if logs else 0
        keys = self.pipeline_config.log_keys
        if keys is not None:
            for key in ([keys] if isinstance(keys, str) else keys):
                if key in batch:
                    logs = logs if logs is not None else {}
                    logs[key] = batch[key][:N] if N > 0 else batch[key]
        return logs

    def sample(self, *a, **k):
        self.finish()
        return self.model.sample(*a, **k)

    # ---- TR:58-74 --------------------------------------------------------------------------------------
    def on_train_start(self):
        self.timer = time.perf_counter()

    def on_train_batch_end(self, outputs, batch, batch_idx):
        self.model.on_train_batch_end(batch)
        if self.global_rank == 0 and batch_idx % 10 == 0 and self.timer is not None:
            delta = time.perf_counter() - self.timer
            logging.info(f"Average time per batch {batch_idx} took {delta / (batch_idx + 1)} seconds")


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
