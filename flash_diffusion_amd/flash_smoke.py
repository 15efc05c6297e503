"""__graft_entry__.smoke(): ONE tiny distillation step (student fwd+bwd, teacher CFG loop, DMD + GAN losses)
on cuda:0 through the HIP path, checked against the CPU oracle on identical weights, latents and random
draws.  The oracle is imported here ONLY as the checker (allowed for smoke(), see oracle/__init__.py)."""
import copy

import torch


def _mi_from_oracle(o, lora_rank=0):
    from .unet import MiUNet2DConditionModel
    cfg = o.cfg
    m = MiUNet2DConditionModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                               down_block_types=tuple(cfg.down_block_types), up_block_types=tuple(cfg.up_block_types),
                               block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
                               cross_attention_dim=cfg.cross_attention_dim,
                               transformer_layers_per_block=cfg.tlayers(), attention_head_dim=cfg.heads())
    sd = {k.replace(".base_layer.", "."): v for k, v in o.state_dict().items()}
    base = {k: v for k, v in sd.items() if ".lora_" not in k}
    m.load_state_dict(base, strict=True)
    m = m.cuda()
    if lora_rank:
        m.add_adapter(lora_rank)
        m.load_state_dict(sd, strict=True)
    else:
        m.freeze()
    return m


def run():
    from oracle.flash_ref import FlashConfigRef, FlashDiffusionRef, TensorConditioner as OC
    from oracle.golden_cases import LORA_RANK, build_models, make_batch
    from oracle.sched_cpu import DPMSolverMultistepSchedulerRef
    from .flash import Draws, FlashDiffusion, FlashDiffusionConfig, TensorConditioner
    from .schedulers import DPMSolverMultistepScheduler
    kw = dict(K=[4], num_iterations_per_K=[10], timestep_distribution="uniform", distill_loss_type="l2",
              gan_loss_type="lsgan", use_dmd_loss=True, dmd_loss_scale=0.3, adversarial_loss_scale=0.1)
    teacher_o, student_o, disc_o = build_models()
    ora = FlashDiffusionRef(FlashConfigRef(**kw), student_denoiser=student_o, teacher_denoiser=teacher_o,
                            teacher_noise_scheduler=DPMSolverMultistepSchedulerRef(), conditioner=OC(), discriminator=disc_o)
    batch = make_batch()
    torch.manual_seed(7)
    ref = ora(batch, step=0)
    ref["loss"][0].backward()
    t2, s2, d2 = build_models()
    m = FlashDiffusion(FlashDiffusionConfig(**kw), student_denoiser=_mi_from_oracle(s2, LORA_RANK),
                       teacher_denoiser=_mi_from_oracle(t2), teacher_noise_scheduler=DPMSolverMultistepScheduler(),
                       conditioner=TensorConditioner(), discriminator=copy.deepcopy(d2).cuda()).cuda()
    m.draws = Draws(ora.last_draws.values)
    out = m({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}, step=0)
    out["loss"][0].backward()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.detach().float().cpu() - b.detach().float()).norm() / (b.detach().float().norm() + 1e-30))
    e_t, e_s = rel(out["teacher_output"], ref["teacher_output"]), rel(out["student_output"], ref["student_output"])
    e_l = abs(float(out["loss"][0]) - float(ref["loss"][0])) / abs(float(ref["loss"][0]))
    og = {n.replace(".base_layer.", "."): p.grad for n, p in student_o.named_parameters() if p.grad is not None}
    names = [n for n, _ in m.student_denoiser.named_parameters() if ".lora_" in n]
    mg = dict(m.student_denoiser.named_parameters())
    ga = torch.cat([mg[n].grad.float().cpu().flatten() for n in names])
    gb = torch.cat([og[n].flatten() for n in names])
    cosg = float(ga @ gb / (ga.norm() * gb.norm()))
    assert e_t < 4e-2 and e_s < 4e-2 and e_l < 6e-2 and cosg > 0.98, (e_t, e_s, e_l, cosg)
    print(f"flash smoke ok: teacher {e_t:.2e} student {e_s:.2e} loss rel {e_l:.2e} LoRA-grad cosine {cosg:.4f}")
