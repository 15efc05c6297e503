"""Where does the host stand relative to the GPU inside one C2 training step?  (dev tool, round 6)

Wraps the phases of the step (teacher loop issue, deferred backward issue, wait for AdamW, student forward, losses) with probes that
take the HOST clock and record a HIP event on the stream the phase runs on.  After a device synchronisation every probe has two
times relative to the step's first probe: when the host passed it and when the GPU reached it.  A phase whose GPU-side START is
later than the previous phase's GPU-side END by about (host start - host of previous end) is one the GPU WAITED for the host in:
the > 100 us idle gaps of profiles/r6_trace_gaps.txt, attributed to named host phases.

  python scripts/host_timeline.py [steps] [warmup]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from flash_diffusion_amd.trainer import TrainingConfig, TrainingPipeline
from flash_diffusion_amd.workloads import SD15, build_flash, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = build_flash(SD15, lora_rank=128, n_teacher_steps=4, device="cuda", seed=0)
pipe = TrainingPipeline(model, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-5], trainable_params=[["student_denoiser"]]))
pipe.configure_optimizers()
batches = [synthetic_batch(16, 64, SD15["cross_attention_dim"], seed=1234 + 1000 * i, vector_dim=0) for i in range(4)]

PROBES = []   # (step, label, host seconds, event)
ON = [False]
STEP = [0]


def probe(label):
    if not ON[0]:
        return
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    PROBES.append((STEP[0], label, time.perf_counter(), ev))


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        probe(label + " >")
        r = f(*a, **k)
        probe(label + " <")
        return r
    setattr(obj, name, g)


wrap(model.teacher_denoiser, "teacher_loop", "teacher_loop issue")
wrap(pipe, "_run_deferred", "deferred backward+step issue")
wrap(pipe, "_wait_pending", "wait_pending")
model.before_student = pipe._before_student   # (re-bind: the hook attribute captured the unwrapped method)
wrap(model.student_denoiser, "forward", "student forward issue")
wrap(model, "_distill_loss", "distill loss")
wrap(model, "_get_conditioning", "conditioning")
wrap(model, "_get_timesteps", "get_timesteps")
wrap(model, "forward", "model.forward")

for i in range(warm):
    pipe.training_step(batches[i % 4], i)
pipe.finish()
torch.cuda.synchronize()
ON[0] = True
t00 = time.perf_counter()
for i in range(steps):
    STEP[0] = i
    probe("training_step >")
    pipe.training_step(batches[i % 4], i)
    probe("training_step <")
STEP[0] = steps
probe("finish >")
pipe.finish()
probe("finish <")
torch.cuda.synchronize()
print(f"{steps} steps: {(time.perf_counter() - t00) / steps * 1e3:.2f} ms per step (probes on)")
e0, h0 = PROBES[0][3], PROBES[0][2]
prev_h = prev_g = 0.0
print(f"{'step':>4s} {'probe':42s} {'host ms':>10s} {'gpu ms':>10s} {'gpu-host':>9s} {'d host':>8s} {'d gpu':>8s}")
for (s, label, h, ev) in PROBES:
    hm, gm = (h - h0) * 1e3, e0.elapsed_time(ev)
    print(f"{s:4d} {label:42s} {hm:10.2f} {gm:10.2f} {gm - hm:9.2f} {hm - prev_h:8.2f} {gm - prev_g:8.2f}")
    prev_h, prev_g = hm, gm
