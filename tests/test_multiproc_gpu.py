"""The data-parallel path on the single GPU of the test box:
  * TrainingPipeline over a world-1 NCCL (= RCCL) process group: the flat LoRA gradient goes through the real collective on the
    side stream, the deferred fused AdamW follows it, and the result equals the run without a process group;
  * `python bench.py --gpus 2` with no launcher: bench.py spawns its own ranks (rendezvous on 127.0.0.1).  Two ranks cannot
    share one GPU under RCCL, so this one run uses the dev switch FDMI_BENCH_BACKEND=gloo; the driver's multi-GPU runs use nccl.
Each body runs in its own interpreter (a process group is per-process state)."""
import json
import os
import subprocess
import sys

import pytest

from tests.isolate import run_isolated

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_pipeline_over_rccl_world_1():
    run_isolated(__name__, "_body_rccl_world_1", ())


def _body_rccl_world_1():
    import copy
    import torch
    import torch.distributed as dist
    from flash_diffusion_amd.trainer import TrainingConfig, TrainingPipeline
    from flash_diffusion_amd.workloads import TINY, build_flash, synthetic_batch

    def run(with_group):
        if with_group:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 1000))
            dist.init_process_group("nccl", rank=0, world_size=1)
        m = build_flash(TINY, lora_rank=8, n_teacher_steps=2, device="cuda", seed=0)
        with torch.no_grad():   # peft's B = 0 would make the first update invisible in the output
            for p in m.student_denoiser.lora_parameters()[1::2]:
                p.normal_(0, 0.02, generator=torch.Generator(device="cuda").manual_seed(3))
        pipe = TrainingPipeline(m, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-3],
                                                  trainable_params=[["student_denoiser"]]), overlap=True)
        pipe.configure_optimizers()
        assert pipe.distributed == with_group
        for i in range(3):
            pipe.training_step(synthetic_batch(2, 16, 64, seed=10 + i), i)
        pipe.finish()
        flat = m.student_denoiser.lora_flat().detach().clone()
        if with_group:
            dist.destroy_process_group()
        return flat
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), orig(*a, **k))[1]
    a, a2 = run(False), run(False)
    assert calls == []
    b = run(True)
    # the step is not bit-reproducible (float atomics in the GroupNorm sums and the weight gradients) and AdamW turns a sign
    # flip of a near-zero gradient into a +-lr move: the yardstick is the distance between two runs WITHOUT a process group
    noise = float((a - a2).abs().max())
    assert len(calls) == 3 and all(n == a.numel() for n in calls), calls      # ONE collective on the flat gradient per step
    assert torch.isfinite(b).all() and float((a - b).abs().max()) <= 3 * noise + 1e-6, (float((a - b).abs().max()), noise)


def test_bench_launches_its_own_ranks():
    env = dict(os.environ, FDMI_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--arch", "tiny", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 2 * 16
    assert j["value"] > 0 and abs(j["config"]["images_per_sec_per_gpu"] * 2 - j["value"]) < 1e-6 * j["value"]
