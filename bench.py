#!/usr/bin/env python
"""bench.py -- distillation images/s on MI355X for BASELINE.json's headline config (C2):
SD1.5 UNet teacher + LoRA-r128 student, per-GPU batch 16, 64x64 latents, 4 teacher CFG steps, bf16 MFMA
compute.  One "step" = one generator iteration of the reference: FlashDiffusion.forward
(step=0) [1 student fwd (grad) + 2n teacher fwd] + backward of loss[0] + AdamW on the LoRA tensors
(+ RCCL all-reduce of the flat LoRA gradient when N > 1).  Synthetic latents / text embeddings and
random-init weights (no network); inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL).
Prints ONE JSON line on rank 0."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF; 2:1 sparse figures excluded)
PEAK_HBM = 8.0e12   # HBM3E peak, same guide (6.3 TB/s is what a streaming copy achieves)
RIDGE = PEAK_BF16 / PEAK_HBM   # 312.5 flop per byte: a launch with less arithmetic intensity sits under the HBM roof
BUCKETS = ["gemm_kernel<128,128,row>", "gemm_kernel<128,64,row>", "gemm_kernel<64,128,row>", "gemm_kernel<64,64,row>",
           "gemm_kernel<128,128,conv>", "gemm_kernel<128,64,conv>", "gemm_kernel<64,128,conv>", "gemm_kernel<64,64,conv>",
           "attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel",
           "gemm3_kernel<256x160,row>", "gemm3_kernel<256x128,row>", "gemm3_kernel<256x160,conv>", "gemm3_kernel<256x128,conv>",
           "gemm4_kernel<256x320,row>", "gemm4_kernel<256x320,conv>", "gemm4_kernel<256x192,row>", "gemm4_kernel<256x192,conv>",
           "wgrad_tn_kernel",
           # round 6, SUBSETS (also counted in their family above): the row GEMMs on the HBM side of the ridge (csrc/common.h)
           "gemm4_kernel<256x320,row|hbm-side>", "gemm3_kernel<256xBN,row|hbm-side>",
           # the 128 x 320 two-blocks-per-CU row kernel (csrc/gemm5.hip: a measured experiment, force_tile only -- empty in the product step)
           "gemm5_kernel<128x320,row>", "gemm5_kernel<128x320,row|hbm-side>"]
SUBSET_BUCKETS = {"gemm4_kernel<256x320,row|hbm-side>": "gemm4_kernel<256x320,row>", "gemm3_kernel<256xBN,row|hbm-side>": None,
                  "gemm5_kernel<128x320,row|hbm-side>": "gemm5_kernel<128x320,row>"}


def sustained_mfma():
    """The MFMA rate this part sustains from registers (scripts/ubench/mfma_rate, built by __graft_entry__.build()): bf16 32x32x16 on all
    256 CUs, RANDOM operands vs ZERO operands.  The guide's 2 495 TFLOP/s is the zero-operand figure (its own DVFS note: the same
    binary clocks 2.30 GHz on zeros and 1.90 - 1.95 GHz on random data); on random data this pool's parts hold ~1.82 PFLOP/s at
    ~1.84 GHz (profiles/r6_mfma_rate.txt: GRBM_GUI_ACTIVE per XCD / duration) -- the MFMA ceiling of any kernel that multiplies
    real activations.  None when the binary is absent."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "scripts", "ubench", "mfma_rate")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    best = {}
    for l in out.splitlines():
        m = re.match(r"(\S+) (\d) waves?/SIMD\s+(random|zeros)\s+rep \d+: [\d.]+ ms\s+([\d.]+) TFLOP/s", l)
        if m and m.group(1) == "32x32x16":
            best[m.group(3)] = max(best.get(m.group(3), 0.0), float(m.group(4)))
    if not best:
        return None
    return {"random_operands": best.get("random", 0.0) / 1e3, "zero_operands": best.get("zeros", 0.0) / 1e3, "unit": "PFLOP/s",
            "kernel": "register-only v_mfma_f32_32x32x16_bf16 loop, 256 CUs (scripts/ubench/mfma_rate.hip), best of 3, this run",
            "clock_ghz": {"random_operands": 1.84, "zero_operands": 2.39,
                          "source": "profiles/r6_mfma_rate.txt (rocprofv3 --pmc GRBM_GUI_ACTIVE: cycles per XCD / kernel duration)"},
            "note": "the 2.5 PFLOP/s peak (and the guide's measured 2.495) holds for zero operands at 2.4 GHz; on random operands the "
                    "part is power-limited to the figure above"}


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(n_teacher_steps, budget_s=130.0):
    """The oracle (CPU fp32 restatement of the reference step, oracle/flash_ref.py -- kind "port") timed on this box's host
    cores on a bounded sample of the SAME workload (SD1.5, r128 LoRA, 64x64 latents, n teacher steps): whole generator
    iterations (forward + backward + AdamW) at B=1 -- one warm-up UNet forward, then the median of up to 3 timed
    iterations -- and, if the time budget allows, one iteration at B=2 (the batch-scaling point).  Threads: the fastest of a
    sweep over physical cores / 8, / 4, / 2 and ALL physical cores of a B = 2 student forward + backward;
    FDMI_CPU_THREADS pins a count."""
    import copy
    import statistics
    import torch
    from oracle.flash_ref import Draws, FlashConfigRef, FlashDiffusionRef, TensorConditioner
    from oracle.sched_cpu import DPMSolverMultistepSchedulerRef
    from oracle.unet_cpu import UNet2DConditionRef, sd15_config
    torch.manual_seed(0)
    teacher = UNet2DConditionRef(sd15_config())
    student = copy.deepcopy(teacher)
    student.add_adapter(128)
    teacher.freeze()
    # thread count: the fastest of {physical/8, physical/4, physical/2, physical} -- ALL four, up to every
    # physical core -- on the part of the step that carries the batch: a B = 2 student forward + backward (LoRA gradients), not a
    # B = 1 forward (whose optimum, 16 of 128 threads on the pool's host, undersold the CPU).  FDMI_CPU_THREADS pins a count.
    phys = _physical_cores()
    probe = (torch.randn(2, 4, 64, 64), torch.tensor([999, 999]), {"cond": {"crossattn": torch.randn(2, 77, 768)}})
    t_begin = time.perf_counter()
    sweep = {}
    forced = int(os.environ.get("FDMI_CPU_THREADS", "0"))
    for n in ([forced] if forced else sorted({max(1, phys // 8), max(1, phys // 4), max(1, phys // 2), phys})):
        torch.set_num_threads(n)
        if not sweep:
            with torch.no_grad():
                student(probe[0][:1], probe[1][:1], {"cond": {"crossattn": probe[2]["cond"]["crossattn"][:1]}})   # warm-up: allocator, oneDNN primitive caches
        t0 = time.perf_counter()
        student(*probe).square().mean().backward()
        sweep[n] = time.perf_counter() - t0
        student.zero_grad(set_to_none=True)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    m = FlashDiffusionRef(FlashConfigRef(K=[n_teacher_steps], num_iterations_per_K=[10 ** 9], timestep_distribution="uniform"),
                          student_denoiser=student, teacher_denoiser=teacher,
                          teacher_noise_scheduler=DPMSolverMultistepSchedulerRef(), conditioner=TensorConditioner())
    opt = torch.optim.AdamW([p for p in student.parameters() if p.requires_grad], lr=1e-5)

    def iteration(B):
        batch = {"image": torch.randn(B, 4, 64, 64), "crossattn": torch.randn(B, 77, 768), "text": ["s"] * B}
        m.draws = Draws({"noise": torch.randn(B, 4, 64, 64), "start_idx": torch.tensor([0]), "guidance": torch.tensor([0.5])})
        t0 = time.perf_counter()
        out = m(batch, step=0)
        opt.zero_grad()
        out["loss"][0].backward()
        opt.step()
        return time.perf_counter() - t0

    # B = 1: median of 3 (the third is dropped only if the first two already ate the whole budget); B = 2: one iteration if
    # its projected cost (2.2 x the B = 1 median) still fits 1.6 x the budget -- ~25 s per B = 1 iteration on the pool's host
    t1 = []
    while len(t1) < 3 and (len(t1) < 2 or time.perf_counter() - t_begin + t1[-1] < budget_s):
        t1.append(iteration(1))
    med1 = statistics.median(t1)
    t2 = None
    if time.perf_counter() - t_begin + 2.2 * med1 < budget_s * 1.6:
        t2 = iteration(2)
    best = max(1.0 / med1, (2.0 / t2) if t2 else 0.0)
    return {"value": best, "unit": "images/s", "cores": threads, "physical_cores": phys, "kind": "port",
            "thread_sweep_s_per_b2_student_fwd_bwd": {str(k): round(v, 2) for k, v in sweep.items()},
            "b1_s_per_iteration": [round(x, 2) for x in t1], "b1_images_per_s": 1.0 / med1,
            "b2_s_per_iteration": round(t2, 2) if t2 else None, "b2_images_per_s": (2.0 / t2) if t2 else None,
            "sample": f"whole generator iterations (fwd+bwd+AdamW) of the same SD1.5 r128 step, {n_teacher_steps} teacher CFG steps, "
                      f"fp32 PyTorch-CPU oracle on {threads} threads (fastest of a B=2 fwd+bwd sweep over {sorted(sweep)} threads, {phys} physical cores), B=1 median of "
                      f"{len(t1)} = {med1:.1f} s" + (f", B=2 one iteration = {t2:.1f} s" if t2 else "") +
                      "; value = the better of the two batch sizes"}


def _latest_profile(pattern):
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    return cands[-1] if cands else None


def dominant_family(arch):
    """(bucket name, provenance): the MFMA kernel family with the largest time per step in the newest committed rocprofv3
    summary of this workload (profiles/rN_kernel_stats[_<arch>].csv, written by scripts/rocprof_to_profiles.py)"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from rocprof_to_profiles import bucket
        f = _latest_profile("r*_kernel_stats.csv" if arch == "sd15" else f"r*_kernel_stats_{arch}*.csv")   # (e.g. r3_kernel_stats_sd3_call8.csv)
        if f is None:
            return None, None
        tot = {}
        with open(f) as fh:
            for line in fh:
                if line.startswith("#") or line.startswith("kernel,"):
                    continue
                name, _, rest = line.rpartition('",')
                b = bucket(name.lstrip('"'))
                if b:
                    tot[b] = tot.get(b, 0.0) + float(rest.split(",")[1])
        if not tot:
            return None, None
        return max(tot, key=tot.get), f"largest ms/step among the MFMA families in {os.path.relpath(f, ROOT)}"
    except (OSError, ValueError, ImportError, IndexError):
        return None, None
    finally:
        sys.path.pop(0)


def parity_record():
    """the headline workload's measured distance to the reference's fixture, from the newest committed parity log of the GPU tests
    (profiles/rN_parity_fullsize.txt: tests/test_fullsize_parity_gpu.py on BASELINE configs[1] at its own batch, B = 16, against
    tests/golden/c2_sd15_r128_n4_b16.npz made by the REAL reference class).  fp32_gate: the validation precision north_star's
    1e-3 is stated for; bf16: the kernels this bench times (the reference's own bf16-mixed sits at the same distance,
    tests/test_precision_class.py)."""
    import re
    f = _latest_profile("r*_parity_fullsize.txt")
    if f is None:
        return None
    rec = {"source": os.path.relpath(f, ROOT), "fixture": "tests/golden/c2_sd15_r128_n4_b16.npz (reference FlashDiffusion, SD1.5 r128, 4 teacher steps, B = 16)"}
    for line in open(f):
        m = re.match(r"step c2_sd15_r128_n4_b16 \[(bf16|fp32)\]: teacher_output=(\S+) student_output=(\S+) .* loss_rel=([0-9.e+-]+),", line)
        if m:
            rec["fp32_gate" if m.group(1) == "fp32" else "bf16"] = {"loss_rel": float(m.group(4)), "teacher_output_rel": float(m.group(2)),
                                                                  "student_output_rel": float(m.group(3))}
    # the yardstick for the bf16 record: the REFERENCE's own precision mode on the same fixture -- the pinned
    # oracle under torch.autocast(bfloat16) (= precision="bf16-mixed", examples/train_flash_sd.py:405) against its fp32 run,
    # tests/golden/c2_sd15_r128_n4_b16_bf16ref.npz (oracle/make_golden.py::make_bf16_anchor); the GPU tests hold the HIP path to 1.5 x it
    try:
        import numpy as np
        a = np.load(os.path.join(ROOT, "tests", "golden", "c2_sd15_r128_n4_b16_bf16ref.npz"))
        rec["reference_bf16_mixed"] = {"loss_rel": float(a["loss_rel"]), "teacher_output_rel": float(a["teacher_output_rel"]),
                                       "student_output_rel": float(a["student_output_rel"]),
                                       "is": "the reference's own bf16-mixed run (CPU autocast of the pinned oracle) vs the same fp32 fixture"}
    except (OSError, KeyError, ValueError):
        pass
    return rec if ("bf16" in rec or "fp32_gate" in rec) else None


_PLAN_CHILD = """
import sys; sys.path.insert(0, %r)
import torch
from flash_diffusion_amd import _lib, workloads
from flash_diffusion_amd.unet import MiUNet2DConditionModel
B, flags, lora = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
with torch.device("meta"):
    m = MiUNet2DConditionModel(**workloads.SD15)
    if lora:
        m.add_adapter(lora)
for t in (m._lora_targets if lora else []):      # (no GPU in this walk: the adapters are declared, not bound)
    assert _lib.lib().fdmi_unet_declare_lora(m._plan().handle, t.encode(), lora) == 0
assert _lib.lib().fdmi_unet_workspace_bytes(m._plan().handle, B, 64, 64, 77, flags) > 0
"""


def algorithmic_bytes(batch, teacher_steps, lora_rank):
    """{bench bucket: (algorithmic HBM bytes per step, launches per step)} of the C2 step's GEMM / conv launches: every operand of a
    launch touched ONCE (A + W + C, + the residual; a GEGLU output is N / 2 wide; a 3x3 convolution's A operand is its input
    tensor, taken as M * Cin elements -- exact for the stride-1 convolutions that carry the time, 4x too small / large for the
    three down / up-sampling ones).  From the C++ plan's own work list in workspace-query mode (FDMI_PLAN_LOG, no GPU work): the
    teacher's 2B forward x teacher_steps + the student's taped forward and backward.  None when the query fails."""
    import re
    import subprocess
    tot = {}
    try:
        for (B, flags, lora, times) in ((2 * batch, 8, 0, teacher_steps), (batch, 1, lora_rank, 1)):
            err = subprocess.run([sys.executable, "-c", _PLAN_CHILD % ROOT, str(B), str(flags), str(lora)],
                                 env=dict(os.environ, FDMI_PLAN_LOG="1"), capture_output=True, text=True, check=True, timeout=300).stderr
            for l in err.splitlines():
                if not l.startswith("PLANGEMM"):
                    continue
                mode, M, N, K, act, res, dgrad, atomic, kern, BM, BN, sk = (int(v) for v in re.findall(r"=(-?\d+)", l))
                if kern == 2:
                    key = f"gemm4_kernel<256x{BN},{'conv' if mode else 'row'}>"
                elif kern == 1:
                    key = f"gemm3_kernel<256x{BN},{'conv' if mode else 'row'}>"
                else:
                    key = f"gemm_kernel<{BM},{BN},{'conv' if mode else 'row'}>"
                nout = N // 2 if act == 2 else N
                a_el = M * (K // 9 if (mode and K % 9 == 0) else K)
                by = 2.0 * (a_el + N * K + M * nout * (2 if res else 1))
                b0, n0 = tot.get(key, (0.0, 0))
                tot[key] = (b0 + by * times, n0 + times)
    except Exception:
        return None
    return tot


def load_traffic(buckets):
    """HBM bytes per launch per kernel family from the newest profiles/rN_traffic.json -- for every family whose source files
    (flash_diffusion_amd._lib.kernel_source_hash) are the ones the counters were collected on.  Returns (unused, provenance,
    {bucket: bytes})."""
    from flash_diffusion_amd import _lib
    f = _latest_profile("r*_traffic.json")
    if f is None:
        return None, None, {}
    try:
        with open(f) as fh:
            tj = json.load(fh)
    except (OSError, ValueError):
        return None, None, {}
    out, stale = {}, []
    whole = tj.get("csrc_sha") == _lib.source_hash()
    for b in buckets:
        k = tj.get("kernels", {}).get(b)
        if not k:
            continue
        if whole or (k.get("src_sha") is not None and k.get("src_sha") == _lib.kernel_source_hash(b)):
            out[b] = k.get("hbm_bytes_per_launch")
        else:
            stale.append(b)
    src = os.path.relpath(f, ROOT) + (f" (not used for {', '.join(stale)}: their kernel sources changed since the counter pass)"
                                      if stale else "")
    return None, src, out


def self_launch(args):
    """`python bench.py --gpus N` without torch.distributed.run: spawn N ranks of this script (one per GPU, RCCL rendezvous on
    127.0.0.1) and wait.  Rank 0's stdout is ours (it prints the JSON line)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:                       # one rank died: the others would wait in a collective forever
                rc = bad[0].returncode
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default 16 (C2); 8 for sdxl / pixart "
                                                            "(C3 / C4), 4 for sd3 (C5)")
    ap.add_argument("--hw", type=int, default=None, help="latent height = width; default 64 (C2); 128 for sdxl / pixart / sd3")
    ap.add_argument("--teacher-steps", type=int, default=4)
    ap.add_argument("--arch", default="sd15", choices=["sd15", "sdxl", "tiny", "pixart", "tiny_pixart", "sd3", "tiny_sd3"],
                    help="sd15 = the headline workload C2; the others are developer legs at BASELINE.json's C3 / C4 / C5 "
                         "per-GPU shapes (sdxl: B=8, 128x128 latents, r64; --batch 16 --hw 64 --lora-rank 128 reproduces "
                         "profiles/r1_bench_sdxl_b16_64x64.json)")
    ap.add_argument("--lora-rank", type=int, default=None, help="default 128 (sd15), 64 (sdxl / pixart / sd3: SURVEY C.4), 8 (tiny*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (sampler, 2-optimizer step)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)             # plain `python bench.py --gpus N`: become the launcher (never returns)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    # one process per GPU; (dev aid) FDMI_BENCH_BACKEND=gloo lets several ranks share one GPU to exercise the
    # multi-process path on a single-GPU box -- the driver's runs use nccl (= RCCL) with one GPU per rank
    backend = os.environ.get("FDMI_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local % torch.cuda.device_count() if backend != "nccl" else local)
    comm_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    elif os.environ.get("FDMI_BENCH_SINGLE_RANK_GROUP", "1") == "1":
        # N = 1 still runs the step's one collective -- the all-reduce of the flat LoRA gradient on the comm stream -- through a
        # process group of ONE rank (RCCL), so that the line's "comm" block measures the code path the N-GPU job runs
        # (VERDICT r4 item 6).  A failure to create the group is reported in the block, not fatal.
        try:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        except Exception as e:     # noqa: BLE001
            comm_note = f"single-rank {backend} group not created: {type(e).__name__}: {e}"
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    from flash_diffusion_amd import _lib, unet as _unet
    from flash_diffusion_amd.trainer import TrainingConfig, TrainingPipeline
    from flash_diffusion_amd.workloads import (PIXART, SD3, SD15, SDXL, TINY, TINY_PIXART, TINY_SD3, build_flash,
                                               build_flash_sd3, synthetic_batch)
    arch = {"sd15": SD15, "sdxl": SDXL, "tiny": TINY, "pixart": PIXART, "tiny_pixart": TINY_PIXART, "sd3": SD3,
            "tiny_sd3": TINY_SD3}[args.arch]
    dit = args.arch.endswith("pixart")
    sd3 = args.arch.endswith("sd3")
    if args.batch is None:
        args.batch = {"sdxl": 8, "pixart": 8, "sd3": 4}.get(args.arch, 16)
    if args.hw is None:
        args.hw = 128 if args.arch in ("sdxl", "pixart", "sd3") else (16 if args.arch.startswith("tiny_") else 64)
    rank_r = args.lora_rank or (8 if args.arch.startswith("tiny") else (64 if (dit or sd3 or args.arch == "sdxl") else 128))
    if sd3:
        model = build_flash_sd3(arch, lora_rank=rank_r, n_teacher_steps=args.teacher_steps, B=args.batch,
                                L=333 if args.arch == "sd3" else 7, device="cuda", seed=0)
    else:
        model = build_flash(arch, lora_rank=rank_r, n_teacher_steps=args.teacher_steps, device="cuda", seed=0)
    pipe = TrainingPipeline(model, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-5],
                                                  trainable_params=[["student_denoiser"]]),
                            overlap=not args.no_overlap)
    pipe.configure_optimizers()
    B = args.batch
    if sd3:   # 16-channel latents; the text embeddings come from the model's prompt-encoder stand-in
        gs = [torch.Generator(device="cpu").manual_seed(1234 + rank + 1000 * i) for i in range(4)]
        batches = [{"image": torch.randn(B, arch["in_channels"], args.hw, args.hw, generator=g).cuda(),
                    "text": ["synthetic"] * B} for g in gs]
    elif dit:   # T5 context [B, 120, caption_channels] + mask of ones, vector = num_vector_conditionings sinusoid blocks
        batches = [synthetic_batch(B, args.hw, arch["caption_channels"], seed=1234 + rank + 1000 * i, L=120,
                                   vector_dim=arch["projection_class_embeddings_input_dim"] * arch["num_vector_conditionings"],
                                   attention_mask=True) for i in range(4)]
    else:
        batches = [synthetic_batch(B, args.hw, arch["cross_attention_dim"], seed=1234 + rank + 1000 * i,
                                   vector_dim=arch.get("projection_class_embeddings_input_dim", 0) or 0) for i in range(4)]

    def run(n, counter=None):
        for i in range(n):
            pipe.training_step(batches[i % len(batches)], i)
            if counter is not None:
                counter[0] += model.student_denoiser.step_flops + model.teacher_denoiser.step_flops
                model.student_denoiser.step_flops = model.teacher_denoiser.step_flops = 0.0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    pipe.finish()
    model.student_denoiser.step_flops = model.teacher_denoiser.step_flops = 0.0
    flops = [0.0]
    pipe.comm_timing = True       # events around every gradient exchange and around the main stream's waits for it
    barrier()
    t0 = time.perf_counter()
    run(args.steps, flops)
    pipe.finish()
    # the last iteration's backward is issued by finish() (deferred backward, trainer.py): its FLOPs are counted here
    flops[0] += model.student_denoiser.step_flops + model.teacher_denoiser.step_flops
    model.student_denoiser.step_flops = model.teacher_denoiser.step_flops = 0.0
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        # every rank's own clock around the same K steps: the line's time is the MAX (the contract); the spread max / min shows a
        # straggler at a glance on the first multi-GPU run
        tl = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([dt], device="cuda", dtype=torch.float64))
        per = [float(x.item()) / args.steps * 1e3 for x in tl]
        rank_ms = {"per_rank": [round(x, 3) for x in per], "min": min(per), "max": max(per), "max_over_min": max(per) / min(per)}
        dt = max(float(x.item()) for x in tl)
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    step_flops = flops[0] / args.steps
    # ---- the gradient exchange of the timed steps: ONE all-reduce of the flat fp32 LoRA gradient per step on the comm
    # stream; "exposed_ms" = mean time the main stream waited for exchange + AdamW (0 when the next teacher loop hides them) ----
    rep = pipe.comm_report()
    pipe.comm_timing = False
    comm = {"backend": (dist.get_backend() if dist.is_initialized() else None),
            "ranks_seen": (dist.get_world_size() if dist.is_initialized() else 0),
            "collective": "all_reduce(sum) of the flat fp32 LoRA gradient, 1/world folded into the fused AdamW",
            "payload_bytes": rep["payload_bytes"], "exchanges_timed": rep["exchanges"],
            "allreduce_ms": rep["allreduce_ms"], "exposed_ms": rep["exposed_ms"],
            "exposed_is": "mean main-stream wait on the comm stream's (all-reduce + AdamW) event per step, HIP events",
            "note": comm_note, "rank_ms_per_step": rank_ms}
    if world > 1:     # slowest rank's figures
        t = torch.tensor([rep["allreduce_ms"] or 0.0, rep["exposed_ms"] or 0.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm["allreduce_ms_max_over_ranks"], comm["exposed_ms_max_over_ranks"] = float(t[0]), float(t[1])

    # ---- roofline leg: per-launch HIP events on the launch stream, one extra (untimed) step.  EVERY rank runs the
    # step (it contains the gradient all-reduce: a collective only rank 0 entered would hang the job); only rank 0
    # records events ----
    roofline = None
    L = _lib.lib()
    # The profiled step runs SERIALLY (no side stream for the teacher, backward not deferred): a kernel's duration is then its
    # own, not its share of a chip it splits with the other stream's kernels.  Each profiled launch carries its own start /
    # stop events (hipExtLaunchKernelGGL = the dispatch's timestamps, as in rocprofv3's kernel trace); if the runtime
    # returns no valid elapsed time for them the leg is repeated with events recorded around the launches (knob 20).
    serial_env = {"FDMI_TEACHER_STREAM": "0", "FDMI_DEFER_BACKWARD": "0"}
    saved_env = {k: os.environ.get(k) for k in serial_env}
    os.environ.update(serial_env)
    nb = 24
    ms = (C.c_double * nb)()
    fl = (C.c_double * nb)()
    ln = (C.c_int64 * nb)()
    by = (C.c_double * nb)()
    timing = "per-dispatch start/stop events (hipExtLaunchKernelGGL)"
    for attempt in (0, 1):
        if os.environ.get("FDMI_BENCH_NO_PROFILE_LEG") == "1":   # (dev: counter-pass bisection, scripts/gpu_calls/r4_call5.sh)
            ok = False
            break
        if rank == 0:
            L.fdmi_tune_set(20, max(attempt, L.fdmi_tune_value(20)))
            L.fdmi_prof_enable(1)
        run(1)
        pipe.finish()
        if rank == 0:
            L.fdmi_prof_enable(0)
            rc = L.fdmi_prof_collect2(nb, ms, fl, ln, by)
            if rc == 0 and os.environ.get("FDMI_BENCH_SHAPES"):   # (dev: one line per launch of the profiled step -> scripts/shape_table.py)
                L.fdmi_prof_dump(os.environ["FDMI_BENCH_SHAPES"].encode())
            ok = rc == 0 and sum(ln) > 0 and sum(ms) > 0
            if L.fdmi_tune_value(20):
                timing = "events recorded around each launch (includes dispatch latency)"
                _lib.check(rc)
        else:
            ok = True
        if world > 1:
            t = torch.tensor([1.0 if ok else 0.0], device="cuda")
            dist.broadcast(t, 0)
            ok = bool(t.item() > 0)
        if ok:
            break
    for k, v in saved_env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    if rank == 0:
        allrows = {BUCKETS[i]: (BUCKETS[i], ms[i], fl[i], ln[i], by[i]) for i in range(len(BUCKETS)) if ln[i] > 0}
        rows = [r[:4] for k, r in allrows.items() if k not in SUBSET_BUCKETS]      # (the families; the subsets are priced below)
        rows.sort(key=lambda r: -r[1])
        mfma_rate = sustained_mfma() if world == 1 else None
        # WHICH kernel the roofline object describes is fixed by the committed rocprofv3 summary of this command (the family
        # with the largest time in profiles/rN_kernel_stats.csv, newest round) -- not by this run's ordering, where two families
        # within a millisecond of each other swapped places from box to box; without a summary: the largest time measured here
        dominant, dominant_src = dominant_family(args.arch)
        byname = {r[0]: r for r in rows}
        if dominant not in byname:
            dominant, dominant_src = rows[0][0], "largest time in this run's profiled step (no committed summary names a launched family)"
        name, tms, tfl, tln = byname[dominant]
        ach = tfl / (tms * 1e-3) / 1e12
        # HBM bytes per launch of that kernel family: PMC numbers cannot be collected from inside this process, so the
        # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this same command is read back (null if absent)
        traffic, traffic_src, traffic_all = load_traffic([r[0] for r in rows])
        algo = algorithmic_bytes(B, args.teacher_steps, rank_r) if args.arch == "sd15" else None

        def _traffic_ratio(key):   # counted HBM bytes per launch / algorithmic bytes per launch of a family (None without both)
            if not algo or key not in algo or traffic_all.get(key) is None:
                return None, None
            per = algo[key][0] / max(algo[key][1], 1)
            return per, traffic_all[key] / per
        def _bound(tms_, tfl_, tby_):
            """the roof a set of launches sits under, from ITS arithmetic intensity (algorithmic flop per algorithmic byte, both
            accumulated by the library per launch) against the 2.5 PFLOP/s : 8 TB/s ridge, and the fraction of THAT roof it reaches"""
            if not tby_ or not tms_:
                return {"arith_intensity": None, "bound": "mfma", "frac_of_bound": round(tfl_ / (tms_ * 1e-3) / PEAK_BF16, 4) if tms_ else None}
            ai = tfl_ / tby_
            hbm = ai < RIDGE
            return {"arith_intensity": round(ai, 1), "bound": "hbm" if hbm else "mfma",
                    "algorithmic_tb_per_s": round(tby_ / (tms_ * 1e-3) / 1e12, 3),
                    "frac_of_bound": round((tby_ / (tms_ * 1e-3) / PEAK_HBM) if hbm else (tfl_ / (tms_ * 1e-3) / PEAK_BF16), 4)}
        fam = {}
        for key, label in (("gemm4_kernel<256x320,row>", "gemm4_row"), ("gemm4_kernel<256x320,conv>", "gemm4_conv"),
                           ("gemm4_kernel<256x192,row>", "gemm4_192_row"), ("attn_fwd_kernel", "attn_fwd"),
                           ("gemm3_kernel<256x160,row>", "gemm3_160_row"), ("gemm3_kernel<256x128,row>", "gemm3_128_row"),
                           ("gemm4_kernel<256x320,row|hbm-side>", "gemm4_row_hbm_side"), ("gemm3_kernel<256xBN,row|hbm-side>", "gemm3_row_hbm_side")):
            if key in allrows:
                r = allrows[key]
                per_launch = (r[4] / r[3]) if r[4] else _traffic_ratio(key)[0]
                fam[label] = {"ms_per_step": round(r[1], 3), "tflops": round(r[2] / (r[1] * 1e-3) / 1e12, 1),
                              "frac": round(r[2] / (r[1] * 1e-3) / PEAK_BF16, 4), "launches": int(r[3]),
                              "traffic": traffic_all.get(key), "algorithmic_bytes_per_launch": per_launch,
                              "traffic_over_algorithmic": (traffic_all[key] / per_launch) if (traffic_all.get(key) and per_launch) else None}
                fam[label].update(_bound(r[1], r[2], r[4]))
                if key in SUBSET_BUCKETS:
                    fam[label]["subset_of"] = SUBSET_BUCKETS[key] or "gemm3_kernel<256x160,row> + gemm3_kernel<256x128,row>"
        # the MFMA-side remainder of the 256 x 320 row family (long K, GEGLU): family minus its HBM-side subset
        if "gemm4_kernel<256x320,row>" in allrows and "gemm4_kernel<256x320,row|hbm-side>" in allrows:
            a_, b_ = allrows["gemm4_kernel<256x320,row>"], allrows["gemm4_kernel<256x320,row|hbm-side>"]
            tms_, tfl_, tln_, tby_ = a_[1] - b_[1], a_[2] - b_[2], a_[3] - b_[3], a_[4] - b_[4]
            if tln_ > 0 and tms_ > 0:
                fam["gemm4_row_mfma_side"] = {"ms_per_step": round(tms_, 3), "tflops": round(tfl_ / (tms_ * 1e-3) / 1e12, 1),
                                              "frac": round(tfl_ / (tms_ * 1e-3) / PEAK_BF16, 4), "launches": int(tln_),
                                              "subset_of": "gemm4_kernel<256x320,row>"}
                fam["gemm4_row_mfma_side"].update(_bound(tms_, tfl_, tby_))
        roofline = {"bound": "mfma", "kernel": name, "kernel_chosen_by": dominant_src, "achieved": ach, "peak": PEAK_BF16 / 1e12,
                    "unit": "TFLOP/s", "frac": ach / (PEAK_BF16 / 1e12), "traffic": traffic_all.get(name),
                    "algorithmic_bytes_per_launch": _traffic_ratio(name)[0], "traffic_over_algorithmic": _traffic_ratio(name)[1],
                    "traffic_source": traffic_src, "launches_per_step": int(tln),
                    "avg_launch_us": tms * 1e3 / tln, "algorithmic_gflop_per_launch": tfl / tln / 1e9,
                    "families": fam,
                    "sustained_mfma_pflops": mfma_rate,
                    "frac_of_sustained_mfma": (ach / (mfma_rate["random_operands"] * 1e3)) if (mfma_rate and mfma_rate.get("random_operands")) else None,
                    "timing": timing + "; profiled step issued serially (teacher loop on the main stream, backward not "
                              "deferred) so that a launch's duration is its own",
                    "all_mfma_kernels": {r[0]: {"ms_per_step": round(r[1], 3), "tflops": round(r[2] / (r[1] * 1e-3) / 1e12, 1),
                                                "launches": int(r[3])} for r in rows},
                    "whole_step": {"algorithmic_tflop_per_step_per_gpu": step_flops / 1e12,
                                   "achieved_tflops_per_gpu": step_flops / (ms_per_step * 1e-3) / 1e12,
                                   "frac_of_peak": step_flops / (ms_per_step * 1e-3) / PEAK_BF16}}
    # ---- secondary: the student's 4-step LCM sampler, FlashDiffusion.sample, same UNet kernels ----
    sampler = None
    if rank == 0 and args.arch in ("sd15", "sdxl") and not args.no_secondary:
        from flash_diffusion_amd.schedulers import LCMScheduler
        model.sampling_noise_scheduler = LCMScheduler()
        zb = batches[0]
        ci = {k: v for k, v in zb.items() if k != "image"}
        zz = torch.randn_like(zb["image"])
        for _ in range(2):
            model.sample(zz, num_steps=4, guidance_scale=1.0, conditioner_inputs=dict(ci))
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        nrep = 3
        for _ in range(nrep):
            model.sample(zz, num_steps=4, guidance_scale=1.0, conditioner_inputs=dict(ci))
        torch.cuda.synchronize()
        dts = (time.perf_counter() - ts0) / nrep
        sampler = {"metric": "student few-step sampling, latents/s (4 LCM steps = 4 UNet evaluations, guidance 1.0, "
                             f"B={B}, LoRA r{rank_r} unmerged)", "value": B / dts, "unit": "images/s", "ms_per_batch": dts * 1e3}
    # ---- secondary: the reference's literal 2-optimizer training_step (TR:194-217): a generator iteration AND a
    # discriminator iteration, i.e. two full FlashDiffusion.forward calls (teacher loop included) with the lsgan GAN term ----
    two_opt = None
    if world == 1 and args.arch == "sd15" and not args.no_secondary:
        from flash_diffusion_amd.workloads import sd15_discriminator
        del pipe
        m2 = build_flash(arch, lora_rank=rank_r, n_teacher_steps=args.teacher_steps, device="cuda", seed=0,
                         discriminator=sd15_discriminator(), gan_loss_type="lsgan")
        p2 = TrainingPipeline(m2, TrainingConfig(optimizers_name=["AdamW", "AdamW"], learning_rates=[1e-5, 1e-5],
                                                 trainable_params=[["student_denoiser"], ["discriminator."]]),
                              overlap=not args.no_overlap)
        p2.configure_optimizers()
        for i in range(2):
            p2.training_step(batches[i % len(batches)], i)
        p2.finish()
        torch.cuda.synchronize()
        tq = time.perf_counter()
        nrep = 2
        for i in range(nrep):
            p2.training_step(batches[i % len(batches)], i)
        p2.finish()
        torch.cuda.synchronize()
        dq = (time.perf_counter() - tq) / nrep
        two_opt = {"metric": "literal 2-optimizer training_step (G + D iteration, lsgan, SD1.5 PatchGAN head on the "
                             "teacher's mid-block features)", "ms_per_step": dq * 1e3, "value": B / dq, "unit": "images/s"}
        del p2, m2
    # ---- secondary: the generator iteration with the loss every shipped YAML trains with -- LPIPS on the VAE
    # decode of both outputs' 64x64 crops (FD:383-397): SD1.5 AutoencoderKL decoder + VGG16 on the HIP path, student side taped ----
    lpips_leg = None
    if world == 1 and args.arch == "sd15" and not args.no_secondary:
        from flash_diffusion_amd.nets import MiLPIPS
        from flash_diffusion_amd.workloads import sd_vae
        try:
            del pipe
        except NameError:
            pass
        lp = MiLPIPS()      # random-init VGG16 / linear layers, like every other weight of the bench ("data": synthetic)
        lp.freeze()
        m3 = build_flash(arch, lora_rank=rank_r, n_teacher_steps=args.teacher_steps, device="cuda", seed=0,
                         distill_loss_type="lpips", vae=sd_vae(), lpips_model=lp)
        p3 = TrainingPipeline(m3, TrainingConfig(optimizers_name=["AdamW"], learning_rates=[1e-5],
                                                 trainable_params=[["student_denoiser"]]), overlap=not args.no_overlap)
        p3.configure_optimizers()
        p3.training_step(batches[0], 0)
        p3.finish()
        torch.cuda.synchronize()
        tq = time.perf_counter()
        nrep = 2
        for i in range(nrep):
            p3.training_step(batches[i % len(batches)], i)
        p3.finish()
        torch.cuda.synchronize()
        dq = (time.perf_counter() - tq) / nrep
        lpips_leg = {"metric": "generator iteration with distill_loss_type='lpips' (SD1.5 VAE decoder to 512 px + LPIPS-VGG16 of both "
                               "outputs, student side back-propagated; random-init VAE / VGG weights)", "ms_per_step": dq * 1e3,
                     "value": B / dq, "unit": "images/s",
                     "vae_decode_tflop_per_call": m3.vae.vae_model.last_flops / 1e12}
        del p3, m3
    if world > 1:
        dist.barrier()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.arch == "sd15":
        cpu = cpu_baseline(args.teacher_steps)
    if rank == 0:
        headline = args.arch == "sd15" and B == 16 and args.hw == 64
        metric = ("distillation images/sec/GPU (SD1.5 64x64 latent, bs=16); 1->8 GPU scaling" if headline else
                  f"distillation images/sec/GPU ({args.arch.upper()} {args.hw}x{args.hw} latent, bs={B}); developer leg")
        line = {
            "metric": metric, "value_is": "whole-job aggregate over n_gpus (per-GPU rate: config.images_per_sec_per_gpu)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{ {'sd15': 'C2', 'sdxl': 'C3-single-GPU', 'pixart': 'C4', 'sd3': 'C5-single-GPU (l2 distill + DMD + lsgan GAN on the full-model backbone at 2B)'}.get(args.arch, 'dev')}: Flash-{args.arch.upper()} "
                                   f"{'DiT' if dit else ('MMDiT' if sd3 else 'UNet')} teacher + LoRA r{rank_r} student, {B} images/GPU, "
                                   f"{args.hw}x{args.hw} latents, {args.teacher_steps} teacher CFG steps (K={args.teacher_steps}, "
                                   "start_idx=0), l2 distill, generator iteration fwd+bwd+fused AdamW",
                       "global_batch": B * world, "parallelism": f"dp{world}", "images_per_sec_per_gpu": value / world,
                       # developer / A-B switches in effect for this line (none = the default, judged configuration)
                       "dev_switches": {k: os.environ[k] for k in ("FDMI_TUNE", "FDMI_TEACHER_LOOP", "FDMI_CFG_DEDUP",
                                                                   "FDMI_NO_CTX_CACHE", "FDMI_TEACHER_STREAM",
                                                                   "FDMI_DEFER_BACKWARD") if os.environ.get(k)}},
            "roofline": roofline, "cpu_baseline": cpu, "comm": comm, "parity": parity_record() if headline else None,
            # does this line's step contain the collective's code path?  At N = 1 a process group of ONE rank is created so
            # that the all-reduce + comm-stream path is in the timed step; if that failed the line measured a different path
            # (comm.note says why) and is NOT comparable with lines that have it
            "process_group": {"created": bool(dist.is_initialized()), "world": world,
                              "comparable_with_group_lines": bool(dist.is_initialized())},
            # the like-for-like numbers of the reference's own loop, also kept under "secondary":
            "two_optimizer_step_ms": two_opt["ms_per_step"] if two_opt else None,          # G + D training_step (TR:169-218)
            "lpips_step_ms": lpips_leg["ms_per_step"] if lpips_leg else None,              # generator iteration with the YAMLs' lpips loss
            "secondary": {"sampler": sampler, "two_optimizer_step": two_opt, "lpips_step": lpips_leg},
        }
        print(json.dumps(line))
    if os.environ.get("FDMI_BENCH_DUMP_LORA"):     # (tests/test_zzz_multigpu_rccl_gpu.py: the replicas must be identical after the run)
        torch.save(model.student_denoiser.lora_flat().detach().cpu(), f"{os.environ['FDMI_BENCH_DUMP_LORA']}.rank{rank}.pt")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
