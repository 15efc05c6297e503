"""RCCL with MORE THAN ONE rank: these tests exist only on a box with >= 2 GPUs -- on the one-GPU test box the
module collects NOTHING (no skip in the report: RCCL cannot put two ranks on one device, tests/test_multiproc_gpu.py covers that box
with world-1 RCCL and two gloo ranks).  On a multi-GPU node they run the code the driver's `bench.py --gpus N` runs:

  * `python bench.py --gpus 2` (self-launched ranks, backend nccl = RCCL over xGMI): the JSON line's "comm" block must report two
    ranks and a timed exchange, and both ranks must end the run with IDENTICAL LoRA buffers (FDMI_BENCH_DUMP_LORA writes them);
  * the torch-free C-ABI collective: `fdmi_comm_unique_id` on rank 0, the 128 bytes handed to rank 1 through a file,
    `fdmi_allreduce_init` on both, `fdmi_allreduce` of a per-rank pattern == the sum (examples/train_flash_sd.py:382-407 is the
    reference's NCCL data-parallel job)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


# (the test functions are only DEFINED on a box with two GPUs: elsewhere nothing is collected and nothing is reported as skipped)
MULTI = _n_gpus() >= 2
pytestmark = pytest.mark.gpu


def _bench_two_ranks(tmp_path, world=2):
    env = dict(os.environ, FDMI_BENCH_DUMP_LORA=str(tmp_path / "lora"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDMI_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--arch", "tiny", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == world and j["comm"]["backend"] == "nccl" and j["comm"]["ranks_seen"] == world
    assert j["comm"]["exchanges_timed"] >= 2 and j["comm"]["allreduce_ms"] > 0 and j["comm"]["payload_bytes"] > 0
    import torch
    bufs = [torch.load(str(tmp_path / f"lora.rank{r_}.pt")) for r_ in range(world)]
    assert all(torch.isfinite(b).all() for b in bufs)
    for b in bufs[1:]:
        assert torch.equal(bufs[0], b), "data-parallel replicas diverged"


_CABI_RANK = """
import sys, os, time
sys.path.insert(0, {root!r})
import ctypes as C
import torch
rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.cuda.set_device(rank)
from flash_diffusion_amd._lib import lib, check, ptr, stream_ptr
L = lib()
uid = (C.c_char * 128)()
if rank == 0:
    check(L.fdmi_comm_unique_id(uid))
    with open(path + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.replace(path + ".tmp", path)
else:
    for _ in range(600):
        if os.path.exists(path):
            break
        time.sleep(0.1)
    uid = (C.c_char * 128).from_buffer_copy(open(path, "rb").read())
check(L.fdmi_allreduce_init(rank, world, uid))
assert L.fdmi_allreduce_world() == world
n = 1 << 20
x = torch.arange(n, device="cuda", dtype=torch.float32) * (rank + 1)
check(L.fdmi_allreduce(ptr(x), n, 0, stream_ptr()))
torch.cuda.synchronize()
want = torch.arange(n, device="cuda", dtype=torch.float32) * sum(r + 1 for r in range(world))
assert torch.equal(x, want), float((x - want).abs().max())
check(L.fdmi_allreduce_destroy())
print("CABI-RANK-OK")
"""


def _c_abi_two_ranks(tmp_path, world=2):
    path = str(tmp_path / "uid.bin")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-c", _CABI_RANK.format(root=ROOT), str(r), str(world), path], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    try:
        res = [p.communicate(timeout=600) for p in procs]
    finally:
        for p in procs:          # (a rank that died leaves its peer waiting in the communicator's bootstrap: never leak it)
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, res):
        assert p.returncode == 0 and "CABI-RANK-OK" in so, se[-4000:]


if MULTI:
    def test_bench_two_ranks_over_rccl_end_with_identical_lora_buffers(tmp_path):
        _bench_two_ranks(tmp_path)

    def test_c_abi_allreduce_two_ranks_with_unique_id_exchange(tmp_path):
        _c_abi_two_ranks(tmp_path)
