"""Attention-forward counter experiment (developer tool).  `python scripts/attn_pmc.py run` launches the two forward kernels
(32x32x16: default; 16x16x32: developer knob 26) a few times on two shapes -- run it under `rocprofv3 --pmc ... --kernel-trace`
(scripts/attn_pmc.sh).  `python scripts/attn_pmc.py table DIR...` prints mean counter values / durations per kernel."""
import csv, glob, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from flash_diffusion_amd import ops, _lib
    L = _lib.lib()
    BF = torch.bfloat16
    for (B, S, H, d) in [(32, 4096, 8, 40), (8, 4096, 10, 64), (16, 4096, 16, 72)]:
        q, k, v = (torch.randn(B, S, H * d, device="cuda").to(BF) for _ in range(3))
        for knob in (0, 1):
            L.fdmi_tune_set(26, knob)
            for _ in range(4):
                ops.attn_fwd(q, k, v, H, d ** -0.5)
            torch.cuda.synchronize()
        L.fdmi_tune_set(26, 0)


def table(dirs):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    n = row.get("Kernel_Name") or row.get("KernelName")
                    if "attn_fwd" not in n:
                        continue
                    n = re.search(r"attn_fwd\w*<[^>]*>", n).group(0) + " grid=" + str(row.get("Grid_Size", ""))
                    c = row.get("Counter_Name") or row.get("CounterName")
                    a = agg[n][c]
                    a[0] += 1
                    a[1] += float(row.get("Counter_Value") or row.get("CounterValue"))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    n = row.get("Kernel_Name") or row.get("KernelName")
                    if "attn_fwd" not in n:
                        continue
                    n = re.search(r"attn_fwd\w*<[^>]*>", n).group(0) + " grid=" + str(row.get("Grid_Size", ""))
                    a = agg[n]["duration_us[" + os.path.basename(d.rstrip("/")) + "]"]
                    a[0] += 1
                    a[1] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
    for n in sorted(agg):
        print(n)
        for c in sorted(agg[n]):
            k, s = agg[n][c]
            print(f"    {c:34s} n={k:3d} mean {s / k:16.1f}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else table(sys.argv[2:])
