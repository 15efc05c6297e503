"""GEMM / conv counter experiment (developer tool).  `python scripts/gemm_pmc.py run` launches the hot GEMM
families of the C2 / C4 steps a few times each, with and without the timing ablations of the DEV instantiation (knob 40 = 32: the
K loop alone; 48: K loop alone with A from L2) -- run it under `rocprofv3 --pmc ... --kernel-trace` (scripts/gemm_pmc.sh).
`python scripts/gemm_pmc.py table DIR...` prints per (kernel, grid, case) the mean counters, the durations of every pass, and the
derived figures: MFMA-pipe busy share, wave-state split (waiting / issue-stalled / issuing), LDS bank-conflict share, clock."""
import csv, glob, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (tag, M, N, K, kind) -- kind: row | res (row + residual) | geglu | conv (3x3, B x hw x hw x Cin -> N)
CASES = [
    ("row320_res  M131072 N320  K320 ", 131072, 320, 320, "res", (256 << 16) | 320),
    ("row320      M131072 N320  K1280", 131072, 320, 1280, "res", (256 << 16) | 320),
    ("row320 qkv  M131072 N960  K320 ", 131072, 960, 320, "row", (256 << 16) | 320),
    ("geglu320    M131072 N2560 K320 ", 131072, 2560, 320, "geglu", (256 << 16) | 320),
    ("conv320 64x64 320->320 B32     ", 131072, 320, 2880, "conv", 0),
    ("conv320 32x32 640->640 B32     ", 32768, 640, 5760, "conv", 0),
    ("row192      M32768 N1152 K1152 ", 32768, 1152, 1152, "row", (256 << 16) | 192),
    ("row192      M32768 N4608 K1152 ", 32768, 4608, 1152, "row", (256 << 16) | 192),
    ("row192      M32768 N1152 K4608 ", 32768, 1152, 4608, "res", (256 << 16) | 192),
    ("g3 160 res  M131072 N320  K320 ", 131072, 320, 320, "res", (256 << 16) | 160),
    ("g3 128 rank M131072 N128  K320 ", 131072, 128, 320, "row", (256 << 16) | 128),
]


def run(reps=4):
    import torch
    from flash_diffusion_amd import ops, _lib
    L = _lib.lib()
    BF = torch.bfloat16
    for (tag, M, N, K, kind, tile) in CASES:
        Nout = N // 2 if kind == "geglu" else N
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, Nout, dtype=BF, device="cuda")
        R = torch.randn(M, Nout, device="cuda").to(BF) if kind == "res" else None
        kw = dict(bias=bias, residual=R, out=out, force_tile=tile)
        if kind == "conv":
            Cin = K // 9
            hw = {131072: 64, 32768: 32}[M]
            A = torch.randn(32, hw, hw, Cin, device="cuda").to(BF)
            kw.update(M=M, conv=dict(Hin=hw, Win=hw, Cin=Cin, Hout=hw, Wout=hw, KH=3, KW=3, stride=1, pad=1))
        else:
            A = torch.randn(M, K, device="cuda").to(BF)
            if kind == "geglu":
                kw["act"] = ops.ACT_GEGLU
        devs = (0, 32, 48) if (tile >> 16 == 256 and (tile & 0xffff) == 320) or kind == "conv" else (0,)
        for dev in devs:
            L.fdmi_tune_set(40, dev)
            for _ in range(reps):
                ops.gemm(A, W, **kw)
            torch.cuda.synchronize()
        L.fdmi_tune_set(40, 0)
        print("ran", tag, flush=True)


def short(n):
    m = re.search(r"(gemm\d?_kernel)<([^>]*)>", n)
    return f"{m.group(1)}<{m.group(2)}>" if m else None


def table(dirs):
    # rocprofv3 gives no case tag: dispatches are matched to (case, dev) by their ORDER within each pass (run() is deterministic)
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    order = []
    for d in dirs:
        base = os.path.basename(d.rstrip("/"))
        trace = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    n = short(row.get("Kernel_Name") or "")
                    if not n:
                        continue
                    trace[int(row["Dispatch_Id"])] = (n, str(row.get("Grid_Size", "")), (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
        ids = sorted(trace)
        # consecutive dispatches of one (kernel, grid) in groups of `reps`
        seq = defaultdict(int)
        key_of = {}
        prev, cnt = None, 0
        for i in ids:
            n, g, us = trace[i]
            if (n, g) != prev:
                seq[(n, g)] += 1
                prev = (n, g)
            key = f"{n} grid={g} #{seq[(n, g)]}"
            key_of[i] = key
            if key not in order:
                order.append(key)
            a = agg[key][f"duration_us[{base}]"]
            a[0] += 1
            a[1] += us
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    i = int(row["Dispatch_Id"])
                    if i not in key_of:
                        continue
                    a = agg[key_of[i]][row["Counter_Name"]]
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    for key in order:
        c = {k: v[1] / v[0] for k, v in agg[key].items()}
        print(key)
        for k in sorted(c):
            print(f"    {k:34s} n={agg[key][k][0]:3d} mean {c[k]:16.1f}")
        der = []
        if "GRBM_GUI_ACTIVE" in c:
            durs = [v for k, v in c.items() if k.startswith("duration_us[p3")]
            if durs:
                der.append(f"clock {c['GRBM_GUI_ACTIVE'] / durs[0] / 1e3:.2f} GHz")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:   # cycles per SIMD summed over the chip's 1024 SIMDs
                der.append(f"MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * 1024):.1f} % of SIMD-cycles")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            der.append("wave cycles: waiting {:.1f} % / issue-stalled {:.1f} % / issuing {:.1f} %".format(
                100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
            der.append(f"LDS bank conflicts {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.1f} % of LDS-active cycles")
        if "SQ_LDS_IDX_ACTIVE" in c and "GRBM_GUI_ACTIVE" in c:
            der.append(f"LDS array active {100 * c['SQ_LDS_IDX_ACTIVE'] / (c['GRBM_GUI_ACTIVE'] * 256):.1f} % of CU-cycles")
        for x in der:
            print("    =>", x)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else table(sys.argv[2:])
