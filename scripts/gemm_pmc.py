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


def run_order(reps=4):
    """the (case tag, developer bits) of every group of `reps` consecutive GEMM dispatches, in the order run() issues them"""
    out = []
    for (tag, M, N, K, kind, tile) in CASES:
        devs = (0, 32, 48) if (tile >> 16 == 256 and (tile & 0xffff) == 320) or kind == "conv" else (0,)
        for dev in devs:
            out.append((tag, {0: "as shipped", 32: "K loop alone (no epilogue: WRONG results, timing ablation)",
                              48: "K loop alone, A from L2 (timing ablation)"}[dev]))
    return out


def table(dirs, reps=4):
    """rocprofv3 gives no case tag: the GEMM dispatches of a pass are matched to (case, developer bits) by their ORDER (run() is
    deterministic: `reps` launches per (case, bits))"""
    order = run_order(reps)
    agg = [defaultdict(lambda: [0, 0.0]) for _ in order]
    kname = [None] * len(order)
    for d in dirs:
        base = os.path.basename(d.rstrip("/"))
        trace = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    n = short(row.get("Kernel_Name") or "")
                    if n:
                        trace[int(row["Dispatch_Id"])] = (n, (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
        ids = sorted(trace)
        if len(ids) != reps * len(order):
            print(f"# pass {base}: {len(ids)} GEMM dispatches, expected {reps * len(order)} -- skipped")
            continue
        grp = {i: k // reps for k, i in enumerate(ids)}
        for i in ids:
            kname[grp[i]] = trace[i][0]
            a = agg[grp[i]][f"duration_us[{base}]"]
            a[0] += 1
            a[1] += trace[i][1]
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    i = int(row["Dispatch_Id"])
                    if i in grp:
                        a = agg[grp[i]][row["Counter_Name"]]
                        a[0] += 1
                        a[1] += float(row["Counter_Value"])
    print("# per launch (mean of %d); SQ_* summed over the chip: SQ_BUSY_CYCLES over its 32 shader engines (so SQ_BUSY_CYCLES / 32 / duration = the\n"
          "# shader clock while the kernel runs), SQ_VALU_MFMA_BUSY_CYCLES over its 1024 SIMDs in clocks (16 per v_mfma_f32_16x16x32_bf16),\n"
          "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* over all waves in quad-cycles.  GRBM_GUI_ACTIVE is summed over the 8 XCDs and\n"
          "# includes the dispatch's ramp: only meaningful for launches of several hundred us." % reps)
    for k, (tag, what) in enumerate(order):
        c = {n: v[1] / v[0] for n, v in agg[k].items()}
        if not c:
            continue
        print(f"{tag.strip()} | {what} | {kname[k]}")
        for n in sorted(c):
            print(f"    {n:34s} n={agg[k][n][0]:3d} mean {c[n]:16.1f}")
        dur = next((v for n, v in sorted(c.items()) if n.startswith("duration_us[p1")), None)
        if "SQ_BUSY_CYCLES" in c and dur:
            clk = c["SQ_BUSY_CYCLES"] / 32 / dur / 1e3
            print(f"    => shader clock during the kernel {clk:.2f} GHz (SQ_BUSY_CYCLES / 32 shader engines / {dur:.1f} us)")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                print(f"    => MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * c['SQ_BUSY_CYCLES']):.1f} % of the SIMD-cycles of the launch "
                      f"({c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 1e3:.1f} k clocks per SIMD of {c['SQ_BUSY_CYCLES'] / 32 / 1e3:.1f} k)")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            print("    => wave cycles: parked at s_waitcnt / s_barrier {:.1f} % | issue-stalled (pipe busy, dependency) {:.1f} % | issuing {:.1f} %".format(
                100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            print(f"    => LDS bank conflicts {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f} % of the LDS-active cycles; "
                  f"SQ_WAIT_INST_LDS {100 * c.get('SQ_WAIT_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 0) or 1, 1):.1f} % of wave cycles" if "SQ_WAVE_CYCLES" in c else
                  f"    => LDS bank conflicts {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f} % of the LDS-active cycles")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else table(sys.argv[2:])
