"""Per-SHAPE table of a profiled step (dev tool, round 6): `FDMI_BENCH_SHAPES=out.csv python bench.py ...` makes the library write one
line per launch of the bench's profiled step (fdmi_prof_dump: per-dispatch HIP events, algorithmic flops and bytes);
`python scripts/shape_table.py out.csv` groups them by (kernel family, shape) and prices each group against the roof its arithmetic
intensity puts it under (2.5 PFLOP/s : 8 TB/s ridge = 312.5 flop/B)."""
import csv, sys
from collections import defaultdict
BUCKETS = {0: "gemm128x128 row", 1: "gemm128x64 row", 2: "gemm64x128 row", 3: "gemm64x64 row", 4: "gemm128x128 conv", 5: "gemm128x64 conv",
           6: "gemm64x128 conv", 7: "gemm64x64 conv", 8: "attn fwd", 9: "attn dQ", 10: "attn dKV", 11: "g3 256x160 row", 12: "g3 256x128 row",
           13: "g3 256x160 conv", 14: "g3 256x128 conv", 15: "g4 256x320 row", 16: "g4 256x320 conv", 17: "g4 256x192 row", 18: "g4 256x192 conv",
           19: "wgrad_tn", 22: "g5 128x320 row"}
PEAK, HBM = 2.5e15, 8.0e12
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (int(r["bucket"]), int(r["kind"]), int(r["s0"]), int(r["s1"]), int(r["s2"]), int(r["s3"]))
    a = agg[k]
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"]); a[3] += float(r["bytes"])
tot = sum(a[1] for a in agg.values())
print(f"# {sum(a[0] for a in agg.values())} profiled launches, {tot:.2f} ms (serial per-dispatch time); flags: 1 residual, 2 GEGLU, 4 dgrad, 8 GN sums, >>8 split-K")
print(f"{'family':18s} {'shape (M N K flags | BH Sq Skv d)':38s} {'n':>4s} {'ms':>8s} {'us/launch':>9s} {'TFLOP/s':>8s} {'TB/s':>6s} {'flop/B':>7s} {'bound':>5s} {'frac':>5s} {'ms at 100%':>10s}")
for k, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    b, kind, s0, s1, s2, s3 = k
    n, ms, fl, by = a
    t = ms * 1e-3
    ai = fl / by if by else float("inf")
    hbm = by > 0 and ai < PEAK / HBM
    frac = (by / t / HBM) if hbm else (fl / t / PEAK)
    ideal = (by / HBM if hbm else fl / PEAK) * 1e3
    print(f"{BUCKETS.get(b, str(b)):18s} {f'{s0} {s1} {s2} {s3}':38s} {n:4d} {ms:8.3f} {ms / n * 1e3:9.1f} {fl / t / 1e12:8.0f} {by / t / 1e12:6.2f} {ai:7.0f} "
          f"{'hbm' if hbm else 'mfma':>5s} {frac:5.2f} {ideal:10.3f}")
