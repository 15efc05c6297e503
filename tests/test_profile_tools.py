"""CPU: scripts/rocprof_to_profiles.py on synthetic rocprofv3-style CSV files -- the per-step kernel table, the gfx950 FETCH_SIZE
correction (x2) and the bench.py bucket names of the traffic summary (the real files come from scripts/profile_c2.sh on the GPU
box; the numbers below are the round-1 conv kernel's: 298397.2 KB reported -> 611.12 MB fetched, 74512 KB -> 76.30 MB written)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = "void (anonymous namespace)::gemm4_kernel<1, false, 320, false>(GemmArgs)"


def test_converter_reproduces_the_round1_figures(tmp_path):
    for d in ("stats/h", "f/h", "w/h", "t/h"):
        os.makedirs(tmp_path / d)
    (tmp_path / "stats/h/1_kernel_stats.csv").write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        f'"{K}",760,179988000,236826,20.0,1,2,3\n"void gn_reduce_kernel<false>(GnArgs, int)",1220,31892000,26141,3.5,1,2,3\n')
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
            '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name",'
            '"Counter_Value","Start_Timestamp","End_Timestamp"\n')
    row = '{i},{i},4,1,10,10,131072,5,"' + K + '",512,147456,0,128,128,112,"{c}",{v},1,2\n'
    (tmp_path / "f/h/1_counter_collection.csv").write_text(head + "".join(row.format(i=i, c="FETCH_SIZE", v=298397.2) for i in (1, 2)))
    (tmp_path / "w/h/1_counter_collection.csv").write_text(head + "".join(row.format(i=i, c="WRITE_SIZE", v=74512.0) for i in (1, 2)))
    (tmp_path / "t/h/1_counter_collection.csv").write_text(
        head + "".join(row.format(i=i, c=c, v=v) for i in (1, 2) for c, v in (("TCC_HIT_sum", 3.0e6), ("TCC_MISS_sum", 1.0e6))))
    pre = os.path.join(ROOT, "profiles", "r987")
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocprof_to_profiles.py"), "--round", "987", "--steps", "4",
                        "--stats-dir", str(tmp_path / "stats"), "--fetch-dir", str(tmp_path / "f"), "--write-dir", str(tmp_path / "w"),
                        "--tcc-dir", str(tmp_path / "t")],
                       check=True, capture_output=True)
        stats = open(pre + "_kernel_stats.csv").read().splitlines()
        assert stats[2] == '"void gemm4_kernel<1, false, 320, false>",190.0,44.997,236.8'
        assert stats[3] == '"void gn_reduce_kernel<false>",305.0,7.973,26.1'
        t = json.load(open(pre + "_traffic.json"))["kernels"]["gemm4_kernel<256x320,conv>"]
        assert t["fetch_MB_corrected"] == 611.12 and t["write_MB"] == 76.3 and t["launches_sampled"] == 2
        assert abs(t["hbm_bytes_per_launch"] - 687.42e6) < 0.01e6
        l2 = open(pre + "_l2_hit_rate.csv").read().splitlines()
        assert l2[2] == '"void gemm4_kernel<1, false, 320, false>",2,3000000,1000000,0.7500'
    finally:
        for suf in ("_kernel_stats.csv", "_pmc_hbm_traffic.csv", "_traffic.json", "_l2_hit_rate.csv"):
            if os.path.exists(pre + suf):
                os.remove(pre + suf)


def test_bench_algorithmic_bytes_walks_the_plan_of_the_headline_step():
    """bench.py's algorithmic bytes per GEMM family come from the C++ plan's work list in workspace-query mode (no GPU): the launch
    counts of the 256 x 320 kernels are the ones rocprofv3 sees per step (profiles/r4_kernel_stats.csv: 423 + 80 + 14 row, 90 + 100
    conv), and one operand pass of a family's mean launch is tens to hundreds of MB."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_algo", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    algo = bench.algorithmic_bytes(16, 4, 128)
    assert algo is not None
    assert algo["gemm4_kernel<256x320,row>"][1] == 517 and algo["gemm4_kernel<256x320,conv>"][1] == 190
    for key, (by, n) in algo.items():
        assert n > 0 and 1e6 < by / n < 1e9, (key, by, n)
