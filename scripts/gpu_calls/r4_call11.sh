#!/bin/bash
# Round 4, call 11: the two test changes that had not run on the GPU yet + a short bench for the new per-family traffic fields.
mkdir -p gpurun_out/c11
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "lean_and_line_wide" > gpurun_out/c11/epi.txt 2>&1; echo "epi rc=$?"
tail -5 gpurun_out/c11/epi.txt
timeout 900 python -m pytest tests/test_multiproc_gpu.py -q > gpurun_out/c11/multiproc.txt 2>&1; echo "multiproc rc=$?"
tail -5 gpurun_out/c11/multiproc.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c11/bench.json 2> gpurun_out/c11/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/c11/bench.json
