#!/bin/bash
# Round 5: grouped LoRA weight-gradient launches (fdmi_wgrad_tn_group) -- parity, isolated rates, in-process A/B on the C2 step.
set -u
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/group
( timeout 300 python -m pytest tests/test_zz_dit_gpu.py tests/test_unet_gpu.py -m gpu -q -x -k "wgrad or plan_matches or unet" 2>&1 | tail -8 ) > gpurun_out/group/tests.txt 2>&1
tail -4 gpurun_out/group/tests.txt
timeout 120 python scripts/wgrad_rates.py groups > gpurun_out/group/rates.txt 2>&1; cat gpurun_out/group/rates.txt | cut -c1-200
timeout 200 python scripts/knob_ab.py --rounds 4 --steps 4 --variants base --extra "ungrouped:47=1" > gpurun_out/group/knob_ab.txt 2>&1; tail -6 gpurun_out/group/knob_ab.txt | cut -c1-200
