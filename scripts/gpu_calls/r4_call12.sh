#!/bin/bash
# Round 4, call 12: the transformer denoisers through their C++ plans (fdmi_dit_*): the existing DiT / MMDiT parity tests.
mkdir -p gpurun_out/c12
timeout 900 python -m pytest tests/test_zz_dit_gpu.py -q -k "dit_frozen or dit_lora" > gpurun_out/c12/dit.txt 2>&1; echo "dit rc=$?"
tail -40 gpurun_out/c12/dit.txt | cut -c1-400
timeout 600 python -m pytest tests/test_pixart_step_gpu.py tests/test_sd3_mmdit_step_gpu.py -q > gpurun_out/c12/steps.txt 2>&1; echo "steps rc=$?"
tail -30 gpurun_out/c12/steps.txt | cut -c1-400
