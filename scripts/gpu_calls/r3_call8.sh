#!/bin/bash
# Round-3 eighth GPU call: T5 text encoder + its kernels, the fused GELU / gate epilogues (kernel + DiT model tests, fp32 gate), PixArt / SD3 bench legs.
set -u
out=gpurun_out/r3c8
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_t5 timeout 900 python -m pytest tests/test_t5_gpu.py -q -rxXsf -p no:cacheprovider
grep -h "FAILED\|passed\|failed\|Error" "$out/01_t5.log" | tail -12
run 02_epi timeout 900 python -m pytest tests/test_kernels_gpu.py -q -rxXsf -p no:cacheprovider -k "gate_and_gelu or epilogues or geglu"
grep -h "FAILED\|passed\|failed" "$out/02_epi.log" | tail -8
run 03_dit timeout 1500 python -m pytest tests/test_zz_dit_gpu.py tests/test_fp32_gate_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_pixart_step_gpu.py -q -rxXsf -p no:cacheprovider -k "dit or DiT or mmdit or pixart or sd3"
grep -h "FAILED\|passed\|failed" "$out/03_dit.log" | tail -12
for arch in pixart sd3; do
  timeout 600 python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_$arch.json" 2> "$out/bench_$arch.err"; tail -c 400 "$out/bench_$arch.json"; echo
done
cp gpurun_out/*.txt "$out/" 2>/dev/null
