import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbench
from flash_diffusion_amd._lib import lib
kbench.gemm_cases = lambda: [("conv3x3 32 1920->640", "conv", 32, 1920, 640), ("linear 65536x320x1280", "lin", 65536, 320, 1280)]
for n in (256, 128, 64, 32, 8):
    lib().fdmi_tune_set(8, n)
    print("ncu", n)
    kbench.run_gemm(10)
