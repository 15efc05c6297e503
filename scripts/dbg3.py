import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbench
kbench.gemm_cases = lambda: [("conv3x3 64x64 320->320", "conv", 64, 320, 320), ("conv3x3 32 640->640", "conv", 32, 640, 640), ("conv3x3 32 1920->640", "conv", 32, 1920, 640), ("linear 65536x320x1280", "lin", 65536, 320, 1280), ("linear 65536x2560x320", "lin", 65536, 2560, 320), ("linear 16384x640x640", "lin", 16384, 640, 640), ("linear 65536x2560x320", "lin", 65536, 2560, 320)]
for ft in ((256 << 16) | 160, (256 << 16) | 128):
    print("force_tile", hex(ft))
    kbench.run_gemm(20, force_tile=ft)
