"""flash_diffusion_amd -- MI355X (gfx950) native Flash-Diffusion distillation hot path.

Host-side mirror of the reference's `src/flash` call contract for the hot path
(denoiser wrapper, FlashDiffusion.forward, TrainingPipeline.training_step) over the C-ABI
library `libfdmi.so` (include/fdmi.h): hand-written HIP kernels for CDNA4, no Triton, no
CUDA shims, no CPU fallback -- importing the ops without the built library raises."""
__version__ = "0.1.0"
