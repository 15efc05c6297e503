"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32) restatement of the Flash-Diffusion distillation hot path of
gojasper/flash-diffusion, used as the parity checker for the MI355X HIP path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from here.  The product package
(``flash_diffusion_amd``) never imports ``oracle`` and fails loudly when its
HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * orchestration + losses (FlashDiffusion.forward, _distill_loss, _dmd_loss,
    _gan_loss, _get_timesteps, _predicted_x_0, boundary scalings): PINNED --
    ``oracle/flash_ref.py`` is checked bit-exactly against the reference's own
    ``flash.models.flash.FlashDiffusion`` imported unmodified from
    /root/reference/src (``oracle/shim_import.py``), and the committed fixtures
    under ``tests/golden/`` were produced by that real reference class.
  * the few-step sampler (FlashDiffusion.sample / log_samples, FD:754-1019): PINNED the same way
    (``FlashDiffusionRef.sample``; fixture ``tests/golden/sample_lcm4.npz`` from the real class).
  * the flow-matching step (FlashDiffusionSD3.forward, _dmd_loss, _gan_loss, get_sigmas;
    flash_sd3/flash_diffusion_model.py): PINNED -- ``oracle/flash_sd3_ref.py`` is bit-identical to the
    reference's own ``FlashDiffusionSD3`` (5 configurations x G/D step x 3 start indices), fixtures
    ``tests/golden/sd3_*.npz`` from the real class.  (its HIP path is
    flash_diffusion_amd/flash_sd3.py over the MMDiT plan, checked against these fixtures by tests/test_flash_sd3_gpu.py.)
  * denoiser / scheduler arithmetic (diffusers UNet2DConditionModel,
    DPMSolverMultistepScheduler, DDPMScheduler, LCMScheduler, FlowMatchEulerDiscreteScheduler): PARITY UNPINNED by the
    reference -- it lives in an un-vendored fork of diffusers
    (requirements.txt:1, ``git+https://github.com/initml/diffusers.git@clement/feature/flash``,
    a branch ref, not installed, no network) and the reference's tests hold no
    golden vectors for it.  ``oracle/unet_cpu.py`` and
    ``oracle/sched_cpu.py`` restate the published upstream diffusers algorithm.
"""
