"""Process-wide switches of the MAL loss path (plain module attributes).

noise_source
    "cpu"  (default) the automask tie-break noise is ``torch.randn(shape)`` drawn from the
           global CPU generator and copied to the device, and ``compute_main_losses`` still
           makes its (dead) draw -- the RNG stream is then exactly the reference's
           (manydepth/loss_utils.py:105-106,178).
    "cuda" the device generator: no host work and no PCIe copy on the step (the reference's
           host-sync hazard listed in SURVEY.md section 5); same distribution, different stream.
    "philox" (whole-step API, ``mal_amd.step.loss_step``) drawn inside the step's first HIP kernel: Philox4x32-10
           keyed by ``noise_seed``, counter = (pixel, step number held in a device word the step advances) + Box-Muller;
           nothing but the library's own kernels runs on the step.  The operator-level API treats it as "cuda".
noise_seed
    key of the "philox" stream (per rank: add the rank).
consistency_target
    write ``outputs["consistency_target/0"]`` (loss_utils.py:212-215, a logging-only map).
"""
noise_source = "cpu"
noise_seed = 0x4d414c5eed
consistency_target = True
