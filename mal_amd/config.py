"""Process-wide switches of the MAL loss path (plain module attributes).

noise_source
    "cpu"  (default) the automask tie-break noise is ``torch.randn(shape)`` drawn from the
           global CPU generator and copied to the device, and ``compute_main_losses`` still
           makes its (dead) draw -- the RNG stream is then exactly the reference's
           (manydepth/loss_utils.py:105-106,178).
    "cuda" the device generator: no host work and no PCIe copy on the step (the reference's
           host-sync hazard listed in SURVEY.md section 5); same distribution, different stream.
consistency_target
    write ``outputs["consistency_target/0"]`` (loss_utils.py:212-215, a logging-only map).
"""
noise_source = "cpu"
consistency_target = True
