"""mal_amd: MI355X-native photometric reprojection + motion-aware loss path of MAL.

Host side: Python on PyTorch-ROCm mirroring the reference's ``manydepth.layers`` /
``manydepth.loss_utils`` / Trainer warp-loss API.  Device side: hand-written HIP kernels
for gfx950 behind the C ABI of ``include/mal_hip.h`` (``mal_amd/lib/libmal_hip.so``).
There is no CPU implementation in this package: without the HIP library and a GPU the
operators raise.
"""
__version__ = "0.1.0"
