"""Target for the rocprofv3 --pmc passes of round 5: the north-star kernel (march_teacher_kernel<false>) in one regime.
usage: gpu_pmc_regime_target.py cold|warm
  cold: 24 launches rotating over 8 batches, each with its own inputs and workspace (~77 MB of operands per launch, 540 MB
        of other batches' traffic between two launches on the same batch: nothing is left in the 256 MiB Infinity Cache)
  warm: 24 launches on ONE batch (its operands stay on the die)
plus, in both, three whole --temporal --distil steps of the rotation (cold: 3 different batches; warm: the same one)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

regime = sys.argv[1] if len(sys.argv) > 1 else "cold"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
from mal_amd import build  # noqa: E402
build.build(verbose=False)
R = 8 if regime == "cold" else 1
plains = bench.Rotation(dev, 1234 + 104729, "distil", R, slot_base=300, graph=False)
enq = bench.teacher_enqueuers(plains.steps, slot_base=100)
for i in range(24):
    enq[i % R](1)
torch.cuda.synchronize()
rot = bench.Rotation(dev, 1234, "step", 6 if regime == "cold" else 1, graph=False)
for _ in range(3 if regime == "warm" else 6):
    rot.eager()
torch.cuda.synchronize()
