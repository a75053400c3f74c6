"""Tiny target for rocprofv3 --pmc runs: three whole loss steps of each kind at B=12 192x640 -- --distil
(mal_loss_step_fwd/_bwd: the north-star teacher kernel) and --temporal --distil (mal_loss_step_warp/_fwd/_bwd)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
for mode in ("distil", "step"):
    step = bench.Step(torch.device("cuda:0"), 1234, mode)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
