"""Tiny target for rocprofv3 --pmc runs: three whole loss steps (mal_loss_step_fwd/_bwd) at B=12 192x640."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
step = bench.Step(torch.device("cuda:0"), 1234, "step")
for _ in range(3):
    step()
torch.cuda.synchronize()
