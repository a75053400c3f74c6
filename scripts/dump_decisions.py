"""Dump the kernels' decisions and gradients of the B=12 192x640 synthetic step for offline analysis."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mal_amd.synthetic import make_batch
from scripts.explore_decisions import run_step_dec
B, H, W = 12, 192, 640
b = make_batch(B, H, W, seed=77)
g = torch.Generator().manual_seed(5)
n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
h = run_step_dec(b, {}, n0)
out = {"g_" + k: v for k, v in h["grads"].items()}
out["dec_teacher"] = h["maps"]["dec_teacher"].numpy()
out["dec_student"] = h["maps"]["dec_student"].numpy()
out["cmask"] = h["maps"]["consistency_mask"].numpy().astype(np.uint8)
out["mono_reproj"] = h["maps"]["mono_reproj"].numpy()
np.savez_compressed("gpurun_out/dec_dump_640.npz", **out)
print("ok", os.path.getsize("gpurun_out/dec_dump_640.npz") / 1e6, "MB")
