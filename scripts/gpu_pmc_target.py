"""Tiny target for rocprofv3 --pmc runs: a few launches of each fused-pass variant (march + tiled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mal_amd import _lib, ops, layers
from mal_amd.synthetic import make_batch
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W = 12, 192, 640
b = make_batch(B, H, W, seed=5)
g = {k: v.to(dev) for k, v in b.items() if torch.is_tensor(v)}
T0 = layers.transformation_from_parameters(g["axisangle_m1"], g["translation_m1"], True)
T1 = layers.transformation_from_parameters(g["axisangle_p1"], g["translation_p1"], False)
srcs = [g["color_m1"], g["color_p1"]]
ident = ops.photo_fwd(g["color0"], srcs, want_argmin=False, want_weight=False)[0]
noise = torch.randn(B, 1, H, W, device=dev)
impls = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,0").split(",")]
for impl in impls:
    lib.mal_set_option(b"pass_impl", impl)
    for it in range(3):
        ops.pass_fused(g["disp_teacher"], g["K"], g["inv_K"], [T0, T1], srcs, g["color0"], flags=0)
        ops.pass_fused(g["disp_teacher"], g["K"], g["inv_K"], [T0, T1], srcs, g["color0"], ident=ident, noise=noise, flags=7)
torch.cuda.synchronize()
