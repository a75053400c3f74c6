"""Sweep the marching kernel's rows-per-task on the GPU box (packed sources)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mal_amd import build, _lib, ops, layers
from mal_amd.synthetic import make_batch
build.build(verbose=False)
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, W = 12, 192, 640
g = {k: v.to(dev) for k, v in make_batch(B, H, W, seed=5).items() if torch.is_tensor(v)}
T0 = layers.transformation_from_parameters(g["axisangle_m1"], g["translation_m1"], True)
T1 = layers.transformation_from_parameters(g["axisangle_p1"], g["translation_p1"], False)
srcs = [g["color_m1"], g["color_p1"]]
ident = ops.photo_fwd(g["color0"], srcs, want_argmin=False, want_weight=False)[0]
noise = torch.randn(B, 1, H, W, device=dev)
_, mono_depth = ops.disp_to_depth(g["disp_teacher"], 0.1, 100.0)
mr = ops.pass_fused(g["disp_teacher"], g["K"], g["inv_K"], [T0, T1], srcs, g["color0"], ident=ident, noise=noise, flags=1)["min_reproj"]
def t(flags, n=30):
    kw = dict(ident=ident, noise=noise) if flags & 1 else {}
    if flags & 32:
        kw.update(ext_mask=g["consistency_mask"].reshape(B, 1, H, W), mono_depth=mono_depth, mono_reproj=mr, ens_reproj=mr)
    f = lambda: ops.pass_fused(g["disp_teacher"], g["K"], g["inv_K"], [T0, T1], srcs, g["color0"], flags=flags, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e6
for rows in [int(x) for x in sys.argv[1].split(",")]:
    lib.mal_set_option(b"march_rows", rows)
    print("rows %2d: fwd %.0f us  teacher %.0f us  student %.0f us" % (rows, t(0), t(7), t(34)))
