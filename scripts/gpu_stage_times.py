"""Per-stage cycle counts of the marching teacher pass (debug bit 64 makes every wave write its stage timers
over min_reproj[task*8..]); prints the mean per wave and per iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mal_amd import _lib, ops, layers
from mal_amd.synthetic import make_batch
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, W = 12, 192, 640
g = {k: v.to(dev) for k, v in make_batch(B, H, W, seed=5).items() if torch.is_tensor(v)}
T0 = layers.transformation_from_parameters(g["axisangle_m1"], g["translation_m1"], True)
T1 = layers.transformation_from_parameters(g["axisangle_p1"], g["translation_p1"], False)
srcs = [g["color_m1"], g["color_p1"]]
ident = ops.photo_fwd(g["color0"], srcs, want_argmin=False, want_weight=False)[0]
noise = torch.randn(B, 1, H, W, device=dev)
names = ["loop+params", "issue(proj+gathers)", "smooth+epi", "gather wait+blend+ring", "hsums", "stats/SSIM", "HC+G", "rolls"]
for flags, label in ((7, "teacher"), (0, "forward-only")):
    lib.mal_set_option(b"debug", int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    for _ in range(2):
        out = ops.pass_fused(g["disp_teacher"], g["K"], g["inv_K"], [T0, T1], srcs, g["color0"],
                             ident=ident if flags & 1 else None, noise=noise if flags & 1 else None, flags=flags)
    torch.cuda.synchronize()
    ntasks = 12 * 11 * 15 if flags else 12 * 11 * 15
    t = out["min_reproj"].flatten()[: ntasks * 8].reshape(ntasks, 8).double().cpu()
    iters = 17 if flags else 15
    m = t.mean(0)
    print(label, "total cycles/wave %.0f  per-iteration %.0f" % (m.sum(), m.sum() / iters))
    for n, v in zip(names, m):
        print("   %-26s %8.0f /iter  (%4.1f%%)" % (n, v / iters, 100 * v / m.sum()))
lib.mal_set_option(b"debug", 0)
