"""A/B on the GPU box: marching (pass_impl=1) vs LDS-tiled (pass_impl=0) fused pass; correctness deltas + timing."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mal_amd import build, _lib, ops, layers
from mal_amd.synthetic import make_batch
build.build(verbose=False)
lib = _lib.load()
L = _lib
dev = torch.device("cuda:0")
B, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (12, 192, 640)))
b = make_batch(B, H, W, seed=5)
g = {k: v.to(dev) for k, v in b.items() if torch.is_tensor(v)}
T0 = layers.transformation_from_parameters(g["axisangle_m1"], g["translation_m1"], True)
T1 = layers.transformation_from_parameters(g["axisangle_p1"], g["translation_p1"], False)
srcs = [g["color_m1"], g["color_p1"]]
ident = ops.photo_fwd(g["color0"], srcs, want_argmin=False, want_weight=False)[0]
noise = torch.randn(B, 1, H, W, device=dev)
_, mono_depth = ops.disp_to_depth(g["disp_teacher"], 0.1, 100.0)

def run(impl, flags, **kw):
    lib.mal_set_option(b"pass_impl", impl)
    return ops.pass_fused(g["disp_student"] if kw.get("student") else g["disp_teacher"], g["K"], g["inv_K"], [T0, T1], srcs, g["color0"],
                          ident=ident if flags & 1 else None, noise=noise if flags & 1 else None,
                          ext_mask=g["consistency_mask"].reshape(B, 1, H, W) if kw.get("student") else None,
                          mono_depth=mono_depth if kw.get("student") else None,
                          mono_reproj=kw.get("mono_reproj"), ens_reproj=kw.get("ens_reproj"), flags=flags)

def cmp(name, a, c):
    a, c = a.cpu().numpy().astype(np.float64), c.cpu().numpy().astype(np.float64)
    d = np.abs(a - c); sc = np.abs(c).max() + 1e-30
    msg = "%-12s max|d|/max %.2e  frac>1e-4 %.2e" % (name, d.max() / sc, (d > 1e-4 * sc).mean())
    if a.ndim == 4 and a.shape[-1] > 8:
        msg += "  | interior %.2e rows0/-1 %.2e %.2e cols0/-1 %.2e %.2e" % (d[..., 2:-2, 2:-2].max() / sc, d[..., 0, :].max() / sc, d[..., -1, :].max() / sc, d[..., :, 0].max() / sc, d[..., :, -1].max() / sc)
    print(msg)

F = L
for nm, flags, kw in (("ensemble", 0, {}), ("teacher", F.F_AUTOMASK | F.F_GRAD | F.F_POSE_GRAD, {}),):
    o0, o1 = run(0, flags, **kw), run(1, flags, **kw)
    print("==", nm, "sums tiled", o0["sums"][:4].tolist(), "march", o1["sums"][:4].tolist())
    cmp("min_reproj", o1["min_reproj"], o0["min_reproj"])
    if flags & F.F_GRAD:
        cmp("g_reproj", o1["g_reproj"], o0["g_reproj"])
        cmp("g_T0", o1["g_T"][0], o0["g_T"][0]); cmp("g_T1", o1["g_T"][1], o0["g_T"][1])
mr = run(0, F.F_AUTOMASK)["min_reproj"]; er = run(0, 0)["min_reproj"]
kw = dict(student=True, mono_reproj=mr, ens_reproj=er)
o0, o1 = run(0, F.F_GRAD | F.F_EPILOGUE, **kw), run(1, F.F_GRAD | F.F_EPILOGUE, **kw)
print("== student sums tiled", o0["sums"][:4].tolist(), "march", o1["sums"][:4].tolist())
for k in ("min_reproj", "g_reproj", "g_cons", "g_distil"):
    cmp(k, o1[k], o0[k])

# timing
def timeit(impl, flags, n=30, **kw):
    for _ in range(3): run(impl, flags, **kw)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): run(impl, flags, **kw)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for pk in (False, True):
  ops.pack_sources = pk
  print("pack_sources", pk)
  for rows in ([16] if len(sys.argv) < 5 else [int(x) for x in sys.argv[4].split(",")]):
    lib.mal_set_option(b"march_rows", rows)
    print("  rows", rows, "ensemble us: tiled %.0f march %.0f tile2 %.0f | teacher: tiled %.0f march %.0f tile2 %.0f | student: tiled %.0f march %.0f tile2 %.0f" % (
          timeit(0, 0), timeit(1, 0), timeit(2, 0), timeit(0, 7), timeit(1, 7), timeit(2, 7), timeit(0, 34, **kw), timeit(1, 34, **kw), timeit(2, 34, **kw)))
# tile2 vs tiled correctness
for nm, flags, kw2 in (("ensemble", 0, {}), ("teacher", 7, {}), ("student", 34, kw)):
    o0, o2 = run(0, flags, **kw2), run(2, flags, **kw2)
    print("== tile2 vs tiled", nm)
    cmp("min_reproj", o2["min_reproj"], o0["min_reproj"])
    if flags & 2:
        cmp("g_reproj", o2["g_reproj"], o0["g_reproj"])
    if flags & 4:
        cmp("g_T0", o2["g_T"][0], o0["g_T"][0])
