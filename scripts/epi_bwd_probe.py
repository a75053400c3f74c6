"""time the lookup's backward call alone (HIP events, back to back) under kernel experiment bits"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from mal_amd import _lib as L, ops
from oracle.gen_golden_epi import make_case
B, Cn, h, w, r, Lv = 8, 128, 48, 160, 8, 3
K, depth, poses, f1, f2 = make_case(B, Cn, h, w, seed=3)
dev = "cuda:0"
lib = L.load()
d1 = 2 * r + 1
D = Lv * d1
f1d = f1.to(dev)
pyr = [f2.to(dev)]
for _ in range(Lv - 1):
    pyr.append(torch.nn.functional.avg_pool2d(pyr[-1], 2, stride=2).contiguous())
coords = torch.empty(B, 2, Lv, d1, h, w, device=dev)
mx, ds = torch.empty(B, 1, h, w, device=dev), torch.empty(B, 1, D, h, w, device=dev)
p = ops._p
dd_, pp_, kk_ = depth.to(dev), poses.to(dev).reshape(B, 16).contiguous(), K.to(dev).reshape(B, 16).contiguous()  # kept alive
L.check(lib.mal_epipolar_coords(p(dd_), p(pp_), p(kk_), B, h, w, r, Lv, 1.3133, 8.0, p(coords), p(mx), p(ds), ops._stream()), "coords")
torch.cuda.synchronize()
print("coords range", float(coords.min()), float(coords.max()))
g_out = torch.randn(B, D, h, w, device=dev)
g_f1 = torch.zeros_like(f1d)
g_pyr = [torch.zeros_like(t) for t in pyr]
g_c = torch.zeros_like(coords)
ws = torch.empty(lib.mal_coord_sample_l1_bwd_workspace_bytes(B), dtype=torch.uint8, device=dev)
def run(want_c=True, want_f=True):
    args = (p(f1d), L.ptr_array([p(t) for t in pyr]), p(coords), p(g_out), B, Cn, h, w, Lv, d1, 1,
            p(g_f1) if want_f else None, L.ptr_array([p(t) if want_f else None for t in g_pyr]), p(g_c) if want_c else None,
            p(ws), ws.numel(), ops._stream())
    for _ in range(2):
        L.check(lib.mal_coord_sample_l1_bwd(*args), "bwd")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.mal_coord_sample_l1_bwd(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
import os
if os.environ.get("ONES"):
    g_out.fill_(1.0)
if os.environ.get("DD"):
    L.check(lib.mal_epipolar_coords(p(depth.to(dev)), p(poses.to(dev).reshape(B, 16).contiguous()), p(K.to(dev).reshape(B, 16).contiguous()),
                                    B, h, w, r, Lv, float(os.environ["DD"]), 8.0, p(coords), p(mx), p(ds), ops._stream()), "coords")
print("LDS planes (fixed point): features only %.2f ms, coords only %.2f ms, everything %.2f ms" % (run(False, True), run(True, False), run(True, True)))
# what the plane kernel's time is made of (mal_set_option("epi_probe"): results are wrong, timings are the point)
for bits, what in ((1, "cheap taps (no divisions / validity tests)"), (2, "no LDS adds"), (4, "no gathers"), (7, "none of the three")):
    lib.mal_set_option(b"epi_probe", bits)
    print("  features only, %-45s %.2f ms" % (what + ":", run(False, True)))
lib.mal_set_option(b"epi_probe", 0)
lib.mal_set_option(b"epi_bwd_planes", 0)
print("global atomics: everything %.2f ms" % run(True, True))
