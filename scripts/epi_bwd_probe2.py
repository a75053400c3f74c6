"""the lookup's backward through autograd, timed per call with events (what bench_epipolar.py runs), vs the direct C call"""
import sys, torch, time
sys.path.insert(0, ".")
from types import SimpleNamespace
from mal_amd import epipolar, _lib as L, ops
from oracle.gen_golden_epi import make_case
B, C, h, w, r, Lv = 8, 128, 48, 160, 8, 3
K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=3)
dev = torch.device("cuda:0")
args = SimpleNamespace(corr_radius=r, disable_pose_updates=True, gap_factor="depth", gap_factor_depth_ratio=8, num_levels=Lv)
R = epipolar.Reprojections(args).to(dev)
g = [t.to(dev) for t in (K, depth, poses, f1, f2)]
R._reg_intrinsics(g[0])
dg, pg = g[1].clone().requires_grad_(True), g[2].clone().requires_grad_(True)
f1g, f2g = g[3].clone().requires_grad_(True), g[4].clone().requires_grad_(True)
def ev():
    return torch.cuda.Event(enable_timing=True)
for it in range(4):
    for t_ in (dg, pg, f1g, f2g):
        t_.grad = None
    c, max_dx, ds = R.depth2epipolarcoords(pg, dg)
    S2 = epipolar.CoordSampler(args)
    S2.register(f1g, f2g, num_levels=Lv)
    out = S2(c, Lv, 1)
    loss = out.sum() if it % 2 == 0 else (out * torch.randn_like(out)).sum()
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    loss.backward()
    e1.record()
    torch.cuda.synchronize()
    print("iter %d (%s cotangent): backward %.2f ms; coords range [%.1f, %.1f], nan %d" % (
        it, "ones" if it % 2 == 0 else "randn", e0.elapsed_time(e1), float(c.min()), float(c.max()), int(torch.isnan(c).sum())))
