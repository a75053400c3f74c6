"""Epipolar correlation lookup (N4, forward) at DualRefine's size: B=8, 128 channels, 48x160 (192x640 / 4), radius 8,
3 levels = 51 hypotheses per pixel.  HIP (mal_epipolar_coords + mal_coord_sample_l1 through mal_amd.epipolar) vs the CPU
checker (the reference's formulation, oracle/epi_oracle.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from mal_amd import epipolar
from oracle import epi_oracle as E
from oracle.gen_golden_epi import make_case
B, C, h, w, r, L = 8, 128, 48, 160, 8, 3
K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=3)
dev = torch.device("cuda:0")
args = SimpleNamespace(corr_radius=r, disable_pose_updates=True, gap_factor="depth", gap_factor_depth_ratio=8, num_levels=L)
R = epipolar.Reprojections(args).to(dev)
S = epipolar.CoordSampler(args)
g = [t.to(dev) for t in (K, depth, poses, f1, f2)]
with torch.no_grad():
    R._reg_intrinsics(g[0])
    S.register(g[3], g[4], num_levels=L)
    def hip():
        c, max_dx, ds = R.depth2epipolarcoords(g[2], g[1])
        return S(c, L, 1)
    for _ in range(3): hip()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): hip()
    torch.cuda.synchronize(); th = (time.perf_counter() - t) / 20
# ---- the same lookup forward + backward (the DEQ solver differentiates through it in training)
dg, pg = g[1].clone().requires_grad_(True), g[2].clone().requires_grad_(True)
f1g, f2g = g[3].clone().requires_grad_(True), g[4].clone().requires_grad_(True)
def hip_fb():
    for t_ in (dg, pg, f1g, f2g):
        t_.grad = None
    c, max_dx, ds = R.depth2epipolarcoords(pg, dg)
    S2 = epipolar.CoordSampler(args)
    S2.register(f1g, f2g, num_levels=L)
    S2(c, L, 1).sum().backward()
for _ in range(3): hip_fb()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): hip_fb()
torch.cuda.synchronize(); tfb = (time.perf_counter() - t) / 10
D = L * (2 * r + 1)
alg = B * h * w * (2 * C * 4 + D * 4 + 2 * D * 4 + 4)  # both feature maps once, the correlation, the coordinates

# ---- the pose refinement step of the same loop: depth2gradcoords + direct_align
g5 = torch.Generator().manual_seed(7)
f2s = (0.8 * f1 + 0.2 * f2).half().float()
src_w, tgt_w, weight = (0.5 + torch.rand(B, 1, h, w, generator=g5) for _ in range(3))
args.disable_fixed_pose_weight, args.robust_pose_loss = True, False
P = epipolar.PoseUpdate(args)
with torch.no_grad():
    P.compute_feat(g[3], f2s.to(dev))
    P.src_w, P.tgt_w = src_w.to(dev), tgt_w.to(dev)
    wd = weight.to(dev)
    def align():
        c_p, P2 = R.depth2gradcoords(g[2], g[1], g[0])
        return P.direct_align(g[2], g[0], c_p, P2, wd)
    for _ in range(3): align()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): align()
    torch.cuda.synchronize(); ta = (time.perf_counter() - t) / 20
# ---- the same step forward + backward (the last unrolled solver step differentiates through it), plain and robust
def align_fb(robust):
    args.robust_pose_loss = robust
    Pg = epipolar.PoseUpdate(args)
    lv = [t_.to(dev).clone().requires_grad_(True) for t_ in (poses, depth, f1, f2s, src_w, tgt_w, weight)]
    def run():
        for t_ in lv:
            t_.grad = None
        c_p, P2 = R.depth2gradcoords(lv[0], lv[1], g[0])
        Pg.compute_feat(lv[2], lv[3])
        Pg.src_w, Pg.tgt_w = lv[4], lv[5]
        new_poses, update = Pg.direct_align(lv[0], g[0], c_p, P2, lv[6])
        (new_poses.sum() + update.sum()).backward()
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10
tafb, tafb_r = align_fb(False), align_fb(True)
args.robust_pose_loss = False
with torch.no_grad():
    # the CPU checkers last: their OpenMP workers keep spinning for a while and would starve the launching thread
    t = time.perf_counter()
    rc, _, _ = E.depth2epipolarcoords(poses, depth, K, torch.tensor([1.0]), r=r, num_levels=L)
    E.coord_sample(f1, E.pyramid(f2, L), rc, L, 1)
    tc = time.perf_counter() - t
    t = time.perf_counter()
    c_p, P2 = E.depth2gradcoords(poses, depth, K)
    E.direct_align(poses, f1, f2s, src_w, tgt_w, K, c_p, P2, weight)
    tca = time.perf_counter() - t
print("HIP %.0f us (%.1f GB/s algorithmic; %d (pixel,hypothesis) pairs -> %.1f G pair-channels/s)   CPU checker %.2f s (%d threads)" %
      (th * 1e6, alg / th / 1e9, B * h * w * D, B * h * w * D * C / th / 1e9, tc, torch.get_num_threads()))

# What bounds these gather kernels is the vector-memory address path, not HBM: scripts/vmem_probe.hip measures, per CU,
# 5.4 cycles per dword wave-load whose 64 lanes fall in one 128-byte line (chip: 102 G wave-loads/s), 16 cycles over 2-4
# lines (37.5 G/s), 32 over 8 (18.8 G/s), 64 over 16 or more (9.5 G/s) -- all L1 hits (profiles/r02_vmem_probe.txt).
WAVE_LOAD_CEILING = {1: 102.0e9, 4: 37.5e9, 8: 18.8e9, 16: 9.5e9}
wave_loads = B * h * w * D * C * 4 / 64.0  # four taps per (pixel, hypothesis, channel), 64 pixels per wave
print("bound: vector-memory address path: %.1f M dword wave-loads -> %.1f G wave-loads/s = %.2f of the coalesced ceiling (%.0f G/s), "
      "%.2f of the ceiling for loads whose lanes spread over 4 cache lines (%.1f G/s): the hypotheses of neighbouring pixels "
      "lie several rows apart under these synthetic poses" %
      (wave_loads / 1e6, wave_loads / th / 1e9, wave_loads / th / WAVE_LOAD_CEILING[1], WAVE_LOAD_CEILING[1] / 1e9,
       wave_loads / th / WAVE_LOAD_CEILING[4], WAVE_LOAD_CEILING[4] / 1e9))
print("lookup forward + backward (VJPs w.r.t. depth, pose, delta, both feature maps; the %.2f G lane-adds of the scatter into the "
      "feature pyramid go to 64-bit fixed-point planes in LDS, one workgroup per (sample, channel)): HIP %.0f us" %
      (B * h * w * D * C * 4 / 1e9, tfb * 1e6))
print("direct_align: HIP %.0f us (gradcoords + normal equations + solve/se3 update kernel)   CPU checker %.2f s" % (ta * 1e6, tca))
print("direct_align forward + backward (VJPs w.r.t. poses, depth, both feature maps, the three weight maps; %.2f G lane-adds "
      "into the target features, LDS planes as above): HIP %.0f us, with --robust_pose_loss %.0f us" % (B * h * w * C * 20 / 1e9, tafb * 1e6, tafb_r * 1e6))

# ---- the second pose regime: a forward-moving camera (KITTI's dominant motion: translation along z, a little sideways, none
# vertically) keeps the epipolar lines of neighbouring pixels close to parallel and almost horizontal, so the 64 lanes of
# a wave load stay within one or two cache lines; the bench's default poses (0.15 in every direction) spread them over ~4
poses_fwd = poses.clone()
poses_fwd[:, 1, 3] = 0.0
poses_fwd[:, 0, 3] *= 0.2
poses_fwd[:, 2, 3] = 0.15
pf = poses_fwd.to(dev)
with torch.no_grad():
    def hip_f():
        c, max_dx, ds = R.depth2epipolarcoords(pf, g[1])
        return S(c, L, 1)
    for _ in range(3): hip_f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): hip_f()
    torch.cuda.synchronize(); thf = (time.perf_counter() - t) / 20
print("forward-motion poses (t = (0.03 N, 0, 0.15), same rotation): HIP %.0f us -> %.1f G wave-loads/s = %.2f of the coalesced ceiling, "
      "%.2f of the 4-line ceiling" % (thf * 1e6, wave_loads / thf / 1e9, wave_loads / thf / WAVE_LOAD_CEILING[1],
                                      wave_loads / thf / WAVE_LOAD_CEILING[4]))

# ---- round 4: two channel planes per workgroup in the lookup's backward (option "epi_bwd_planes" 2, default) against one
from mal_amd import _lib
lib = _lib.load()
for mode in (1, 2, 1, 2):
    lib.mal_set_option(b"epi_bwd_planes", mode)
    for _ in range(3): hip_fb()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): hip_fb()
    torch.cuda.synchronize()
    print("lookup forward + backward, epi_bwd_planes=%d (%s): %.0f us" %
          (mode, "two channel planes per workgroup, 1024 threads" if mode == 2 else "one plane per workgroup, 512 threads",
           (time.perf_counter() - t) / 10 * 1e6))
lib.mal_set_option(b"epi_bwd_planes", 2)
