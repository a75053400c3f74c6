"""GPU parity of the epipolar correlation lookup (mal_amd.epipolar -> mal_epipolar_coords / mal_coord_sample_l1) against
the golden run of the reference's own Reprojections / CoordSampler (tests/golden/epi_*.npz) and, at DualRefine's size
(B=8, 128 channels, 48x160, radius 8, 3 levels), against the CPU oracle.  fp32, tolerance 1e-4 (north_star): the pose
product and the bilinear blend are summed in a different order than ATen's; a sample whose position lies within 1e-4 px
of a tap boundary may take the neighbouring taps (the blend is continuous there, so the value still agrees)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.test_epi_oracle import CASES, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _args(r, L):
    return SimpleNamespace(corr_radius=r, disable_pose_updates=True, gap_factor="depth", gap_factor_depth_ratio=8, num_levels=L)


def run(K, depth, poses, f1, f2, r, L, heads, delta):
    from mal_amd import epipolar
    d = lambda t: t.to(DEV)
    R = epipolar.Reprojections(_args(r, L)).to(DEV)
    with torch.no_grad():
        R.delta.fill_(float(delta))
        R._reg_intrinsics(d(K))
        c, max_dx, ds = R.depth2epipolarcoords(d(poses), d(depth))
        S = epipolar.CoordSampler(_args(r, L))
        S.register(d(f1), d(f2), num_levels=L)
        corr = S(c, L, heads)
    return c.cpu(), max_dx.cpu(), ds.cpu(), corr.cpu()


def check(got, ref):
    c, max_dx, ds, corr = got
    rc, rmax, rds, rcorr = ref
    assert torch.allclose(ds, rds, rtol=1e-6, atol=1e-6)
    assert torch.allclose(max_dx, rmax, rtol=1e-6, atol=1e-7)
    # coordinates in pixels: 1e-4 relative to the image size
    assert (c - rc).abs().max() <= 1e-4 * max(1.0, float(rc.abs().max()))
    assert (corr - rcorr).abs().max() <= 1e-4 * max(1.0, float(rcorr.abs().max()))


@pytest.mark.parametrize("tag", CASES)
def test_golden(tag):
    z, K, depth, poses, f1, f2, r, L, heads, delta = load(tag)
    t = lambda k: torch.from_numpy(z[k])
    check(run(K, depth, poses, f1, f2, r, L, heads, float(delta)),
          (t("out/coords"), t("out/max_dx"), t("out/depths"), t("out/corr")))


def test_dualrefine_size_against_the_cpu_checker():
    from oracle import epi_oracle as E
    from oracle.gen_golden_epi import make_case
    B, C, h, w, r, L = 8, 128, 48, 160, 8, 3
    K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=3)
    delta = torch.tensor([0.7])
    with torch.no_grad():
        rc, rmax, rds = E.depth2epipolarcoords(poses, depth, K, delta, r=r, num_levels=L)
        rcorr = E.coord_sample(f1, E.pyramid(f2, L), rc, L, 1)
    check(run(K, depth, poses, f1, f2, r, L, 1, 0.7), (rc, rmax, rds, rcorr))


def test_forward_only_is_enforced_where_no_vjp_exists():
    """the lookup is differentiable; the pose-refinement step is not yet and refuses tensors that require grad"""
    from mal_amd import epipolar, _lib
    R = epipolar.Reprojections(_args(2, 1)).to(DEV)
    K = torch.eye(4, device=DEV).repeat(1, 1, 1)
    R._reg_intrinsics(K)
    with pytest.raises(_lib.MalError):
        R.depth2gradcoords(torch.eye(4, device=DEV)[None], torch.ones(1, 1, 8, 8, device=DEV, requires_grad=True))


def _align_case(i):
    from mal_amd import epipolar
    d = lambda t: t.to(DEV)
    a = SimpleNamespace(corr_radius=2, disable_pose_updates=False, gap_factor="depth", gap_factor_depth_ratio=8, num_levels=1,
                        disable_fixed_pose_weight=True, robust_pose_loss=False)
    R = epipolar.Reprojections(a).to(DEV)
    P = epipolar.PoseUpdate(a)
    with torch.no_grad():
        R._reg_intrinsics(d(i["K"]))
        c_p, P2 = R.depth2gradcoords(d(i["poses"]), d(i["depth"]), d(i["K"]))
        P.compute_feat(d(i["f1"]), d(i["f2"]))
        P.src_w, P.tgt_w = d(i["src_w"]), d(i["tgt_w"])
        H, b = P.normal_equations(d(i["K"]), c_p, P2, d(i["weight"]))
        new_poses, update = P.direct_align(d(i["poses"]), d(i["K"]), c_p, P2, d(i["weight"]))
    return c_p.cpu(), P2.cpu(), H.cpu(), b.cpu(), new_poses.cpu(), update.cpu()


def _check_align(i, got, ref_cp, ref_P2, ref_new, ref_update):
    from oracle import epi_oracle as E
    c_p, P2, H, b, new_poses, update = got
    assert (c_p - ref_cp).abs().max() <= 1e-4 * max(1.0, float(ref_cp.abs().max()))
    assert torch.allclose(P2, ref_P2, rtol=1e-5, atol=1e-5)
    rH, rb = E.normal_equations(i["f1"], i["f2"], i["src_w"], i["tgt_w"], i["K"], ref_cp, ref_P2, i["weight"])
    assert (H - rH).abs().max() <= 1e-4 * float(rH.abs().max()) and (b - rb).abs().max() <= 1e-4 * float(rb.abs().max())
    # the update solves a 6x6 system: its error is the normal equations' times the conditioning
    assert (update - ref_update).abs().max() <= 2e-3 * max(1e-3, float(ref_update.abs().max()))
    assert (new_poses - ref_new).abs().max() <= 2e-3


@pytest.mark.parametrize("tag", ["epi_align_b2_c16_12x20_r4_l3", "epi_align_b1_c8_9x13_r2_l2_h2"])
def test_direct_align_golden(tag):
    from tests.test_epi_oracle import load_align
    z, i = load_align(tag)
    t = lambda k: torch.from_numpy(z[k])
    _check_align(i, _align_case(i), t("out/c_p"), t("out/P2"), t("out/new_poses"), t("out/update"))


def test_direct_align_dualrefine_size():
    from oracle import epi_oracle as E
    from oracle.gen_golden_epi import make_case
    B, C, h, w = 8, 128, 48, 160
    K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=4, trans=0.05)
    g = torch.Generator().manual_seed(7)
    i = dict(K=K, depth=depth, poses=poses, f1=f1, f2=(0.8 * f1 + 0.2 * f2).half().float(),
             src_w=0.5 + torch.rand(B, 1, h, w, generator=g), tgt_w=0.5 + torch.rand(B, 1, h, w, generator=g),
             weight=0.5 + torch.rand(B, 1, h, w, generator=g))
    with torch.no_grad():
        c_p, P2 = E.depth2gradcoords(poses, depth, K)
        new_poses, update = E.direct_align(poses, i["f1"], i["f2"], i["src_w"], i["tgt_w"], K, c_p, P2, i["weight"])
    _check_align(i, _align_case(i), c_p, P2, new_poses, update)


def test_depthbins_lookup_golden():
    """the masking lookup of depth_pose.py:561-580: depthbins2coords (both branches) + CoordSampler.__corr__"""
    from tests.test_epi_oracle import BINS_CASES, load_bins
    from mal_amd import epipolar
    for tag in BINS_CASES:
        z, K, depth, poses, f1, f2, (dmin, dmax, bmin, bmax) = load_bins(tag)
        d = lambda t: t.to(DEV)
        for name, flag in (("lin", False), ("bins", True)):
            a = SimpleNamespace(corr_radius=2, disable_pose_updates=True, gap_factor="depth", gap_factor_depth_ratio=8,
                                num_levels=2, min_depth=dmin, max_depth=dmax, use_depth_bins_for_masking=flag)
            R = epipolar.Reprojections(a).to(DEV)
            S = epipolar.CoordSampler(a)
            with torch.no_grad():
                R._reg_intrinsics(d(K))
                R.update_depth_bins(bmax, bmin, 4.0, 4.0)
                S.register(d(f1), d(f2), num_levels=2)
                c0, ds0 = R.depthbins2coords(d(poses), d(depth))
                corr0 = S.__corr__(c0)
            rc, rd, rcorr = (torch.from_numpy(z[k + name]) for k in ("out/c0_", "out/ds0_", "out/corr0_"))
            assert torch.allclose(ds0.cpu(), rd, rtol=1e-6, atol=1e-6)
            # far hypotheses project far outside the image: compare where the reference coordinate is near the image
            near = (rc.abs() < 1e4)
            assert ((c0.cpu() - rc).abs()[near] <= 1e-4 * rc.abs()[near].clamp(min=1.0)).all()
            assert (corr0.cpu() - rcorr).abs().max() <= 1e-4 * max(1.0, float(rcorr.abs().max()))


# ------------------------------------------------------------------ VJPs of the lookup (round 2)
def run_grads(K, depth, poses, f1, f2, r, L, heads, delta, w_corr, w_ds, w_mx):
    from mal_amd import epipolar
    d = lambda t: t.to(DEV)
    R = epipolar.Reprojections(_args(r, L)).to(DEV)
    with torch.no_grad():
        R.delta.fill_(float(delta))
    R._reg_intrinsics(d(K))
    dg, pg = d(depth).clone().requires_grad_(True), d(poses).clone().requires_grad_(True)
    f1g, f2g = d(f1).clone().requires_grad_(True), d(f2).clone().requires_grad_(True)
    c, max_dx, ds = R.depth2epipolarcoords(pg, dg)
    S = epipolar.CoordSampler(_args(r, L))
    S.register(f1g, f2g, num_levels=L)
    corr = S(c, L, heads)
    ((corr * d(w_corr)).sum() + (ds * d(w_ds)).sum() + (max_dx * d(w_mx)).sum()).backward()
    torch.cuda.synchronize()
    return {"depth": dg.grad.cpu(), "poses": pg.grad.cpu(), "delta": R.delta.grad.cpu(), "f1": f1g.grad.cpu(), "f2": f2g.grad.cpu()}


def check_grads(got, ref, n_pix):
    for k, r in ref.items():
        g = got[k].reshape(r.shape)
        sc = float(r.abs().max())
        if k in ("poses", "delta"):  # sums over all pixels and hypotheses
            assert float((g - r).abs().max()) <= 1e-4 * sc + 1e-6, (k, float((g - r).abs().max()), sc)
        else:
            # a sample within rounding distance of a tap boundary takes the neighbouring taps (value continuous, slope
            # not): a handful of elements, the rest at 1e-4 of the map's scale
            bad = ((g - r).abs() > 1e-4 * sc).float().mean().item()
            assert bad <= 2e-4 + 4.0 / g.numel(), (k, bad)
            assert float(np.linalg.norm((g - r).numpy().ravel()) / np.linalg.norm(r.numpy().ravel())) <= 2e-3, k


@pytest.mark.parametrize("tag", ["epi_grad_b2_c16_12x20_r4_l3", "epi_grad_b1_c8_9x13_r2_l2_h2"])
def test_lookup_vjp_golden(tag):
    """gradients of the lookup against those autograd takes through the reference's own classes (the fixture)"""
    import os
    from tests.test_epi_oracle import GOLDEN
    z, K, depth, poses, f1, f2, r, L, heads, delta = load(tag.replace("epi_grad_", "epi_"))
    zg = np.load(os.path.join(GOLDEN, tag + ".npz"))
    t = lambda k: torch.from_numpy(zg[k])
    got = run_grads(K, depth, poses, f1, f2, r, L, heads, float(delta), t("in/w_corr"), t("in/w_ds"), t("in/w_mx"))
    check_grads(got, {k: t("grad/" + k) for k in ("depth", "poses", "delta", "f1", "f2")}, depth.numel())


def test_lookup_vjp_dualrefine_size():
    """B=8, 128 channels, 48x160, radius 8, 3 levels: against autograd through the CPU checker"""
    from oracle.gen_golden_epi import make_case
    from tests.test_epi_oracle import oracle_lookup_grads
    B, C, h, w, r, L, heads = 8, 128, 48, 160, 8, 3, 1
    K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=31)
    g = torch.Generator().manual_seed(32)
    D = L * (2 * r + 1)
    w_corr, w_ds, w_mx = torch.randn(B, D * heads, h, w, generator=g), 0.1 * torch.randn(B, 1, D, h, w, generator=g), torch.randn(B, 1, h, w, generator=g)
    delta = torch.tensor([0.7])
    ref = oracle_lookup_grads(K, depth, poses, f1, f2, r, L, heads, delta, w_corr, w_ds, w_mx)
    got = run_grads(K, depth, poses, f1, f2, r, L, heads, 0.7, w_corr, w_ds, w_mx)
    check_grads(got, ref, depth.numel())
