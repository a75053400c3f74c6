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
    """the lookup and the pose-refinement step are differentiable; the masking lookup (run under no_grad upstream,
    depth_pose.py:522) refuses tensors that require grad instead of silently detaching them"""
    from mal_amd import epipolar, _lib
    a = _args(2, 1)
    a.use_depth_bins_for_masking, a.min_depth, a.max_depth = False, 0.1, 100.0
    R = epipolar.Reprojections(a).to(DEV)
    R._reg_intrinsics(torch.eye(4, device=DEV).repeat(1, 1, 1))
    with pytest.raises(_lib.MalError):
        R.depthbins2coords(torch.eye(4, device=DEV)[None], torch.ones(1, 1, 8, 8, device=DEV, requires_grad=True))
    c_p, P2 = R.depth2gradcoords(torch.eye(4, device=DEV)[None], torch.ones(1, 1, 8, 8, device=DEV, requires_grad=True))
    assert c_p.requires_grad and P2.requires_grad


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


# ---------------------------------------------------------------- VJP of the pose-refinement step
def _align_objects(robust):
    from mal_amd import epipolar
    a = SimpleNamespace(corr_radius=2, disable_pose_updates=False, gap_factor="depth", gap_factor_depth_ratio=8, num_levels=1,
                        disable_fixed_pose_weight=True, robust_pose_loss=robust)
    return epipolar.Reprojections(a).to(DEV), epipolar.PoseUpdate(a)


def run_align_grads(i, Wn, Wu, robust):
    """depth2gradcoords + direct_align on the device, backward of sum(new_poses Wn) + sum(update Wu)"""
    from tests.test_epi_oracle import ALIGN_LEAVES
    R, P = _align_objects(robust)
    d = lambda t: t.to(DEV)
    lv = {k: d(i[k]).clone().requires_grad_(True) for k in ALIGN_LEAVES}
    R._reg_intrinsics(d(i["K"]))
    c_p, P2 = R.depth2gradcoords(lv["poses"], lv["depth"], d(i["K"]))
    P.compute_feat(lv["f1"], lv["f2"])
    P.src_w, P.tgt_w = lv["src_w"], lv["tgt_w"]
    new_poses, update = P.direct_align(lv["poses"], d(i["K"]), c_p, P2, lv["weight"])
    ((new_poses * d(Wn)).sum() + (update * d(Wu)).sum()).backward()
    torch.cuda.synchronize()
    return new_poses.detach().cpu(), update.detach().cpu(), {k: v.grad.cpu() for k, v in lv.items()}


def _check_maps(got, ref, tol, what):
    for k, r in ref.items():
        g = got[k].reshape(r.shape)
        sc = float(r.abs().max())
        if g.numel() <= 64:
            assert float((g - r).abs().max()) <= tol * sc + 1e-7, (what, k, float((g - r).abs().max()), sc)
        else:  # a sample within rounding distance of a tap boundary takes the neighbouring taps: a handful of elements
            bad = ((g - r).abs() > tol * sc).float().mean().item()
            assert bad <= 2e-4 + 4.0 / g.numel(), (what, k, bad)
            assert float(np.linalg.norm((g - r).numpy().ravel()) / (np.linalg.norm(r.numpy().ravel()) + 1e-30)) <= 20 * tol, (what, k)


@pytest.mark.parametrize("tag", [t + r for t in ("epi_aligngrad_b2_c16_12x20_r4_l3", "epi_aligngrad_b1_c8_9x13_r2_l2_h2")
                                 for r in ("", "_robust")])
def test_direct_align_vjp_golden(tag):
    """end to end against the gradients autograd takes through the reference's own classes (the fixture).  The chain holds
    a 6x6 solve: errors are the pieces' (1e-4, next test) times its conditioning"""
    from tests.test_epi_oracle import load_aligngrad, ALIGN_LEAVES
    i, g, robust = load_aligngrad(tag)
    new_poses, update, grads = run_align_grads(i, g["in/Wn"], g["in/Wu"], robust)
    assert (update - g["out/update"]).abs().max() <= 2e-3 * max(1e-3, float(g["out/update"].abs().max()))
    assert (new_poses - g["out/new_poses"]).abs().max() <= 2e-3
    _check_maps(grads, {k: g["grad/" + k] for k in ALIGN_LEAVES}, 5e-3, tag)


@pytest.mark.parametrize("robust", [False, True], ids=["plain", "robust_pose_loss"])
def test_direct_align_vjp_pieces_dualrefine_size(robust):
    """B=8, 128 channels, 48x160: each of the three VJPs on its own against autograd through the CPU checker, 1e-4"""
    from mal_amd import epipolar
    from oracle import epi_oracle as E
    from oracle.gen_golden_epi import make_case
    B, C, h, w = 8, 128, 48, 160
    K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=4, trans=0.05)
    g = torch.Generator().manual_seed(17)
    rnd = lambda *s: torch.randn(*s, generator=g)
    f2 = (0.8 * f1 + 0.2 * f2).half().float()
    src_w, tgt_w, weight = (0.5 + torch.rand(B, 1, h, w, generator=g) for _ in range(3))
    d = lambda t: t.to(DEV)
    leaf = lambda t, dev="cpu": t.to(dev).clone().requires_grad_(True)
    # ---- depth2gradcoords
    w_cp, w_P2 = rnd(B, 2, 1, 5, h, w), rnd(B, 4, h * w)
    ol = dict(depth=leaf(depth), poses=leaf(poses))
    c_p, P2 = E.depth2gradcoords(ol["poses"], ol["depth"], K)
    ((c_p * w_cp).sum() + (P2 * w_P2).sum()).backward()
    R, P = _align_objects(robust)
    R._reg_intrinsics(d(K))
    hl = dict(depth=leaf(depth, DEV), poses=leaf(poses, DEV))
    hc, hP = R.depth2gradcoords(hl["poses"], hl["depth"], d(K))
    ((hc * d(w_cp)).sum() + (hP * d(w_P2)).sum()).backward()
    _check_maps({k: v.grad.cpu() for k, v in hl.items()}, {k: v.grad for k, v in ol.items()}, 1e-4, "gradcoords")
    # ---- normal equations (same p2 / P2 on both sides)
    c_p, P2 = c_p.detach(), P2.detach()
    g_H, g_b = rnd(B, 6, 6), rnd(B, 6)
    names = ("f1", "f2", "src_w", "tgt_w", "weight", "p2", "P2")
    vals = (f1, f2, src_w, tgt_w, weight, c_p, P2)
    ol = {k: leaf(v) for k, v in zip(names, vals)}
    H, b = E.normal_equations(ol["f1"], ol["f2"], ol["src_w"], ol["tgt_w"], K, ol["p2"], ol["P2"], ol["weight"], robust=robust)
    ((H * g_H).sum() + (b * g_b).sum()).backward()
    hl = {k: leaf(v, DEV) for k, v in zip(names, vals)}
    P.compute_feat(hl["f1"], hl["f2"])
    P.src_w, P.tgt_w = hl["src_w"], hl["tgt_w"]
    hH, hb = P.normal_equations(d(K), hl["p2"], hl["P2"], hl["weight"])
    assert (hH.detach().cpu() - H.detach()).abs().max() <= 1e-4 * float(H.detach().abs().max())
    assert (hb.detach().cpu() - b.detach()).abs().max() <= 1e-4 * float(b.detach().abs().max())
    ((hH * d(g_H)).sum() + (hb * d(g_b)).sum()).backward()
    _check_maps({k: v.grad.cpu() for k, v in hl.items()}, {k: v.grad for k, v in ol.items()}, 1e-4, "normal_equations")
    # ---- solve + se3_exp + pose product (H, b from the checker; torch.linalg.cholesky's backward symmetrises d/dH)
    H0, b0 = H.detach(), b.detach()
    g_new, g_up = rnd(B, 4, 4), rnd(B, 6, 1)
    ol = dict(H=leaf(H0), b=leaf(b0), poses=leaf(poses))
    Lc = torch.linalg.cholesky(ol["H"])
    up = torch.cholesky_solve(ol["b"][..., None], Lc)
    new = torch.bmm(E.se3_exp(up), ol["poses"])
    ((new * g_new).sum() + (up * g_up).sum()).backward()
    hl = dict(H=leaf(H0, DEV), b=leaf(b0, DEV), poses=leaf(poses, DEV))
    hn, hu = epipolar.AlignUpdateFn.apply(hl["H"], hl["b"], hl["poses"])
    ((hn * d(g_new)).sum() + (hu * d(g_up)).sum()).backward()
    assert (hu.detach().cpu() - up.detach()).abs().max() <= 2e-3 * max(1e-3, float(up.abs().max()))
    ref = {k: v.grad for k, v in ol.items()}
    ref["H"] = 0.5 * (ref["H"] + ref["H"].transpose(1, 2))
    got = {k: v.grad.cpu() for k, v in hl.items()}
    for k in ref:  # per sample: the conditioning of the 6x6 systems differs
        for s in range(B):
            sc = float(ref[k][s].abs().max())
            assert float((got[k][s] - ref[k][s]).abs().max()) <= 2e-3 * sc + 1e-7, (k, s)


def test_non_finite_cotangent_poisons_the_feature_gradient():
    """the feature cotangents are accumulated in fixed point scaled by the largest |cotangent| of the sample: a NaN / Inf
    there once turned the scale into zero and the result into silent zeros.  Upstream's autograd (and the float-atomic path)
    carry the non-finite value into the result: the sample whose cotangent holds it must come back non-finite, the other
    sample untouched."""
    from tests.test_epi_oracle import GOLDEN
    import os
    tag = "epi_grad_b2_c16_12x20_r4_l3"
    z, K, depth, poses, f1, f2, r, L, heads, delta = load(tag.replace("epi_grad_", "epi_"))
    zg = np.load(os.path.join(GOLDEN, tag + ".npz"))
    t = lambda k: torch.from_numpy(zg[k])
    clean = run_grads(K, depth, poses, f1, f2, r, L, heads, float(delta), t("in/w_corr"), t("in/w_ds"), t("in/w_mx"))
    for bad_value in (float("nan"), float("inf")):
        w = t("in/w_corr").clone()
        w[1, 3, 2, 5] = bad_value
        got = run_grads(K, depth, poses, f1, f2, r, L, heads, float(delta), w, t("in/w_ds"), t("in/w_mx"))
        assert not torch.isfinite(got["f2"][1]).all(), bad_value          # not silently zero / finite garbage
        assert torch.isfinite(got["f2"][0]).all() and torch.equal(got["f2"][0], clean["f2"][0])
