"""GPU parity of the module-level drop-ins (manydepth/layers.py names) against the golden
vectors produced by the reference's own layer objects (tests/golden/layers_*.npz)."""
import numpy as np
import pytest
import torch

from tests import golden_io as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _c(a):
    return torch.from_numpy(np.array(a)).to(DEV)


def _l2rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def _sample_ok(grid, H, W, ac, tol=1e-3):
    """pixels whose sampling position is not within tol of an integer / the border clip"""
    g = grid.astype(np.float64)
    if ac:
        ix, iy = (g[..., 0] + 1) / 2 * (W - 1), (g[..., 1] + 1) / 2 * (H - 1)
    else:
        ix, iy = ((g[..., 0] + 1) * W - 1) / 2, ((g[..., 1] + 1) * H - 1) / 2
    ok = np.ones(ix.shape, bool)
    for v, hi in ((ix, W - 1), (iy, H - 1)):
        ok &= np.abs(v - np.round(v)) > tol
        ok &= (v > tol) & (v < hi - tol) | (v < -tol) | (v > hi + tol)
    return ok[:, None]


@pytest.mark.parametrize("tag", G.LAYER_CASES)
def test_geometry_modules(tag):
    from mal_amd import layers
    z = G.load(tag)
    b = {k: v.to(DEV) for k, v in G.batch_from_golden(z).items() if torch.is_tensor(v)}
    B, _, H, W = b["color0"].shape
    disp = b["disp_teacher"].clone().requires_grad_(True)
    sd, depth = layers.disp_to_depth(disp, 0.1, 100.0)
    G.assert_close(sd.detach().cpu(), z["scaled_disp"], 1e-6), G.assert_close(depth.detach().cpu(), z["depth"], 1e-6)
    for inv in (False, True):
        T = layers.transformation_from_parameters(b["axisangle_m1"], b["translation_m1"], invert=inv)
        G.assert_close(T.cpu(), z["T_inv%d" % inv], 1e-5, "T", floor=1e-3)
    G.assert_close(layers.rot_from_axisangle(b["axisangle_p1"]).cpu(), z["rot"], 1e-5, floor=1e-3)
    G.assert_close(layers.get_translation_matrix(b["translation_p1"]).cpu(), z["trans"], 0)
    T = _c(z["T_inv1"]).requires_grad_(True)
    pts = layers.BackprojectDepth(B, H, W)(depth, b["inv_K"])
    G.assert_close(pts.detach().cpu(), z["cam_points"], 1e-6, "cam_points", floor=1e-4)
    grid, zc = layers.Project3D(B, H, W, dc=True)(pts, b["K"], T)
    G.assert_close(grid.detach().cpu(), z["grid_A"], 1e-5, "grid_A", floor=1e-1)
    G.assert_close(zc.detach().cpu(), z["proj_depth"], 1e-5, "proj_depth", floor=1e-3)
    warped = layers.grid_sample(b["color_m1"], grid, padding_mode="border", align_corners=True)
    G.assert_close(warped.detach().cpu(), z["warped_A"], 1e-4, "warped_A", floor=1e-1)
    (warped * _c(z["in/g_warped"])).sum().backward()
    # against the reference's own gradients AWAY from the pixels where its taps may differ from the kernel's; every pixel
    # and the pose gradient are held at 1e-4 with the taps forced in test_warp_gradient_with_forced_taps below
    ok = _sample_ok(z["grid_A"], H, W, True)
    sc = np.abs(z["grad_disp_A"]).max()
    assert (np.abs(disp.grad.cpu().numpy() - z["grad_disp_A"])[ok] > 1e-4 * sc).mean() <= 1e-3
    # DualRefine convention
    disp2 = b["disp_teacher"].clone().requires_grad_(True)
    T2 = _c(z["T_inv1"]).requires_grad_(True)
    pts2 = layers.BackprojectDepth(B, H, W)(layers.disp_to_depth(disp2, 0.1, 100.0)[1], b["inv_K"])
    grid2 = layers.Project3DDualRefine(B, H, W)(pts2, b["K"], T2)
    G.assert_close(grid2.detach().cpu(), z["grid_B"], 1e-5, "grid_B", floor=1e-1)
    warped2 = layers.grid_sample(b["color_m1"], grid2, padding_mode="border", align_corners=False)
    G.assert_close(warped2.detach().cpu(), z["warped_B"], 1e-4, "warped_B", floor=1e-1)
    (warped2 * _c(z["in/g_warped"])).sum().backward()
    ok = _sample_ok(z["grid_B"], H, W, False)
    sc = np.abs(z["grad_disp_B"]).max()
    assert (np.abs(disp2.grad.cpu().numpy() - z["grad_disp_B"])[ok] > 1e-4 * sc).mean() <= 1e-3


@pytest.mark.parametrize("tag", G.LAYER_CASES)
def test_photometric_modules(tag):
    from mal_amd import layers, loss_utils
    z = G.load(tag)
    b = {k: v.to(DEV) for k, v in G.batch_from_golden(z).items() if torch.is_tensor(v)}
    x = _c(z["warped_A"]).requires_grad_(True)
    y = b["color0"].clone().requires_grad_(True)
    ssim = layers.SSIM()
    s = ssim(x, y)
    d = np.abs(s.detach().cpu().numpy() - z["ssim"])
    assert d.max() <= 1e-4 and (d > 2e-5 + 1e-4 * z["ssim"]).mean() <= 1e-3
    (s * _c(z["in/g_ssim"])).sum().backward()
    # SSIM's gradient inherits the sigma cancellation: compare at the scale of the map
    assert _l2rel(x.grad.cpu().numpy(), z["grad_ssim_x"]) <= 2e-4
    assert _l2rel(y.grad.cpu().numpy(), z["grad_ssim_y"]) <= 2e-4
    x2 = _c(z["warped_A"]).requires_grad_(True)
    r = loss_utils.compute_reprojection_loss(ssim, x2, b["color0"])
    d = np.abs(r.detach().cpu().numpy() - z["reproj"])
    assert d.max() <= 1e-4
    (r * _c(z["in/g_reproj"])).sum().backward()
    assert _l2rel(x2.grad.cpu().numpy(), z["grad_reproj_pred"]) <= 2e-4
    ident = loss_utils.compute_reprojection_loss(ssim, b["color_m1"], b["color0"])
    assert np.abs(ident.cpu().numpy() - z["identity_reproj"]).max() <= 1e-4
    m = loss_utils.compute_loss_masks(_c(z["reproj"]), _c(z["identity_reproj"]))
    G.assert_close(m.cpu(), z["automask"], 0)
    G.assert_close(loss_utils.compute_loss_masks(_c(z["reproj"]), None).cpu(), z["automask_none"], 0)
    disp3 = b["disp_student"].clone().requires_grad_(True)
    sm = layers.get_smooth_loss(disp3, b["color0"])
    sm.backward()
    assert abs(float(sm) - float(z["smooth"])) <= 1e-5 * abs(float(z["smooth"]))
    g, r = disp3.grad.cpu().numpy(), z["grad_smooth"]
    # sign(d_q - d_q') of two fp32 numbers is exact (0 at a tie, as torch.abs's gradient): every pixel, nothing exempted
    assert np.abs(g - r).max() <= 1e-4 * np.abs(r).max()


@pytest.mark.parametrize("invert", [False, True])
def test_pose_kernel_forward_backward(invert):
    """mal_pose_fwd/bwd (a16) against the oracle's tensor-op restatement of layers.py:26-100."""
    from mal_amd import layers
    from oracle import mal_oracle as O
    torch.manual_seed(11)
    aa, tr, w = 0.3 * torch.randn(5, 1, 3), 0.5 * torch.randn(5, 1, 3), torch.randn(5, 4, 4)
    aa[0] = 0.01 * aa[0]  # the small-angle regime the pose decoder lives in (pose_decoder.py:47)
    a0, t0 = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    Tc = O.transformation_from_parameters(a0, t0, invert)
    (Tc * w).sum().backward()
    a1, t1 = aa.to(DEV).requires_grad_(True), tr.to(DEV).requires_grad_(True)
    Tg = layers.transformation_from_parameters(a1, t1, invert)
    (Tg * w.to(DEV)).sum().backward()
    assert torch.allclose(Tg.detach().cpu(), Tc.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(a1.grad.cpu(), a0.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(t1.grad.cpu(), t0.grad, rtol=1e-4, atol=1e-5)


def test_identity_min_and_fused_ensemble_agree():
    """generate_images_pred_ensemble (fused, no grad) == warp + reprojection + min on explicit images."""
    from mal_amd import layers, loss_utils, trainer, ops
    z = G.load("step_b3_37x50_distil")
    b = {k: v.to(DEV) for k, v in G.batch_from_golden(z).items() if torch.is_tensor(v)}
    B, _, H, W = b["color0"].shape
    opt = trainer.default_options(height=H, width=W, batch_size=B)
    lp = trainer.LossPath(opt)
    inputs = {("color", 0, 0): b["color0"], ("color", -1, 0): b["color_m1"], ("color", 1, 0): b["color_p1"],
              ("K", 0): b["K"], ("inv_K", 0): b["inv_K"]}
    T0 = layers.transformation_from_parameters(b["axisangle_m1"], b["translation_m1"], True)
    T1 = layers.transformation_from_parameters(b["axisangle_p1"], b["translation_p1"], False)
    fused = lp.generate_images_pred_ensemble(inputs, T0, T1, b["disp_teacher"])
    _, _, warped = ops.warp_fwd(b["disp_teacher"], b["K"], b["inv_K"], [T0, T1], [b["color_m1"], b["color_p1"]], 0.1,
                                100.0, 1e-7, 0, want_depth=False, want_grid=False)
    r = torch.minimum(loss_utils.compute_reprojection_loss(None, warped[0], b["color0"]),
                      loss_utils.compute_reprojection_loss(None, warped[1], b["color0"]))
    # same arithmetic up to the association of the 3x3 window sums (the fused kernel sums
    # horizontally then vertically); SSIM's sigma cancellation turns that into <= ~1e-4 abs
    assert (fused - r).abs().max().item() <= 1e-4
    from mal_amd import _lib
    lib = _lib.load()
    if not lib.mal_build_has_experiments():  # the default library holds the marching formulation only
        assert lib.mal_set_option(b"pass_impl", 0) != 0 and lib.mal_set_option(b"pass_impl", 1) == 0
        return
    outs = []
    for impl in (0, 1, 2):  # MAL_EXPERIMENTS build: the three formulations of the fused pass agree with each other
        assert lib.mal_set_option(b"pass_impl", impl) == 0
        outs.append(lp.generate_images_pred_ensemble(inputs, T0, T1, b["disp_teacher"]))
    lib.mal_set_option(b"pass_impl", 1)
    assert (outs[0] - r).abs().max().item() <= 2e-6  # the LDS-tiled v1 keeps ATen's row-major sum order
    assert (outs[1] - outs[0]).abs().max().item() <= 1e-4 and (outs[2] - outs[0]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("fuse,avg,shape", [(True, False, (2, 40, 72)), (False, False, (2, 40, 72)),
                                            (False, True, (2, 40, 72)), (True, False, (8, 192, 640))],
                         ids=["fused", "explicit", "explicit-avg", "fused-b8_192x640"])
def test_dualrefine_loss_path(fuse, avg, shape):
    """a17: DualRefine's per-(scale, deq_iter) loops (dualrefine/trainer.py:395-451,530-633) with the
    align_corners=False convention, against the oracle's restatement of the same lines; the last case is
    BASELINE.json configs[4] (B=8 192x640)."""
    from mal_amd import dualrefine, layers
    from mal_amd.synthetic import make_batch
    from oracle import mal_oracle as O
    from tests import hip_harness as HH
    B, H, W = shape
    batch = make_batch(B, H, W, seed=321)
    torch.manual_seed(5)
    noises = [torch.randn(B, 1, H, W) for _ in range(2)]
    kw = dict(height=H, width=W, batch_size=B, n_losses=1, avg_reprojection=avg)

    def build(dev, pose_fn):
        mv = lambda t: t.to(dev).contiguous()
        inputs = {("color", f, 0): mv(batch[k]) for f, k in ((0, "color0"), (-1, "color_m1"), (1, "color_p1"))}
        inputs[("K", 0)], inputs[("inv_K", 0)] = mv(batch["K"]), mv(batch["inv_K"])
        leaves = {k: mv(batch[k]).clone().requires_grad_(True) for k in HH.LEAVES}
        T_m1 = pose_fn(leaves["axisangle_m1"], leaves["translation_m1"], True)
        T_p1 = pose_fn(leaves["axisangle_p1"], leaves["translation_p1"], False)
        outputs = {("disp", 0, 0): leaves["disp_teacher"], ("disp", 0, 1): leaves["disp_student"],
                   ("cam_T_cam", 0, -1): T_m1, ("cam_T_cam", 0, 1): T_p1, ("cam_T_cam", 0, -1, 1): T_m1 * 1.0,
                   "consistency_mask": mv(batch["consistency_mask"]).unsqueeze(1)}
        return inputs, outputs, leaves

    inputs, outputs, leaves = build("cpu", O.transformation_from_parameters)
    opt = O.dr_default_opt(**kw)
    O.dr_generate_images_pred(opt, inputs, outputs)
    ref = O.dr_compute_losses(opt, inputs, outputs, noises=[n.clone() for n in noises])
    ref["loss"].backward()
    inputs, outputs, gl = build(DEV, layers.transformation_from_parameters)
    lp = dualrefine.DualRefineLossPath(dualrefine.default_options(**kw), fuse=fuse)
    from mal_amd import ops
    ops.DECISION_SINK = [] if fuse else None  # the fused passes export their decisions (mal_decisions_next_pass)
    try:
        lp.generate_images_pred(inputs, outputs)
        got = lp.compute_losses(inputs, outputs, noises=[n.to(DEV) for n in noises])
        got["loss"].backward()
        torch.cuda.synchronize()
        sink = [d.cpu() for d in (ops.DECISION_SINK or [])]
    finally:
        ops.DECISION_SINK = None
    assert set(got) == set(ref)
    for k, v in ref.items():  # the free-running oracle: scalars move by what a few near-tie pixels carry
        assert abs(float(got[k].detach()) - float(v)) <= 2e-4 * abs(float(v)) + 2e-5, (k, float(got[k].detach()), float(v))
    # ---- gradients, decision-exact (round 5; rounds 1-4: 2e-2 on the poses against the free-running oracle): the route's own
    # decisions -- exported by the fused passes, re-derived from the dicts the explicit route leaves (its sampling grids, its
    # warped images through the same deterministic min kernel) -- are forced on the oracle, fp64 is the yardstick
    from oracle import aten_restated as AR
    from tests.test_gpu_decisions import _dr_decode, _dr_oracle, _dr_hold_against_forced_oracle, _to64
    if fuse:
        assert len(sink) == 2
        forced = {(0, it): _dr_decode(sink[it]) for it in (0, 1)}
    else:
        from mal_amd import _lib as L
        target = inputs[("color", 0, 0)]
        sources = [inputs[("color", f, 0)] for f in (-1, 1)]
        flags = L.F_AVG if avg else 0
        ident = lp._identity(target, sources, flags)
        forced = {}
        for it in (0, 1):
            cands = [outputs[("color", f, 0, it)].detach() for f in (-1, 1)]
            _, am, wt, _ = ops.photo_fwd(target, cands, ident, noises[it].to(DEV), None, flags | L.F_AUTOMASK)
            win = am.long().cpu()
            pred = torch.where(win == 1, cands[1].cpu(), cands[0].cpu())
            forced[(0, it)] = dict(win=win, automask=wt.cpu(), l1=torch.sign(pred - target.cpu()),
                                   taps={f: AR.taps_of(outputs[("sample", f, 0, it)].detach().cpu(), H, W, align_corners=False)
                                         for f in (-1, 1)})
            if avg:  # the mean over both frames: no argmin, and each candidate's L1 term has its own signs
                forced[(0, it)].pop("l1")
    f32, g32, _, _ = _dr_oracle(batch, kw, noises, forced=forced)
    _, g64, _, _ = _dr_oracle(batch, kw, noises, forced=_to64(forced), dtype=torch.float64)
    _dr_hold_against_forced_oracle({k: gl[k].grad.cpu().numpy() for k in HH.LEAVES}, {k: batch[k].numpy() for k in HH.LEAVES},
                                   f32, g32, g64, {k: float(v.detach()) for k, v in got.items()})


@pytest.mark.parametrize("n_cand", [1, 2, 3, 4])
def test_photo_marching_kernels_match_the_pixel_kernels(n_cand):
    """min-reprojection / automask over materialised candidates (loss_utils.py:84-113) and its backward:
    the marching formulation (two candidates per launch, running min across launches) against the
    one-pixel-per-thread kernels that keep ATen's summation order."""
    from mal_amd import ops, _lib
    from mal_amd.synthetic import make_batch
    lib = _lib.load()
    B, H, W = 3, 45, 139  # ragged against the 62/60-column strips and the row segments
    bt = make_batch(B, H, W, seed=11)
    tgt = bt["color0"].to(DEV)
    g = torch.Generator().manual_seed(3)
    cands = [(bt["color_m1"] if i % 2 == 0 else bt["color_p1"]).clone() for i in range(n_cand)]
    cands = [(c + 0.02 * i * torch.rand(c.shape, generator=g)).clamp(0, 1).to(DEV) for i, c in enumerate(cands)]
    ident = (0.25 * torch.rand(B, 1, H, W, generator=g)).to(DEV)
    noise = torch.randn(B, 1, H, W, generator=g).to(DEV)
    ext = (torch.rand(B, 1, H, W, generator=g) > 0.2).float().to(DEV)
    scale = torch.tensor([0.7], device=DEV)
    res = []
    for impl in (0, 1):
        assert lib.mal_set_option(b"photo_impl", impl) == 0
        mn, am, wt, sums = ops.photo_fwd(tgt, cands, ident=ident, noise=noise, ext_mask=ext, flags=_lib.F_AUTOMASK)
        gr = ops.photo_bwd(tgt, cands, am, wt, scale, sums, 0, [True] * n_cand)
        res.append((mn, am, wt, sums.clone(), gr))
    lib.mal_set_option(b"photo_impl", 1)
    (mn0, am0, wt0, s0, g0), (mn1, am1, wt1, s1, g1) = res
    assert (mn0 - mn1).abs().max().item() <= 1e-4
    flip = (am0 != am1) | (wt0 != wt1)  # candidates / automask at rounding distance of each other
    assert flip.float().mean().item() <= 2e-3
    for k in range(2):
        assert abs(float(s0[k]) - float(s1[k])) <= 1e-4 * abs(float(s0[k])) + 2.0 * float(flip.sum())
    near = torch.nn.functional.max_pool2d(flip.float(), 5, 1, 2) > 0  # a flipped pixel moves its neighbours' gradients
    for a, c in zip(g0, g1):
        d = (a - c).abs()[~near.expand_as(a)]
        assert d.max().item() <= 2e-4 * a.abs().max().item() * (1 + 4 * float(flip.sum()))


# ------------------------------------------------------------------ module-level gradients, decision-exact (round 3)
# The loose bounds above (pose gradient of the bare warp at 2e-3, smoothness gradient with 2 % of the pixels exempt) came
# from comparing against a reference that takes ITS OWN discontinuous decisions.  Below, the only such decisions of the
# bare warp -- the bilinear tap cell and the border clip per pixel -- are derived from the kernel's own sampling grid and
# forced on the float64 restatement (oracle.aten_restated.grid_sample_forced_taps): both sides then evaluate the same
# smooth function and every element is held at 1e-4 of the map's scale, no pixel exempted.
@pytest.mark.parametrize("ac", [True, False], ids=["manydepth_align_corners", "dualrefine"])
@pytest.mark.parametrize("tag", G.LAYER_CASES)
def test_warp_gradient_with_forced_taps(tag, ac):
    from mal_amd import layers
    from oracle import aten_restated as AR, mal_oracle as O
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    dev = lambda t: t.to(DEV)
    # 1. the sampler alone: d/d grid on the kernel's own grid
    grid_ref = torch.from_numpy(z["grid_A" if ac else "grid_B"].copy())
    grid_h = dev(grid_ref).clone().requires_grad_(True)
    cot = torch.from_numpy(z["in/g_warped"].copy())
    warped = layers.grid_sample(dev(b["color_m1"]), grid_h, padding_mode="border", align_corners=ac)
    (warped * dev(cot)).sum().backward()
    x0, y0, clipx, clipy = AR.taps_of(grid_ref, H, W, align_corners=ac)            # fp32, ATen's unnormalisation = the kernel's
    g64 = grid_ref.double().clone().requires_grad_(True)
    out64 = AR.grid_sample_forced_taps(b["color_m1"].double(), g64, x0, y0, clipx, clipy, align_corners=ac)
    (out64 * cot.double()).sum().backward()
    assert float((warped.detach().cpu().double() - out64.detach()).abs().max()) <= 2e-6
    gh, gr = grid_h.grad.cpu().double(), g64.grad
    assert float((gh - gr).abs().max()) <= 1e-4 * float(gr.abs().max()), float((gh - gr).abs().max()) / float(gr.abs().max())
    # 2. the geometry in front of it (no decisions): disp, T -> grid, with a fixed cotangent on the grid
    G_grid = gr.float()
    disp = dev(b["disp_teacher"]).clone().requires_grad_(True)
    T = dev(torch.from_numpy(z["T_inv1"].copy())).requires_grad_(True)
    pts = layers.BackprojectDepth(B, H, W)(layers.disp_to_depth(disp, 0.1, 100.0)[1], dev(b["inv_K"]))
    grid = (layers.Project3D if ac else layers.Project3DDualRefine)(B, H, W)(pts, dev(b["K"]), T)
    (grid * dev(G_grid)).sum().backward()
    d64 = b["disp_teacher"].double().clone().requires_grad_(True)
    T64 = torch.from_numpy(z["T_inv1"].copy()).double().requires_grad_(True)
    p64 = O.backproject_depth(O.disp_to_depth(d64, 0.1, 100.0)[1], b["inv_K"].double())
    gr64 = O.project_3d(p64, b["K"].double(), T64, H, W, convention="manydepth" if ac else "dualrefine")
    (gr64 * G_grid.double()).sum().backward()
    for name, h, r in (("disp", disp.grad.cpu().double(), d64.grad), ("T", T.grad.cpu().double()[:, :3], T64.grad[:, :3])):
        assert float((h - r).abs().max()) <= 1e-4 * float(r.abs().max()), (name, float((h - r).abs().max()) / float(r.abs().max()))


@pytest.mark.parametrize("tag", G.LAYER_CASES)
def test_smoothness_gradient_every_pixel(tag):
    """get_smooth_loss (layers.py:210-223): sign(d_x - d_x') of two fp32 numbers is exact (0 at a tie, as torch.abs's
    gradient), so there is nothing to exempt: every pixel at 1e-4 of the map's scale against the reference's own gradient"""
    from mal_amd import layers
    z = G.load(tag)
    b = G.batch_from_golden(z)
    d = b["disp_student"].to(DEV).clone().requires_grad_(True)
    sm = layers.get_smooth_loss(d, b["color0"].to(DEV))
    sm.backward()
    g, r = d.grad.cpu().numpy(), z["grad_smooth"]
    assert abs(float(sm) - float(z["smooth"])) <= 1e-5 * abs(float(z["smooth"]))
    assert np.abs(g - r).max() <= 1e-4 * np.abs(r).max(), np.abs(g - r).max() / np.abs(r).max()
