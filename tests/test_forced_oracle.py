"""Decision-forced mode of the CPU oracle (``forced=`` of oracle.mal_oracle.mal_loss_step): told the decisions it
would take by itself, it must reproduce the free-running oracle -- so that, told the HIP kernels' decisions, any
remaining difference to the kernels is arithmetic, not a re-decided near-tie (tests/test_gpu_decisions.py)."""
import numpy as np
import pytest
import torch

from tests import golden_io as G
from tests import hip_harness as HH


@pytest.mark.parametrize("tag", ["step_b2_32x64_distil", "step_b2_32x64_noens", "step_b3_37x50_distil"])
def test_forced_with_own_decisions_reproduces_free_run(tag):
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    kw = G.opt_kwargs(z)
    o = HH.run_oracle(b, kw, n0, n1)
    dec = HH.oracle_decisions(o, b, n0, no_ens=bool(kw.get("no_ens")))
    f = HH.run_oracle(b, kw, n0, n1, forced=dec)
    for k, v in o["losses"].items():
        assert abs(f["losses"][k] - v) <= 2e-6 * abs(v), k
    for k in HH.LEAVES:
        a, r = f["grads"][k], o["grads"][k]
        # the restated sampler (explicit taps) and ATen's differ by fp32 association only
        assert np.abs(a - r).max() <= 1e-4 * np.abs(r).max(), (k, np.abs(a - r).max() / np.abs(r).max())
    d = HH.decision_differences(dec, dec)
    assert not any(v.any() for v in d.values())


def test_forcing_changes_only_the_forced_choice():
    """flip one teacher winner: the value moves by the candidates' gap at that pixel, nothing else changes"""
    z = G.load("step_b2_32x64_distil")
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    o = HH.run_oracle(b, {}, n0, n1)
    dec = HH.oracle_decisions(o, b, n0)
    m = dec["teacher"]["automask"][0, 0]
    y, x = [int(v[0]) for v in torch.nonzero(m)[:1].T]
    dec["teacher"]["win"][0, 0, y, x] ^= 1
    del dec["teacher"]["l1"]  # the signs belong to the old winner; without them |t-p| is re-decided
    f = HH.run_oracle(b, {}, n0, n1, forced=dec)
    gap = abs(float(o["mono_cands"][0, 0, y, x] - o["mono_cands"][0, 1, y, x]))
    expect = gap / float(dec["teacher"]["automask"].sum())
    got = f["mono_losses"]["reproj_loss/0"] - o["mono_losses"]["reproj_loss/0"]
    assert abs(got - expect) <= 1e-2 * expect + 1e-7, (got, expect)  # the loss itself is an fp32 number near 0.1


@pytest.mark.parametrize("temporal", [False, True], ids=["plain", "temporal"])
def test_four_scale_forced_with_own_decisions_reproduces_free_run(temporal):
    """the same for the non-distil four-scale path (oracle.mal_oracle.compute_losses / generate_images_pred with one decision
    set per scale): what tests/test_gpu_multiscale.py forces the kernels' decisions through"""
    from mal_amd.synthetic import make_batch, fake_image_synthesis
    B, H, W, sclm = 2, 32, 64, 2
    batch = make_batch(B, H, W, seed=81, with_syn=temporal)
    g = torch.Generator().manual_seed(13)
    nt = [torch.randn(B, 1, H, W, generator=g) for _ in range(sclm + 1)]
    kw = dict(height=H, width=W, batch_size=B, sclm=sclm, distil=False, temporal=temporal)
    synth = fake_image_synthesis(batch["syn_rects"]) if temporal else None
    o = HH.ms_run_oracle(batch, kw, nt, nt, True, synth=synth)
    dec = HH.ms_oracle_decisions(o, batch, nt, sclm)
    f = HH.ms_run_oracle(batch, kw, nt, nt, True, synth=synth, forced=dec)
    for who in ("teacher", "student"):
        for k, v in o[who].items():
            assert abs(f[who][k] - v) <= 2e-6 * abs(v) + 1e-9, (who, k, f[who][k], v)
    for k, r in o["grads"].items():
        a = f["grads"][k]
        assert np.abs(a - r).max() <= 1e-4 * np.abs(r).max(), (k, np.abs(a - r).max() / np.abs(r).max())
    d = HH.ms_decision_differences(dec, dec, sclm)
    assert not any(v.any() for sc in d for v in sc.values())
    # ... and in float64 it is the same function
    f64 = HH.ms_run_oracle(batch, kw, nt, nt, True, synth=synth, forced=dec, double=True)
    for k, r in o["grads"].items():
        assert np.linalg.norm((f64["grads"][k] - r).ravel()) <= 2e-3 * np.linalg.norm(r.ravel()), k


@pytest.mark.parametrize("kw_extra", [{}, {"Tstar_D0_pair": True}, {"no_ssim": True}], ids=["default", "Tstar_D0", "no_ssim"])
def test_dualrefine_pose_update_forced_with_own_decisions_reproduces_free_run(kw_extra):
    """DualRefine's pose-update losses (oracle.mal_oracle.dr_pose_update_generate_images_pred / dr_compute_pose_update_losses,
    dualrefine/trainer.py:457-480,699-767) with ``forced=``: told its own taps, winner, automask and L1 signs, the oracle
    reproduces its free run -- what tests/test_gpu_decisions.py forces the FRAMED marching pass's decisions through"""
    from mal_amd.synthetic import make_batch
    from oracle import aten_restated as AR
    from oracle import mal_oracle as O
    B, H, W = 2, 24, 40
    batch = make_batch(B, H, W, seed=17)
    torch.manual_seed(3)
    noises = [torch.randn(B, 1, H, W) for _ in range(2)]
    nz = torch.randn(B, 1, H, W)
    kw = dict(height=H, width=W, batch_size=B, n_losses=1)
    kw.update(kw_extra)

    def run(forced=None, dtype=torch.float32):
        mv = lambda t: t.to(dtype) if t.is_floating_point() else t
        inputs = {("color", f, 0): mv(batch[k]) for f, k in ((0, "color0"), (-1, "color_m1"), (1, "color_p1"))}
        inputs[("K", 0)], inputs[("inv_K", 0)] = mv(batch["K"]), mv(batch["inv_K"])
        leaves = {k: mv(batch[k]).clone().requires_grad_(True) for k in HH.LEAVES}
        pf = O.transformation_from_parameters
        outputs = {("disp", 0, 0): leaves["disp_teacher"], ("disp", 0, 1): leaves["disp_student"],
                   ("cam_T_cam", 0, -1): pf(leaves["axisangle_m1"], leaves["translation_m1"], True),
                   ("cam_T_cam", 0, 1): pf(leaves["axisangle_p1"], leaves["translation_p1"], False),
                   ("cam_T_cam", 0, -1, 1): pf(leaves["axisangle_m1"] * 1.05 + 0.002, leaves["translation_m1"] * 0.95 - 0.003, True),
                   "consistency_mask": mv(batch["consistency_mask"]).unsqueeze(1)}
        opt = O.dr_default_opt(**kw)
        O.dr_generate_images_pred(opt, inputs, outputs)
        O.dr_pose_update_generate_images_pred(opt, inputs, outputs, forced=forced)
        pl = O.dr_compute_pose_update_losses(opt, inputs, outputs, noise=nz.clone().to(dtype), forced=forced)
        pl["loss"].backward()
        return pl, {k: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy() for k, t in leaves.items()}, inputs, outputs

    pl, g, inputs, outputs = run()
    target = inputs[("color", 0, 0)]
    cands = [outputs[("color", -1, 0, 0, 1)].detach(), outputs[("color", 1, 0, 0)].detach()]
    R = torch.cat([O.compute_reprojection_loss(c, target, kw.get("no_ssim", False)) for c in cands], 1)
    win = R.argmin(1, keepdim=True)
    I = torch.cat([O.compute_reprojection_loss(inputs[("color", f, 0)], target, kw.get("no_ssim", False)) for f in (-1, 1)], 1)
    ident = I.min(1, keepdim=True)[0] + nz * 0.00001
    automask = O.compute_loss_masks(R.min(1, keepdim=True)[0], ident)
    chosen = torch.where(win.bool(), cands[1], cands[0])
    forced = dict(taps={-1: AR.taps_of(outputs[("sample", -1, 0, 0, 1)], H, W, align_corners=False)}, win=win, automask=automask,
                  l1=torch.sign(chosen - target))
    fl, fg, _, _ = run(forced)
    for k, v in pl.items():
        assert abs(float(fl[k].detach()) - float(v.detach())) <= 2e-6 * abs(float(v.detach())), k
    moved = 0
    for k, r in g.items():
        if np.abs(r).max() > 0:
            moved += 1
            assert np.abs(fg[k] - r).max() <= 1e-4 * np.abs(r).max(), (k, np.abs(fg[k] - r).max() / np.abs(r).max())
        else:
            assert not fg[k].any(), k
    assert moved >= 5  # both disparities (frame -1's: unless detached), both frames' pose parameters
    if kw.get("Tstar_D0_pair"):
        assert not g["disp_student"].any()  # the refined pose is paired with iteration 0's depth, detached (:463-465)
    # ... and in float64 it is the same function
    _, g64, _, _ = run({k: ({f: t for f, t in v.items()} if k == "taps" else (v.double() if v.dtype == torch.float32 else v))
                        for k, v in forced.items()}, torch.float64)
    for k, r in g.items():
        if np.abs(r).max() > 0:
            assert np.linalg.norm((g64[k] - r).ravel()) <= 2e-3 * np.linalg.norm(r.ravel()), k
