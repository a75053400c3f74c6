"""GPU parity of the trainer-level methods that the step tests do not reach: the non-distillation
``compute_losses`` fallback (manydepth/trainer.py:1248-1475) with two scales, ``--no_ssim``,
``materialize_warps`` and the dictionary contract of ``generate_images_pred``."""
import numpy as np
import pytest
import torch

from mal_amd.synthetic import make_batch, to_dicts
from oracle import mal_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _l2rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def _two_scale(batch, dev, pose_fn):
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, pose_fn, device=dev)
    H, W = batch["color0"].shape[-2:]
    # scale 1: half-resolution disparity and colour (mono_dataset.py resizes per scale)
    lo = torch.nn.functional.avg_pool2d(batch["disp_student"], 2).to(dev).clone().requires_grad_(True)
    leaves["disp_lo"] = lo
    for d in (mono_outputs, outputs):
        d[("disp", 1)] = lo
    inputs[("color", 0, 1)] = torch.nn.functional.avg_pool2d(batch["color0"], 2).to(dev)
    return inputs, mono_outputs, outputs, leaves


@pytest.mark.parametrize("fuse,no_ssim", [(True, False), (False, False), (False, True)],
                         ids=["fused", "explicit", "explicit-no_ssim"])
def test_non_distil_losses_two_scales(fuse, no_ssim):
    from mal_amd import layers, trainer, config
    B, H, W = 2, 48, 80
    batch = make_batch(B, H, W, seed=77)
    torch.manual_seed(9)
    noises = [torch.randn(B, 1, H, W) for _ in range(2)]
    kw = dict(height=H, width=W, batch_size=B, sclm=1, distil=False, no_ssim=no_ssim)
    # ---- oracle
    opt = O.default_opt(**kw)
    inputs, mono_outputs, outputs, leaves = _two_scale(batch, "cpu", O.transformation_from_parameters)
    O.generate_images_pred(opt, inputs, mono_outputs)
    ref_t = O.compute_losses(opt, inputs, mono_outputs, is_multi=False, noises=[n.clone() for n in noises])
    for key in list(mono_outputs.keys()):
        if isinstance(key, tuple) and key[0] in ("depth", "disp"):
            outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
    O.generate_images_pred(opt, inputs, outputs, is_multi=True)
    ref_s = O.compute_losses(opt, inputs, outputs, is_multi=True)
    (ref_t["loss"] + ref_s["loss"]).backward()
    # ---- HIP
    hopt = trainer.default_options(**kw)
    hin, hmono, hout, hl = _two_scale(batch, DEV, layers.transformation_from_parameters)
    lp = trainer.LossPath(hopt, fuse=fuse)
    old = config.noise_source
    config.noise_source = "given"
    try:
        lp.generate_images_pred(hin, hmono)
        got_t, _ = lp.compute_losses(hin, hmono, is_multi=False, noises=[n.to(DEV) for n in noises])
        for key in list(hmono.keys()):
            if isinstance(key, tuple) and key[0] in ("depth", "disp"):
                hout[("mono_" + key[0],) + tuple(key[1:])] = hmono[key]
        lp.generate_images_pred(hin, hout, is_multi=True)
        got_s, _ = lp.compute_losses(hin, hout, is_multi=True)
    finally:
        config.noise_source = old
    (got_t["loss"] + got_s["loss"]).backward()
    for ref, got in ((ref_t, got_t), (ref_s, got_s)):
        assert set(ref) == set(got), (sorted(ref), sorted(got))
        for k, v in ref.items():
            # the masked mean sum(rp*m)/sum(m) moves by at most (rp_i + mean)/sum(m) <~ 1/(B*H*W) per automask pixel that
            # sits at rounding distance of its threshold and takes the other side: two such pixels allowed here (the
            # whole-step route is checked decision-exactly instead, tests/test_gpu_decisions.py)
            tie = 2.0 / (B * H * W) if ("reproj" in k or k.startswith("loss")) else 0.0
            assert abs(float(got[k].detach()) - float(v)) <= 2e-4 * abs(float(v)) + 1e-6 + tie, (k, float(got[k].detach()), float(v))
    for k in ("disp_teacher", "disp_student", "disp_lo"):
        g, r = hl[k].grad.cpu().numpy(), leaves[k].grad.numpy()
        # one near-tie pixel taking the other branch moves its own gradient, and through the bilinear
        # upsampling's adjoint up to 9 pixels of the half-resolution map
        # ... and an automask pixel that flips rescales EVERY teacher gradient by sum(m)/(sum(m) +- 1): two flips over a
        # mask that covers at least half of the pixels (renorm; the whole-step route forces the decisions instead)
        renorm = 4.0 / (B * H * W)
        assert (np.abs(g - r) > (2e-4 + renorm) * np.abs(r).max()).mean() <= (2e-2 if k == "disp_lo" else 5e-3), k
    for k in ("axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1"):
        assert _l2rel(hl[k].grad.cpu().numpy(), leaves[k].grad.numpy()) <= 2e-2, k


def test_generate_images_pred_dictionary_contract():
    """keys written by generate_images_pred (manydepth/trainer.py:1100,1117,1122,1157) and the lazy record."""
    from mal_amd import layers, trainer
    B, H, W = 2, 32, 48
    batch = make_batch(B, H, W, seed=3)
    opt = trainer.default_options(height=H, width=W, batch_size=B)
    inputs, mono_outputs, _, _ = to_dicts(batch, layers.transformation_from_parameters, device=DEV)
    eager = trainer.LossPath(opt, fuse=False)
    eager.generate_images_pred(inputs, mono_outputs)
    for k in (("depth", 0, 0), ("sample", -1, 0), ("sample", 1, 0), ("color", -1, 0), ("color", 1, 0),
              ("color_identity", -1, 0), ("color_identity", 1, 0)):
        assert k in mono_outputs, k
    assert mono_outputs[("sample", 1, 0)].shape == (B, H, W, 2) and mono_outputs[("color", 1, 0)].shape == (B, 3, H, W)
    inputs2, lazy_out, _, _ = to_dicts(batch, layers.transformation_from_parameters, device=DEV)
    lazy = trainer.LossPath(opt, fuse=True)
    lazy.generate_images_pred(inputs2, lazy_out)
    assert ("mal_ctx", 0) in lazy_out and ("color", 1, 0) not in lazy_out and ("depth", 0, 0) in lazy_out
    lazy.materialize_warps(lazy_out)
    assert torch.allclose(lazy_out[("color", 1, 0)], mono_outputs[("color", 1, 0)].detach(), atol=1e-6)
    assert torch.allclose(lazy_out[("sample", -1, 0)], mono_outputs[("sample", -1, 0)].detach(), atol=1e-6)
    # oracle agreement of the materialised warp
    oin, oout, _, _ = to_dicts(batch, O.transformation_from_parameters)
    O.generate_images_pred(O.default_opt(height=H, width=W, batch_size=B), oin, oout)
    assert np.abs(mono_outputs[("color", 1, 0)].detach().cpu().numpy() - oout[("color", 1, 0)].detach().numpy()).max() <= 1e-5
    assert np.abs(mono_outputs[("depth", 0, 0)].detach().cpu().numpy() / oout[("depth", 0, 0)].detach().numpy() - 1).max() <= 1e-6


def _run_multiscale(inputs, mono_outputs, outputs, leaves, opt_kw, nt, ns, hip, fuse=True):
    """both networks' compute_losses over sclm+1 scales (trainer.py:573-612 with not opt.distil), backward of the sum"""
    if hip:
        from mal_amd import trainer, config
        lp = trainer.LossPath(trainer.default_options(**opt_kw), fuse=fuse)
        old = config.noise_source
        config.noise_source = "given"
        try:
            lp.generate_images_pred(inputs, mono_outputs)
            lt, _ = lp.compute_losses(inputs, mono_outputs, is_multi=False, noises=[n.to(DEV) for n in nt])
            for key in list(mono_outputs.keys()):
                if isinstance(key, tuple) and key[0] in ("depth", "disp"):
                    outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
            lp.generate_images_pred(inputs, outputs, is_multi=True)
            ls, _ = lp.compute_losses(inputs, outputs, is_multi=True)
        finally:
            config.noise_source = old
    else:
        opt = O.default_opt(**opt_kw)
        O.generate_images_pred(opt, inputs, mono_outputs)
        lt = O.compute_losses(opt, inputs, mono_outputs, is_multi=False, noises=[n.clone() for n in nt])
        for key in list(mono_outputs.keys()):
            if isinstance(key, tuple) and key[0] in ("depth", "disp"):
                outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
        O.generate_images_pred(opt, inputs, outputs, is_multi=True)
        ls = O.compute_losses(opt, inputs, outputs, is_multi=True, noises=[n.clone() for n in ns])
    (lt["loss"] + ls["loss"]).backward()
    return lt, ls


@pytest.mark.parametrize("fuse", [True, False], ids=["fused", "explicit"])
def test_four_scales_against_the_reference_fixture(fuse):
    """sclm=3 (BASELINE configs[1]'s "4 scales"): the fixture holds the reference's own numbers for both networks'
    compute_losses over four disparity scales (oracle/gen_golden.py run_reference_multiscale)"""
    from mal_amd import layers
    from tests import golden_io as G
    z = G.load(G.MULTISCALE_CASE)
    b, sclm, inputs, mono_outputs, outputs, leaves = G.multiscale_dicts(z, layers.transformation_from_parameters, DEV)
    B, _, H, W = b["color0"].shape
    nt, ns = G.multiscale_noises(z, (B, 1, H, W), sclm)
    kw = dict(height=H, width=W, batch_size=B, sclm=sclm, distil=False)
    lt, ls = _run_multiscale(inputs, mono_outputs, outputs, leaves, kw, nt, ns, hip=True, fuse=fuse)
    N = B * H * W
    for who, d in (("teacher", lt), ("student", ls)):
        for k, v in d.items():
            ref = float(z["%s/%s" % (who, k)])
            # an automask pixel at rounding distance of its threshold moves a masked mean by <~ 1/N: two allowed per scale
            tie = 2.0 * (sclm + 1) / N if (who == "teacher" and ("reproj" in k or k.startswith("loss"))) else 0.0
            assert abs(float(v.detach()) - ref) <= 2e-4 * abs(ref) + 1e-6 + tie, (who, k, float(v.detach()), ref)
    renorm = 4.0 / N
    for k, t in leaves.items():
        g, r = t.grad.cpu().numpy(), z["grad/" + k].reshape(t.shape)
        if g.ndim == 4:
            bad = (np.abs(g - r) > (2e-4 + renorm) * np.abs(r).max()).mean()
            assert bad <= (2e-2 if k[-1].isdigit() else 5e-3), (k, bad)
        else:
            assert _l2rel(g, r) <= 2e-2, k


def test_four_scales_at_baseline_size():
    """B=12 192x640, sclm=3 against the oracle (whose glue the fixture above pins)"""
    from mal_amd import layers
    B, H, W, sclm = 12, 192, 640, 3
    batch = make_batch(B, H, W, seed=79)
    g = torch.Generator().manual_seed(12)
    nt = [torch.randn(B, 1, H, W, generator=g) for _ in range(sclm + 1)]
    ns = [torch.randn(B, 1, H, W, generator=g) for _ in range(sclm + 1)]
    kw = dict(height=H, width=W, batch_size=B, sclm=sclm, distil=False)

    def build(dev, pose_fn):
        inputs, mono_outputs, outputs, leaves = to_dicts(batch, pose_fn, device=dev)
        for s in range(1, sclm + 1):
            inputs[("color", 0, s)] = torch.nn.functional.avg_pool2d(batch["color0"], 2 ** s).to(dev)
            for name, outs in (("disp_teacher", mono_outputs), ("disp_student", outputs)):
                leaf = torch.nn.functional.avg_pool2d(batch[name], 2 ** s).to(dev).clone().requires_grad_(True)
                leaves["%s_s%d" % (name, s)] = leaf
                outs[("disp", s)] = leaf
        return inputs, mono_outputs, outputs, leaves

    oi, om, oo, ol = build("cpu", O.transformation_from_parameters)
    rt, rs = _run_multiscale(oi, om, oo, ol, kw, nt, ns, hip=False)
    hi, hm, ho, hl = build(DEV, layers.transformation_from_parameters)
    lt, ls = _run_multiscale(hi, hm, ho, hl, kw, nt, ns, hip=True)
    N = B * H * W
    for ref, got, who in ((rt, lt, "teacher"), (rs, ls, "student")):
        assert set(ref) == set(got)
        for k, v in ref.items():
            tie = 40.0 * (sclm + 1) / N if (who == "teacher" and ("reproj" in k or k.startswith("loss"))) else 0.0
            assert abs(float(got[k].detach()) - float(v)) <= 1e-4 * abs(float(v)) + 1e-6 + tie, (who, k, float(got[k].detach()), float(v))
    for k in hl:
        gq, r = hl[k].grad.cpu().numpy(), ol[k].grad.numpy()
        if gq.ndim == 4:  # near-tie pixels (a few tens per 1.5 M, tests/test_gpu_decisions.py) and their neighbourhoods
            bad = (np.abs(gq - r) > 3e-4 * np.abs(r).max()).mean()
            sc = int(k[-1]) if k[-1].isdigit() else 0  # a pixel of scale s collects the gradient of 4**s full-resolution pixels
            assert bad <= 1e-3 * (1 + 4 ** sc / 8.0), (k, bad)
        else:
            assert _l2rel(gq, r) <= 2e-2, (k, _l2rel(gq, r))
