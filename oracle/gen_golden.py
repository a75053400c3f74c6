"""Generate the golden vectors under tests/golden/ from the REFERENCE's own functions.

TEST INFRASTRUCTURE ONLY.  Run in the authoring container only (it needs
/root/reference, which never travels):

    python -m oracle.gen_golden

What is imported from the reference: ``manydepth.layers``,
``manydepth.loss_utils`` (after registering an empty stand-in for the sibling module
``manydepth.pareto``, which upstream never committed and only the ``opt.pareto`` branch
uses) and ``dualrefine.layers``.  ``manydepth.trainer`` is not importable (cv2, wandb,
detectron2, torchmetrics, absent vis.py), so the few lines of glue between the layers
and the loss functions (trainer.py:573-629,1078-1207) are restated here, calling the
reference's layer objects and loss functions for all arithmetic.

The vectors are data: inputs (colours as uint8, disparities as fp16 so the stored
values are exact), the CPU-generator noise, and the reference's outputs / gradients.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    sys.path.insert(0, REF)
    stub = types.ModuleType("manydepth.pareto")

    def pareto_fn(*a, **k):
        raise NotImplementedError("manydepth/pareto.py is absent upstream")

    stub.pareto_fn = pareto_fn
    sys.modules["manydepth.pareto"] = stub
    import manydepth.layers as ML
    import manydepth.loss_utils as MLU
    import dualrefine.layers as DL
    return ML, MLU, DL


def quantize_batch(batch):
    """Make every stored input exactly representable in the on-disk dtype."""
    q = dict(batch)
    for k in ("color0", "color_m1", "color_p1"):
        q[k] = torch.round(batch[k] * 255).clamp(0, 255) / 255
    for k in ("disp_teacher", "disp_student", "lowest_cost"):
        q[k] = batch[k].half().float()
    for k in ("axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1"):
        q[k] = batch[k].clone()
    return q


def pack_inputs(q):
    d = {}
    for k in ("color0", "color_m1", "color_p1"):
        d["in/" + k] = torch.round(q[k] * 255).to(torch.uint8).numpy()
    for k in ("disp_teacher", "disp_student", "lowest_cost"):
        d["in/" + k] = q[k].half().numpy()
    for k in ("axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1", "K", "inv_K"):
        d["in/" + k] = q[k].numpy()
    d["in/consistency_mask"] = q["consistency_mask"].to(torch.uint8).numpy()
    d["in/augmentation_mask"] = q["augmentation_mask"].to(torch.uint8).numpy()
    if "syn_rects" in q:
        d["in/syn_rects"] = np.array(q["syn_rects"], dtype=np.int64)
    return d


def summarize(name, t, d, full):
    """Full tensor for small cases; for big ones sums + an 8x8-strided subsample."""
    a = t.detach().double()
    if full:
        d[name] = t.detach().numpy()
        return
    d[name + "#sum"] = np.float64(a.sum())
    d[name + "#abs"] = np.float64(a.abs().sum())
    d[name + "#sq"] = np.float64((a * a).sum())
    if t.dim() >= 2 and t.shape[-1] >= 16 and t.shape[-2] >= 16:
        d[name + "#sub"] = t.detach()[..., ::8, ::8].contiguous().numpy()
    else:
        d[name] = t.detach().numpy()


def run_reference_step(ML, MLU, q, opt_kw, noise_seed, full, tag):
    from mal_amd.synthetic import to_dicts, fake_image_synthesis
    from oracle.mal_oracle import default_opt

    B, _, H, W = q["color0"].shape
    opt = default_opt(height=H, width=W, batch_size=B, **opt_kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(q, ML.transformation_from_parameters)
    ssim = ML.SSIM()
    backproject = ML.BackprojectDepth(B, H, W)
    project = ML.Project3D(B, H, W)
    synth = fake_image_synthesis(q["syn_rects"]) if "syn_rects" in q else None

    def gen_pred(outs, is_multi):  # glue: trainer.py:1078-1170
        disp = F.interpolate(outs[("disp", 0)], [H, W], mode="bilinear", align_corners=False)
        _, depth = ML.disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outs[("depth", 0, 0)] = depth
        for f in (-1, 1):
            T = outs[("cam_T_cam", 0, f)]
            if is_multi:
                T = T.detach()
            pts = backproject(depth, inputs[("inv_K", 0)])
            grid = project(pts, inputs[("K", 0)], T)
            outs[("sample", f, 0)] = grid
            outs[("color", f, 0)] = F.grid_sample(inputs[("color", f, 0)], grid, padding_mode="border",
                                                  align_corners=True)
        has = False
        if synth is not None and ((not is_multi and opt.temporal) or (is_multi and opt.main_temporal)):
            has = synth(inputs, outs, 0)
        return has

    d = {}
    shape = (B, 1, H, W)
    torch.manual_seed(noise_seed)
    noise_mono = torch.randn(shape)
    noise_main = torch.randn(shape)
    if full:
        d["in/noise_mono"] = noise_mono.numpy()
        d["in/noise_main"] = noise_main.numpy()
    else:  # regenerated from the seed by the tests (torch.manual_seed; randn; randn), checked by sum
        d["in/noise_mono#sum"] = np.float64(noise_mono.double().sum())
        d["in/noise_main#sum"] = np.float64(noise_main.double().sum())
    d["in/noise_seed"] = np.int64(noise_seed)

    # pass A (teacher)
    has_ins = gen_pred(mono_outputs, False) and opt.temporal
    torch.manual_seed(noise_seed)
    mono_losses, mono_reproj = MLU.compute_mono_losses(ssim, inputs, mono_outputs, opt.temporal, has_ins)
    for key in list(mono_outputs.keys()):
        if isinstance(key, tuple) and key[0] in ("depth", "disp"):
            outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
    # trainer.py:1066-1076 / :592-593
    mono_d = outputs[("mono_depth", 0, 0)]
    matching = 1 / outputs["lowest_cost"].unsqueeze(1)
    mm = ((matching - mono_d) / mono_d) < 1.0
    mm = mm * (((mono_d - matching) / matching) < 1.0)
    d["matching_mask"] = mm[:, 0].to(torch.uint8).numpy() if full else np.int64(mm.sum())
    outputs["consistency_mask"] = outputs["consistency_mask"] * mm[:, 0]
    # pass B (ensemble): trainer.py:594-600,1172-1207
    ensemble_reproj = None
    if not opt.no_ens:
        if opt.learn_ens:  # trainer.py:596-597: the learnt ensemble head's disparity (a leaf here), not detached
            ens_leaf = q["disp_ens"].clone().requires_grad_(True)
            outputs["ens_disp"] = ens_leaf
            leaves["disp_ens"] = ens_leaf
            disp_e = ens_leaf
        else:
            disp_e = (mono_outputs[("disp", 0)].detach() + outputs[("disp", 0)].detach()) / 2.0
        disp_e = F.interpolate(disp_e, [H, W], mode="bilinear", align_corners=False)
        _, depth_e = ML.disp_to_depth(disp_e, opt.min_depth, opt.max_depth)
        rr = []
        for f in (-1, 1):
            pts = backproject(depth_e, inputs[("inv_K", 0)])
            grid = project(pts, inputs[("K", 0)], outputs[("cam_T_cam", 0, f)].detach())
            pred = F.grid_sample(inputs[("color", f, 0)], grid, padding_mode="border", align_corners=True)
            rr.append(MLU.compute_reprojection_loss(ssim, pred, inputs[("color", 0, 0)]))
        ensemble_reproj = torch.min(torch.cat(rr, 1), dim=1, keepdim=True)[0]
    # pass C (student)
    multi_has = gen_pred(outputs, True) and opt.main_temporal
    w_list = [0.7, 0.3]
    losses, w_out, loss_list = MLU.compute_main_losses(ssim, inputs, outputs, mono_reproj, ensemble_reproj, opt,
                                                       None, w_list, multi_has)
    main_only = {k: v.detach().clone() for k, v in losses.items()}
    for k, v in mono_losses.items():
        losses[k] = losses[k] + v
    if opt.loss_blc:
        loss_list[0] = loss_list[0] + mono_losses["loss"]
        # loss_utils.py:303-318 (restated: .cuda() there): bs * sum_i w_i L_i
        final = B * (w_list[0] * loss_list[0] + w_list[1] * loss_list[1])
        d["loss_list0"] = np.float64(loss_list[0].item())
        d["loss_list1"] = np.float64(loss_list[1].item())
    else:
        final = losses["loss"]
    final.backward()

    d["final_loss"] = np.float64(final.item())
    for k, v in mono_losses.items():
        d["mono_losses/" + k] = np.float64(v.item())
    for k, v in main_only.items():
        d["main_losses/" + k] = np.float64(v.item())
    for k, v in losses.items():
        d["losses/" + k] = np.float64(v.item())
    summarize("mono_reproj", mono_reproj, d, full)
    if ensemble_reproj is not None:
        summarize("ensemble_reproj", ensemble_reproj, d, full)
    summarize("mono/depth", mono_outputs[("depth", 0, 0)], d, full)
    summarize("multi/depth", outputs[("depth", 0, 0)], d, full)
    summarize("consistency_target", outputs["consistency_target/0"], d, full)
    for f in (-1, 1):
        fn = "m1" if f < 0 else "p1"
        summarize("mono/color_" + fn, mono_outputs[("color", f, 0)], d, full)
        summarize("multi/color_" + fn, outputs[("color", f, 0)], d, full)
        summarize("mono/sample_" + fn, mono_outputs[("sample", f, 0)].permute(0, 3, 1, 2), d, full)
        if ("syn", f, 0) in mono_outputs:
            summarize("mono/syn_" + fn, mono_outputs[("syn", f, 0)], d, full)
    for k, t in leaves.items():
        g = t.grad if t.grad is not None else torch.zeros_like(t)
        summarize("grad/" + k, g, d, full)
    d.update(pack_inputs(q))
    if "disp_ens" in q:
        d["in/disp_ens"] = q["disp_ens"].half().numpy()
    d["opt"] = np.array(repr(sorted(opt_kw.items())))
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "final_loss", d["final_loss"])


def multiscale_inputs(q, sclm):
    """the per-scale inputs of the sclm>0 path derived from a quantised batch: ("color", 0, s) = the target pooled by
    2**s (mono_dataset.py:150-166 resizes; pooling keeps the fixture self-contained) and per-scale disparities =
    the full-resolution ones pooled and rounded to fp16 (exactly representable on disk)"""
    color = {s: F.avg_pool2d(q["color0"], 2 ** s) for s in range(1, sclm + 1)}
    disp = {name: {s: F.avg_pool2d(q[name], 2 ** s).half().float() for s in range(1, sclm + 1)}
            for name in ("disp_teacher", "disp_student")}
    return color, disp


def run_reference_multiscale(ML, MLU, q, sclm, noise_seed, tag, temporal=False):
    """The non-distillation ``Trainer.compute_losses`` over ``sclm+1`` disparity scales for both networks
    (manydepth/trainer.py:1078-1170 per-scale upsample + warp, :1248-1475 per-scale loss / 2**scale, total / (sclm+1)):
    the glue restated here, every arithmetic step through the reference's own objects (``SSIM``,
    ``compute_reprojection_loss``, ``compute_loss_masks``, ``get_smooth_loss``, ``disp_to_depth``, ``BackprojectDepth``,
    ``Project3D``) and ATen's ``interpolate`` / ``grid_sample``.  ``temporal`` (--temporal on this path): the producer is
    called once per scale on the teacher's warp of that scale, has_ins is the last call's (trainer.py:1161-1162), and the
    teacher's min of every scale takes r(syn_f, target) in (:1279-1283); the producer is the rectangle stand-in of
    ``q["syn_rects"]`` (mal_amd.synthetic.fake_image_synthesis: Mask2Former is out of scope)."""
    from mal_amd.synthetic import to_dicts, fake_image_synthesis
    from oracle.mal_oracle import default_opt
    B, _, H, W = q["color0"].shape
    opt = default_opt(height=H, width=W, batch_size=B, sclm=sclm, distil=False)
    inputs, mono_outputs, outputs, leaves = to_dicts(q, ML.transformation_from_parameters)
    synth = fake_image_synthesis(q["syn_rects"]) if temporal else None
    state = {"has_ins": False}
    color_s, disp_s = multiscale_inputs(q, sclm)
    for s in range(1, sclm + 1):
        inputs[("color", 0, s)] = color_s[s]
        for name, outs in (("disp_teacher", mono_outputs), ("disp_student", outputs)):
            leaf = disp_s[name][s].clone().requires_grad_(True)
            leaves["%s_s%d" % (name, s)] = leaf
            outs[("disp", s)] = leaf
    ssim = ML.SSIM()
    backproject, project = ML.BackprojectDepth(B, H, W), ML.Project3D(B, H, W)

    def gen_pred(outs, is_multi):  # trainer.py:1088-1125, not v1_multiscale: every scale is warped at full resolution
        for scale in range(sclm + 1):
            disp = F.interpolate(outs[("disp", scale)], [H, W], mode="bilinear", align_corners=False)
            _, depth = ML.disp_to_depth(disp, opt.min_depth, opt.max_depth)
            outs[("depth", 0, scale)] = depth
            for f in (-1, 1):
                T = outs[("cam_T_cam", 0, f)]
                if is_multi:
                    T = T.detach()
                pts = backproject(depth, inputs[("inv_K", 0)])
                grid = project(pts, inputs[("K", 0)], T)
                outs[("color", f, scale)] = F.grid_sample(inputs[("color", f, 0)], grid, padding_mode="border",
                                                          align_corners=True)
            if not is_multi and temporal:  # trainer.py:1161-1162
                state["has_ins"] = synth(inputs, outs, scale)

    def losses_of(outs, is_multi, noises):  # trainer.py:1248-1475
        losses, total = {}, 0
        target = inputs[("color", 0, 0)]
        for scale in range(sclm + 1):
            disp, color = outs[("disp", scale)], inputs[("color", 0, scale)]
            R = [MLU.compute_reprojection_loss(ssim, outs[("color", f, scale)], target) for f in (-1, 1)]
            if not is_multi and temporal and state["has_ins"]:  # trainer.py:1279-1283
                R += [MLU.compute_reprojection_loss(ssim, outs[("syn", f, scale)], target) for f in (-1, 1)]
            R = torch.cat(R, 1)
            I = torch.cat([MLU.compute_reprojection_loss(ssim, inputs[("color", f, 0)], target) for f in (-1, 1)], 1)
            ident, _ = torch.min(I, dim=1, keepdim=True)
            rp, _ = torch.min(R, dim=1, keepdim=True)
            ident = ident + noises[scale] * 0.00001
            mask = MLU.compute_loss_masks(rp, ident)
            cons = 0
            if is_multi:
                mask = torch.ones_like(mask)
                mask = mask * outs["consistency_mask"].unsqueeze(1)
                mask = mask * (1 - outs["augmentation_mask"][:B])
                cmask = (1 - mask).float()
            reproj = (rp * mask).sum() / (mask.sum() + 1e-7)
            if is_multi:
                multi_depth, mono_depth = outs[("depth", 0, scale)], outs[("mono_depth", 0, scale)].detach()
                cons = (torch.abs(multi_depth - mono_depth) * cmask).mean()
                losses["consistency_loss/%d" % scale] = cons
            losses["reproj_loss/%d" % scale] = reproj
            loss = reproj + cons
            mean_disp = disp.mean(2, True).mean(3, True)
            loss = loss + opt.disparity_smoothness * ML.get_smooth_loss(disp / (mean_disp + 1e-7), color) / (2 ** scale)
            total = total + loss
            losses["loss/%d" % scale] = loss
        losses["loss"] = total / (sclm + 1)
        return losses

    torch.manual_seed(noise_seed)
    noises_t = [torch.randn(B, 1, H, W) for _ in range(sclm + 1)]
    noises_s = [torch.randn(B, 1, H, W) for _ in range(sclm + 1)]  # drawn upstream too; dead for the student's value
    gen_pred(mono_outputs, False)
    lt = losses_of(mono_outputs, False, noises_t)
    for key in list(mono_outputs.keys()):
        if isinstance(key, tuple) and key[0] in ("depth", "disp"):
            outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
    gen_pred(outputs, True)
    ls = losses_of(outputs, True, noises_s)
    (lt["loss"] + ls["loss"]).backward()
    d = {"in/noise_seed": np.int64(noise_seed), "sclm": np.int64(sclm)}
    for k, v in lt.items():
        d["teacher/" + k] = np.float64(v.item())
    for k, v in ls.items():
        d["student/" + k] = np.float64(v.item())
    for k, t in leaves.items():
        d["grad/" + k] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
    for s in range(1, sclm + 1):
        for name in ("disp_teacher", "disp_student"):
            d["in/%s_s%d" % (name, s)] = disp_s[name][s].half().numpy()
    d.update(pack_inputs(q))
    d["opt"] = np.array(repr(sorted(dict({"sclm": sclm, "distil": False}, **({"temporal": True} if temporal else {})).items())))
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "teacher", d["teacher/loss"], "student", d["student/loss"])


def run_reference_layers(ML, MLU, DL, tag, B=2, H=24, W=40, seed=77):
    from mal_amd.synthetic import make_batch
    q = quantize_batch(make_batch(B, H, W, seed))
    d = pack_inputs(q)
    g = torch.Generator().manual_seed(seed + 1)
    disp = q["disp_teacher"].clone().requires_grad_(True)
    sd, depth = ML.disp_to_depth(disp, 0.1, 100.0)
    d["scaled_disp"], d["depth"] = sd.detach().numpy(), depth.detach().numpy()
    for inv in (False, True):
        aa, tr = q["axisangle_m1"], q["translation_m1"]
        d["T_inv%d" % inv] = ML.transformation_from_parameters(aa, tr, invert=inv).numpy()
    d["rot"] = ML.rot_from_axisangle(q["axisangle_p1"]).numpy()
    d["trans"] = ML.get_translation_matrix(q["translation_p1"]).numpy()
    T = ML.transformation_from_parameters(q["axisangle_m1"], q["translation_m1"], invert=True).requires_grad_(True)
    pts = ML.BackprojectDepth(B, H, W)(depth, q["inv_K"])
    d["cam_points"] = pts.detach().numpy()
    grid, zc = ML.Project3D(B, H, W, dc=True)(pts, q["K"], T)
    d["grid_A"], d["proj_depth"] = grid.detach().numpy(), zc.detach().numpy()
    warped = F.grid_sample(q["color_m1"], grid, padding_mode="border", align_corners=True)
    d["warped_A"] = warped.detach().numpy()
    gw = torch.randn(warped.shape, generator=g)
    d["in/g_warped"] = gw.numpy()
    (warped * gw).sum().backward()
    d["grad_disp_A"], d["grad_T_A"] = disp.grad.numpy().copy(), T.grad.numpy().copy()
    # DualRefine convention (dualrefine/layers.py:216-226 + align_corners=False)
    disp2 = q["disp_teacher"].clone().requires_grad_(True)
    T2 = T.detach().clone().requires_grad_(True)
    _, depth2 = DL.disp_to_depth(disp2, 0.1, 100.0)
    pts2 = DL.BackprojectDepth(B, H, W)(depth2, q["inv_K"])
    grid2 = DL.Project3D(B, H, W)(pts2, q["K"], T2)
    warped2 = F.grid_sample(q["color_m1"], grid2, padding_mode="border", align_corners=False)
    d["grid_B"], d["warped_B"] = grid2.detach().numpy(), warped2.detach().numpy()
    (warped2 * gw).sum().backward()
    d["grad_disp_B"], d["grad_T_B"] = disp2.grad.numpy().copy(), T2.grad.numpy().copy()
    # photometric primitives
    x = warped.detach().clone().requires_grad_(True)
    y = q["color0"].clone().requires_grad_(True)
    ssim = ML.SSIM()
    s = ssim(x, y)
    d["ssim"] = s.detach().numpy()
    gs = torch.randn(s.shape, generator=g)
    d["in/g_ssim"] = gs.numpy()
    (s * gs).sum().backward()
    d["grad_ssim_x"], d["grad_ssim_y"] = x.grad.numpy().copy(), y.grad.numpy().copy()
    x2 = warped.detach().clone().requires_grad_(True)
    r = MLU.compute_reprojection_loss(ssim, x2, q["color0"])
    d["reproj"] = r.detach().numpy()
    gr = torch.randn(r.shape, generator=g)
    d["in/g_reproj"] = gr.numpy()
    (r * gr).sum().backward()
    d["grad_reproj_pred"] = x2.grad.numpy().copy()
    ident = MLU.compute_reprojection_loss(ssim, q["color_m1"], q["color0"])
    d["identity_reproj"] = ident.numpy()
    d["automask"] = MLU.compute_loss_masks(r.detach(), ident).numpy()
    d["automask_none"] = MLU.compute_loss_masks(r.detach(), None).numpy()
    disp3 = q["disp_student"].clone().requires_grad_(True)
    sm = ML.get_smooth_loss(disp3, q["color0"])
    sm.backward()
    d["smooth"], d["grad_smooth"] = np.float64(sm.item()), disp3.grad.numpy().copy()
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def run_reference_loss_balancing(MLU, tag="loss_balancing"):
    """a15: the reference's OWN ``LossBalancing`` (manydepth/loss_utils.py:283-345) on the CPU.  ``compute_loss`` cannot run
    there (hard-coded ``.cuda("cuda:0")``, :304), so its one side effect on the object -- the two loss scalars of a step
    written into ``train_scores`` for the step's bs records, :311-316 -- is applied directly, exactly as that loop writes them;
    ``update_weight`` (:320-345) is plain numpy and runs as is.  Recorded: a score sequence that takes the initialisation
    branch, ordinary re-weightings, both clamps (adjust term held at 2.0 and at 0.5), an update over a step that runs off
    the end of the dataset (records beyond ``num_data`` are never written), and the running state after every update."""
    bs, num_data, num_loss = 4, 30, 2
    lb = MLU.LossBalancing(num_loss, num_data, bs)
    rng = np.random.RandomState(20261005)
    # (L0, L1) per step and the lambda handed to update_weight after it (options.py: lambda_for_adjust_* ~ 0.1 .. 3)
    scores = np.stack([0.30 + 0.05 * rng.rand(8), 0.020 + 0.004 * rng.rand(8)], axis=1)
    scores[3] = (0.02, 0.30)    # the main term collapses: its adjust term would exceed 2.0 -> clamped to 2.0, the other to 0.5
    scores[5] = (3.00, 0.001)   # ... and the other way round
    lambdas = np.array([0.1, 0.3, 0.3, 3.0, 0.1, 3.0, 1.0, 0.2])
    w, prev_total, prev_loss = [], [], []
    for it in range(len(scores)):
        for b in range(bs):  # compute_loss's record loop (:306-316)
            rec = bs * it + b
            if rec < lb.num_data:
                lb.train_scores[rec, :] = scores[it]
        w.append(lb.update_weight(it, float(lambdas[it])))
        prev_total.append(float(lb.previous_total_loss))
        prev_loss.append(np.array(lb.previous_loss, dtype=np.float64))
    out = dict(bs=np.int64(bs), num_data=np.int64(num_data), num_loss=np.int64(num_loss), scores=scores, lambdas=lambdas,
               weights=np.array(w, dtype=np.float64), previous_total_loss=np.array(prev_total), previous_loss=np.stack(prev_loss),
               train_scores=np.array(lb.train_scores), initial_w=np.array([1.0 / num_loss] * num_loss))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print(tag, "weights", np.array(w)[:, 0].round(5).tolist(), np.array(w)[:, 1].round(5).tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    ML, MLU, DL = import_reference()
    from mal_amd.synthetic import make_batch
    torch.set_num_threads(8)
    small = quantize_batch(make_batch(2, 32, 64, seed=1234))
    if sys.argv[1:] == ["blc"]:  # `python -m oracle.gen_golden blc` writes that one case only
        run_reference_loss_balancing(MLU)
        return
    if sys.argv[1:] == ["ms_temporal"]:  # `python -m oracle.gen_golden ms_temporal` writes that one case only
        run_reference_multiscale(ML, MLU, quantize_batch(make_batch(2, 48, 96, seed=1238, with_syn=True)), 3, 1009,
                                 "multiscale_b2_48x96_sclm3_temporal", temporal=True)
        return
    if sys.argv[1:] != ["learnens"]:  # `python -m oracle.gen_golden learnens` writes that one case only
        run_reference_layers(ML, MLU, DL, "layers_b2_24x40")
        run_reference_layers(ML, MLU, DL, "layers_b1_19x33", B=1, H=19, W=33, seed=78)
        run_reference_step(ML, MLU, small, {}, 1000, True, "step_b2_32x64_distil")
        run_reference_step(ML, MLU, small, {"no_ens": True}, 1001, True, "step_b2_32x64_noens")
        run_reference_step(ML, MLU, small, {"loss_blc": True}, 1002, True, "step_b2_32x64_lossblc")
        run_reference_step(ML, MLU, small, {"no_ens": True, "dual_distil": True}, 1003, True, "step_b2_32x64_dual")
        syn = quantize_batch(make_batch(2, 32, 64, seed=1235, with_syn=True))
        run_reference_step(ML, MLU, syn, {"temporal": True}, 1004, True, "step_b2_32x64_temporal")
        run_reference_step(ML, MLU, syn, {"temporal": True, "main_temporal": True}, 1005, True,
                           "step_b2_32x64_temporal_main")
    only = sys.argv[1:]
    le = dict(small)  # --learn_ens (loss_utils.py:240-241, trainer.py:596-597): a third disparity map stands in for the head's output
    g = torch.Generator().manual_seed(99)
    le["disp_ens"] = (0.55 * small["disp_teacher"] + 0.45 * small["disp_student"] + 0.01 * torch.randn(small["disp_teacher"].shape, generator=g)).clamp(0.01, 0.99).half().float()
    run_reference_step(ML, MLU, le, {"learn_ens": True}, 1008, True, "step_b2_32x64_learnens")
    if only == ["learnens"]:
        return
    ragged = quantize_batch(make_batch(3, 37, 50, seed=1236))
    run_reference_step(ML, MLU, ragged, {}, 1006, True, "step_b3_37x50_distil")
    big = quantize_batch(make_batch(2, 192, 640, seed=1234))
    run_reference_step(ML, MLU, big, {}, 2000, False, "step_b2_192x640_distil")
    run_reference_multiscale(ML, MLU, quantize_batch(make_batch(2, 48, 96, seed=1237)), 3, 1007, "multiscale_b2_48x96_sclm3")
    run_reference_multiscale(ML, MLU, quantize_batch(make_batch(2, 48, 96, seed=1238, with_syn=True)), 3, 1009,
                             "multiscale_b2_48x96_sclm3_temporal", temporal=True)
    run_reference_loss_balancing(MLU)


if __name__ == "__main__":
    main()
