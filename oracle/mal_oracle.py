"""CPU oracle for the MAL photometric-reprojection / motion-aware-loss hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mal_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the timed CPU baseline -- never as the product path.

What it is: a restatement, in our own words, of the algorithm the reference executes on
its PyTorch-CPU path for SURVEY.md section 8 rows a1..a17.  Every function cites the
reference lines it follows (paths relative to /root/reference).  The arithmetic the
reference delegates to PyTorch ATen (``F.grid_sample``, ``AvgPool2d``, ``ReflectionPad2d``,
``min``/``argmin``; PyTorch 2.10.0 in this image, the reference pins no version) is
called through the same ATen entry points here (``aten=True``, the default, also what
the CPU baseline times), and is restated independently as explicit index arithmetic in
``oracle/aten_restated.py`` (``aten=False``) so the two can be checked against each
other.

Parity status: PINNED.  The reference ships no tests or golden vectors for this path
(SURVEY.md section 4 / 8c); the oracle is pinned against outputs of the reference's own
functions (``manydepth.layers``, ``manydepth.loss_utils``, ``dualrefine.layers``) imported
from /root/reference in the authoring container by ``oracle/gen_golden.py``; the
resulting vectors are committed under ``tests/golden/`` and re-checked by
``tests/test_oracle_golden.py`` on every run (no reference needed at test time).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from . import aten_restated as AR

# ----------------------------------------------------------------------------------
# a1  disp -> depth
# ----------------------------------------------------------------------------------


def disp_to_depth(disp, min_depth, max_depth):
    """manydepth/layers.py:14-23 (== dualrefine/layers.py:17-26).

    scaled = 1/max + (1/min - 1/max) * disp ; depth = 1/scaled.
    """
    lo = 1 / max_depth
    hi = 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


# ----------------------------------------------------------------------------------
# a16  pose parameters -> 4x4
# ----------------------------------------------------------------------------------


def rot_from_axisangle(vec):
    """manydepth/layers.py:61-100.  vec (B,1,3) -> (B,4,4) Rodrigues rotation.

    angle = |vec|, axis = vec/(angle+1e-7); entries built from x,y,z,cos,sin, 1-cos.
    """
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    rows = [
        [x * xC + ca, xyC - zs, zxC + ys],
        [xyC + zs, y * yC + ca, yzC - xs],
        [zxC - ys, yzC + xs, z * zC + ca],
    ]
    B = vec.shape[0]
    R = torch.zeros(B, 4, 4, dtype=vec.dtype)
    for i in range(3):
        for j in range(3):
            R[:, i, j] = rows[i][j].reshape(B)
    R[:, 3, 3] = 1
    return R


def get_translation_matrix(t):
    """manydepth/layers.py:45-58.  (B,1,3) -> (B,4,4) identity with last column t."""
    B = t.shape[0]
    M = torch.zeros(B, 4, 4, dtype=t.dtype)
    for i in range(4):
        M[:, i, i] = 1
    M[:, :3, 3] = t.contiguous().view(B, 3)
    return M


def transformation_from_parameters(axisangle, translation, invert=False):
    """manydepth/layers.py:26-42.  M = T*R, or for invert: R^T * T(-t)."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


# ----------------------------------------------------------------------------------
# a2/a3/a4  back-projection, projection, sampling
# ----------------------------------------------------------------------------------


def pixel_grid(B, H, W, dtype=torch.float32):
    """manydepth/layers.py:149-161: homogeneous pixel coords [x, y, 1], row-major."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=dtype)], 0)
    return pix.unsqueeze(0).repeat(B, 1, 1)


def backproject_depth(depth, inv_K):
    """manydepth/layers.py:163-168.  depth (B,1,H,W), inv_K (B,4,4) -> (B,4,HW)."""
    B, _, H, W = depth.shape
    pix = pixel_grid(B, H, W, depth.dtype)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    return torch.cat([cam, torch.ones(B, 1, H * W, dtype=depth.dtype)], 1)


def project_3d(points, K, T, H, W, eps=1e-7, convention="manydepth", with_depth=False):
    """manydepth/layers.py:184-199 (convention "manydepth": grid = (u/(W-1)-0.5)*2, for
    align_corners=True sampling) and dualrefine/layers.py:216-226 (convention
    "dualrefine": grid = 2*(u+0.5)/W - 1, for align_corners=False sampling).
    Returns (B,H,W,2) [+ the projected z (B,1,H,W) when with_depth, layers.py:195-197].
    """
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
    if convention == "manydepth":
        gx = (pix[..., 0] / (W - 1) - 0.5) * 2
        gy = (pix[..., 1] / (H - 1) - 0.5) * 2
    elif convention == "dualrefine":
        gx = 2 * (pix[..., 0] + 0.5) / W - 1
        gy = 2 * (pix[..., 1] + 0.5) / H - 1
    else:
        raise ValueError(convention)
    grid = torch.stack([gx, gy], -1)
    if with_depth:
        return grid, cam[:, 2, :].unsqueeze(1).view(B, 1, H, W)
    return grid


def grid_sample_border(src, grid, convention="manydepth", aten=True):
    """manydepth/trainer.py:1122-1125 (padding_mode="border", align_corners=True) and
    dualrefine/trainer.py:444-447 (align_corners=False)."""
    ac = convention == "manydepth"
    if aten:
        return F.grid_sample(src, grid, padding_mode="border", align_corners=ac)
    return AR.grid_sample_bilinear_border(src, grid, align_corners=ac)


# ----------------------------------------------------------------------------------
# a7/a8/a9/a12  photometric primitives
# ----------------------------------------------------------------------------------

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def ssim(x, y, aten=True):
    """manydepth/layers.py:243-257.  3x3 reflect-padded box statistics per channel,
    out = clamp((1 - n/d)/2, 0, 1)."""
    if aten:
        xp, yp = F.pad(x, (1, 1, 1, 1), mode="reflect"), F.pad(y, (1, 1, 1, 1), mode="reflect")
        pool = lambda t: F.avg_pool2d(t, 3, 1)
    else:
        xp, yp = AR.reflection_pad1(x), AR.reflection_pad1(y)
        pool = AR.avg_pool3
    mu_x, mu_y = pool(xp), pool(yp)
    sig_x = pool(xp ** 2) - mu_x ** 2
    sig_y = pool(yp ** 2) - mu_y ** 2
    sig_xy = pool(xp * yp) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    d = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def compute_reprojection_loss(pred, target, no_ssim=False, aten=True, l1_sign=None):
    """manydepth/loss_utils.py:46-55 (== manydepth/trainer.py:1211-1223 with
    ``no_ssim``, dualrefine/trainer.py:487-499).  0.85*mean_c SSIM + 0.15*mean_c |t-p|.
    ``l1_sign`` (decision-forced parity tests): |t-p| is taken as s*(p-t) with the given sign(p-t)."""
    l1 = (torch.abs(target - pred) if l1_sign is None else l1_sign * (pred - target)).mean(1, True)
    if no_ssim:
        return l1
    return 0.85 * ssim(pred, target, aten=aten).mean(1, True) + 0.15 * l1


def compute_loss_masks(reprojection_loss, identity_reprojection_loss):
    """manydepth/loss_utils.py:27-44: automask = (argmin([reproj, identity]) == 0)."""
    if identity_reprojection_loss is None:
        return torch.ones_like(reprojection_loss)
    both = torch.cat([reprojection_loss, identity_reprojection_loss], 1)
    return (torch.argmin(both, 1, keepdim=True) == 0).float()


def get_smooth_loss(disp, img, signs=None):
    """manydepth/layers.py:210-223: edge-aware first-difference smoothness.  ``signs`` = (sx, sy) (decision-forced
    parity tests): |a-b| is taken as s*(a-b) with the given sign instead of re-deciding it where a ~ b."""
    dx = disp[:, :, :, :-1] - disp[:, :, :, 1:]
    dy = disp[:, :, :-1, :] - disp[:, :, 1:, :]
    dx, dy = (torch.abs(dx), torch.abs(dy)) if signs is None else (dx * signs[0], dy * signs[1])
    ix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    iy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (dx * torch.exp(-ix)).mean() + (dy * torch.exp(-iy)).mean()


def normalized_smooth_loss(disp, color, signs=None):
    """manydepth/loss_utils.py:119-121: disp / (mean_HW disp + 1e-7), then a12."""
    mean_disp = disp.mean(2, True).mean(3, True)
    return get_smooth_loss(disp / (mean_disp + 1e-7), color, signs)


def _draw_noise(shape, noise):
    """The reference draws the tie-break noise from the global CPU generator
    (manydepth/loss_utils.py:105-106,178).  ``noise`` overrides the draw for tests."""
    return torch.randn(shape) if noise is None else noise


# ----------------------------------------------------------------------------------
# a5/a6/a13  trainer glue (the reference's Trainer is not importable here: cv2, wandb,
# detectron2, torchmetrics and an absent vis.py -- restated from the text)
# ----------------------------------------------------------------------------------


def default_opt(**kw):
    """The option fields the hot path reads (manydepth/options.py:62-97,130-162,372)."""
    o = dict(height=192, width=640, batch_size=12, min_depth=0.1, max_depth=100.0,
             frame_ids=[0, -1, 1], sclm=0, v1_multiscale=False, disable_automasking=False,
             no_ssim=False, disparity_smoothness=1e-3, temporal=False, main_temporal=False,
             distil=True, no_ens=False, learn_ens=False, dual_distil=False, ensemble=False,
             loss_blc=False, pareto=False, disable_motion_masking=False,
             no_matching_augmentation=False, loss_pct=False)
    o.update(kw)
    return SimpleNamespace(**o)


def generate_images_pred(opt, inputs, outputs, is_multi=False, synth=None, aten=True, forced=None):
    """manydepth/trainer.py:1078-1170.

    Per scale: upsample disp to full res (bilinear, align_corners=False) unless
    v1_multiscale, a1, then for f in frame_ids[1:]: a2 -> a3 -> a4.  T is detached for
    the multi-frame (student) pass (:1107-1109).  ``synth(inputs, outputs, scale) ->
    bool`` stands in for dyn_utils.image_synthesis (:1161-1165); returns has_ins.
    ``forced`` (decision-forced parity tests): {"taps": {f: (x0, y0, clipx, clipy)}} -- the bilinear taps and border
    clips are taken as given (aten_restated.grid_sample_forced_taps) -- for scale 0, or a list of such dicts, one per scale
    (the four-scale path: every scale warps the full-resolution sources with its own upsampled disparity).
    """
    has_ins = False
    for scale in range(opt.sclm + 1):
        disp = outputs[("disp", scale)]
        if opt.v1_multiscale:
            src_scale = scale
        else:
            disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
            src_scale = 0
        _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outputs[("depth", 0, scale)] = depth
        H, W = depth.shape[-2:]
        for f in opt.frame_ids[1:]:
            T = outputs[("cam_T_cam", 0, f)]
            if is_multi:
                T = T.detach()
            pts = backproject_depth(depth, inputs[("inv_K", src_scale)])
            grid = project_3d(pts, inputs[("K", src_scale)], T, H, W)
            outputs[("sample", f, scale)] = grid
            if forced is not None:
                fs = forced[scale] if isinstance(forced, (list, tuple)) else forced
                outputs[("color", f, scale)] = AR.grid_sample_forced_taps(inputs[("color", f, src_scale)], grid,
                                                                          *fs["taps"][f])
            else:
                outputs[("color", f, scale)] = grid_sample_border(inputs[("color", f, src_scale)], grid, aten=aten)
            if not opt.disable_automasking:
                outputs[("color_identity", f, scale)] = inputs[("color", f, src_scale)]
        if synth is not None and ((not is_multi and opt.temporal) or (is_multi and opt.main_temporal)):
            has_ins = bool(synth(inputs, outputs, scale))
    return has_ins


def generate_images_pred_ensemble(opt, inputs, T_l, T_n, disp, aten=True):
    """manydepth/trainer.py:1172-1207: warp both sources with the given poses and
    the ensemble disparity, return min_f r(warp_f, target), (B,1,H,W)."""
    disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
    _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
    H, W = depth.shape[-2:]
    target = inputs[("color", 0, 0)]
    r = []
    for T, f in zip((T_l, T_n), opt.frame_ids[1:]):
        pts = backproject_depth(depth, inputs[("inv_K", 0)])
        grid = project_3d(pts, inputs[("K", 0)], T, H, W)
        pred = grid_sample_border(inputs[("color", f, 0)], grid, aten=aten)
        r.append(compute_reprojection_loss(pred, target, opt.no_ssim, aten=aten))
    return torch.min(torch.cat(r, 1), dim=1, keepdim=True)[0]


def compute_matching_mask(outputs):
    """manydepth/trainer.py:1066-1076: where cost-volume depth and teacher depth agree
    to within a factor two (both relative differences < 1)."""
    mono = outputs[("mono_depth", 0, 0)]
    matching = 1 / outputs["lowest_cost"].unsqueeze(1)
    mask = ((matching - mono) / mono) < 1.0
    mask = mask * (((mono - matching) / matching) < 1.0)
    return mask[:, 0]


# ----------------------------------------------------------------------------------
# a10/a11  MAL losses
# ----------------------------------------------------------------------------------


def _candidate_losses(inputs, outputs, with_syn, no_ssim=False, aten=True, l1_sign=None):
    """``l1_sign`` holds the signs of the WINNING candidate's L1 term; it is applied to every candidate, which is
    harmless because the forced argmin then selects the winner's value only."""
    target = inputs[("color", 0, 0)]
    cand = [compute_reprojection_loss(outputs[("color", f, 0)], target, no_ssim, aten, l1_sign) for f in (-1, 1)]
    if with_syn:
        cand += [compute_reprojection_loss(outputs[("syn", f, 0)], target, no_ssim, aten, l1_sign) for f in (-1, 1)]
    ident = [compute_reprojection_loss(inputs[("color", f, 0)], target, no_ssim, aten) for f in (-1, 1)]
    return torch.cat(cand, 1), torch.cat(ident, 1)


def compute_mono_losses(inputs, outputs, temporal, has_ins, noise=None, aten=True, forced=None):
    """manydepth/loss_utils.py:57-129 (teacher).  Returns (losses, min_c R (B,1,H,W)).
    ``forced`` (decision-forced parity tests): {"win": long (B,1,H,W), "automask": (B,1,H,W), "smooth": (sx, sy)} --
    the argmin over candidates, the automask comparison and the signs inside the smoothness term are taken as given."""
    R, I = _candidate_losses(inputs, outputs, bool(temporal and has_ins), aten=aten,
                             l1_sign=None if forced is None else forced.get("l1"))
    ident = torch.min(I, dim=1, keepdim=True)[0]
    rp = torch.min(R, dim=1, keepdim=True)[0] if forced is None else torch.gather(R, 1, forced["win"])
    ident = ident + _draw_noise(ident.shape, noise) * 0.00001
    mask = compute_loss_masks(rp, ident) if forced is None else forced["automask"].to(rp.dtype)
    reproj = (rp * mask).sum() / (mask.sum() + 1e-7)
    smooth = normalized_smooth_loss(outputs[("disp", 0)], inputs[("color", 0, 0)], None if forced is None else forced["smooth"])
    loss = reproj + 1e-3 * smooth / (2 ** 0)
    losses = {"reproj_loss/0": reproj, "loss/0": loss, "loss": loss}
    return losses, (torch.min(R, dim=1, keepdim=True)[0] if forced is None else rp)


def compute_main_losses(inputs, outputs, mono_reproj, ensemble_reproj, opt, w_list, multi_has_ins,
                        noise=None, aten=True, forced=None):
    """manydepth/loss_utils.py:131-281 (student).  Returns (losses, new_w_list, loss_list).

    The automask is computed and then replaced by ones (:181-192): the identity terms
    and the noise do not reach any value, but the CPU generator still advances (:178).
    The pareto branch (:256-265) needs manydepth/pareto.py, which upstream never
    committed: not restated.
    ``forced`` (decision-forced parity tests): {"win", "distil": long (B,1,H,W), "smooth": (sx, sy)}.
    """
    R, I = _candidate_losses(inputs, outputs, bool(multi_has_ins), aten=aten,
                             l1_sign=None if forced is None else forced.get("l1"))
    ident = torch.min(I, dim=1, keepdim=True)[0]
    rp = torch.min(R, dim=1, keepdim=True)[0] if forced is None else torch.gather(R, 1, forced["win"])
    multi_reproj = rp.clone()
    ident = ident + _draw_noise(ident.shape, noise) * 0.00001
    m = torch.ones_like(compute_loss_masks(rp, ident))
    m = m * outputs["consistency_mask"].unsqueeze(1)
    m = m * (1 - outputs["augmentation_mask"][:opt.batch_size])
    cmask = (1 - m).float()

    reproj = (rp * m).sum() / (m.sum() + 1e-7)

    multi_depth = outputs[("depth", 0, 0)]
    mono_depth = outputs[("mono_depth", 0, 0)].detach()
    consistency = (torch.abs(multi_depth - mono_depth) * cmask).mean()
    outputs["consistency_target/0"] = 1 / (mono_depth.detach() * cmask + multi_depth.detach() * (1 - cmask))

    losses = {"consistency_loss/0": consistency, "reproj_loss/0": reproj}
    loss = reproj + consistency
    loss = loss + 1e-3 * normalized_smooth_loss(outputs[("disp", 0)], inputs[("color", 0, 0)],
                                                None if forced is None else forced["smooth"]) / (2 ** 0)

    if ensemble_reproj is None:
        idx = torch.min(torch.cat([mono_reproj, multi_reproj], 1), dim=1, keepdim=True)[1]
        if forced is not None:
            idx = (forced["distil"] != 0).long()  # the kernels number the 2-way argmin 0 (teacher) / 2 (student)
        teacher = outputs[("mono_depth", 0, 0)] if opt.dual_distil else mono_depth
        distil_depth = torch.where(idx == 0, teacher, multi_depth)
    else:
        idx = torch.min(torch.cat([mono_reproj, ensemble_reproj, multi_reproj], 1), dim=1, keepdim=True)[1]
        if forced is not None:
            idx = forced["distil"]
        if opt.learn_ens:
            _, ens_depth = disp_to_depth(outputs["ens_disp"], opt.min_depth, opt.max_depth)
        else:
            ens_depth = (mono_depth + multi_depth) / 2.0
        distil_depth = torch.where(idx == 0, mono_depth, ens_depth)
        distil_depth = torch.where(idx == 2, multi_depth, distil_depth)
    distil = (torch.abs(distil_depth - multi_depth) * (1 - cmask)).mean()

    if opt.pareto:
        raise NotImplementedError("manydepth/pareto.py is absent upstream (loss_utils.py:3)")
    if opt.loss_blc:
        loss_list = [loss.clone(), distil]
        losses["distil_loss"] = distil
        new_w = w_list
    else:
        losses["distil_loss"] = distil
        loss = loss + distil
        new_w, loss_list = None, None
    losses["loss/0"] = loss
    losses["loss"] = loss
    return losses, new_w, loss_list


def compute_losses(opt, inputs, outputs, is_multi=False, has_ins=False, noises=None, aten=True, forced=None):
    """manydepth/trainer.py:1248-1475: the non-distillation fallback, looping scales
    0..sclm; per-scale loss / 2**scale for smoothness, total / (sclm+1).
    ``forced`` (decision-forced parity tests): one dict per scale, {"win": long (B,1,H,W), "l1": (B,3,H,W), "smooth": (sx, sy)
    at the scale's own size, and for the teacher "automask": (B,1,H,W)} -- the argmin over the candidates, the signs inside
    the winner's L1 term and inside the smoothness term, and the automask comparison are taken as given."""
    losses = {}
    total = 0
    for scale in range(opt.sclm + 1):
        src_scale = scale if opt.v1_multiscale else 0
        disp = outputs[("disp", scale)]
        color = inputs[("color", 0, scale)]
        target = inputs[("color", 0, src_scale)]
        fids = opt.frame_ids[1:]
        fd = None if forced is None else forced[scale]
        l1s = None if fd is None else fd.get("l1")  # the winner's signs, applied to every candidate: the forced argmin picks the winner's value only
        R = [compute_reprojection_loss(outputs[("color", f, scale)], target, opt.no_ssim, aten, l1s) for f in fids]
        if (not is_multi) and opt.temporal and has_ins:
            R += [compute_reprojection_loss(outputs[("syn", f, scale)], target, opt.no_ssim, aten, l1s) for f in fids]
        R = torch.cat(R, 1)
        I = torch.cat([compute_reprojection_loss(inputs[("color", f, src_scale)], target, opt.no_ssim, aten)
                       for f in fids], 1)
        ident = torch.min(I, dim=1, keepdim=True)[0]
        rp = torch.min(R, dim=1, keepdim=True)[0] if fd is None else torch.gather(R, 1, fd["win"])
        if not opt.disable_automasking:
            nz = None if noises is None else noises[scale]
            ident = ident + _draw_noise(ident.shape, nz) * 0.00001
        # NB (:1309-1310) the identity term is passed even with disable_automasking
        mask = compute_loss_masks(rp, ident) if (fd is None or is_multi) else fd["automask"].to(rp.dtype)
        if is_multi:
            mask = torch.ones_like(mask)
            if not opt.disable_motion_masking:
                mask = mask * outputs["consistency_mask"].unsqueeze(1)
            if not opt.no_matching_augmentation:
                mask = mask * (1 - outputs["augmentation_mask"][:opt.batch_size])
            cmask = (1 - mask).float()
        reproj = (rp * mask).sum() / (mask.sum() + 1e-7)
        consistency = 0
        ensemble = 0
        if is_multi:
            multi_depth = outputs[("depth", 0, scale)]
            mono_depth = outputs[("mono_depth", 0, scale)].detach()
            consistency = (torch.abs(multi_depth - mono_depth) * cmask).mean()
            outputs["consistency_target/{}".format(scale)] = 1 / (
                mono_depth.detach() * cmask + multi_depth.detach() * (1 - cmask))
            losses["consistency_loss/{}".format(scale)] = consistency
            if opt.ensemble:
                ens_t = (mono_depth + multi_depth) / 2.0
                ensemble = (torch.abs(ens_t - multi_depth) * mask).mean()
                losses["ensemble_loss/{}".format(scale)] = ensemble
        losses["reproj_loss/{}".format(scale)] = reproj
        loss = reproj + consistency + ensemble
        loss = loss + opt.disparity_smoothness * normalized_smooth_loss(disp, color, None if fd is None else fd["smooth"]) / (2 ** scale)
        total = total + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total / (opt.sclm + 1)
    return losses


# ----------------------------------------------------------------------------------
# a15  loss balancing (host)
# ----------------------------------------------------------------------------------


class LossBalancing:
    """manydepth/loss_utils.py:283-345.  ``compute_loss`` cannot run on CPU upstream
    (hard-coded .cuda("cuda:0") at :304, whose result is unused); restated from
    :303-318: the accumulation is inside the per-sample loop, so the returned scalar is
    bs * sum_i w_i * loss_i (for records inside the dataset)."""

    def __init__(self, num_loss, num_train_data, bs):
        self.num_loss = num_loss
        self.last_rebalancing_iter = 0
        self.previous_total_loss = 0
        self.previous_loss = 0
        self.w_list = np.array([1.0 / num_loss, 1.0 / num_loss])
        self.loss_initialize_scale = np.array([1.0 / num_loss, 1.0 / num_loss])
        self.train_scores = np.zeros((num_train_data, num_loss))
        self.num_data = num_train_data
        self.bs = bs
        self.weight_initialization = True
        self.weight_initialization_done = False

    def compute_loss(self, loss_list, index_iter):
        loss = 0
        for b in range(self.bs):
            rec = self.bs * index_iter + b
            if rec < self.num_data:
                for i in range(self.num_loss):
                    loss = loss + self.w_list[i] * loss_list[i]
                for i in range(self.num_loss):
                    self.train_scores[rec, i] = float(loss_list[i])
        return loss

    def update_weight(self, i, lam):
        window = self.train_scores[self.last_rebalancing_iter * self.bs:(i + 1) * self.bs, :].mean(axis=0)
        total = np.sum(window * self.w_list)
        if self.weight_initialization and not self.weight_initialization_done:
            for k in range(self.num_loss):
                self.w_list[k] = (total * self.loss_initialize_scale[k]) / window[k]
            self.weight_initialization_done = True
        else:
            prev_w = np.array(self.w_list)
            if self.previous_total_loss > 0:
                for k in range(self.num_loss):
                    adj = 1 + lam * ((total / self.previous_total_loss) * (self.previous_loss[k] / window[k]) - 1)
                    adj = min(max(adj, 0.5), 2.0)
                    self.w_list[k] = prev_w[k] * adj
        self.previous_total_loss = np.sum(window * self.w_list)
        self.previous_loss = window
        return self.w_list[0], self.w_list[1]


# ----------------------------------------------------------------------------------
# process_batch: passes A (teacher), B (ensemble), C (student)
# ----------------------------------------------------------------------------------


def mal_loss_step(opt, inputs, mono_outputs, outputs, noise_mono=None, noise_main=None, w_list=None,
                  synth=None, aten=True, freeze_tp=False, forced=None):
    """manydepth/trainer.py:573-629, the loss part of process_batch with --distil.

    ``mono_outputs`` / ``outputs`` hold what the networks would have produced
    (("disp",0), ("cam_T_cam",0,f), and for the student "consistency_mask",
    "augmentation_mask", "lowest_cost").  Returns (losses, loss_list, mono_losses,
    mono_reproj, ensemble_reproj).
    ``forced`` (decision-forced parity tests): {"teacher": {...}, "student": {...}, "cmask": (B,H,W)} -- every
    discontinuous per-pixel choice of the two gradient passes is taken from the caller (the HIP kernels' own
    decisions) instead of being re-decided; everything else is the same arithmetic.
    """
    ft, fs = (None, None) if forced is None else (forced["teacher"], forced["student"])
    has_ins = generate_images_pred(opt, inputs, mono_outputs, synth=synth, aten=aten, forced=ft)
    if not opt.temporal:
        has_ins = False
    mono_losses, mono_reproj = compute_mono_losses(inputs, mono_outputs, opt.temporal, has_ins, noise_mono, aten, ft)
    for key in list(mono_outputs.keys()):
        if isinstance(key, tuple) and key[0] in ("depth", "disp"):
            outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
    outputs["consistency_mask"] = (outputs["consistency_mask"] * compute_matching_mask(outputs) if forced is None
                                   else forced["cmask"].to(outputs["consistency_mask"].dtype))
    ensemble_reproj = None
    if not opt.no_ens:
        if opt.learn_ens:  # trainer.py:596-597: the learnt ensemble head's disparity, not detached
            disp_ens = outputs["ens_disp"]
        else:
            disp_ens = (mono_outputs[("disp", 0)].detach() + outputs[("disp", 0)].detach()) / 2.0
        ensemble_reproj = generate_images_pred_ensemble(
            opt, inputs, outputs[("cam_T_cam", 0, -1)].detach(), outputs[("cam_T_cam", 0, 1)].detach(), disp_ens, aten)
    multi_has_ins = generate_images_pred(opt, inputs, outputs, is_multi=True, synth=synth, aten=aten, forced=fs)
    if not opt.main_temporal:
        multi_has_ins = False
    losses, w_list, loss_list = compute_main_losses(inputs, outputs, mono_reproj, ensemble_reproj, opt, w_list,
                                                    multi_has_ins, noise_main, aten, fs)
    if not freeze_tp:
        for k, v in mono_losses.items():
            losses[k] = losses[k] + v
        if opt.loss_blc:
            loss_list[0] = loss_list[0] + mono_losses["loss"]
    return losses, loss_list, mono_losses, mono_reproj, ensemble_reproj


# ----------------------------------------------------------------------------------
# a17  DualRefine variant (4-tuple keys, align_corners=False convention)
# ----------------------------------------------------------------------------------


def dr_default_opt(**kw):
    o = dict(height=192, width=640, batch_size=8, min_depth=0.1, max_depth=100.0, frame_ids=[0, -1, 1],
             scales=[0], n_losses=1, v1_multiscale=False, disable_automasking=False, no_ssim=False,
             disparity_smoothness=1e-3, avg_reprojection=False, disable_motion_masking=False,
             Dstar_T0_pair=False, Tstar_D0_pair=False)
    o.update(kw)
    return SimpleNamespace(**o)


def dr_generate_images_pred(opt, inputs, outputs, aten=True, forced=None):
    """dualrefine/trainer.py:395-451 (the debug prints at :452-455 are not behaviour).
    ``forced`` (decision-forced parity tests): {(scale, it): {"taps": {f: (x0, y0, clipx, clipy)}, ...}}."""
    for scale in opt.scales:
        n = opt.n_losses + 1 if scale in (0, 1, 2) else 1
        for it in range(n):
            if scale == 1:
                continue
            disp = outputs[("disp", scale, it)]
            if not opt.v1_multiscale:
                disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
            _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
            outputs[("depth", 0, scale, it)] = depth
            H, W = depth.shape[-2:]
            for f in opt.frame_ids[1:]:
                if f == 1:
                    T = outputs[("cam_T_cam", 0, f)]
                    if it > 0:
                        T = T.detach()
                elif it > 0:
                    T = outputs[("cam_T_cam", 0, f)].detach() if opt.Dstar_T0_pair else outputs[("cam_T_cam", 0, f, 1)]
                else:
                    T = outputs[("cam_T_cam", 0, f)]
                pts = backproject_depth(depth, inputs[("inv_K", 0)])
                grid = project_3d(pts, inputs[("K", 0)], T, H, W, convention="dualrefine")
                outputs[("sample", f, scale, it)] = grid
                if forced is not None and (scale, it) in forced:
                    outputs[("color", f, scale, it)] = AR.grid_sample_forced_taps(
                        inputs[("color", f, 0)], grid, *forced[(scale, it)]["taps"][f], align_corners=False)
                else:
                    outputs[("color", f, scale, it)] = grid_sample_border(
                        inputs[("color", f, 0)], grid, convention="dualrefine", aten=aten)
                if not opt.disable_automasking:
                    outputs[("color_identity", f, scale, it)] = inputs[("color", f, 0)]


def dr_compute_losses(opt, inputs, outputs, noises=None, aten=True, forced=None):
    """dualrefine/trainer.py:530-633 (the f_thres > 0 branch: per (scale, deq_iter)).
    ``forced`` (decision-forced parity tests): {(scale, it): {"win", "automask", "l1"}} -- min over the two warped
    candidates, automask comparison and L1 signs taken as given (avg_reprojection has no argmin to force)."""
    losses = {}
    total = 0
    k = 0
    for scale in opt.scales:
        loss = 0
        n = opt.n_losses + 1 if scale in (0, 1, 2) else 1
        for it in range(n):
            if scale == 1:
                continue
            disp = outputs[("disp", scale, it)]
            color = inputs[("color", 0, scale)]
            target = inputs[("color", 0, 0)]
            fids = opt.frame_ids[1:]
            fd = None if forced is None else forced.get((scale, it))
            R = torch.cat([compute_reprojection_loss(outputs[("color", f, scale, it)], target, opt.no_ssim, aten,
                                                     None if fd is None else fd.get("l1")) for f in fids], 1)
            ident = None
            if not opt.disable_automasking:
                I = torch.cat([compute_reprojection_loss(inputs[("color", f, 0)], target, opt.no_ssim, aten)
                               for f in fids], 1)
                ident = I.mean(1, keepdim=True) if opt.avg_reprojection else torch.min(I, dim=1, keepdim=True)[0]
            rp = R.mean(1, keepdim=True) if opt.avg_reprojection else torch.min(R, dim=1, keepdim=True)[0]
            if fd is not None and not opt.avg_reprojection:
                rp = torch.gather(R, 1, fd["win"])
            if not opt.disable_automasking:
                nz = None if noises is None else noises[k]
                ident = ident + _draw_noise(ident.shape, nz) * 0.00001
            k += 1
            mask = compute_loss_masks(rp, ident) if fd is None else fd["automask"].to(rp.dtype)
            if it > 0:
                if not opt.disable_motion_masking:
                    mask = mask * outputs["consistency_mask"]
                cmask = (1 - mask).float()
            reproj = (rp * mask).sum() / (mask.sum() + 1e-7)
            consistency = 0
            if it > 0:
                multi_depth = outputs[("depth", 0, scale, it)]
                mono_depth = outputs[("depth", 0, scale, 0)].detach()
                consistency = (torch.abs(multi_depth - mono_depth) * cmask).mean()
                outputs["consistency_target/{}_{}".format(scale, it)] = 1 / (
                    mono_depth.detach() * cmask + multi_depth.detach() * (1 - cmask))
                losses["consistency_loss/{}_{}".format(scale, it)] = consistency
            losses["reproj_loss/{}".format(scale)] = reproj
            loss = loss + reproj + consistency
            loss = loss + opt.disparity_smoothness * normalized_smooth_loss(disp, color) / (2 ** scale)
            total = total + loss
            losses["loss/{}_{}".format(scale, it)] = loss
        # upstream's running `loss` is updated IN PLACE (`loss += ...`, :624,630), so every "loss/{scale}_{it}" entry of a
        # scale aliases one tensor and reads as the sum over the scale's iterations; `total_loss` took the value each
        # iteration had at the time (:631)
        for key in [k for k in losses if k.startswith("loss/{}_".format(scale))]:
            losses[key] = loss
    losses["loss"] = total / len(opt.scales)
    return losses


def dr_pose_update_generate_images_pred(opt, inputs, outputs, aten=True, forced=None):
    """dualrefine/trainer.py:457-480: frame -1 warped once more with the refined pose ``("cam_T_cam", 0, -1, 1)`` and the
    last iteration's depth (the iteration-0 depth, detached, with --Tstar_D0_pair).  The debug prints and ``exit(0)`` of
    :481-484 are not behaviour.  ``forced`` (decision-forced parity tests): {"taps": {-1: (x0, y0, clipx, clipy)}, ...}."""
    if getattr(opt, "Tstar_D0_pair", False):
        depth = outputs[("depth", 0, 0, 0)].clone().detach()
    else:
        depth = outputs[("depth", 0, 0, opt.n_losses)]
    H, W = depth.shape[-2:]
    pts = backproject_depth(depth, inputs[("inv_K", 0)])
    grid = project_3d(pts, inputs[("K", 0)], outputs[("cam_T_cam", 0, -1, 1)], H, W, convention="dualrefine")
    outputs[("sample", -1, 0, 0, 1)] = grid
    if forced is not None:
        outputs[("color", -1, 0, 0, 1)] = AR.grid_sample_forced_taps(inputs[("color", -1, 0)], grid, *forced["taps"][-1],
                                                                     align_corners=False)
    else:
        outputs[("color", -1, 0, 0, 1)] = grid_sample_border(inputs[("color", -1, 0)], grid, convention="dualrefine", aten=aten)


def dr_compute_pose_update_losses(opt, inputs, outputs, noise=None, aten=True, forced=None):
    """dualrefine/trainer.py:699-767: min (or mean, --avg_reprojection) over {frame -1 under the refined pose,
    frame +1 of iteration 0}, automask against the raw sources, masked mean; no smoothness / consistency term.
    ``forced`` (decision-forced parity tests): {"win", "automask", "l1"} as in dr_compute_losses."""
    target = inputs[("color", 0, 0)]
    R = []
    for f in opt.frame_ids[1:]:
        pred = outputs[("color", -1, 0, 0, 1)] if f == -1 else outputs[("color", f, 0, 0)]
        R.append(compute_reprojection_loss(pred, target, opt.no_ssim, aten, None if forced is None else forced.get("l1")))
    R = torch.cat(R, 1)
    ident = None
    if not opt.disable_automasking:
        I = torch.cat([compute_reprojection_loss(inputs[("color", f, 0)], target, opt.no_ssim, aten)
                       for f in opt.frame_ids[1:]], 1)
        ident = I.mean(1, keepdim=True) if opt.avg_reprojection else torch.min(I, dim=1, keepdim=True)[0]
    rp = R.mean(1, keepdim=True) if opt.avg_reprojection else torch.min(R, dim=1, keepdim=True)[0]
    if forced is not None and not opt.avg_reprojection:
        rp = torch.gather(R, 1, forced["win"])
    if not opt.disable_automasking:
        ident = ident + _draw_noise(ident.shape, noise) * 0.00001
    mask = compute_loss_masks(rp, ident) if forced is None else forced["automask"].to(rp.dtype)
    reproj = (rp * mask).sum() / (mask.sum() + 1e-7)
    return {"reproj_loss/pose_0": reproj, "loss/pose_0_0": reproj, "loss": reproj}
