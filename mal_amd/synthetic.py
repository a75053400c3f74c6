"""Synthetic KITTI-shaped batches for the MAL loss path.

No dataset or checkpoint exists in the build or GPU containers, so tests, ``bench.py``
and the golden-vector script all draw from this generator.  Everything is produced on
the CPU from a private ``torch.Generator`` (seed given by the caller), so the global
RNG stream -- which the reference consumes for its tie-break noise
(manydepth/loss_utils.py:105-106) -- is left untouched.

Key/shape contract of the ``inputs`` dict: manydepth/datasets/mono_dataset.py:125-215
(``("color", f, 0)`` (B,3,H,W) in [0,1]; ``("K", 0)`` / ``("inv_K", 0)`` (B,4,4));
intrinsics: manydepth/datasets/kitti_dataset.py:26-29 (normalised fx 0.58, fy 1.92,
cx = cy = 0.5) scaled by the image size, ``inv_K = pinv(K)``
(mono_dataset.py:181-190).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def kitti_intrinsics(B, H, W, dtype=torch.float32):
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= W
    K[1, :] *= H
    inv_K = np.linalg.pinv(K)
    K = torch.from_numpy(K).to(dtype).unsqueeze(0).repeat(B, 1, 1).contiguous()
    inv_K = torch.from_numpy(inv_K).to(dtype).unsqueeze(0).repeat(B, 1, 1).contiguous()
    return K, inv_K


def _lowpass(g, B, C, H, W, factor=8):
    h, w = max(H // factor, 2), max(W // factor, 2)
    small = torch.rand(B, C, h, w, generator=g)
    return F.interpolate(small, size=(H, W), mode="bilinear", align_corners=False)


def _lowpass_randn(g, B, C, H, W, factor=8):
    h, w = max(H // factor, 2), max(W // factor, 2)
    small = torch.randn(B, C, h, w, generator=g)
    return F.interpolate(small, size=(H, W), mode="bilinear", align_corners=False)


def _shift(img, dx, dy):
    """Integer translate with edge replication."""
    B, C, H, W = img.shape
    xs = (torch.arange(W) - dx).clamp(0, W - 1)
    ys = (torch.arange(H) - dy).clamp(0, H - 1)
    return img[:, :, ys][:, :, :, xs]


def make_batch(B=12, H=192, W=640, seed=1234, with_syn=False):
    """Returns a dict of CPU fp32 tensors:

    color0, color_m1, color_p1 (B,3,H,W); K, inv_K (B,4,4);
    disp_teacher, disp_student (B,1,H,W) in (0,1);
    axisangle_m1/p1, translation_m1/p1 (B,1,3);
    consistency_mask (B,H,W) {0,1}; augmentation_mask (B,1,1,1) {0,1};
    lowest_cost (B,H,W) (a disparity, as the cost volume's argmin is);
    [syn_shift: list of rectangles used to fake the temporal-hint producer].
    """
    g = torch.Generator().manual_seed(int(seed))
    tex = _lowpass(g, B, 3, H, W, 8) * 0.7 + _lowpass(g, B, 3, H, W, 2) * 0.3
    color0 = (tex + 0.05 * torch.rand(B, 3, H, W, generator=g)).clamp(0, 1)
    dxm = int(torch.randint(2, 7, (1,), generator=g))
    dxp = int(torch.randint(2, 7, (1,), generator=g))
    color_m1 = (_shift(tex, dxm, 1) + 0.05 * torch.rand(B, 3, H, W, generator=g)).clamp(0, 1)
    color_p1 = (_shift(tex, -dxp, -1) + 0.05 * torch.rand(B, 3, H, W, generator=g)).clamp(0, 1)
    K, inv_K = kitti_intrinsics(B, H, W)
    disp_t = torch.sigmoid(1.2 * _lowpass_randn(g, B, 1, H, W, 8) - 2.2)
    disp_s = torch.sigmoid(1.2 * _lowpass_randn(g, B, 1, H, W, 8) - 2.2)
    # keep the student correlated with the teacher, as a trained pair would be
    disp_s = (0.7 * disp_t + 0.3 * disp_s).contiguous()
    batch = dict(
        color0=color0.contiguous(), color_m1=color_m1.contiguous(), color_p1=color_p1.contiguous(),
        K=K, inv_K=inv_K, disp_teacher=disp_t.contiguous(), disp_student=disp_s,
        axisangle_m1=0.01 * torch.randn(B, 1, 3, generator=g), translation_m1=0.05 * torch.randn(B, 1, 3, generator=g),
        axisangle_p1=0.01 * torch.randn(B, 1, 3, generator=g), translation_p1=0.05 * torch.randn(B, 1, 3, generator=g),
        consistency_mask=(torch.rand(B, H, W, generator=g) < 0.8).float(),
        augmentation_mask=(torch.rand(B, 1, 1, 1, generator=g) < 0.5).float(),
    )
    # cost-volume disparity: teacher's scaled disparity times a factor, so that the
    # matching mask (manydepth/trainer.py:1066-1076) is neither all-true nor all-false
    scaled = 0.01 + 9.99 * disp_t[:, 0]
    batch["lowest_cost"] = (scaled * torch.exp(0.6 * torch.randn(B, H, W, generator=g))).contiguous()
    if with_syn:
        rects = []
        for _ in range(3):
            y0 = int(torch.randint(0, max(H - 8, 1), (1,), generator=g))
            x0 = int(torch.randint(0, max(W - 12, 1), (1,), generator=g))
            hh = int(torch.randint(4, max(H // 3, 5), (1,), generator=g))
            ww = int(torch.randint(4, max(W // 4, 5), (1,), generator=g))
            sx = int(torch.randint(-8, 9, (1,), generator=g))
            sy = int(torch.randint(-3, 4, (1,), generator=g))
            rects.append((y0, x0, hh, ww, sy, sx))
        batch["syn_rects"] = rects
    return batch


def fake_image_synthesis(rects):
    """Stand-in for manydepth/dyn_utils.py:121-170 (the Mask2Former-driven producer is out
    of scope): returns ``synth(inputs, outputs, scale) -> has_ins`` which writes
    ``outputs[("syn", f, scale)]`` = the warped image with a few rectangles shifted,
    built with clone / slice-assign so it stays differentiable wrt the warped image the
    way the reference's ``where``/``clone`` composition is (dyn_utils.py:127-128,163-168).
    """

    def synth(inputs, outputs, scale):
        for f in (-1, 1):
            img = outputs[("color", f, scale)]
            syn = img.clone()
            H, W = img.shape[-2:]
            for (y0, x0, hh, ww, sy, sx) in rects:
                sy_, sx_ = (sy, sx) if f < 0 else (-sy, -sx)
                ys0, ys1 = max(y0, 0), min(y0 + hh, H)
                xs0, xs1 = max(x0, 0), min(x0 + ww, W)
                yd0, xd0 = ys0 + sy_, xs0 + sx_
                yd1, xd1 = ys1 + sy_, xs1 + sx_
                # clip destination to the image, shrink source accordingly
                cy0, cx0 = max(-yd0, 0), max(-xd0, 0)
                cy1, cx1 = max(yd1 - H, 0), max(xd1 - W, 0)
                if ys1 - cy1 <= ys0 + cy0 or xs1 - cx1 <= xs0 + cx0:
                    continue
                syn[:, :, yd0 + cy0:yd1 - cy1, xd0 + cx0:xd1 - cx1] = img[:, :, ys0 + cy0:ys1 - cy1, xs0 + cx0:xs1 - cx1]
            outputs[("syn", f, scale)] = syn
        return True

    return synth


class _Instances:
    """the slice of detectron2's ``Instances`` that ``image_synthesis`` touches (scores, pred_masks, len, indexing)"""

    def __init__(self, scores, masks):
        self.scores, self.pred_masks = scores, masks

    def __len__(self):
        return len(self.scores)
    def __getitem__(self, sel):
        if torch.is_tensor(sel) and sel.dtype == torch.bool and not sel.is_cuda and bool(sel.all()):
            return self  # every instance kept: no device-side selection (a step stays capturable in a HIP graph)
        return _Instances(self.scores[sel], self.pred_masks[sel])


def instance_stub(B, H, W, n_inst=3, seed=0, device="cpu"):
    """Stand-ins for the two external models of the temporal-hint producer -- the Mask2Former segmenter and the
    Hungarian matcher (manydepth/dyn_utils.py:172-190, matcher.py:89-173) -- so that ``dyn_utils.image_synthesis``
    itself can be driven on synthetic data: every sample has ``n_inst`` confident instances (elliptic blobs) that
    moved by a few pixels between the two warped frames and are all matched.  Returns ``(ins_model, matcher)``.
    All tensors live on ``device`` already (nothing is copied from the host inside a step)."""
    g = torch.Generator().manual_seed(int(seed))
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    masks = []
    for b in range(B):
        pair = [[], []]
        for _ in range(n_inst):
            cy = float(torch.randint(H // 5, max(4 * H // 5, H // 5 + 1), (1,), generator=g))
            cx = float(torch.randint(W // 8, max(7 * W // 8, W // 8 + 1), (1,), generator=g))
            ry = float(torch.randint(max(H // 16, 2), max(H // 6, 3), (1,), generator=g))
            rx = float(torch.randint(max(W // 32, 2), max(W // 10, 3), (1,), generator=g))
            sx = float(torch.randint(-8, 9, (1,), generator=g))
            sy = float(torch.randint(-3, 4, (1,), generator=g))
            for f, sgn in ((0, -0.5), (1, 0.5)):
                pair[f].append((((yy - cy - sgn * sy) / ry) ** 2 + ((xx - cx - sgn * sx) / rx) ** 2) <= 1.0)
        masks.append((torch.stack(pair[0]).to(device), torch.stack(pair[1]).to(device)))
    scores = torch.full((n_inst,), 0.9)  # host side, as the thresholding of dyn_utils.py:133 is a host decision
    empty = torch.zeros(n_inst, H, W, dtype=torch.bool, device=device)
    every = torch.arange(n_inst, device=device)
    state = {"b": 0}

    def ins_model(images):
        if images.shape[0] != 2:  # the target frames: only the scores are read (dyn_utils.py:131-133)
            state["b"] = 0
            return [{"instances": _Instances(scores, empty)} for _ in range(images.shape[0])]
        b = state["b"] % B        # the (warped last, warped next) pair of the next sample, in batch order
        state["b"] += 1
        return [{"instances": _Instances(scores, masks[b][0])}, {"instances": _Instances(scores, masks[b][1])}]

    def matcher(ins_last, ins_next, cur):
        return every, every

    return ins_model, matcher


def to_dicts(batch, pose_fn, device=None, requires_grad=True):
    """Arrange a ``make_batch`` result into the reference's dict contract.

    ``pose_fn(axisangle, translation, invert)`` builds the 4x4 (the caller passes its own
    ``transformation_from_parameters`` -- product, oracle or reference).  Returns
    (inputs, mono_outputs, outputs, leaves) where leaves are the tensors gradients are
    taken with respect to.
    """
    dev = device if device is not None else torch.device("cpu")
    mv = lambda t: t.to(dev).contiguous()
    inputs = {("color", 0, 0): mv(batch["color0"]), ("color", -1, 0): mv(batch["color_m1"]),
              ("color", 1, 0): mv(batch["color_p1"]), ("K", 0): mv(batch["K"]), ("inv_K", 0): mv(batch["inv_K"])}
    leaves = {}
    for k in ("disp_teacher", "disp_student", "axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1"):
        t = mv(batch[k]).clone()
        t.requires_grad_(requires_grad)
        leaves[k] = t
    T_m1 = pose_fn(leaves["axisangle_m1"], leaves["translation_m1"], True)
    T_p1 = pose_fn(leaves["axisangle_p1"], leaves["translation_p1"], False)
    mono_outputs = {("disp", 0): leaves["disp_teacher"], ("cam_T_cam", 0, -1): T_m1, ("cam_T_cam", 0, 1): T_p1}
    outputs = {("disp", 0): leaves["disp_student"], ("cam_T_cam", 0, -1): T_m1, ("cam_T_cam", 0, 1): T_p1,
               "consistency_mask": mv(batch["consistency_mask"]), "augmentation_mask": mv(batch["augmentation_mask"]),
               "lowest_cost": mv(batch["lowest_cost"])}
    if "disp_ens" in batch:  # --learn_ens: the ensemble head's disparity (trainer.py:596-597, loss_utils.py:240-241)
        leaves["disp_ens"] = mv(batch["disp_ens"]).clone().requires_grad_(requires_grad)
        outputs["ens_disp"] = leaves["disp_ens"]
    return inputs, mono_outputs, outputs, leaves
