"""CPU restatement of the temporal-hint producer's per-sample arithmetic
(manydepth/dyn_utils.py:6-119: ``fill_dynamic_obj`` and ``generate_dynamic_instance``).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under mal_amd/).  Pinned against the reference's
own TorchScript functions through tests/golden/dyn_*.npz (oracle/gen_golden_dyn.py).

Stated per output pixel p = (r, c) rather than as slice assignments:

    extents of an instance mask (dyn_utils.py:53-83): a row r counts as present iff r >= 1 and the mask has a
    pixel in it (the reference weighs the mask by the row index, so row 0 is invisible; likewise column 0);
    low / top = largest / smallest present row (0 if none), right / left for columns.
    delta (":86-93): of (low_next-low_last, top_next-top_last) the one of larger magnitude (the first on a
    tie), halved and rounded half-to-even; the same for columns; "last" moves by +delta, "next" by -delta;
    with replace=True displacements of magnitude < 3 are zeroed (":95-99).
    synthesis of "last" (":6-36, 106-119"):
        A(p)   = sum_i [p - d_i inside the image and mask_last_i(p - d_i)] * img_last(p - d_i)
        any(p) = or_i  [ ... ]
        bg(p)  = img_next(p) where or_i (mask_last_i & ~mask_next_i)(p), else img_last(p)
        ori_last(p) = (any(p) ? A(p) : bg(p)) where or_i (mask_last_i | mask_next_i)(p), else img_last(p)
    and symmetrically for "next" with -d_i, mask_next, and the roles of the images swapped.
"""
from __future__ import annotations

import torch


def extents(mask):
    """(num,H,W) bool -> int64 (num,4): low, top, right, left   (dyn_utils.py:53-83)"""
    num, H, W = mask.shape
    rows = mask.any(dim=2).clone()
    cols = mask.any(dim=1).clone()
    rows[:, 0] = False
    cols[:, 0] = False
    out = torch.zeros(num, 4, dtype=torch.int64)
    for i in range(num):
        r = torch.nonzero(rows[i]).flatten()
        c = torch.nonzero(cols[i]).flatten()
        if len(r):
            out[i, 0], out[i, 1] = r.max(), r.min()
        if len(c):
            out[i, 2], out[i, 3] = c.max(), c.min()
    return out


def deltas(mask_last, mask_next, replace):
    """-> (dx, dy) int64 (num,) each: the row / column displacement of the "last" copy (dyn_utils.py:86-103)"""
    el, en = extents(mask_last), extents(mask_next)

    def pick(a, b):  # the one of larger magnitude, the first on a tie; halved, half-to-even
        sel = torch.where(b.abs() > a.abs(), b, a)
        return torch.round(sel.to(torch.float32) / 2).long()

    dx = pick(en[:, 0] - el[:, 0], en[:, 1] - el[:, 1])
    dy = pick(en[:, 2] - el[:, 2], en[:, 3] - el[:, 3])
    if replace:
        dx = torch.where(dx.abs() < 3, torch.zeros_like(dx), dx)
        dy = torch.where(dy.abs() < 3, torch.zeros_like(dy), dy)
    return dx, dy


def _shift(t, dr, dc, fill):
    """t[..., r, c] -> out[..., r, c] = t[..., r - dr, c - dc] inside the image, `fill` outside"""
    H, W = t.shape[-2:]
    out = torch.full_like(t, fill)
    r0, r1 = max(0, dr), min(H, H + dr)
    c0, c1 = max(0, dc), min(W, W + dc)
    if r1 > r0 and c1 > c0:
        out[..., r0:r1, c0:c1] = t[..., r0 - dr:r1 - dr, c0 - dc:c1 - dc]
    return out


def _synth(mask, dx, dy, source, background, replaced_region):
    any_ = torch.zeros(mask.shape[1:], dtype=torch.bool)
    acc = torch.zeros_like(source)
    for i in range(mask.shape[0]):
        m = _shift(mask[i], int(dx[i]), int(dy[i]), False)
        acc = acc + m.unsqueeze(0) * _shift(source, int(dx[i]), int(dy[i]), 0.0)
        any_ = any_ | m
    syn = torch.where(any_, acc, background)
    return torch.where(replaced_region, syn, source)


def generate_dynamic_instance(mask_last, mask_next, img_last, img_next, replace=False):
    """masks (num,H,W) bool, images (C,H,W) -> (ori_last, ori_next); differentiable w.r.t. the images"""
    dx, dy = deltas(mask_last, mask_next, replace)
    region = (mask_last | mask_next).any(dim=0)
    bg_last = torch.where((mask_last & ~mask_next).any(dim=0), img_next, img_last)
    bg_next = torch.where((mask_next & ~mask_last).any(dim=0), img_last, img_next)
    ori_last = _synth(mask_last, dx, dy, img_last, bg_last, region)
    ori_next = _synth(mask_next, -dx, -dy, img_next, bg_next, region)
    return ori_last, ori_next


def image_synthesis(inputs, outputs, scale, thres, ins_model, matcher, replace=False):
    """manydepth/dyn_utils.py:121-170 restated on the CPU around ``generate_dynamic_instance`` above: the producer of the
    temporal hint as ``compute_mono_losses`` consumes it (loss_utils.py:84-88).  Same control flow: the segmenter is asked
    for the target frames' instances (only their scores are read, :131-133), then, per sample with a confident instance,
    for the (warped last, warped next) pair (:139-143); the matcher picks the instance rows (:144); samples without a
    match keep the warped images (:146-147).  ``ins_model(images) -> [{"instances": obj with .scores / .pred_masks}]`` and
    ``matcher(ins_last, ins_next, cur) -> (rows_last, rows_next)`` stand in for Mask2Former and the Hungarian matcher, as
    in ``mal_amd.dyn_utils.image_synthesis`` (the BGR x 255 conversion of :172-190 belongs to the segmenter's side).
    Differentiable w.r.t. ``outputs[("color", f, scale)]`` through clone / index assignment like upstream (:127-128,163-164).
    Writes ``outputs[("syn", f, scale)]`` when any sample matched; returns has_ins."""
    bs = inputs[("color", 0, 0)].shape[0]
    instances = ins_model(inputs[("color", 0, 0)])
    syn_last = outputs[("color", -1, scale)].clone()
    syn_next = outputs[("color", 1, scale)].clone()
    has_ins = False
    for b in range(bs):
        cur = instances[b]["instances"]
        instances_cur = cur[cur.scores > thres]
        if len(instances_cur) == 0:
            continue
        img_last = outputs[("color", -1, scale)][b].clone()
        img_next = outputs[("color", 1, scale)][b].clone()
        both = ins_model(torch.cat([img_last.detach().unsqueeze(0), img_next.detach().unsqueeze(0)], dim=0))
        ins_last, ins_next = both[0]["instances"], both[1]["instances"]
        slice_last, slice_next = matcher(ins_last, ins_next, instances_cur)
        if len(slice_last) + len(slice_next) == 0:
            continue
        has_ins = True
        mask_last = ins_last.pred_masks[slice_last].bool()
        mask_next = ins_next.pred_masks[slice_next].bool()
        tmp_last, tmp_next = generate_dynamic_instance(mask_last, mask_next, img_last, img_next, replace=replace)
        syn_last[b] = tmp_last
        syn_next[b] = tmp_next
    if has_ins:
        outputs[("syn", -1, scale)] = syn_last
        outputs[("syn", 1, scale)] = syn_next
    return has_ins
