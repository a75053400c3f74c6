"""CPU restatement of ManyDepth's cost-volume construction as MAL uses it
(manydepth/networks/resnet_encoder.py:121-233 ``compute_depth_bins`` / ``match_features`` and the lines of
``ResnetEncoderMatching.forward`` that consume them, :296-312; all under ``torch.no_grad()`` upstream).

TEST INFRASTRUCTURE ONLY.  ``manydepth.networks`` cannot be imported here (torchvision is absent), so this is a
restatement of those lines; the geometry inside it (``BackprojectDepth``, ``Project3D``) and ``F.grid_sample``
are the reference's own / ATen's, and tests/golden/costvol_*.npz pins the whole function to a run that uses
the reference's layer objects for that arithmetic (oracle/gen_golden_costvol.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mal_oracle as O


def depth_bins(min_depth_bin, max_depth_bin, num_bins, binning="linear"):
    """resnet_encoder.py:121-141"""
    if binning == "inverse":
        return torch.from_numpy((1 / np.linspace(1 / max_depth_bin, 1 / min_depth_bin, num_bins)[::-1]).copy()).float()
    if binning == "linear":
        return torch.linspace(float(min_depth_bin), float(max_depth_bin), num_bins)
    if binning == "log":
        base, it = np.log(min_depth_bin), np.log(max_depth_bin / min_depth_bin)
        return torch.exp(torch.tensor([base + it * i / num_bins for i in range(num_bins)], dtype=torch.float32))
    raise NotImplementedError(binning)


def match_features(current_feats, lookup_feats, relative_poses, K, invK, bins, set_missing_to_max=True,
                   backproject=None, project=None):
    """resnet_encoder.py:152-233.  current (B,C,h,w), lookup (B,F,C,h,w), poses (B,F,4,4), K/invK (B,4,4) at the
    matching resolution, bins (D,) -> (cost_volume (B,D,h,w), missing_mask (B,D,h,w)).
    ``backproject(depth, invK)`` / ``project(points, K, T)`` default to the restated layers; the golden generator
    passes the reference's layer objects instead."""
    B, C, h, w = current_feats.shape
    D = bins.numel()
    backproject = backproject or (lambda depth, ik: O.backproject_depth(depth, ik))
    project = project or (lambda pts, k, t: O.project_3d(pts, k, t, h, w))
    warp_depths = bins.view(D, 1, 1, 1).expand(D, 1, h, w).contiguous().float()
    vols, masks = [], []
    for b in range(B):
        cost = torch.zeros(D, h, w)
        counts = torch.zeros(D, h, w)
        world = backproject(warp_depths, invK[b:b + 1].expand(D, 4, 4))
        for f in range(lookup_feats.shape[1]):
            pose = relative_poses[b:b + 1, f]
            if pose.sum() == 0:  # a missing lookup frame
                continue
            feat = lookup_feats[b:b + 1, f].repeat(D, 1, 1, 1)
            pix = project(world, K[b:b + 1].expand(D, 4, 4), pose.expand(D, 4, 4))
            warped = F.grid_sample(feat, pix, padding_mode="zeros", mode="bilinear", align_corners=True)
            x = (pix[..., 0] / 2 + 0.5) * (w - 1)
            y = (pix[..., 1] / 2 + 0.5) * (h - 1)
            edge = ((x >= 2.0) * (x <= w - 2) * (y >= 2.0) * (y <= h - 2)).float()
            cur = torch.zeros_like(edge)
            cur[:, 2:-2, 2:-2] = 1.0
            diffs = torch.abs(warped - current_feats[b:b + 1]).mean(1) * (edge * cur)
            cost = cost + diffs
            counts = counts + (diffs > 0).float()
        cost = cost / (counts + 1e-7)
        missing = (cost == 0).float()
        if set_missing_to_max:
            cost = cost * (1 - missing) + cost.max(0)[0].unsqueeze(0) * missing
        vols.append(cost)
        masks.append(missing)
    return torch.stack(vols, 0), torch.stack(masks, 0)


def encoder_outputs(cost_volume, missing_mask, bins):
    """ResnetEncoderMatching.forward :299-312 -> (masked cost volume, lowest_cost (B,h,w), confidence_mask (B,h,w))"""
    D = bins.numel()
    confidence = ((cost_volume * (1 - missing_mask) > 0).sum(1) == D).float()
    viz = cost_volume.clone()
    viz[viz == 0] = 100
    argmin = torch.min(viz, 1)[1]
    lowest_cost = 1 / bins[argmin.reshape(-1)].reshape(argmin.shape)
    return cost_volume * confidence.unsqueeze(1), lowest_cost, confidence
