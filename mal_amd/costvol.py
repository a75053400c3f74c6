"""ManyDepth's cost volume as MAL's student encoder builds it:
``ResnetEncoderMatching.match_features`` (manydepth/networks/resnet_encoder.py:152-233) and the lines of its
``forward`` that turn the volume into ``lowest_cost`` / ``confidence_mask`` (:296-312), as two HIP launches
(``mal_cost_volume``) instead of a Python loop over the batch with 96-fold feature replication.  Forward only,
as upstream (``torch.no_grad()``).  The kernels read the encoder's own NCHW feature maps.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops


def _run(current_feats, lookup_feats, relative_poses, K, invK, depth_bins, set_missing_to_max, want):
    cur = ops._req(current_feats.detach(), "current_feats")
    look = ops._req(lookup_feats.detach(), "lookup_feats")
    B, C, h, w = cur.shape
    if look.dim() != 5 or look.shape[0] != B or tuple(look.shape[2:]) != (C, h, w):
        raise L.MalError("lookup_feats must be (B,F,C,h,w) matching current_feats (B,C,h,w)")
    F_ = look.shape[1]
    dev = cur.device
    bins = torch.as_tensor(depth_bins, dtype=torch.float32).to(dev).contiguous().reshape(-1)
    D = bins.numel()
    if L.load().mal_costvol_channel_last():  # first formulation (mal_set_option("costvol_impl", 0)): relayout
        cl = cur.permute(0, 2, 3, 1).contiguous()
        ll = look.permute(0, 1, 3, 4, 2).contiguous()
    else:                                    # default: the kernels read the encoder's own (B,C,h,w) layout
        cl, ll = cur, ops._req(look, "lookup_feats")
    poses = ops._req(relative_poses.detach().reshape(B, F_, 16).contiguous(), "relative_poses")
    Kc, iKc = ops._req(K.detach().reshape(B, 16).contiguous(), "K"), ops._req(invK.detach().reshape(B, 16).contiguous(), "invK")
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    cv = new(B, D, h, w)
    miss = new(B, D, h, w) if want["missing"] else None
    masked = new(B, D, h, w) if want["masked"] else None
    low = new(B, h, w) if want["lowest"] else None
    conf = new(B, h, w) if want["confidence"] else None
    p = ops._p
    L.check(L.load().mal_cost_volume(p(cl), p(ll), p(poses), p(Kc), p(iKc), p(bins), B, F_, C, D, h, w, 1e-7,
                                     1 if set_missing_to_max else 0, p(cv), p(miss), p(masked), p(low), p(conf),
                                     ops._stream()), "mal_cost_volume")
    return cv, miss, masked, low, conf


def match_features(current_feats, lookup_feats, relative_poses, K, invK, depth_bins, set_missing_to_max=True):
    """resnet_encoder.py:152-233 -> (cost_volume (B,D,h,w), missing_mask (B,D,h,w)).  ``depth_bins``: the D depth
    hypotheses (``self.depth_bins`` upstream, from ``compute_depth_bins``)."""
    cv, miss, _, _, _ = _run(current_feats, lookup_feats, relative_poses, K, invK, depth_bins, set_missing_to_max,
                             dict(missing=True, masked=False, lowest=False, confidence=False))
    return cv, miss


def cost_volume_outputs(current_feats, lookup_feats, relative_poses, K, invK, depth_bins, set_missing_to_max=True):
    """match_features + :299-312 of the encoder's forward in the same two launches ->
    (cost_volume x confidence (B,D,h,w), lowest_cost (B,h,w), confidence_mask (B,h,w))."""
    _, _, masked, low, conf = _run(current_feats, lookup_feats, relative_poses, K, invK, depth_bins, set_missing_to_max,
                                   dict(missing=False, masked=True, lowest=True, confidence=True))
    return masked, low, conf
