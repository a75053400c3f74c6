"""Drop-in for the hot-path names of ``manydepth/layers.py`` (and ``dualrefine/layers.py``).

Same names, argument meaning and shapes as the reference, so its import lines
(manydepth/trainer.py:29-30, dualrefine/trainer.py:21-22) can point here unchanged; the
arithmetic runs in libmal_hip.so.  CPU tensors raise: there is no fallback path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn

__all__ = ["disp_to_depth", "depth_to_disp", "transformation_from_parameters", "get_translation_matrix",
           "rot_from_axisangle", "BackprojectDepth", "Project3D", "Project3DDualRefine", "grid_sample",
           "get_smooth_loss", "SSIM"]


def disp_to_depth(disp, min_depth, max_depth):
    """manydepth/layers.py:14-23 -> (scaled_disp, depth)."""
    return Fn.DispToDepthFn.apply(disp, float(min_depth), float(max_depth))


def depth_to_disp(depth, min_depth, max_depth):
    """dualrefine/layers.py:10-14 (host-trivial elementwise; plain tensor ops)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    return (1 / depth - min_disp) / (max_disp - min_disp)


def rot_from_axisangle(vec):
    """manydepth/layers.py:61-100.  (B,1,3) axis-angle -> (B,4,4).  A dozen scalar formulas
    on B values: kept as tensor ops on the caller's device (host-trivial,
    autograd to the pose network needed)."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = axis[..., 0:1], axis[..., 1:2], axis[..., 2:3]
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    zero, one = torch.zeros_like(ca), torch.ones_like(ca)
    rows = [x * xC + ca, xyC - zs, zxC + ys, zero,
            xyC + zs, y * yC + ca, yzC - xs, zero,
            zxC - ys, yzC + xs, z * zC + ca, zero,
            zero, zero, zero, one]
    return torch.cat(rows, 2).view(vec.shape[0], 4, 4)


def get_translation_matrix(translation_vector):
    """manydepth/layers.py:45-58."""
    B = translation_vector.shape[0]
    T = torch.eye(4, dtype=translation_vector.dtype, device=translation_vector.device).repeat(B, 1, 1)
    t = translation_vector.contiguous().view(B, 3, 1)
    return torch.cat([T[:, :, :3], torch.cat([t, T[:, 3:, 3:]], 1)], 2)


def transformation_from_parameters(axisangle, translation, invert=False):
    """manydepth/layers.py:26-42.  On the device: one HIP launch forward, one backward
    (mal_pose_fwd/bwd) instead of ~40 tiny tensor ops each way; on CPU tensors (host-side
    tools, tests) the same formulas as plain tensor ops."""
    if axisangle.is_cuda and axisangle.dim() == 3 and axisangle.shape[1:] == (1, 3):
        return Fn.PoseFn.apply(axisangle, translation, bool(invert))
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


class BackprojectDepth(nn.Module):
    """manydepth/layers.py:138-168.  Constructor arguments are accepted as upstream; the
    pixel grid is derived from thread indices in the kernel instead of a (B,3,HW) buffer."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        return Fn.BackprojectFn.apply(depth.reshape(-1, 1, self.height, self.width), inv_K)


class Project3D(nn.Module):
    """manydepth/layers.py:171-199 (``convention`` 0) / dualrefine/layers.py:204-226 (1)."""

    convention = Fn.MANYDEPTH

    def __init__(self, batch_size, height, width, dc=False, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.dc, self.eps = batch_size, height, width, dc, eps

    def forward(self, points, K, T):
        grid, z = Fn.Project3DFn.apply(points, K, T, self.height, self.width, float(self.eps), self.convention,
                                       bool(self.dc))
        return (grid, z) if self.dc else grid


class Project3DDualRefine(Project3D):
    convention = Fn.DUALREFINE

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__(batch_size, height, width, False, eps)


def grid_sample(input, grid, padding_mode="border", align_corners=True, mode="bilinear"):
    """The one F.grid_sample configuration the path uses (manydepth/trainer.py:1122-1125;
    dualrefine/trainer.py:444-447)."""
    if padding_mode != "border" or mode != "bilinear":
        raise ValueError("mal_amd.grid_sample implements mode='bilinear', padding_mode='border' only")
    return Fn.GridSampleFn.apply(input, grid, bool(align_corners))


def get_smooth_loss(disp, img):
    """manydepth/layers.py:210-223."""
    return Fn.SmoothLossFn.apply(disp, img, False)


class SSIM(nn.Module):
    """manydepth/layers.py:226-257."""

    def forward(self, x, y):
        return Fn.SSIMFn.apply(x, y)
