"""DualRefine's epipolar correlation lookup, forward, on the HIP kernels of
``csrc/mal_epipolar.hip``: the two objects ``DEQDepthPose`` calls in every fixed-point iteration
(dualrefine/networks/depth_pose.py:433-435), with the reference's names and signatures --

    Reprojections.depth2epipolarcoords(poses, depths)      dualrefine/networks/utils/utils.py:112-217
    CoordSampler.register / __call__(coords, levels, heads) dualrefine/networks/corr.py:6-50

Forward only (inference; training differentiates through both inside the DEQ solver -- their VJPs are not built yet,
so tensors that require grad are refused rather than silently detached).  No CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops


def _no_grad(*ts):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        raise L.MalError("mal_amd.epipolar is forward-only: call it under torch.no_grad() (the VJPs of the epipolar lookup "
                         "are not implemented)")


class Reprojections(torch.nn.Module):
    """utils.py:112-217 (the members the correlation lookup uses)."""

    def __init__(self, args):
        super().__init__()
        self.r = args.corr_radius
        self.delta = torch.nn.Parameter(torch.tensor([1.]))
        if not getattr(args, "disable_pose_updates", False):
            self.delta_p = torch.nn.Parameter(torch.tensor([1.]))
        self.num_depth_bins = 96
        self.args = args
        self.K = None

    def update_depth_bins(self, max_depth_bin, min_depth_bin, mean_depth_bin, median_depth_bin):
        self.max_depth_bin, self.min_depth_bin = max_depth_bin, min_depth_bin
        self.mean_depth_bin, self.median_depth_bin = mean_depth_bin, median_depth_bin

    def _reg_intrinsics(self, intrinsics):
        self.K = intrinsics
        self.fx, self.fy, self.cx, self.cy = intrinsics[:, 0, 0], intrinsics[:, 1, 1], intrinsics[:, 0, 2], intrinsics[:, 1, 2]

    def depth2epipolarcoords(self, poses, depths):
        """-> (coords (B,2,L,2r+1,h,w), max_dx (B,1,h,w), depths (B,1,L*(2r+1),h,w)), utils.py:180-217"""
        if getattr(self.args, "gap_factor", "depth") != "depth":
            raise NotImplementedError("--gap_factor minmax evaluates self.minmax(r), which does not exist upstream "
                                      "(dualrefine/networks/utils/utils.py:177,193)")
        if self.K is None:
            raise L.MalError("Reprojections: call _reg_intrinsics(K) first (depth_pose.py:471)")
        _no_grad(poses, depths, self.delta)
        d = ops._req(depths.detach(), "depths")
        B, _, h, w = d.shape
        dev = d.device
        T = ops._req(poses.detach().float().reshape(B, 16).contiguous(), "poses")
        K = ops._req(self.K.detach().float().reshape(B, 16).contiguous(), "K")
        Lv, d1 = self.args.num_levels, 2 * self.r + 1
        dd = float(F.softplus(self.delta.detach().float().cpu()))
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        coords, max_dx, ds = new(B, 2, Lv, d1, h, w), new(B, 1, h, w), new(B, 1, Lv * d1, h, w)
        p = ops._p
        L.check(L.load().mal_epipolar_coords(p(d), p(T), p(K), B, h, w, self.r, Lv, dd,
                                             float(self.args.gap_factor_depth_ratio), p(coords), p(max_dx), p(ds),
                                             ops._stream()), "mal_epipolar_coords")
        return coords, max_dx, ds


class CoordSampler(torch.nn.Module):
    """corr.py:6-50"""

    def __init__(self, args):
        super().__init__()
        self.args = args

    def register(self, fmap1, fmap2, num_levels=4):
        _no_grad(fmap1, fmap2)
        self.num_levels = num_levels
        self.fmap1 = ops._req(fmap1.detach().float(), "fmap1").clone()
        f2 = ops._req(fmap2.detach().float(), "fmap2").clone()
        self.f2_pyramid = [f2]
        for _ in range(num_levels - 1):  # corr.py:19-23 (a dense pooling: torch)
            f2 = F.avg_pool2d(f2, 2, stride=2)
            self.f2_pyramid.append(f2.contiguous())

    def _update_fmap1(self, fmap1):
        self.fmap1 = ops._req(fmap1.detach().float(), "fmap1").clone()

    def __call__(self, coords, num_levels=1, num_head=1):
        _no_grad(coords)
        c = ops._req(coords.detach(), "coords")
        B, two, n1, d1, h, w = c.shape
        C = self.fmap1.shape[1]
        if two != 2 or n1 != num_levels or tuple(self.fmap1.shape) != (B, C, h, w) or num_levels > len(self.f2_pyramid):
            raise L.MalError("CoordSampler: coords must be (B,2,num_levels,d,h,w) matching the registered feature maps")
        out = torch.empty(B, num_levels * num_head * d1, h, w, dtype=torch.float32, device=c.device)
        p = ops._p
        L.check(L.load().mal_coord_sample_l1(p(self.fmap1), L.ptr_array([p(f) for f in self.f2_pyramid[:num_levels]]), p(c),
                                             B, C, h, w, num_levels, d1, num_head, p(out), ops._stream()),
                "mal_coord_sample_l1")
        return out
