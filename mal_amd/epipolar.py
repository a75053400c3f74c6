"""DualRefine's epipolar correlation lookup, forward, on the HIP kernels of
``csrc/mal_epipolar.hip``: the two objects ``DEQDepthPose`` calls in every fixed-point iteration
(dualrefine/networks/depth_pose.py:433-435), with the reference's names and signatures --

    Reprojections.depth2epipolarcoords(poses, depths)      dualrefine/networks/utils/utils.py:112-217
    CoordSampler.register / __call__(coords, levels, heads) dualrefine/networks/corr.py:6-50

The correlation lookup (``depth2epipolarcoords`` + ``CoordSampler.__call__``) is differentiable (round 2: the DEQ solver
differentiates through both in training, depth_pose.py:426-455): autograd Functions over ``mal_epipolar_coords_bwd`` /
``mal_coord_sample_l1_bwd``.  The pose-refinement step (``depth2gradcoords``, ``direct_align``) and the masking lookup
are forward-only: tensors that require grad are refused there rather than silently detached.  No CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops


class EpipolarCoordsFn(torch.autograd.Function):
    """(depths (B,1,h,w), poses (B,4,4), K, softplus(delta) (1,)) -> (coords, max_dx, depth hypotheses);
    utils.py:180-217 and its VJP (d/d depths, d/d poses, d/d softplus(delta))."""

    @staticmethod
    def forward(ctx, depths, poses, K, dd_t, r, Lv, ratio):
        d = ops._req(depths.float(), "depths")
        B, _, h, w = d.shape
        dev = d.device
        T = ops._req(poses.float().reshape(B, 16).contiguous(), "poses")
        Kc = ops._req(K.float().reshape(B, 16).contiguous(), "K")
        dd = float(dd_t.detach().float().cpu())
        d1 = 2 * r + 1
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        coords, max_dx, ds = new(B, 2, Lv, d1, h, w), new(B, 1, h, w), new(B, 1, Lv * d1, h, w)
        p = ops._p
        L.check(L.load().mal_epipolar_coords(p(d), p(T), p(Kc), B, h, w, r, Lv, dd, float(ratio), p(coords), p(max_dx), p(ds),
                                             ops._stream()), "mal_epipolar_coords")
        ctx.save_for_backward(d, T, Kc)
        ctx.cfg = (r, Lv, dd, float(ratio), tuple(poses.shape), tuple(dd_t.shape), dd_t.device)
        ctx.set_materialize_grads(False)
        return coords, max_dx, ds

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_coords, g_max_dx, g_ds):
        d, T, Kc = ctx.saved_tensors
        r, Lv, dd, ratio, pshape, dshape, ddev = ctx.cfg
        B, _, h, w = d.shape
        dev = d.device
        if g_coords is None:
            g_coords = torch.zeros(B, 2, Lv, 2 * r + 1, h, w, dtype=torch.float32, device=dev)
        gc = ops._req(g_coords.float(), "g_coords")
        gm = None if g_max_dx is None else ops._req(g_max_dx.float(), "g_max_dx")
        gs = None if g_ds is None else ops._req(g_ds.float(), "g_depths")
        g_depth = torch.empty_like(d)
        g_poses = torch.empty(B, 16, dtype=torch.float32, device=dev)
        g_dd = torch.empty(1, dtype=torch.float32, device=dev)
        lib, p = L.load(), ops._p
        ws = torch.empty(lib.mal_epipolar_coords_bwd_workspace_bytes(B, h, w), dtype=torch.uint8, device=dev)
        L.check(lib.mal_epipolar_coords_bwd(p(d), p(T), p(Kc), p(gc), p(gm), p(gs), B, h, w, r, Lv, dd, ratio, p(g_depth),
                                            p(g_poses), p(g_dd), p(ws), ws.numel(), ops._stream()), "mal_epipolar_coords_bwd")
        return g_depth, g_poses.reshape(pshape), None, g_dd.reshape(dshape).to(ddev), None, None, None


class CoordSampleFn(torch.autograd.Function):
    """(fmap1, coords, pyramid levels of fmap2...) -> the L1 correlation lookup (corr.py:25-50) and its VJP"""

    @staticmethod
    def forward(ctx, fmap1, coords, num_levels, num_head, *pyramid):
        f1 = ops._req(fmap1.float(), "fmap1")
        c = ops._req(coords.float(), "coords")
        pyr = [ops._req(f.float(), "fmap2 level") for f in pyramid]
        B, two, n1, d1, h, w = c.shape
        C = f1.shape[1]
        out = torch.empty(B, num_levels * num_head * d1, h, w, dtype=torch.float32, device=c.device)
        p = ops._p
        L.check(L.load().mal_coord_sample_l1(p(f1), L.ptr_array([p(f) for f in pyr]), p(c), B, C, h, w, num_levels, d1,
                                             num_head, p(out), ops._stream()), "mal_coord_sample_l1")
        ctx.save_for_backward(f1, c, *pyr)
        ctx.cfg = (num_levels, num_head)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        f1, c, *pyr = ctx.saved_tensors
        num_levels, num_head = ctx.cfg
        B, two, n1, d1, h, w = c.shape
        C = f1.shape[1]
        need = ctx.needs_input_grad
        g_f1 = torch.zeros_like(f1) if need[0] else None
        g_c = torch.zeros_like(c) if need[1] else None
        g_pyr = [torch.zeros_like(f) if need[4 + i] else None for i, f in enumerate(pyr)]
        p = ops._p
        L.check(L.load().mal_coord_sample_l1_bwd(p(f1), L.ptr_array([p(f) for f in pyr]), p(c), p(ops._req(g_out.float(), "g_out")),
                                                 B, C, h, w, num_levels, d1, num_head, p(g_f1),
                                                 L.ptr_array([p(g) for g in g_pyr]), p(g_c), ops._stream()),
                "mal_coord_sample_l1_bwd")
        return (g_f1, g_c, None, None, *g_pyr)


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
        return EpipolarCoordsFn.apply(depths, poses, self.K.detach(), F.softplus(self.delta.float()), self.r,
                                      self.args.num_levels, float(self.args.gap_factor_depth_ratio))


    def depth2gradcoords(self, poses, depths, intrinsics=None):
        """-> (c1 (B,2,1,5,h,w), X1 (B,4,h*w)), utils.py:219-236 (``intrinsics`` is unused upstream too: the registered
        ones are)"""
        if self.K is None:
            raise L.MalError("Reprojections: call _reg_intrinsics(K) first (depth_pose.py:471)")
        _no_grad(poses, depths)
        d = ops._req(depths.detach(), "depths")
        B, _, h, w = d.shape
        T = ops._req(poses.detach().float().reshape(B, 16).contiguous(), "poses")
        K = ops._req(self.K.detach().float().reshape(B, 16).contiguous(), "K")
        c_p = torch.empty(B, 2, 1, 5, h, w, dtype=torch.float32, device=d.device)
        P2 = torch.empty(B, 4, h * w, dtype=torch.float32, device=d.device)
        p = ops._p
        L.check(L.load().mal_epipolar_gradcoords(p(d), p(T), p(K), B, h, w, p(c_p), p(P2), ops._stream()),
                "mal_epipolar_gradcoords")
        return c_p, P2


    def depthbins2coords(self, poses, depths):
        """-> (coords (B,2,1,num_depth_bins,h,w), depth hypotheses (B,1,num_depth_bins,h,w)), utils.py:231-255.  The
        hypotheses are a few elementwise tensor ops (as upstream); their projection is the kernel."""
        if self.K is None:
            raise L.MalError("Reprojections: call _reg_intrinsics(K) first (depth_pose.py:471)")
        _no_grad(poses, depths)
        dep = ops._req(depths.detach(), "depths")
        B, _, h, w = dep.shape
        dev, n = dep.device, self.num_depth_bins
        a = self.args
        if getattr(a, "use_depth_bins_for_masking", False):
            d = torch.linspace(float(self.min_depth_bin), float(self.max_depth_bin), n, device=dev)
            d = d[None, None, :, None, None].repeat(B, 1, 1, h, w)
        else:
            lin = torch.linspace(0, 1, n, device=dev)
            depths_ = 8 * (dep - a.min_depth) + a.min_depth
            depths_ = torch.clamp(depths_, max=a.max_depth)
            lin_ = (depths_ - a.min_depth) / (dep - a.min_depth)
            lin = lin[None, None, :, None, None] * lin_[:, None]
            d = lin * (dep[:, None] - a.min_depth) + a.min_depth
        d = d.contiguous()
        T = ops._req(poses.detach().float().reshape(B, 16).contiguous(), "poses")
        K = ops._req(self.K.detach().float().reshape(B, 16).contiguous(), "K")
        coords = torch.empty(B, 2, 1, n, h, w, dtype=torch.float32, device=dev)
        p = ops._p
        L.check(L.load().mal_epipolar_coords_of_depths(p(d), p(T), p(K), B, n, h, w, p(coords), ops._stream()),
                "mal_epipolar_coords_of_depths")
        return coords, d


def se3_exp(vec):
    """dualrefine/layers.py:29-55 (a handful of 3x3 tensor ops on (B,6,1): plain torch on the device)"""
    rho, phi = vec[:, :3], vec[:, 3:]
    theta = torch.norm(phi, 2, 1, keepdim=True)
    a = phi / theta
    a_skew = torch.zeros((vec.shape[0], 3, 3), device=vec.device)
    a_skew[:, 0, 1] = -a[:, 2, 0]
    a_skew[:, 0, 2] = a[:, 1, 0]
    a_skew[:, 1, 0] = a[:, 2, 0]
    a_skew[:, 1, 2] = -a[:, 0, 0]
    a_skew[:, 2, 0] = -a[:, 1, 0]
    a_skew[:, 2, 1] = a[:, 0, 0]
    eye = torch.eye(3, device=vec.device).unsqueeze(0)
    aat = torch.bmm(a, a.permute(0, 2, 1))
    R = torch.cos(theta) * eye + (1 - torch.cos(theta)) * aat + torch.sin(theta) * a_skew
    J = (torch.sin(theta) / theta) * eye + (1 - (torch.sin(theta) / theta)) * aat + (1 - torch.cos(theta)) / theta * a_skew
    T = torch.eye(4, device=vec.device).unsqueeze(0).repeat(vec.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, -1:] = torch.bmm(J, rho.type(J.dtype))
    return T


class PoseUpdate(torch.nn.Module):
    """The geometric half of utils.py:258-407: ``compute_feat``, the pixel weights and ``direct_align`` (one
    feature-metric Gauss-Newton step).  The learned weight / feature heads of upstream's module (its ``weights`` and
    ``feats`` convolution stacks) are networks and stay with the caller: assign ``src_w`` / ``tgt_w`` (B,1,h,w), or call
    ``compute_uncertainty`` with ``--disable_fixed_pose_weight`` for ones, as upstream."""

    def __init__(self, args, inp_dim=None, norm_fn="batch"):
        super().__init__()
        self.args = args
        if getattr(args, "robust_pose_loss", False):
            raise NotImplementedError("--robust_pose_loss (utils.py:334-338) is not built")

    def compute_uncertainty(self, feats):
        if not getattr(self.args, "disable_fixed_pose_weight", False):
            raise NotImplementedError("the learned pose-weight head (utils.py:289-292) is a network: compute it with the "
                                      "caller's module and assign src_w / tgt_w")
        bsz, _, ht, wd = feats.shape
        self.src_w, self.tgt_w = feats.new_ones((bsz // 2, 1, ht, wd)), feats.new_ones((bsz // 2, 1, ht, wd))

    def compute_feat(self, fmap1, fmap2):
        _no_grad(fmap1, fmap2)
        self.src_feat, self.tgt_feat = fmap1.detach().float(), fmap2.detach().float()

    def normal_equations(self, calib_K, p2, P2, weight):
        """utils.py:303-355 -> H (B,6,6), b (B,6) in two launches"""
        _no_grad(p2, P2, weight, self.src_w, self.tgt_w)
        src, tgt = ops._req(self.src_feat, "src_feat"), ops._req(self.tgt_feat, "tgt_feat")
        B, C, h, w = src.shape
        dev = src.device
        sw, tw = ops._req(self.src_w.detach().float(), "src_w"), ops._req(self.tgt_w.detach().float(), "tgt_w")
        wt = ops._req(weight.detach().float(), "weight") if weight is not None else None
        K = ops._req(calib_K.detach().float().reshape(B, 16).contiguous(), "K")
        c, X1 = ops._req(p2.detach(), "p2"), ops._req(P2.detach(), "P2")
        if tuple(c.shape) != (B, 2, 1, 5, h, w) or tuple(X1.shape) != (B, 4, h * w):
            raise L.MalError("direct_align: p2 must be (B,2,1,5,h,w) and P2 (B,4,h*w) as depth2gradcoords returns them")
        H, b = torch.empty(B, 6, 6, dtype=torch.float32, device=dev), torch.empty(B, 6, dtype=torch.float32, device=dev)
        lib, p = L.load(), ops._p
        ws = torch.empty(lib.mal_direct_align_workspace_bytes(B, h, w), dtype=torch.uint8, device=dev)
        L.check(lib.mal_direct_align_normal_eq(p(src), p(tgt), p(sw), p(tw), p(wt), p(K), p(c), p(X1), B, C, h, w, p(H), p(b),
                                               p(ws), ws.numel(), ops._stream()), "mal_direct_align_normal_eq")
        return H, b

    def direct_align(self, poses, calib_K, p2, P2, weight):
        """utils.py:303-368 -> (new poses (B,4,4), update (B,6,1)): normal equations (two launches), then the 6x6 solve
        with upstream's fall-backs, se3_exp and the pose product in one more"""
        H, b = self.normal_equations(calib_K, p2, P2, weight)
        B = H.shape[0]
        T = ops._req(poses.detach().float().reshape(B, 16).contiguous(), "poses")
        new_poses = torch.empty(B, 4, 4, dtype=torch.float32, device=H.device)
        update = torch.empty(B, 6, 1, dtype=torch.float32, device=H.device)
        p = ops._p
        L.check(L.load().mal_direct_align_update(p(H), p(b), p(T), B, p(new_poses), p(update), ops._stream()),
                "mal_direct_align_update")
        return new_poses.type(poses.dtype), update


class CoordSampler(torch.nn.Module):
    """corr.py:6-50"""

    def __init__(self, args):
        super().__init__()
        self.args = args

    def register(self, fmap1, fmap2, num_levels=4):
        self.num_levels = num_levels
        self.fmap1 = ops._req(fmap1.float(), "fmap1")
        f2 = ops._req(fmap2.float(), "fmap2")
        self.f2_pyramid = [f2]
        for _ in range(num_levels - 1):  # corr.py:19-23 (a dense pooling: torch, differentiable)
            f2 = F.avg_pool2d(f2, 2, stride=2)
            self.f2_pyramid.append(f2.contiguous())

    def _update_fmap1(self, fmap1):
        self.fmap1 = ops._req(fmap1.float(), "fmap1")

    def __corr__(self, coords, num_levels=1, num_head=1):
        """corr.py:52-75: the mean over all channels = the lookup with one head"""
        return self(coords, num_levels, 1)

    def __call__(self, coords, num_levels=1, num_head=1):
        B, two, n1, d1, h, w = coords.shape
        C = self.fmap1.shape[1]
        if two != 2 or n1 != num_levels or tuple(self.fmap1.shape) != (B, C, h, w) or num_levels > len(self.f2_pyramid):
            raise L.MalError("CoordSampler: coords must be (B,2,num_levels,d,h,w) matching the registered feature maps")
        return CoordSampleFn.apply(self.fmap1, coords, num_levels, num_head, *self.f2_pyramid[:num_levels])
