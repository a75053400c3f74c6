"""DualRefine's epipolar correlation lookup, forward, on the HIP kernels of
``csrc/mal_epipolar.hip``: the two objects ``DEQDepthPose`` calls in every fixed-point iteration
(dualrefine/networks/depth_pose.py:433-435), with the reference's names and signatures --

    Reprojections.depth2epipolarcoords(poses, depths)      dualrefine/networks/utils/utils.py:112-217
    CoordSampler.register / __call__(coords, levels, heads) dualrefine/networks/corr.py:6-50

The correlation lookup (``depth2epipolarcoords`` + ``CoordSampler.__call__``) and the pose-refinement step
(``depth2gradcoords`` + ``PoseUpdate.direct_align``, with or without ``--robust_pose_loss``) are differentiable (the last
unrolled solver step differentiates through all of them in training, depth_pose.py:426-455): autograd Functions over
``mal_epipolar_coords_bwd`` / ``mal_coord_sample_l1_bwd`` / ``mal_epipolar_gradcoords_bwd`` /
``mal_direct_align_normal_eq_bwd`` / ``mal_direct_align_update_bwd``.  The masking lookup (``depthbins2coords``, run
under no_grad upstream) is forward-only: tensors that require grad are refused there rather than silently detached.
No CPU fallback.
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
        lib, p = L.load(), ops._p
        go = ops._req(g_out.float(), "g_out")
        ws = torch.empty(lib.mal_coord_sample_l1_bwd_workspace_bytes(B), dtype=torch.uint8, device=c.device)
        L.check(lib.mal_coord_sample_l1_bwd(p(f1), L.ptr_array([p(f) for f in pyr]), p(c), p(go), B, C, h, w, num_levels, d1,
                                            num_head, p(g_f1), L.ptr_array([p(g) for g in g_pyr]), p(g_c), p(ws), ws.numel(),
                                            ops._stream()), "mal_coord_sample_l1_bwd")
        return (g_f1, g_c, None, None, *g_pyr)


class GradCoordsFn(torch.autograd.Function):
    """(depths (B,1,h,w), poses (B,4,4), K) -> (c_p (B,2,1,5,h,w), P2 (B,4,h*w)); utils.py:219-236 and its VJP"""

    @staticmethod
    def forward(ctx, depths, poses, K):
        d = ops._req(depths.float(), "depths")
        B, _, h, w = d.shape
        T = ops._req(poses.float().reshape(B, 16).contiguous(), "poses")
        Kc = ops._req(K.float().reshape(B, 16).contiguous(), "K")
        c_p = torch.empty(B, 2, 1, 5, h, w, dtype=torch.float32, device=d.device)
        P2 = torch.empty(B, 4, h * w, dtype=torch.float32, device=d.device)
        p = ops._p
        L.check(L.load().mal_epipolar_gradcoords(p(d), p(T), p(Kc), B, h, w, p(c_p), p(P2), ops._stream()),
                "mal_epipolar_gradcoords")
        ctx.save_for_backward(d, T, Kc)
        ctx.pshape = tuple(poses.shape)
        ctx.set_materialize_grads(False)
        return c_p, P2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_cp, g_P2):
        d, T, Kc = ctx.saved_tensors
        B, _, h, w = d.shape
        dev = d.device
        if g_cp is None and g_P2 is None:
            return None, None, None
        gc = torch.zeros(B, 2, 1, 5, h, w, dtype=torch.float32, device=dev) if g_cp is None else ops._req(g_cp.float(), "g_c_p")
        gP = None if g_P2 is None else ops._req(g_P2.float(), "g_P2")
        g_depth = torch.empty_like(d)
        g_poses = torch.empty(B, 16, dtype=torch.float32, device=dev)
        lib, p = L.load(), ops._p
        ws = torch.empty(lib.mal_epipolar_gradcoords_bwd_workspace_bytes(B, h, w), dtype=torch.uint8, device=dev)
        L.check(lib.mal_epipolar_gradcoords_bwd(p(d), p(T), p(Kc), p(gc), p(gP), B, h, w, p(g_depth), p(g_poses), p(ws),
                                                ws.numel(), ops._stream()), "mal_epipolar_gradcoords_bwd")
        return g_depth, g_poses.reshape(ctx.pshape), None


class NormalEquationsFn(torch.autograd.Function):
    """(src_feat, tgt_feat, src_w, tgt_w, weight or None, p2, P2, K, robust) -> (H (B,6,6), b (B,6)); utils.py:303-355"""

    @staticmethod
    def forward(ctx, src_feat, tgt_feat, src_w, tgt_w, weight, p2, P2, K, robust):
        src, tgt = ops._req(src_feat.float(), "src_feat"), ops._req(tgt_feat.float(), "tgt_feat")
        B, C, h, w = src.shape
        dev = src.device
        sw, tw = ops._req(src_w.float(), "src_w"), ops._req(tgt_w.float(), "tgt_w")
        wt = ops._req(weight.float(), "weight") if weight is not None else None
        Kc = ops._req(K.float().reshape(B, 16).contiguous(), "K")
        c, X1 = ops._req(p2.float(), "p2"), ops._req(P2.float(), "P2")
        if tuple(c.shape) != (B, 2, 1, 5, h, w) or tuple(X1.shape) != (B, 4, h * w):
            raise L.MalError("direct_align: p2 must be (B,2,1,5,h,w) and P2 (B,4,h*w) as depth2gradcoords returns them")
        H, b = torch.empty(B, 6, 6, dtype=torch.float32, device=dev), torch.empty(B, 6, dtype=torch.float32, device=dev)
        lib, p = L.load(), ops._p
        ws = torch.empty(lib.mal_direct_align_workspace_bytes(B, h, w), dtype=torch.uint8, device=dev)
        L.check(lib.mal_direct_align_normal_eq(p(src), p(tgt), p(sw), p(tw), p(wt), p(Kc), p(c), p(X1), B, C, h, w,
                                               1 if robust else 0, p(H), p(b), p(ws), ws.numel(), ops._stream()),
                "mal_direct_align_normal_eq")
        ctx.save_for_backward(src, tgt, sw, tw, wt, Kc, c, X1)
        ctx.robust = bool(robust)
        return H, b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_H, g_b):
        src, tgt, sw, tw, wt, Kc, c, X1 = ctx.saved_tensors
        B, C, h, w = src.shape
        need = ctx.needs_input_grad
        gH, gb = ops._req(g_H.float().reshape(B, 36).contiguous(), "g_H"), ops._req(g_b.float().contiguous(), "g_b")
        new = lambda t, zero=False: (torch.zeros_like(t) if zero else torch.empty_like(t))
        g_src = new(src) if need[0] else None
        g_tgt = new(tgt, True) if need[1] else None        # scattered into with atomics
        g_sw = new(sw) if need[2] else None
        g_tw = new(tw, True) if need[3] else None
        g_wt = new(wt) if (need[4] and wt is not None) else None
        g_c = new(c) if need[5] else None
        g_X1 = new(X1) if need[6] else None
        lib, p = L.load(), ops._p
        ws = torch.empty(lib.mal_direct_align_bwd_workspace_bytes(B, h, w), dtype=torch.uint8, device=src.device) if need[1] else None
        L.check(lib.mal_direct_align_normal_eq_bwd(p(src), p(tgt), p(sw), p(tw), p(wt), p(Kc), p(c), p(X1), p(gH), p(gb), B,
                                                   C, h, w, 1 if ctx.robust else 0, p(g_src), p(g_tgt), p(g_sw), p(g_tw),
                                                   p(g_wt), p(g_c), p(g_X1), p(ws), ws.numel() if ws is not None else 0,
                                                   ops._stream()),
                "mal_direct_align_normal_eq_bwd")
        return g_src, g_tgt, g_sw, g_tw, g_wt, g_c, g_X1, None, None


class AlignUpdateFn(torch.autograd.Function):
    """(H (B,6,6), b (B,6), poses (B,4,4)) -> (new_poses (B,4,4), update (B,6,1)); utils.py:357-368"""

    @staticmethod
    def forward(ctx, H, b, poses):
        B = H.shape[0]
        Hc, bc = ops._req(H.float().reshape(B, 36).contiguous(), "H"), ops._req(b.float().contiguous(), "b")
        T = ops._req(poses.float().reshape(B, 16).contiguous(), "poses")
        new_poses = torch.empty(B, 4, 4, dtype=torch.float32, device=Hc.device)
        update = torch.empty(B, 6, 1, dtype=torch.float32, device=Hc.device)
        p = ops._p
        L.check(L.load().mal_direct_align_update(p(Hc), p(bc), p(T), B, p(new_poses), p(update), ops._stream()),
                "mal_direct_align_update")
        ctx.save_for_backward(Hc, bc, T)
        ctx.set_materialize_grads(False)
        return new_poses, update

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_new, g_update):
        Hc, bc, T = ctx.saved_tensors
        B = Hc.shape[0]
        dev = Hc.device
        if g_new is None and g_update is None:
            return None, None, None
        gn = torch.zeros(B, 16, dtype=torch.float32, device=dev) if g_new is None else \
            ops._req(g_new.float().reshape(B, 16).contiguous(), "g_new_poses")
        gu = None if g_update is None else ops._req(g_update.float().reshape(B, 6).contiguous(), "g_update")
        g_H, g_b = torch.empty(B, 36, dtype=torch.float32, device=dev), torch.empty(B, 6, dtype=torch.float32, device=dev)
        g_T = torch.empty(B, 16, dtype=torch.float32, device=dev)
        p = ops._p
        L.check(L.load().mal_direct_align_update_bwd(p(Hc), p(bc), p(T), p(gn), p(gu), B, p(g_H), p(g_b), p(g_T), ops._stream()),
                "mal_direct_align_update_bwd")
        return g_H.reshape(B, 6, 6), g_b, g_T.reshape(B, 4, 4)


def _no_grad(*ts):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        raise L.MalError("the masking lookup (depthbins2coords) is forward-only: call it under torch.no_grad() as upstream does "
                         "(depth_pose.py:522); the correlation lookup and the pose-refinement step are differentiable")


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
        return GradCoordsFn.apply(depths, poses, self.K.detach())

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

    def compute_uncertainty(self, feats):
        if not getattr(self.args, "disable_fixed_pose_weight", False):
            raise NotImplementedError("the learned pose-weight head (utils.py:289-292) is a network: compute it with the "
                                      "caller's module and assign src_w / tgt_w")
        bsz, _, ht, wd = feats.shape
        self.src_w, self.tgt_w = feats.new_ones((bsz // 2, 1, ht, wd)), feats.new_ones((bsz // 2, 1, ht, wd))

    def compute_feat(self, fmap1, fmap2):
        self.src_feat, self.tgt_feat = fmap1.float(), fmap2.float()

    def normal_equations(self, calib_K, p2, P2, weight):
        """utils.py:303-355 -> H (B,6,6), b (B,6) in two launches (``--robust_pose_loss``: :344-355 inside the same)"""
        return NormalEquationsFn.apply(self.src_feat, self.tgt_feat, self.src_w, self.tgt_w, weight, p2, P2, calib_K.detach(),
                                       bool(getattr(self.args, "robust_pose_loss", False)))

    def direct_align(self, poses, calib_K, p2, P2, weight):
        """utils.py:303-368 -> (new poses (B,4,4), update (B,6,1)): normal equations (two launches), then the 6x6 solve
        with upstream's fall-backs, se3_exp and the pose product in one more; differentiable end to end"""
        H, b = self.normal_equations(calib_K, p2, P2, weight)
        new_poses, update = AlignUpdateFn.apply(H, b, poses)
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
