"""torch.autograd bindings of the HIP kernels (host plumbing only; the arithmetic is in
libmal_hip.so).  Scalars produced by the kernels stay on the device: the backward
passes hand device pointers of ``grad_output`` scalars and of the f64 reduction results to
``mal_axpy_maps`` instead of reading them on the host (the reference's own host-sync
hazards are listed in SURVEY.md section 5).
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import ops

MANYDEPTH, DUALREFINE = 0, 1


def _scalar(t):
    """0-dim float32 contiguous device tensor usable as a device scalar pointer."""
    return t.detach().to(torch.float32).reshape(()).contiguous()


class DispToDepthFn(Function):
    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        ctx.save_for_backward(disp)
        ctx.rng = (min_depth, max_depth)
        return ops.disp_to_depth(disp, min_depth, max_depth)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_scaled, g_depth):
        (disp,) = ctx.saved_tensors
        return ops.disp_to_depth_bwd(disp, g_scaled, g_depth, *ctx.rng), None, None


class PoseFn(Function):
    """transformation_from_parameters (manydepth/layers.py:26-42) for one frame, one launch."""

    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        ctx.save_for_backward(axisangle, translation)
        ctx.invert = bool(invert)
        return ops.pose_fwd([axisangle], [translation], [invert])[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_T):
        axisangle, translation = ctx.saved_tensors
        g_aa, g_tr = ops.pose_bwd([axisangle], [translation], [ctx.invert], [g_T], [ctx.needs_input_grad[0]],
                                  [ctx.needs_input_grad[1]])
        return g_aa[0], g_tr[0], None


class BackprojectFn(Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        ctx.save_for_backward(inv_K)
        ctx.shape = depth.shape
        return ops.backproject(depth, inv_K)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_points):
        (inv_K,) = ctx.saved_tensors
        B, _, H, W = ctx.shape
        return ops.backproject_bwd(g_points, inv_K, B, H, W), None


class Project3DFn(Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, eps, convention, want_z):
        ctx.save_for_backward(points, K, T)
        ctx.cfg = (H, W, eps, convention)
        grid, z = ops.project3d(points, K, T, H, W, eps, convention, want_z)
        if z is None:
            z = points.new_empty(0)
            ctx.mark_non_differentiable(z)
        return grid, z

    @staticmethod
    @once_differentiable
    def backward(ctx, g_grid, g_z):
        points, K, T = ctx.saved_tensors
        H, W, eps, convention = ctx.cfg
        if g_z is not None and g_z.numel() == 0:
            g_z = None
        g_pts, g_T = ops.project3d_bwd(points, K, T, g_grid, g_z, H, W, eps, convention,
                                       need_points=ctx.needs_input_grad[0], need_T=ctx.needs_input_grad[2])
        return g_pts, None, g_T, None, None, None, None, None


class GridSampleFn(Function):
    """F.grid_sample(mode=bilinear, padding_mode=border); gradient wrt the grid only."""

    @staticmethod
    def forward(ctx, src, grid, align_corners):
        ctx.save_for_backward(src, grid)
        ctx.ac = align_corners
        return ops.grid_sample(src, grid, align_corners)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        src, grid = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise L.MalError("mal_amd.grid_sample: gradient wrt the sampled image is not part of the MAL loss path "
                             "(sources carry no gradient, manydepth/trainer.py:1122-1125)")
        return None, ops.grid_sample_bwd(src, grid, g_out, ctx.ac), None


class SSIMFn(Function):
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return ops.ssim(x, y)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return ops.ssim_bwd(x, y, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


class ReprojectionLossFn(Function):
    @staticmethod
    def forward(ctx, pred, target, no_ssim):
        ctx.save_for_backward(pred, target)
        ctx.no_ssim = no_ssim
        return ops.reprojection_loss(pred, target, no_ssim)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        gp, gt = ops.reprojection_loss_bwd(pred, target, g, ctx.no_ssim, ctx.needs_input_grad[0],
                                           ctx.needs_input_grad[1])
        return gp, gt, None


class SmoothLossFn(Function):
    """get_smooth_loss(disp[/mean], img) -> 0-dim float32; exact gradient wrt disp."""

    @staticmethod
    def forward(ctx, disp, img, normalise):
        loss, g = ops.smooth_loss(disp, img, normalise, disp.requires_grad)
        ctx.save_for_backward(g)
        return loss.to(torch.float32)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        (g,) = ctx.saved_tensors
        return ops.axpy_maps([g], scales=[_scalar(g_out)]), None, None


class WarpFn(Function):
    """disp, T_f -> depth, grid_f, warped_f  (Trainer.generate_images_pred, materialising)."""

    @staticmethod
    def forward(ctx, disp, K, inv_K, cfg, n_frames, *Ts_and_srcs):
        Ts, srcs = list(Ts_and_srcs[:n_frames]), list(Ts_and_srcs[n_frames:])
        ctx.save_for_backward(disp, K, inv_K, *Ts, *srcs)
        ctx.cfg, ctx.F = cfg, n_frames
        ctx.set_materialize_grads(False)
        depth, grids, warped = ops.warp_fwd(disp, K, inv_K, Ts, srcs, *cfg)
        return (depth, *grids, *warped)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_depth, *g_rest):
        F = ctx.F
        saved = ctx.saved_tensors
        disp, K, inv_K = saved[:3]
        Ts, srcs = list(saved[3:3 + F]), list(saved[3 + F:3 + 2 * F])
        g_grid, g_warped = list(g_rest[:F]), list(g_rest[F:2 * F])
        need_T = [ctx.needs_input_grad[5 + f] for f in range(F)]
        g_disp, g_T = ops.warp_bwd(disp, K, inv_K, Ts, srcs, g_warped, g_grid, g_depth, *ctx.cfg, need_T=need_T)
        return (g_disp, None, None, None, None, *g_T, *([None] * F))


class PhotoLossFn(Function):
    """Materialised candidates -> (sum(rp*w)/(sum(w)+1e-7), min_reproj map, weight map).

    compute_mono_losses / compute_main_losses' reprojection term on explicit images
    (manydepth/loss_utils.py:79-113,146-199); gradient wrt every candidate image.
    """

    @staticmethod
    def forward(ctx, target, ident, noise, ext_mask, flags, *cands):
        mn, am, wt, sums = ops.photo_fwd(target, cands, ident, noise, ext_mask, flags)
        ctx.save_for_backward(target, am, wt, sums, *cands)
        ctx.flags = flags
        ctx.set_materialize_grads(False)
        loss = ops.finish_scalars(sums[0:1], sums[1:2], 1e-7).reshape(())
        ctx.mark_non_differentiable(mn, wt)
        return loss, mn, wt

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, _g_mn, _g_wt):
        target, am, wt, sums = ctx.saved_tensors[:4]
        cands = ctx.saved_tensors[4:]
        need = [ctx.needs_input_grad[5 + i] for i in range(len(cands))]
        if g_loss is None:
            return (None,) * (5 + len(cands))
        g = ops.photo_bwd(target, cands, am, wt, _scalar(g_loss), sums, ctx.flags, need)
        return (None, None, None, None, None, *g)


class FusedPassFn(Function):
    """One launch of mal_pass_fused for a whole pass of manydepth/trainer.py:573-612.

    Inputs (tensors): disp, T_m1, T_p1, then constants.  Outputs:
      reproj  = sum(rp*w)/(sum(w)+1e-7)                      (differentiable)
      cons    = mean(|d_multi-d_mono|*(1-w))  (epilogue)     (differentiable)
      distil  = mean(|d_target-d_multi|*w)    (epilogue)     (differentiable)
      min_reproj, consistency_target, depth                  (not differentiable)
    """

    @staticmethod
    def forward(ctx, disp, T_m1, T_p1, K, inv_K, src_m1, src_p1, target, ident, noise, ext_mask, mono_depth,
                mono_reproj, ens_reproj, cfg, sample_scale=None):
        min_depth, max_depth, eps, convention, automask, epilogue, want_cons_target = cfg
        need_disp = disp.requires_grad
        need_T = T_m1.requires_grad or T_p1.requires_grad
        flags = 0
        if automask:
            flags |= L.F_AUTOMASK
        if epilogue:
            flags |= L.F_EPILOGUE
        if need_disp or need_T:
            flags |= L.F_GRAD
        if need_T:
            flags |= L.F_POSE_GRAD
        out = ops.pass_fused(disp, K, inv_K, [T_m1, T_p1], [src_m1, src_p1], target, ident, noise, ext_mask,
                             mono_depth, mono_reproj, ens_reproj, min_depth, max_depth, eps, convention, flags,
                             want_min=True, want_cons_target=bool(epilogue and want_cons_target), want_depth=False,
                             sample_scale=sample_scale)
        sums = out["sums"]
        B, _, H, W = disp.shape
        reproj = ops.finish_scalars(sums[0:1], sums[1:2], 1e-7).reshape(())
        if epilogue:
            means = ops.finish_scalars(sums[2:4], None, 0.0, 1.0 / float(B * H * W))
            cons, distil = means[0], means[1]
        else:
            cons = distil = torch.zeros((), dtype=torch.float32, device=disp.device)
        saved = [sums]
        for k in ("g_reproj", "g_cons", "g_distil"):
            saved.append(out[k] if out[k] is not None else disp.new_empty(0))
        saved += [g if g is not None else disp.new_empty(0) for g in out["g_T"]]
        ctx.save_for_backward(*saved)
        ctx.epilogue, ctx.n = bool(epilogue), B * H * W
        ctx.set_materialize_grads(False)
        mn = out["min_reproj"]
        ct = out["cons_target"] if out["cons_target"] is not None else disp.new_empty(0)
        ctx.mark_non_differentiable(mn, ct)
        return reproj, cons, distil, mn, ct

    @staticmethod
    @once_differentiable
    def backward(ctx, g_reproj, g_cons, g_distil, _g_mn, _g_ct):
        sums, G_r, G_c, G_d, gT0, gT1 = ctx.saved_tensors
        g_disp = g_T0 = g_T1 = None
        if ctx.needs_input_grad[0] and G_r.numel():
            maps, scales, denoms, mults, eps = [], [], [], [], []
            if g_reproj is not None:
                maps, scales, denoms, mults, eps = [G_r], [_scalar(g_reproj)], [sums[1:2]], [1.0], [1e-7]
            if ctx.epilogue:
                inv = 1.0 / float(ctx.n)
                for G, g in ((G_c, g_cons), (G_d, g_distil)):
                    if g is not None:
                        maps.append(G), scales.append(_scalar(g)), denoms.append(None), mults.append(inv), eps.append(0.0)
            g_disp = ops.axpy_maps(maps, scales, denoms, mults, eps) if maps else torch.zeros_like(G_r)
        if g_reproj is not None:
            if ctx.needs_input_grad[1] and gT0.numel():
                g_T0 = ops.axpy_maps([gT0], [_scalar(g_reproj)], [sums[1:2]], [1.0], [1e-7])
            if ctx.needs_input_grad[2] and gT1.numel():
                g_T1 = ops.axpy_maps([gT1], [_scalar(g_reproj)], [sums[1:2]], [1.0], [1e-7])
        return (g_disp, g_T0, g_T1) + (None,) * 13


class DistilFn(Function):
    """Consistency + distillation terms on materialised depth maps (loss_utils.py:193-254).  ``ens_depth`` (--learn_ens,
    :240-241): the learnt ensemble's depth, which then receives gradient where it wins the three-way min."""

    @staticmethod
    def forward(ctx, multi_depth, mono_depth, multi_reproj, mono_reproj, ens_reproj, ext_mask, dual_distil, ens_depth=None):
        flags = L.F_DUAL_DISTIL if dual_distil else 0
        need = multi_depth.requires_grad or (dual_distil and mono_depth.requires_grad) or \
            (ens_depth is not None and ens_depth.requires_grad)
        sums, g_cons, g_dist, g_third, ct = ops.distil_epilogue(multi_depth, mono_depth, multi_reproj, mono_reproj,
                                                                ens_reproj, ext_mask, flags, need_grad=need, ens_depth=ens_depth)
        n = multi_depth.numel()
        means = ops.finish_scalars(sums[2:4], None, 0.0, 1.0 / float(n))
        e = multi_depth.new_empty(0)
        ctx.save_for_backward(g_cons if g_cons is not None else e, g_dist if g_dist is not None else e,
                              g_third if g_third is not None else e)
        ctx.n = n
        ctx.learned = ens_depth is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(ct)
        return means[0], means[1], ct

    @staticmethod
    @once_differentiable
    def backward(ctx, g_c, g_d, _g_ct):
        G_c, G_d, G_3 = ctx.saved_tensors
        inv = 1.0 / float(ctx.n)
        g_multi = g_mono = g_ens = None
        terms = [(G, g) for G, g in ((G_c, g_c), (G_d, g_d)) if g is not None]
        if ctx.needs_input_grad[0] and terms and G_c.numel():
            g_multi = ops.axpy_maps([t[0] for t in terms], [_scalar(t[1]) for t in terms], None, [inv] * len(terms))
        if G_3.numel() and g_d is not None:
            third = ops.axpy_maps([G_3], [_scalar(g_d)], None, [inv])
            if ctx.learned and ctx.needs_input_grad[7]:
                g_ens = third
            elif not ctx.learned and ctx.needs_input_grad[1]:
                g_mono = third
        return g_multi, g_mono, None, None, None, None, None, g_ens
