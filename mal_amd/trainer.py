"""The warp/loss methods of ``manydepth.trainer.Trainer`` (manydepth/trainer.py:555-644,
1066-1475) as a mixin with the same method names, arguments and ``outputs`` keys, so the
reference Trainer can inherit them in place of its own.

``MALLossPath`` needs ``self.opt`` (the fields listed in SURVEY.md section 5) and, for the
temporal hint, ``self.image_synthesis(inputs, outputs, scale) -> bool`` (upstream:
``dyn_utils.image_synthesis`` bound to the Mask2Former model, trainer.py:1161-1165; the
segmenter itself is out of scope).

``fuse`` (default True): ``generate_images_pred`` records what a pass needs in
``outputs[("mal_ctx", scale)]`` and materialises only ``("depth", 0, scale)``; the loss
functions then run the single fused kernel.  ``("sample", f, s)`` / ``("color", f, s)`` are
produced when ``fuse=False``, when the temporal hint needs them, or on demand with
``materialize_warps``.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import _lib as L
from . import config
from . import functional as Fn
from . import layers
from . import loss_utils
from . import ops


class WarpContext(SimpleNamespace):
    """Inputs of one warp pass: disp (full res), T[2], K, inv_K, depth range, convention."""


class MALLossPath:
    convention = Fn.MANYDEPTH
    fuse = True
    has_ins = False
    multi_has_ins = False
    freeze_tp = False
    w_list = None
    loss_blc = None

    # ---------------------------------------------------------------- warp
    def _full_res_disp(self, disp):
        if self.opt.v1_multiscale:
            return disp
        if tuple(disp.shape[-2:]) == (self.opt.height, self.opt.width):
            return disp  # bilinear, align_corners=False at scale 1 is the identity
        return F.interpolate(disp, [self.opt.height, self.opt.width], mode="bilinear", align_corners=False)

    def _needs_images(self, is_multi):
        if not self.fuse:
            return True
        return bool((not is_multi and self.opt.temporal) or (is_multi and self.opt.main_temporal))

    def generate_images_pred(self, inputs, outputs, is_multi=False):
        """manydepth/trainer.py:1078-1170."""
        opt = self.opt
        for scale in range(opt.sclm + 1):
            disp = self._full_res_disp(outputs[("disp", scale)])
            source_scale = scale if opt.v1_multiscale else 0
            fids = opt.frame_ids[1:]
            Ts = [outputs[("cam_T_cam", 0, f)] for f in fids]
            if is_multi:
                Ts = [t.detach() for t in Ts]  # don't update posenet from the multi-frame pass (:1107-1109)
            K, inv_K = inputs[("K", source_scale)], inputs[("inv_K", source_scale)]
            srcs = [inputs[("color", f, source_scale)] for f in fids]
            cfg = (float(opt.min_depth), float(opt.max_depth), 1e-7, self.convention)
            if self._needs_images(is_multi):
                res = Fn.WarpFn.apply(disp, K, inv_K, cfg, len(fids), *Ts, *srcs)
                outputs[("depth", 0, scale)] = res[0]
                for i, f in enumerate(fids):
                    outputs[("sample", f, scale)] = res[1 + i]
                    outputs[("color", f, scale)] = res[1 + len(fids) + i]
            else:
                _, outputs[("depth", 0, scale)] = layers.disp_to_depth(disp, opt.min_depth, opt.max_depth)
                outputs[("mal_ctx", scale)] = WarpContext(disp=disp, T=Ts, K=K, inv_K=inv_K, min_depth=cfg[0],
                                                          max_depth=cfg[1], eps=cfg[2], convention=cfg[3],
                                                          srcs=srcs, fids=fids)
            if not opt.disable_automasking:
                for f in fids:
                    outputs[("color_identity", f, scale)] = inputs[("color", f, source_scale)]
            if is_multi is False and opt.temporal:
                self.has_ins = self.image_synthesis(inputs, outputs, scale)
            if is_multi and opt.main_temporal:
                self.multi_has_ins = self.image_synthesis(inputs, outputs, scale)

    def materialize_warps(self, outputs, scale=0):
        """Produce ("sample", f, s) / ("color", f, s) for a pass recorded lazily (no grad)."""
        ctx = outputs.get(("mal_ctx", scale))
        if ctx is None:
            return
        with torch.no_grad():
            _, grids, warped = ops.warp_fwd(ctx.disp.detach(), ctx.K, ctx.inv_K, [t.detach() for t in ctx.T], ctx.srcs,
                                            ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention, want_depth=False)
        for i, f in enumerate(ctx.fids):
            outputs[("sample", f, scale)] = grids[i]
            outputs[("color", f, scale)] = warped[i]
        del outputs[("mal_ctx", scale)]

    def image_synthesis(self, inputs, outputs, scale):
        raise NotImplementedError("bind the temporal-hint producer (manydepth/dyn_utils.py:121-170) to "
                                  "self.image_synthesis; the Mask2Former segmenter is outside this package")

    def generate_images_pred_ensemble(self, inputs, T_l, T_n, disp, disp2=None):
        """manydepth/trainer.py:1172-1207 -> min_f r(warp_f, target), (B,1,H,W), no gradient.
        ``disp2``: when given the disparity is (disp+disp2)/2, formed inside the kernel (:598)."""
        opt = self.opt
        if tuple(disp.shape[-2:]) != (opt.height, opt.width):
            if disp2 is not None:
                disp, disp2 = (disp + disp2) / 2.0, None
            disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
        if disp2 is not None and getattr(opt, "no_ssim", False):
            disp, disp2 = (disp + disp2) / 2.0, None
        fids = opt.frame_ids[1:]
        srcs = [inputs[("color", f, 0)] for f in fids]
        with torch.no_grad():
            if getattr(opt, "no_ssim", False):
                _, _, warped = ops.warp_fwd(disp.detach(), inputs[("K", 0)], inputs[("inv_K", 0)],
                                            [T_l.detach(), T_n.detach()], srcs, float(opt.min_depth),
                                            float(opt.max_depth), 1e-7, self.convention, want_depth=False,
                                            want_grid=False)
                mn, _, _, _ = ops.photo_fwd(inputs[("color", 0, 0)], warped, flags=L.F_NO_SSIM, want_argmin=False,
                                            want_weight=False)
                return mn
            out = ops.pass_fused(disp.detach(), inputs[("K", 0)], inputs[("inv_K", 0)], [T_l.detach(), T_n.detach()],
                                 srcs, inputs[("color", 0, 0)], min_depth=float(opt.min_depth),
                                 max_depth=float(opt.max_depth), eps=1e-7, convention=self.convention, flags=0,
                                 disp2=None if disp2 is None else disp2.detach())
        return out["min_reproj"]

    # ---------------------------------------------------------------- small ops
    def compute_reprojection_loss(self, pred, target):
        """manydepth/trainer.py:1211-1223."""
        return loss_utils.compute_reprojection_loss(None, pred, target, getattr(self.opt, "no_ssim", False))

    @staticmethod
    def compute_loss_masks(reprojection_loss, identity_reprojection_loss):
        """manydepth/trainer.py:1225-1243."""
        return loss_utils.compute_loss_masks(reprojection_loss, identity_reprojection_loss)

    def compute_matching_mask(self, outputs):
        """manydepth/trainer.py:1066-1076 -> (B,H,W) {0,1} float mask."""
        return ops.matching_mask(outputs["lowest_cost"], outputs[("mono_depth", 0, 0)].detach()[:, 0].contiguous())

    # ---------------------------------------------------------------- non-distillation losses
    def compute_losses(self, inputs, outputs, is_multi=False, noises=None):
        """manydepth/trainer.py:1248-1475 (the ``not opt.distil`` fallback, scales 0..sclm)."""
        opt = self.opt
        losses = {}
        total = 0
        no_ssim = bool(getattr(opt, "no_ssim", False))
        for scale in range(opt.sclm + 1):
            source_scale = scale if opt.v1_multiscale else 0
            disp = outputs[("disp", scale)]
            color = inputs[("color", 0, scale)]
            target = inputs[("color", 0, source_scale)]
            fids = opt.frame_ids[1:]
            sources = [inputs[("color", f, source_scale)] for f in fids]
            B, _, H, W = target.shape
            with_syn = bool((not is_multi) and opt.temporal and self.has_ins)
            ident = noise = ext = None
            flags = L.F_NO_SSIM if no_ssim else 0
            if is_multi:
                m = torch.ones(B, 1, H, W, dtype=torch.float32, device=target.device)
                if not opt.disable_motion_masking:
                    m = m * outputs["consistency_mask"].unsqueeze(1)
                if not opt.no_matching_augmentation:
                    m = m * (1 - outputs["augmentation_mask"][:opt.batch_size])
                ext = m.contiguous()
                if not opt.disable_automasking and config.noise_source == "cpu" and noises is None:
                    torch.randn((B, 1, H, W))  # drawn and discarded upstream (:1305-1308,1325)
            else:
                ident = loss_utils.identity_min(target, sources, no_ssim)
                if not opt.disable_automasking:
                    noise = noises[scale] if noises is not None else loss_utils.draw_noise((B, 1, H, W), target.device)
                flags |= L.F_AUTOMASK  # the identity term is compared even with disable_automasking (:1309-1311)
            ctx = None if (with_syn or no_ssim) else loss_utils._ctx(outputs, scale)
            if ctx is not None:
                cfg = (ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention, ident is not None, False, False)
                reproj, _, _, rp_map, _ = Fn.FusedPassFn.apply(ctx.disp, ctx.T[0], ctx.T[1], ctx.K, ctx.inv_K,
                                                               sources[0], sources[1], target, ident, noise, ext,
                                                               None, None, None, cfg)
            else:
                if ("color", fids[0], scale) not in outputs:
                    raise L.MalError("compute_losses: warped images missing; call generate_images_pred first")
                cands = [outputs[("color", f, scale)] for f in fids]
                if with_syn:
                    cands += [outputs[("syn", f, scale)] for f in fids]
                reproj, rp_map, _ = Fn.PhotoLossFn.apply(target, ident, noise, ext, flags, *cands)
            loss = reproj
            if is_multi:
                multi_depth = outputs[("depth", 0, scale)]
                mono_depth = outputs[("mono_depth", 0, scale)].detach()
                # consistency term only (no distillation here): |d_multi - d_mono| * (1 - m)
                cons, _, ct = Fn.DistilFn.apply(multi_depth, mono_depth, rp_map, rp_map, None, ext, False)
                if config.consistency_target:
                    outputs["consistency_target/{}".format(scale)] = ct
                losses["consistency_loss/{}".format(scale)] = cons
                loss = loss + cons
                if getattr(opt, "ensemble", False):
                    ens = (torch.abs((mono_depth + multi_depth) / 2.0 - multi_depth) * ext).mean()
                    losses["ensemble_loss/{}".format(scale)] = ens
                    loss = loss + ens
            losses["reproj_loss/{}".format(scale)] = reproj
            loss = loss + opt.disparity_smoothness * loss_utils._smooth(disp, color) / (2 ** scale)
            total = total + loss
            losses["loss/{}".format(scale)] = loss
        losses["loss"] = total / (opt.sclm + 1)
        return losses, []

    # ---------------------------------------------------------------- process_batch (loss half)
    def compute_batch_losses(self, inputs, mono_outputs, outputs, index_iter=0, noise_mono=None):
        """manydepth/trainer.py:573-642: everything in ``process_batch`` after the network
        forward.  ``mono_outputs`` / ``outputs`` are what ``self.model`` returned."""
        opt = self.opt
        self.generate_images_pred(inputs, mono_outputs)
        if not opt.temporal:
            self.has_ins = False
        if opt.distil:
            mono_losses, mono_reproj = loss_utils.compute_mono_losses(None, inputs, mono_outputs, opt.temporal,
                                                                      self.has_ins, noise=noise_mono)
        else:
            mono_losses, _ = self.compute_losses(inputs, mono_outputs, is_multi=False)
        for key in list(mono_outputs.keys()):
            if isinstance(key, tuple) and key[0] in ("depth", "disp"):
                outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
        outputs["consistency_mask"] = ops.matching_mask(
            outputs["lowest_cost"], outputs[("mono_depth", 0, 0)].detach()[:, 0].contiguous(),
            outputs["consistency_mask"])
        ensemble_reproj = None
        if opt.distil and not opt.no_ens:
            if opt.learn_ens:
                d1, d2 = outputs["ens_disp"], None
            else:  # (mono + multi) / 2 (:598), averaged inside the kernel
                d1, d2 = mono_outputs[("disp", 0)].detach(), outputs[("disp", 0)].detach()
            ensemble_reproj = self.generate_images_pred_ensemble(inputs, outputs[("cam_T_cam", 0, -1)].detach(),
                                                                 outputs[("cam_T_cam", 0, 1)].detach(), d1, d2)
        self.generate_images_pred(inputs, outputs, is_multi=True)
        loss_list = None
        if opt.distil:
            if not opt.main_temporal:
                self.multi_has_ins = False
            losses, self.w_list, loss_list = loss_utils.compute_main_losses(
                None, inputs, outputs, mono_reproj, ensemble_reproj, opt, getattr(self, "model", None), self.w_list,
                self.multi_has_ins)
        else:
            losses, _ = self.compute_losses(inputs, outputs, is_multi=True)
        if not self.freeze_tp:
            for key, val in mono_losses.items():
                losses[key] = losses[key] + val
            if opt.loss_blc:
                loss_list[0] = loss_list[0] + mono_losses["loss"]
        if opt.loss_blc and self.loss_blc is not None:
            losses["loss"] = self.loss_blc.compute_loss(loss_list, index_iter)
            losses["w_ori"], losses["w_distil"] = self.loss_blc.update_weight(
                index_iter, getattr(self, "current_lambda_for_adjust", 3.0))
        return outputs, losses, loss_list


def default_options(**kw):
    """The hot-path fields of manydepth/options.py with their upstream defaults
    (options.py:62-97,130-162,296-302,372,434-452)."""
    o = dict(height=192, width=640, batch_size=12, min_depth=0.1, max_depth=100.0, frame_ids=[0, -1, 1], sclm=0,
             v1_multiscale=False, disable_automasking=False, no_ssim=False, disparity_smoothness=1e-3,
             temporal=False, main_temporal=False, distil=True, no_ens=False, learn_ens=False, dual_distil=False,
             ensemble=False, loss_blc=False, pareto=False, disable_motion_masking=False,
             no_matching_augmentation=False)
    o.update(kw)
    return SimpleNamespace(**o)


class LossPath(MALLossPath):
    """Stand-alone holder of the loss path (what bench.py and the tests drive)."""

    def __init__(self, opt, fuse=True, image_synthesis=None):
        self.opt = opt
        self.fuse = fuse
        if image_synthesis is not None:
            self.image_synthesis = image_synthesis
        self.ssim = layers.SSIM()
        self.backproject_depth = {0: layers.BackprojectDepth(opt.batch_size, opt.height, opt.width)}
        self.project_3d = {0: layers.Project3D(opt.batch_size, opt.height, opt.width)}
        if opt.loss_blc:
            self.w_list = [0.5, 0.5]
