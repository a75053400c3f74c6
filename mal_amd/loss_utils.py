"""Drop-in for ``manydepth/loss_utils.py``: same function names, arguments, dictionary keys
and return tuples, computed by libmal_hip.so.

Two routes per call, chosen by what ``generate_images_pred`` left in ``outputs``:
  * fused   -- ``outputs[("mal_ctx", 0)]`` is present and the warped images were not
               materialised: warp + SSIM + L1 + min + masks + gradient in ONE kernel
               launch (mal_pass_fused);
  * explicit -- ``outputs[("color", f, 0)]`` exist (always the case with the temporal hint,
               whose ``("syn", f, 0)`` images are built from them): mal_photo_fwd/bwd on
               the materialised candidates, gradients flow back through them to the warp.
In the fused route ``losses["loss"]`` / ``loss_list`` carry the gradient; the per-term
entries are the same graph-connected scalars, so backward through any of them works too.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import config
from . import functional as Fn
from . import ops

__all__ = ["compute_loss_masks", "compute_reprojection_loss", "compute_mono_losses", "compute_main_losses",
           "LossBalancing", "identity_min"]


def compute_reprojection_loss(ssim, pred, target, no_ssim=False):
    """manydepth/loss_utils.py:46-55.  ``ssim`` is accepted for signature parity; the SSIM
    arithmetic is inside the kernel (layers.py:243-257)."""
    return Fn.ReprojectionLossFn.apply(pred, target, bool(no_ssim))


def compute_loss_masks(reprojection_loss, identity_reprojection_loss):
    """manydepth/loss_utils.py:27-44: (argmin([reproj, identity]) == 0).float().
    argmin of two maps with first-index tie-break is ``reproj <= identity``."""
    if identity_reprojection_loss is None:
        return torch.ones_like(reprojection_loss)
    return (reprojection_loss <= identity_reprojection_loss).float()


def identity_min(target, sources, no_ssim=False):
    """min_f r(source_f, target), (B,1,H,W): the identity reprojection term of
    loss_utils.py:92-101.  Sources carry no gradient."""
    flags = L.F_NO_SSIM if no_ssim else 0
    mn, _, _, _ = ops.photo_fwd(target, [s.detach() for s in sources], None, None, None, flags, want_argmin=False,
                                want_weight=False)
    return mn


def draw_noise(shape, device):
    """The tie-break noise of loss_utils.py:105-106,178.  ``config.noise_source``:
    "cpu"  -- ``torch.randn(shape)`` from the global CPU generator then H2D, exactly the
              reference's stream (default);
    "cuda" -- the device generator (no host work, no PCIe copy; same distribution);
    "philox" -- only the whole-step API draws inside its kernels (mal_amd.step); here as "cuda"."""
    if config.noise_source == "cpu":
        return torch.randn(shape).to(device, non_blocking=True)
    return torch.randn(shape, device=device)


def _ctx(outputs, scale=0):
    ctx = outputs.get(("mal_ctx", scale))
    if ctx is not None and ("color", -1, scale) not in outputs:
        return ctx
    return None


def _candidates(outputs, with_syn, scale=0):
    c = [outputs[("color", f, scale)] for f in (-1, 1)]
    if with_syn:
        c += [outputs[("syn", f, scale)] for f in (-1, 1)]
    return c


def _smooth(disp, color):
    """loss_utils.py:119-121: get_smooth_loss(disp / (mean_HW disp + 1e-7), color)."""
    return Fn.SmoothLossFn.apply(disp, color, True)


def compute_mono_losses(ssim, inputs, outputs, temporal, has_ins, noise=None):
    """manydepth/loss_utils.py:57-129 (teacher).  -> (losses, min_c R (B,1,H,W))."""
    target = inputs[("color", 0, 0)]
    sources = [inputs[("color", -1, 0)], inputs[("color", 1, 0)]]
    B, _, H, W = target.shape
    ident = identity_min(target, sources)
    if noise is None:
        noise = draw_noise((B, 1, H, W), target.device)
    with_syn = bool(temporal and has_ins)
    ctx = None if with_syn else _ctx(outputs)
    if ctx is not None:
        cfg = (ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention, True, False, False)
        reproj, _, _, min_reproj, _ = Fn.FusedPassFn.apply(ctx.disp, ctx.T[0], ctx.T[1], ctx.K, ctx.inv_K, sources[0],
                                                           sources[1], target, ident, noise, None, None, None, None,
                                                           cfg)
    else:
        reproj, min_reproj, _ = Fn.PhotoLossFn.apply(target, ident, noise, None, L.F_AUTOMASK,
                                                     *_candidates(outputs, with_syn))
    loss = reproj + 1e-3 * _smooth(outputs[("disp", 0)], inputs[("color", 0, 0)]) / (2 ** 0)
    losses = {"reproj_loss/0": reproj, "loss/0": loss, "loss": loss}
    return losses, min_reproj


def compute_main_losses(ssim, inputs, outputs, mono_reproj, ensemble_reproj, opt, model, w_list, multi_has_ins,
                        noise=None):
    """manydepth/loss_utils.py:131-281 (student).  -> (losses, new_w_list, loss_list).

    The reference computes the identity losses, draws the tie-break noise and builds an
    automask, then overwrites the mask with ones (:178-192).  None of that reaches a value,
    so it is not computed; with ``config.noise_source == "cpu"`` the draw is still made so
    the CPU generator advances exactly as upstream.
    """
    if getattr(opt, "pareto", False):
        raise NotImplementedError("opt.pareto needs manydepth/pareto.py, which upstream never committed "
                                  "(manydepth/loss_utils.py:3,256-265)")
    learned = bool(getattr(opt, "learn_ens", False)) and ensemble_reproj is not None
    if learned and "ens_disp" not in outputs:
        raise KeyError("opt.learn_ens reads outputs['ens_disp'], the learnt ensemble head's disparity (loss_utils.py:240-241; "
                       "the shipped RepDepth has no such head: the caller's network provides it)")
    target = inputs[("color", 0, 0)]
    sources = [inputs[("color", -1, 0)], inputs[("color", 1, 0)]]
    B, _, H, W = target.shape
    if noise is None and config.noise_source == "cpu":
        torch.randn((B, 1, H, W))  # dead value upstream (:178,192); keeps the RNG stream aligned
    cmask = outputs["consistency_mask"].to(torch.float32).reshape(B, 1, H, W)
    keep = (1 - outputs["augmentation_mask"][:opt.batch_size]).to(torch.float32).reshape(B)
    m = None  # consistency_mask * (1 - augmentation_mask), (B,1,H,W): built only where a tensor is needed
    mono_depth = outputs[("mono_depth", 0, 0)]
    dual = bool(getattr(opt, "dual_distil", False)) and ensemble_reproj is None
    mono_reproj = mono_reproj.detach()
    ens = ensemble_reproj.detach() if ensemble_reproj is not None else None
    with_syn = bool(multi_has_ins)
    ctx = None if with_syn else _ctx(outputs)
    want_ct = config.consistency_target
    if ctx is not None and not dual and not learned:
        cfg = (ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention, False, True, want_ct)
        reproj, cons, distil, multi_reproj, ct = Fn.FusedPassFn.apply(
            ctx.disp, ctx.T[0], ctx.T[1], ctx.K, ctx.inv_K, sources[0], sources[1], target, None, None, cmask,
            mono_depth.detach(), mono_reproj, ens, cfg, keep)
    else:
        m = (cmask * keep.reshape(B, 1, 1, 1)).contiguous()
        if ctx is not None:
            cfg = (ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention, False, False, False)
            reproj, _, _, multi_reproj, _ = Fn.FusedPassFn.apply(
                ctx.disp, ctx.T[0], ctx.T[1], ctx.K, ctx.inv_K, sources[0], sources[1], target, None, None, m, None,
                None, None, cfg)
        else:
            reproj, multi_reproj, _ = Fn.PhotoLossFn.apply(target, None, None, m, 0, *_candidates(outputs, with_syn))
        teacher = mono_depth if dual else mono_depth.detach()
        ens_depth = None
        if learned:  # the learnt ensemble's depth replaces (mono + multi) / 2 and receives gradient where it wins (:240-245)
            from .layers import disp_to_depth
            _, ens_depth = disp_to_depth(outputs["ens_disp"], opt.min_depth, opt.max_depth)
        cons, distil, ct = Fn.DistilFn.apply(outputs[("depth", 0, 0)], teacher, multi_reproj, mono_reproj, ens, m, dual,
                                             ens_depth)
    if want_ct and ct.numel():
        outputs["consistency_target/0"] = ct
    losses = {"consistency_loss/0": cons, "reproj_loss/0": reproj}
    loss = reproj + cons
    loss = loss + 1e-3 * _smooth(outputs[("disp", 0)], inputs[("color", 0, 0)]) / (2 ** 0)
    if getattr(opt, "loss_blc", False):
        loss_list = [loss.clone(), distil]
        losses["distil_loss"] = distil
        new_w_list = w_list
    else:
        losses["distil_loss"] = distil
        loss = loss + distil
        new_w_list, loss_list = None, None
    losses["loss/0"] = loss
    losses["loss"] = loss
    return losses, new_w_list, loss_list


class LossBalancing:
    """manydepth/loss_utils.py:283-345 (host-side scalar logic, a15).

    ``compute_loss`` returns ``bs * sum_i w_i * loss_i`` exactly as upstream's loop does
    (:303-318), but records the two scalars with ONE device->host copy per step instead of
    ``bs * num_loss`` item reads (:316); upstream's stray ``.cuda("cuda:0")`` (:304) is gone.
    """

    def __init__(self, num_loss, num_train_data, bs):
        self.num_loss = num_loss
        self.weight_initialization_done = False
        self.last_rebalancing_iter = 0
        self.previous_total_loss = 0
        self.previous_loss = 0
        self.w_list = np.array([1. / num_loss, 1. / num_loss])
        self.loss_initialize_scale = np.array([1. / num_loss, 1. / num_loss])
        self.train_scores = np.zeros((num_train_data, num_loss))
        self.train_metrics = np.zeros((num_train_data, 7))
        self.num_data = num_train_data
        self.bs = bs
        self.weight_initialization = True

    def compute_loss(self, loss_list, index_iter):
        first = self.bs * index_iter
        n_in = max(0, min(self.bs, self.num_data - first))
        loss = 0
        if n_in > 0:
            combined = 0
            for i in range(self.num_loss):
                combined = combined + float(self.w_list[i]) * loss_list[i]
            loss = n_in * combined
            vals = torch.stack([l.detach().reshape(()) for l in loss_list]).cpu().numpy()
            self.train_scores[first:first + n_in, :] = vals[None, :]
        return loss

    def update_weight(self, i, current_lambda_for_adjust):
        window = self.train_scores[self.last_rebalancing_iter * self.bs:(i + 1) * self.bs, :].mean(axis=0)
        total_loss = np.sum(window * self.w_list)
        if self.weight_initialization and not self.weight_initialization_done:
            for k in range(self.num_loss):
                self.w_list[k] = (total_loss * self.loss_initialize_scale[k]) / window[k]
            self.weight_initialization_done = True
            self.previous_total_loss = np.sum(window * self.w_list)
            self.previous_loss = window
        else:
            prev_w = np.array(self.w_list)
            if self.previous_total_loss > 0:
                for k in range(self.num_loss):
                    adj = 1 + current_lambda_for_adjust * (
                        (total_loss / self.previous_total_loss) * (self.previous_loss[k] / window[k]) - 1)
                    adj = min(max(adj, 1.0 / 2.0), 2.0 / 1.0)
                    self.w_list[k] = prev_w[k] * adj
            self.previous_total_loss = np.sum(window * self.w_list)
            self.previous_loss = window
        return self.w_list[0], self.w_list[1]
