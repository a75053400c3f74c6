"""DualRefine's warp / loss methods on the same kernels.

Mirrors ``dualrefine/trainer.py``: ``generate_images_pred`` (:395-451),
``pose_update_generate_images_pred`` (:457-480), ``compute_reprojection_loss`` (:487-499),
``compute_losses`` (:530-697) and ``compute_pose_update_losses`` (:699-767) -- same method
names, 4-tuple ``outputs`` keys ``(name, frame, scale, deq_iter)`` and loss-dict keys.
Differences from the ManyDepth path that the kernels take as a flag: ``Project3D`` normalises
``2*(u+0.5)/W-1`` and samples with ``align_corners=False`` (dualrefine/layers.py:224-225,
trainer.py:444-447) -> ``convention=1``; the automask is multiplied by ``consistency_mask`` for
deq_iter > 0 (:593-597); ``--avg_reprojection`` averages candidates instead of taking the min
(:579-583), which runs on the explicit kernels.  The debug ``print``s and the ``exit(0)`` the
shipped file carries (:452-455,481-484) are not behaviour and are not reproduced.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn.functional as F

import ctypes as C

from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import config
from . import functional as Fn
from . import layers
from . import loss_utils
from . import ops
from .trainer import WarpContext


def default_options(**kw):
    """The fields these methods read, with dualrefine/options.py defaults -- except ``scales``: upstream's default is
    [0, 1, 2, 3] (options.py:65-69; the loops then visit scale 0 and 2 with n_losses+1 iterations, skip scale 1 and take
    iteration 0 of scale 3, trainer.py:403-407,536-547); this helper defaults to [0], the refinement iterations of the full
    resolution, which is what ``loss_step`` (one library call per direction) covers and bench.py's dualrefine mode times.
    Pass ``scales=[0, 1, 2, 3]`` for upstream's list: ``generate_images_pred`` + ``compute_losses`` take any."""
    o = dict(height=192, width=640, batch_size=8, min_depth=0.1, max_depth=100.0, frame_ids=[0, -1, 1], scales=[0],
             n_losses=1, v1_multiscale=False, disable_automasking=False, no_ssim=False, disparity_smoothness=1e-3,
             avg_reprojection=False, disable_motion_masking=False, Dstar_T0_pair=False, Tstar_D0_pair=False,
             # upstream's default is False (options.py:160-162) -- and its shipped branch then ends the process (exit(0),
             # trainer.py:484); the default here is the configuration upstream can run.  False: loss_step adds the pose-update losses
             disable_pose_updates=True)
    o.update(kw)
    return SimpleNamespace(**o)


_DR_WS = {}


def _dr_workspace(dev, B, H, W, n, slot=0):
    """``slot``: the scale of opt.scales the call belongs to -- every scale's call keeps its own maps until its backward"""
    need = L.load().mal_dr_workspace_bytes(B, H, W, n)
    from . import step as _step
    key = (dev.index, ops._stream(), B, H, W, n, slot, _step._WS_SLOT)  # (step.workspace_slot: see there)
    ws = _DR_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _DR_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return ws


class DrLossStepFn(Function):
    """generate_images_pred + compute_losses of DualRefine's trainer over the deq iterations of ONE scale as one library call
    per direction (``mal_dr_loss_fwd/_bwd``).  Leaves: ``disp[it]`` (n of them, at the scale's own size), then ``T_m1[it]``,
    then ``T_p1[it]`` ((B,4,4) each; a pose the trainer detaches simply arrives without ``requires_grad``).  For a scale > 0
    the disparities are upsampled here (trainer.py:411-412) and the adjoint is applied to what the library hands back.
    ``cfg[9]`` (a dict, scale 0 only): the pose-update losses (trainer.py:457-480,699-767) ride on the call as one more marching
    pass -- four more leaves follow: the disparity paired with the refined pose for frame -1, iteration 0's disparity (frame
    +1's candidate is ("color", 1, 0, 0)), ("cam_T_cam", 0, -1, 1), ("cam_T_cam", 0, 1); the term is the SECOND output (its own
    cotangent: upstream adds it after the division by len(scales))."""

    @staticmethod
    def forward(ctx, consts, cfg, *leaves):
        color0, color_m1, color_p1, K, inv_K, cmask, noises = consts[:7]
        color0_s = consts[7] if len(consts) > 7 else None
        min_depth, max_depth, smooth_weight, flags, n, philox = cfg[:6]
        scale = cfg[6] if len(cfg) > 6 else 0
        texels_from = cfg[7] if len(cfg) > 7 else None  # the workspace of the step's first call (texels + identity term)
        want_dec = len(cfg) > 8 and cfg[8]               # parity instrumentation (tests): decision planes per iteration
        pu = cfg[9] if len(cfg) > 9 else None            # the pose-update losses: {"noise": (B,1,H,W) N(0,1) or None}
        req, p = ops._req, ops._p
        tens = [None if t is None else req(t, "leaf") for t in leaves]  # (None: a pose-update operand that IS an iteration's)
        cons = [req(t, "input") for t in (color0, color_m1, color_p1, K, inv_K)]
        cm = None if cmask is None else req(cmask, "consistency_mask")
        nz = [None if t is None else req(t, "noise") for t in (noises or [None] * n)]
        B, _, H, W = cons[0].shape
        dev = tens[0].device
        a = L.DrArgs()
        a.B, a.H, a.W, a.n_iters, a.scale = B, H, W, n, int(scale)
        a.min_depth, a.max_depth, a.smooth_weight, a.flags = float(min_depth), float(max_depth), float(smooth_weight), int(flags)
        a.color0, a.color_m1, a.color_p1, a.K, a.inv_K = (p(t) for t in cons)
        up = []
        if scale:
            cs = req(color0_s, "inputs[('color', 0, scale)]")
            cons.append(cs)
            a.color0_s = p(cs)
            up = [ops.upsample_bilinear(t, H, W) for t in tens[:n]]
        for it in range(n):
            a.disp[it], a.T_m1[it], a.T_p1[it] = p(up[it] if scale else tens[it]), p(tens[n + it]), p(tens[2 * n + it])
            if scale:
                a.disp_lo[it] = p(tens[it])
            a.noise[it] = p(nz[it])
        a.consistency_mask = p(cm)
        if philox is not None:  # drawn in the step's first launch; the device counter advances with every (replayed) step
            from . import step as _step
            ctr = _step.noise_counter(dev)
            a.flags |= L.DR_NOISE_PHILOX
            a.noise_seed, a.noise_counter = int(philox) & 0xFFFFFFFFFFFFFFFF, ctr.data_ptr()
        losses = torch.empty(4 * L.DR_MAX_ITERS + 4, dtype=torch.float32, device=dev)
        total = torch.empty(1, dtype=torch.float32, device=dev)
        a.losses, a.loss_total = p(losses), p(total)
        pu_total, pu_nz = None, None
        if pu is not None:
            if scale or len(tens) != 3 * n + 4:
                raise L.MalError("DrLossStepFn: the pose-update losses ride on the scale-0 call, with four more leaves")
            pu_total = torch.empty(1, dtype=torch.float32, device=dev)
            pu_nz = None if pu.get("noise") is None else req(pu["noise"], "pose-update noise")
            a.flags |= L.DR_POSE_UPDATE
            # an operand that is one of the iterations' own (pu["into"][k] = that iteration, else None) arrives as None: its
            # pose-update gradient is added into the iteration's output by the assembly launch, not by autograd afterwards
            into = pu.get("into") or (None,) * 4
            ops_ = []
            for k, base in enumerate((0, 0, n, 2 * n)):
                t = tens[3 * n + k]
                if (t is None) != (into[k] is not None):
                    raise L.MalError("DrLossStepFn: a pose-update operand is either a leaf of its own or one of the iterations'")
                ops_.append(tens[base + into[k]] if t is None else t)
            a.pu_disp_m1, a.pu_disp_p1, a.pu_T_m1, a.pu_T_p1 = (p(t) for t in ops_)
            a.pu_disp_m1_into, a.pu_disp_p1_into, a.pu_T_m1_into, a.pu_T_p1_into = (0 if i is None else i + 1 for i in into)
            a.pu_noise, a.pu_loss_total = p(pu_nz), p(pu_total)
        ws = _dr_workspace(dev, B, H, W, n, slot=int(scale))
        a.ws, a.ws_bytes, a.stream = p(ws), ws.numel(), ops._stream()
        if texels_from is not None:
            a.texels_from = p(texels_from)
        decs = []
        if want_dec:
            decs = [torch.zeros((L.DEC_PLANES, B, H, W), dtype=torch.int32, device=dev) for _ in range(n)]
            for it in range(n):
                a.dec[it] = p(decs[it])
            if pu is not None:
                decs.append(torch.zeros((L.DEC_PLANES, B, H, W), dtype=torch.int32, device=dev))
                a.pu_dec = p(decs[-1])
        L.check(L.load().mal_dr_loss_fwd(C.byref(a)), "mal_dr_loss_fwd")
        ctx.args, ctx.keep, ctx.n, ctx.scale, ctx.up = a, (tens, cons, cm, nz, ws, losses, total, pu_total, pu_nz), n, int(scale), up
        ctx.pu = pu is not None
        ctx.texels_from = texels_from  # (kept alive: the passes of this call read it)
        ctx.ws_token = ops.claim_workspace(ws)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(losses, ws, *decs)
        return (total, pu_total, losses, ws, *decs)  # (ws: what a later scale's call of the same step takes the texels from)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, g_pu=None, _g_losses=None, _g_ws=None, *_g_decs):
        n, tens = ctx.n, ctx.keep[0]
        if g_total is None and (g_pu is None or not ctx.pu):
            return (None,) * (2 + len(tens))
        ops.check_workspace(ctx.keep[4], ctx.ws_token, "DualRefineLossPath.loss_step backward")
        zero = lambda g: torch.zeros(1, dtype=torch.float32, device=tens[0].device) if g is None else g.reshape(1).contiguous()
        g_total = zero(g_total)
        a = ctx.args
        grads = [torch.empty_like(t) if (t is not None and ctx.needs_input_grad[2 + i]) else None for i, t in enumerate(tens)]
        a.g_total = ops._p(g_total)
        if ctx.pu:
            g_pu = zero(g_pu)
            a.g_pu_total = ops._p(g_pu)
            a.g_pu_disp_m1, a.g_pu_disp_p1, a.g_pu_T_m1, a.g_pu_T_p1 = (ops._p(g) for g in grads[3 * n:3 * n + 4])
        full = [None] * n  # scale > 0: d total / d (upsampled disparity) without the smoothness term, which arrives in grads[it]
        for it in range(n):
            if ctx.scale and grads[it] is not None:
                full[it] = torch.empty_like(ctx.up[it])
                a.g_disp[it], a.g_disp_lo[it] = ops._p(full[it]), ops._p(grads[it])
            else:
                a.g_disp[it] = ops._p(grads[it])
            a.g_T_m1[it], a.g_T_p1[it] = ops._p(grads[n + it]), ops._p(grads[2 * n + it])
        L.check(L.load().mal_dr_loss_bwd(C.byref(a)), "mal_dr_loss_bwd")
        for it in range(n):
            if full[it] is not None:
                grads[it].add_(ops.upsample_bilinear_adjoint(full[it], tens[it].shape[-2], tens[it].shape[-1]))
        return (None, None, *grads)


class DualRefineLossPath:
    convention = Fn.DUALREFINE
    fuse = True
    f_thres = 1  # compute_losses takes the per-deq-iteration branch when f_thres > 0 (:543)

    def __init__(self, opt, fuse=True):
        self.opt = opt
        self.fuse = fuse
        self.num_scales = len(opt.scales)
        self.ssim = layers.SSIM()

    # ------------------------------------------------------------------ warp
    def _pose_for(self, outputs, frame_id, deq_iter):
        """dualrefine/trainer.py:420-435."""
        if frame_id == 1:
            T = outputs[("cam_T_cam", 0, frame_id)]
            return T.detach() if deq_iter > 0 else T
        if deq_iter > 0:
            if self.opt.Dstar_T0_pair:
                return outputs[("cam_T_cam", 0, frame_id)].detach()
            return outputs[("cam_T_cam", 0, frame_id, 1)]
        return outputs[("cam_T_cam", 0, frame_id)]

    def _iters(self, scale):
        return self.opt.n_losses + 1 if scale in (0, 1, 2) else 1

    def generate_images_pred(self, inputs, outputs):
        """dualrefine/trainer.py:395-451."""
        opt = self.opt
        self._ident_cache = None  # a new batch: the identity term is evaluated once per call of this method, never carried over
        for scale in opt.scales:
            for deq_iter in range(self._iters(scale)):
                if scale == 1:
                    continue
                disp = outputs[("disp", scale, deq_iter)]
                if not opt.v1_multiscale and tuple(disp.shape[-2:]) != (opt.height, opt.width):
                    disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
                fids = opt.frame_ids[1:]
                Ts = [self._pose_for(outputs, f, deq_iter) for f in fids]
                K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
                srcs = [inputs[("color", f, 0)] for f in fids]
                cfg = (float(opt.min_depth), float(opt.max_depth), 1e-7, self.convention)
                if self.fuse and not opt.avg_reprojection and not opt.no_ssim:
                    _, depth = layers.disp_to_depth(disp, opt.min_depth, opt.max_depth)
                    outputs[("depth", 0, scale, deq_iter)] = depth
                    outputs[("mal_ctx", scale, deq_iter)] = WarpContext(
                        disp=disp, T=Ts, K=K, inv_K=inv_K, min_depth=cfg[0], max_depth=cfg[1], eps=cfg[2],
                        convention=cfg[3], srcs=srcs, fids=fids)
                else:
                    res = Fn.WarpFn.apply(disp, K, inv_K, cfg, len(fids), *Ts, *srcs)
                    outputs[("depth", 0, scale, deq_iter)] = res[0]
                    for i, f in enumerate(fids):
                        outputs[("sample", f, scale, deq_iter)] = res[1 + i]
                        outputs[("color", f, scale, deq_iter)] = res[1 + len(fids) + i]
                if not opt.disable_automasking:
                    for f in fids:
                        outputs[("color_identity", f, scale, deq_iter)] = inputs[("color", f, 0)]

    def pose_update_generate_images_pred(self, inputs, outputs):
        """dualrefine/trainer.py:457-480: frame -1 re-warped with the refined pose."""
        opt = self.opt
        if opt.Tstar_D0_pair:
            depth = outputs[("depth", 0, 0, 0)].clone().detach()
        else:
            depth = outputs[("depth", 0, 0, opt.n_losses)]
        T = outputs[("cam_T_cam", 0, -1, 1)]
        pts = layers.BackprojectDepth(opt.batch_size, opt.height, opt.width)(depth, inputs[("inv_K", 0)])
        grid = layers.Project3DDualRefine(opt.batch_size, opt.height, opt.width)(pts, inputs[("K", 0)], T)
        outputs[("color", -1, 0, 0, 1)] = layers.grid_sample(inputs[("color", -1, 0)], grid, padding_mode="border",
                                                             align_corners=False)

    # ------------------------------------------------------------------ losses
    def compute_reprojection_loss(self, pred, target):
        """dualrefine/trainer.py:487-499."""
        return loss_utils.compute_reprojection_loss(None, pred, target, self.opt.no_ssim)

    @staticmethod
    def compute_loss_masks(reprojection_loss, identity_reprojection_loss):
        return loss_utils.compute_loss_masks(reprojection_loss, identity_reprojection_loss)

    def _identity(self, target, sources, flags):
        """min (or mean, :568-573) over the raw sources of r(source, target): a function of the batch only, but upstream's
        loops evaluate it once per (scale, deq_iter) and once more for the consistency weights -- three marching launches
        per step at n_losses = 1.  Cached per batch (generate_images_pred resets it): the entry HOLDS the tensors it was
        computed from and is hit only by the very same tensor objects at the same version, so a new batch that the caching
        allocator happens to place at the old address (with _version 0 again) can never be mistaken for the old one."""
        tensors = [target] + list(sources)
        hit = getattr(self, "_ident_cache", None)
        if hit is not None and hit[0] == flags and len(hit[1]) == len(tensors) and \
                all(a is b and v == b._version for (a, v), b in zip(hit[1], tensors)):
            return hit[2]
        ident, _, _, _ = ops.photo_fwd(target, [s.detach() for s in sources], None, None, None, flags,
                                       want_argmin=False, want_weight=False)
        self._ident_cache = (flags, [(t, t._version) for t in tensors], ident)
        return ident

    def _reproj_term(self, inputs, outputs, key_tail, cands_keys, ext_mask, noise):
        """masked min/avg reprojection for one (scale, deq_iter): the fused pass when the warp was
        recorded lazily, else the explicit kernels.  -> (loss scalar, per-pixel map)."""
        opt = self.opt
        target = inputs[("color", 0, 0)]
        fids = opt.frame_ids[1:]
        sources = [inputs[("color", f, 0)] for f in fids]
        B, _, H, W = target.shape
        flags = (L.F_NO_SSIM if opt.no_ssim else 0) | (L.F_AVG if opt.avg_reprojection else 0)
        ident = None
        if not opt.disable_automasking:
            ident = self._identity(target, sources, flags)
            flags |= L.F_AUTOMASK
        ctx = outputs.get(("mal_ctx",) + key_tail) if cands_keys is None else None
        if ctx is not None:
            cfg = (ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention, ident is not None, False, False)
            reproj, _, _, rp_map, _ = Fn.FusedPassFn.apply(ctx.disp, ctx.T[0], ctx.T[1], ctx.K, ctx.inv_K, sources[0],
                                                           sources[1], target, ident, noise, ext_mask, None, None,
                                                           None, cfg)
        else:
            keys = cands_keys if cands_keys is not None else [("color", f) + key_tail for f in fids]
            reproj, rp_map, _ = Fn.PhotoLossFn.apply(target, ident, noise, ext_mask, flags,
                                                     *[outputs[k] for k in keys])
        return reproj, rp_map

    def compute_losses(self, inputs, outputs, noises=None):
        """dualrefine/trainer.py:530-633 (per scale, per deq iteration; losses accumulate across
        the iterations of a scale exactly as upstream's running ``loss`` does)."""
        opt = self.opt
        losses = {}
        total = 0
        k = 0
        for scale in opt.scales:
            loss = 0
            for it in range(self._iters(scale) if self.f_thres > 0 else 1):
                if scale == 1 and self.f_thres > 0:
                    continue
                disp = outputs[("disp", scale, it)]
                color = inputs[("color", 0, scale)]
                ext = None
                if it > 0 and not opt.disable_motion_masking:
                    ext = outputs["consistency_mask"].to(torch.float32).contiguous()
                noise = None
                if not opt.disable_automasking:  # one draw per (scale, deq_iter), as :586-587
                    target = inputs[("color", 0, 0)]
                    noise = noises[k] if noises is not None else loss_utils.draw_noise(
                        (target.shape[0], 1) + tuple(target.shape[-2:]), target.device)
                k += 1
                reproj, rp_map = self._reproj_term(inputs, outputs, (scale, it), None, ext, noise)
                losses["reproj_loss/{}".format(scale)] = reproj
                loss = loss + reproj
                if it > 0:
                    multi_depth = outputs[("depth", 0, scale, it)]
                    mono_depth = outputs[("depth", 0, scale, 0)].detach()
                    # consistency_mask = 1 - (automask * consistency_mask): needs the combined weight map
                    wmap = self._weight_map(inputs, outputs, scale, it, ext, rp_map, noise)
                    cons, _, ct = Fn.DistilFn.apply(multi_depth, mono_depth, rp_map, rp_map, None, wmap, False)
                    if config.consistency_target:
                        outputs["consistency_target/{}_{}".format(scale, it)] = ct
                    losses["consistency_loss/{}_{}".format(scale, it)] = cons
                    loss = loss + cons
                loss = loss + opt.disparity_smoothness * loss_utils._smooth(disp, color) / (2 ** scale)
                total = total + loss
                losses["loss/{}_{}".format(scale, it)] = loss
            # upstream updates its running `loss` in place (dualrefine/trainer.py:624,630): the per-iteration entries of a
            # scale alias one tensor and all read as the sum over the scale's iterations
            for key in [k for k in losses if k.startswith("loss/{}_".format(scale))]:
                losses[key] = loss
        losses["loss"] = total / self.num_scales
        return losses

    def loss_step(self, inputs, outputs, noises=None, want_decisions=False, pose_update=None, pose_noise=None):
        """``generate_images_pred`` + ``compute_losses`` (dualrefine/trainer.py:395-451,530-633) in ONE library call per
        direction AND scale of ``opt.scales`` (upstream's default list is [0,1,2,3]: scale 0 and 2 with the deq iterations
        0..n_losses, scale 1 skipped, scale 3 iteration 0 only, :403-407,536-547; a lower scale's disparities are upsampled
        around the call), with --avg_reprojection / --no_ssim when set; same ``losses`` keys and values as the two methods
        called one after the other (they remain the route for --v1_multiscale and for the ("color", ...) / ("sample", ...)
        outputs, which this call does not materialise).  ``noises``: one (B,1,H,W) N(0,1) map per visited
        (scale, iteration), in loop order (default: drawn as ``config.noise_source`` says).
        ``pose_update`` (default: ``not opt.disable_pose_updates``): the pose-update losses of ``process_batch`` (:337-343;
        ``pose_update_generate_images_pred`` + ``compute_pose_update_losses``, :457-480,699-767) ride on the scale-0 call as one
        more marching pass -- "reproj_loss/pose_0" / "loss/pose_0_0" appear and "loss" includes the term, as upstream's merged
        dictionary has them; ``pose_noise``: its (B,1,H,W) N(0,1) map (default: drawn like the others)."""
        opt = self.opt
        if pose_update is None:
            pose_update = not getattr(opt, "disable_pose_updates", True)
        if pose_update and 0 not in opt.scales:
            raise L.MalError("loss_step: the pose-update losses ride on the scale-0 call; use pose_update_generate_images_pred + "
                             "compute_pose_update_losses for a scale list without scale 0")
        n_full = opt.n_losses + 1
        scales = list(opt.scales)
        if any(s_ not in (0, 1, 2, 3) for s_ in scales) or len(set(scales)) != len(scales) or not scales \
                or opt.v1_multiscale or n_full > L.DR_MAX_ITERS \
                or list(opt.frame_ids) != [0, -1, 1] or self.f_thres <= 0:
            raise L.MalError("DualRefineLossPath.loss_step covers scales out of [0,1,2,3], frames [0,-1,1], not --v1_multiscale, "
                             "n_losses < %d; use generate_images_pred + compute_losses otherwise" % L.DR_MAX_ITERS)
        target = inputs[("color", 0, 0)]
        B, _, H, W = target.shape
        flags = (L.DR_NO_AUTOMASK if opt.disable_automasking else 0) | (L.DR_NO_MOTION_MASK if opt.disable_motion_masking else 0) | \
                (L.DR_AVG if opt.avg_reprojection else 0) | (L.DR_NO_SSIM if opt.no_ssim else 0)
        units = [(s_, it) for s_ in scales if s_ != 1 for it in range(n_full if s_ in (0, 1, 2) else 1)]
        philox = None
        if noises is None and not opt.disable_automasking:
            if config.noise_source == "philox":  # drawn inside each call's first launch: no RNG launch, no host work
                philox = config.noise_seed
            else:
                noises = [loss_utils.draw_noise((B, 1, H, W), target.device) for _ in units]  # one draw per visit (:586-587)
        if pose_update and pose_noise is None and not opt.disable_automasking and philox is None:
            pose_noise = loss_utils.draw_noise((B, 1, H, W), target.device)
        losses, total, k = {}, None, 0
        pu_total = None
        decisions = {}  # want_decisions: {(scale, it): (MAL_DEC_PLANES,B,H,W) int32}
        first_ws = None  # the first call's workspace: later scales of this step take the texels and the identity term from it
        for scale in scales:
            if scale == 1:
                continue  # trainer.py:406-407,546-547
            n = n_full if scale in (0, 1, 2) else 1
            disps = [outputs[("disp", scale, it)] for it in range(n)]
            for d in disps:
                if tuple(d.shape) != (B, 1, H >> scale, W >> scale):
                    raise L.MalError("loss_step: the disparities of scale %d must be (B,1,%d,%d)" % (scale, H >> scale, W >> scale))
            T_m1 = [self._pose_for(outputs, -1, it) for it in range(n)]
            T_p1 = [self._pose_for(outputs, 1, it) for it in range(n)]
            cmask = None
            if n > 1 and not opt.disable_motion_masking:
                cmask = outputs["consistency_mask"].to(torch.float32)
            nz = None if noises is None else list(noises[k:k + n])
            k += n
            consts = (target, inputs[("color", -1, 0)], inputs[("color", 1, 0)], inputs[("K", 0)], inputs[("inv_K", 0)], cmask, nz,
                      inputs[("color", 0, scale)] if scale else None)
            pu, pu_leaves = None, ()
            if pose_update and scale == 0:
                # frame -1: the refined pose paired with the last iteration's depth -- or iteration 0's, detached (:463-469);
                # frame +1: ("color", 1, 0, 0), i.e. iteration 0's depth under ("cam_T_cam", 0, 1) with its graph (:420-423)
                d_m1 = outputs[("disp", 0, 0)].detach() if opt.Tstar_D0_pair else outputs[("disp", 0, opt.n_losses)]
                pu_leaves = [d_m1, outputs[("disp", 0, 0)], outputs[("cam_T_cam", 0, -1, 1)], outputs[("cam_T_cam", 0, 1)]]
                if philox is not None and pose_noise is not None:
                    raise L.MalError("loss_step: pose_noise given while the other maps are drawn in the kernels (pass `noises` too)")
                # upstream's default pairing re-uses the iterations' own tensors (the last disparity, the refined pose, frame
                # +1's pose of iteration 0): their pose-update gradients then join the iteration's inside the assembly launch
                into = []
                for k, own in enumerate((disps, disps, T_m1, T_p1)):
                    hit = [i for i, t in enumerate(own) if t is pu_leaves[k]]
                    into.append(hit[0] if hit else None)
                    if hit:
                        pu_leaves[k] = None
                pu = {"noise": None if opt.disable_automasking else pose_noise, "into": tuple(into)}
            cfg = (opt.min_depth, opt.max_depth, opt.disparity_smoothness / (2 ** scale), flags, n, philox, scale, first_ws,
                   bool(want_decisions), pu)
            tot_s, pu_s, v, ws_s, *decs_s = DrLossStepFn.apply(consts, cfg, *disps, *T_m1, *T_p1, *pu_leaves)
            for it, d in enumerate(decs_s):
                decisions[(scale, it) if it < n else ("pose", 0)] = d
            if pu is not None:
                pu_total = pu_s.reshape(())
            if first_ws is None:
                first_ws = ws_s
            total = tot_s.reshape(()) if total is None else total + tot_s.reshape(())
            losses["reproj_loss/%d" % scale] = v[4 * (n - 1)]
            for it in range(n):
                losses["loss/%d_%d" % (scale, it)] = v[4 * L.DR_MAX_ITERS + 1]  # upstream's entries alias the running loss (:624,630)
                if it > 0:
                    losses["consistency_loss/%d_%d" % (scale, it)] = v[4 * it + 1]
        losses["loss"] = total / self.num_scales if self.num_scales != 1 else total
        if pu_total is not None:  # process_batch merges the two dictionaries: "loss" is in both and is summed (:337-343)
            losses["reproj_loss/pose_0"] = losses["loss/pose_0_0"] = pu_total
            losses["loss"] = losses["loss"] + pu_total
        if want_decisions:
            return losses, decisions
        return losses

    def _weight_map(self, inputs, outputs, scale, it, ext, rp_map, noise):
        """reprojection_loss_mask of dualrefine/trainer.py:589-597 as a (B,1,H,W) map."""
        opt = self.opt
        w = torch.ones_like(rp_map)
        if not opt.disable_automasking:
            target = inputs[("color", 0, 0)]
            sources = [inputs[("color", f, 0)] for f in opt.frame_ids[1:]]
            flags = (L.F_NO_SSIM if opt.no_ssim else 0) | (L.F_AVG if opt.avg_reprojection else 0)
            ident = self._identity(target, sources, flags)
            w = loss_utils.compute_loss_masks(rp_map.detach(), ident + noise * 0.00001 if noise is not None else ident)
        if ext is not None:
            w = w * ext
        return w.contiguous()

    def compute_pose_update_losses(self, inputs, outputs, noise=None):
        """dualrefine/trainer.py:699-767."""
        keys = [("color", -1, 0, 0, 1), ("color", 1, 0, 0)]
        if keys[1] not in outputs:
            ctx = outputs.get(("mal_ctx", 0, 0))
            if ctx is None:
                raise L.MalError("compute_pose_update_losses: call generate_images_pred first")
            cfg = (ctx.min_depth, ctx.max_depth, ctx.eps, ctx.convention)
            res = Fn.WarpFn.apply(ctx.disp, ctx.K, ctx.inv_K, cfg, 2, *ctx.T, *ctx.srcs)
            outputs[keys[1]] = res[4]  # (depth, grid_-1, grid_+1, warped_-1, warped_+1)
        if noise is None and not self.opt.disable_automasking:
            target = inputs[("color", 0, 0)]
            noise = loss_utils.draw_noise((target.shape[0], 1) + tuple(target.shape[-2:]), target.device)
        reproj, _ = self._reproj_term(inputs, outputs, (0, 0), keys, None, noise)
        losses = {"reproj_loss/pose_0": reproj, "loss/pose_0_0": reproj, "loss": reproj}
        return losses
