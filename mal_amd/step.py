"""The whole loss half of ``Trainer.process_batch`` (manydepth/trainer.py:573-642, ``--distil``)
as one C call forward (``mal_loss_step_fwd``, 5 kernels) and one backward
(``mal_loss_step_bwd``, 2 kernels): no Python between the kernels, no per-op autograd nodes.

``loss_step(...)`` returns the same ``losses`` dict keys as the reference's ``process_batch``
(views of one 16-float device vector, so reading them launches nothing) plus the per-pixel maps.
The operator-level API (``mal_amd.loss_utils`` / ``MALLossPath``) computes the same thing op by
op; ``tests/test_gpu_step.py`` holds the two against each other and against the CPU checker.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import ops

_WS = {}
_NOISE_COUNTER = {}  # per device: the step number of the "philox" noise stream, a device word the step itself advances
MAP_NAMES = ("mono_reproj", "multi_reproj", "consistency_mask", "ens_reproj", "noise", "dec_teacher", "dec_student")


def noise_counter(dev):
    """(1,) int64 device tensor: how many steps of the in-kernel ("philox") tie-break noise stream have been drawn"""
    t = _NOISE_COUNTER.get(dev.index)
    if t is None:
        t = _NOISE_COUNTER[dev.index] = torch.zeros(1, dtype=torch.int64, device=dev)
    return t


_WS_SLOT = 0


class workspace_slot:
    """``with workspace_slot(k): loss_step(...)``: the steps issued inside keep their intermediate maps in workspace number
    ``k`` of this (device, stream, shape) instead of the shared one -- k steps in flight on ONE stream then do not overwrite
    each other's maps (a trainer that runs the backward of a step before the next forward never needs it; bench.py rotates
    over several batches with it, so that no step finds the previous replay's working set in the Infinity Cache)."""

    def __init__(self, slot):
        self.slot = int(slot)

    def __enter__(self):
        global _WS_SLOT
        self.prev, _WS_SLOT = _WS_SLOT, self.slot
        return self

    def __exit__(self, *exc):
        global _WS_SLOT
        _WS_SLOT = self.prev
        return False


def _workspace(dev, B, H, W):
    need = L.load().mal_step_workspace_bytes(B, H, W)
    key = (dev.index, ops._stream(), B, H, W, _WS_SLOT)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _WS[key] = ws
    return ws


def _build_args(disp_t, disp_s, aa_m1, tr_m1, aa_p1, tr_p1, consts, cfg, temporal=False, ens_disp=None, main_temporal=False):
    """the mal_step_args block of one step, the tensors it points to (kept alive by the caller) and the map dict;
    ``ens_disp`` (--learn_ens): a seventh leaf, appended to the kept tensors"""
    color0, color_m1, color_p1, K, inv_K, cmask, keep, lowest, noise = consts
    min_depth, max_depth, no_ens, w_main, w_distil, want_maps, aug_is_mask, want_dec, philox = cfg[:9]
    dual = len(cfg) > 9 and cfg[9]
    req = ops._req
    tens = [req(t, n) for t, n in ((disp_t, "disp_teacher"), (disp_s, "disp_student"), (aa_m1, "axisangle"),
                                   (tr_m1, "translation"), (aa_p1, "axisangle"), (tr_p1, "translation"))]
    if ens_disp is not None:
        tens.append(req(ens_disp, "ens_disp"))
        if tuple(tens[6].shape) != tuple(tens[0].shape):
            raise L.MalError("loss_step: outputs['ens_disp'] must have the shape of the disparities")
    # Zero-copy texels: three (B,3,H,W) images in torch.channels_last memory format ARE the (B,H,W,3) texel images the passes
    # gather from -- the step then skips the re-layout of its first sweep (MAL_STEP_TEXEL_INPUTS; same results bit for bit).
    # A loader gets there with `.contiguous(memory_format=torch.channels_last)` on the host tensor.
    texel_inputs = all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] == 3 and not t.is_contiguous()
                       and t.is_contiguous(memory_format=torch.channels_last) for t in (color0, color_m1, color_p1))
    cons = [t if texel_inputs else req(t, "input") for t in (color0, color_m1, color_p1)]
    cons += [req(t, "input") for t in (K, inv_K, cmask, keep, lowest)]
    cons.append(None if noise is None else req(noise, "input"))
    B, _, H, W = tens[0].shape
    dev = tens[0].device
    a = L.StepArgs()
    a.B, a.H, a.W = B, H, W
    a.min_depth, a.max_depth = float(min_depth), float(max_depth)
    a.flags = (L.STEP_NO_ENS if no_ens else 0) | (L.STEP_AUG_MASK if aug_is_mask else 0) | (L.STEP_TEMPORAL if temporal else 0) | \
              (L.STEP_DUAL_DISTIL if (dual and no_ens) else 0) | (L.STEP_MAIN_TEMPORAL if main_temporal else 0) | \
              (L.STEP_TEXEL_INPUTS if texel_inputs else 0)  # (upstream reads dual_distil on the two-way branch only)
    if philox is not None:  # (seed, want the drawn values back)
        a.flags |= L.STEP_NOISE_PHILOX
        a.noise_seed = int(philox[0]) & 0xFFFFFFFFFFFFFFFF
        ctr = noise_counter(tens[0].device)
        a.noise_counter = ctr.data_ptr()
    a.w_main, a.w_distil = float(w_main), float(w_distil)
    p = ops._p
    a.disp_teacher, a.disp_student = p(tens[0]), p(tens[1])
    a.axisangle_m1, a.translation_m1, a.axisangle_p1, a.translation_p1 = (p(t) for t in tens[2:6])
    if ens_disp is not None:
        a.ens_disp = p(tens[6])
    (a.color0, a.color_m1, a.color_p1, a.K, a.inv_K, a.consistency_mask, a.augmentation_keep, a.lowest_cost,
     a.noise) = (p(t) for t in cons)
    losses = torch.empty(16, dtype=torch.float32, device=dev)
    total = torch.empty(1, dtype=torch.float32, device=dev)  # its own tensor: no select/copy nodes in the backward
    a.losses, a.loss_total = p(losses), p(total)
    maps = {}
    if want_maps:
        new = lambda shape: torch.empty(shape, dtype=torch.float32, device=dev)
        maps = dict(mono_reproj=new((B, 1, H, W)), multi_reproj=new((B, 1, H, W)),
                    consistency_mask=new((B, H, W)))
        if not no_ens:
            maps["ens_reproj"] = new((B, 1, H, W))
        a.mono_reproj, a.multi_reproj = p(maps["mono_reproj"]), p(maps["multi_reproj"])
        a.consistency_mask_out = p(maps["consistency_mask"])
        a.ens_reproj = p(maps.get("ens_reproj"))
    if philox is not None and philox[1]:
        maps["noise"] = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        a.noise_out = p(maps["noise"])
    if want_dec:  # parity instrumentation (tests): the kernels' per-pixel decisions, MAL_DEC_* planes
        for k in ("dec_teacher", "dec_student"):
            maps[k] = torch.zeros((L.DEC_PLANES, B, H, W), dtype=torch.int32, device=dev)
        a.dec_teacher, a.dec_student = p(maps["dec_teacher"]), p(maps["dec_student"])
    ws = _workspace(dev, B, H, W)
    a.ws, a.ws_bytes, a.stream = p(ws), ws.numel(), ops._stream()
    return a, (tens, cons, ws, losses, total), maps


def _run_bwd(ctx, g_total):
    tens = ctx.keep[0]
    ops.check_workspace(ctx.keep[2], ctx.ws_token, "loss_step backward")
    # only the total is differentiable through this node: the 16 slots are its terms, for logging
    g_total = g_total.reshape(1).contiguous()
    a = ctx.args
    grads = [torch.empty_like(t) if ctx.needs_input_grad[i] else None for i, t in enumerate(tens[:6])]
    a.g_total = ops._p(g_total)
    (a.g_disp_teacher, a.g_disp_student, a.g_axisangle_m1, a.g_translation_m1, a.g_axisangle_p1,
     a.g_translation_p1) = (ops._p(g) for g in grads)
    g_ens = None
    if len(tens) > 6 and ctx.needs_input_grad[ctx.ens_index]:  # --learn_ens: the ensemble head's disparity
        g_ens = torch.empty_like(tens[6])
        a.g_ens_disp = ops._p(g_ens)
    L.check(L.load().mal_loss_step_bwd(C.byref(a)), "mal_loss_step_bwd")
    ctx.g_ens = g_ens
    return grads


class LossStepFn(Function):
    """leaves: disp_teacher, disp_student, axisangle_m1, translation_m1, axisangle_p1, translation_p1."""

    @staticmethod
    def forward(ctx, disp_t, disp_s, aa_m1, tr_m1, aa_p1, tr_p1, consts, cfg, ens_disp=None):
        a, keep, maps = _build_args(disp_t, disp_s, aa_m1, tr_m1, aa_p1, tr_p1, consts, cfg, ens_disp=ens_disp)
        L.check(L.load().mal_loss_step_fwd(C.byref(a)), "mal_loss_step_fwd")
        if _CAPTURE is not None:
            _CAPTURE.append((a, keep))
        ctx.args = a
        ctx.ens_index = 8
        ctx.keep = keep  # the C struct holds raw pointers: keep the tensors alive
        ctx.ws_token = ops.claim_workspace(keep[2])
        ctx.set_materialize_grads(False)
        outs = [keep[4], keep[3]] + [maps[k] for k in MAP_NAMES if k in maps]
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, *_):
        if g_total is None:
            return (None,) * 9
        grads = _run_bwd(ctx, g_total)
        return (*grads, None, None, ctx.g_ens)


class _Hint:
    """One pass's exchange with the temporal hint's producer: the teacher's (``--temporal``, mal_step_args.warp_* / syn_* /
    g_syn_* / g_warp_* / syn_region) or the student's (``--main_temporal``, the ``*_s_*`` members)."""

    def __init__(self, a, student, B, H, W, dev, scale=None):
        """``scale`` (mal_ms_args, one hint per scale): the members are arrays indexed by it and the sparse bit lives in
        ``syn_sparse``; None: mal_step_args"""
        self.a, self.tag, self.student, self.scale = a, "_s" if student else "", student, scale
        self.sparse_flag = L.STEP_SYN_S_SPARSE if student else L.STEP_SYN_SPARSE
        self.sparse = False
        self.shape, self.dev = (B, 3, H, W), dev
        # the two warped images of a sample side by side: the (2,3,H,W) pair the instance segmenter is fed is then a VIEW
        # (upstream stacks it per sample, dyn_utils.py:139-140); ("color", f, 0) are the two batch-strided halves
        self.pair = torch.empty((B, 2, 3, H, W), dtype=torch.float32, device=dev)
        self.warp = [self.pair[:, 0], self.pair[:, 1]]
        self._set("warp", self.warp)
        # The buffers the synthesised images are made in are NOT filled: a producer that knows the key
        # ("syn_sparse_buffers", scale) -- mal_amd.dyn_utils.image_synthesis -- writes only the pixels of its instances'
        # regions into them instead of cloning every sample first (dyn_utils.py:127-128) and says so (("syn_sparse", scale),
        # with the region map); the sweep then reads the warped images everywhere else (MAL_STEP_SYN_SPARSE), so the pass in
        # front of the producer writes them once, not twice
        self.pre = [torch.empty(self.shape, dtype=torch.float32, device=dev) for _ in range(2)]
        self.has_ins, self.region, self.snap, self.syn, self.syn_data, self.leaf, self.g_syn = False, None, None, None, None, None, None

    def _put(self, name, ptr):
        if self.scale is None:
            setattr(self.a, name, ptr)
        else:
            getattr(self.a, name)[self.scale] = ptr

    def _set(self, base, tensors, tail=("_m1", "_p1")):
        for t, sfx in zip(tensors, tail):
            self._put(base + self.tag + sfx, t.data_ptr())

    def _mark_sparse(self):
        self.sparse = True
        if self.scale is None:
            self.a.flags |= self.sparse_flag
        else:
            self.a.syn_sparse |= 1 << self.scale

    def produce(self, synth, inputs, defer=False):
        """between mal_loss_step_warp and mal_loss_step_fwd; anything raised here is the caller's to abort the step on.
        ``defer``: the caller calls ``finish`` itself (with the has_ins every scale is to use)"""
        B, _, H, W = self.shape
        sc = self.scale or 0
        with torch.enable_grad():
            self.leaf = [w.detach().requires_grad_(True) for w in self.warp]
            local = {("color", -1, sc): self.leaf[0], ("color", 1, sc): self.leaf[1], ("color_pair", sc): self.pair,
                     ("syn_sparse_buffers", sc): (self.pre[0], self.pre[1])}
            self.has_ins = bool(synth(inputs, local, sc))
        self.local = local
        if not defer:
            self.finish()

    def finish(self, has_ins=None):
        """what the producer left -> the argument block (``has_ins``: overrides the producer's own answer -- the multi-scale
        path keeps the LAST scale's for all of them, trainer.py:1162)"""
        B, _, H, W = self.shape
        sc, local = self.scale or 0, self.local
        if has_ins is not None:
            if has_ins and not self.has_ins:
                raise KeyError(("syn", -1, sc))  # as upstream: compute_losses reads a key this scale's producer call never wrote
            self.has_ins = bool(has_ins)
        if self.has_ins:
            self.syn = [local[("syn", -1, sc)], local[("syn", 1, sc)]]
            self.syn_data = [ops._req(t.detach(), "syn") for t in self.syn]
            region = local.get(("syn_region", sc))
            sparse = bool(local.get(("syn_sparse", sc)))
            if sparse and (region is None or any(t.data_ptr() != q.data_ptr() for t, q in zip(self.syn_data, self.pre))):
                raise L.MalError("loss_step: ('syn_sparse', %d) needs the region map and the buffers of ('syn_sparse_buffers', %d)" % (sc, sc))
            if region is not None and not (region.is_cuda and region.dtype == torch.uint8 and tuple(region.shape) == (B, H, W)
                                           and region.is_contiguous()):
                raise L.MalError("loss_step: ('syn_region', %d) must be a contiguous (B,H,W) uint8 device tensor" % sc)
            self.region = region
            if region is not None:
                self._put("syn" + self.tag + "_region", region.data_ptr())
                if sparse:
                    self._mark_sparse()
                # ... and with the map the sweep leaves a second copy of d/d syn at the touched pixels: what the
                # producer's in-place backward gathers from
                self.snap = [torch.empty(self.shape, dtype=torch.float32, device=self.dev) for _ in range(2)]
                self._set("g_syn", self.snap, ("_region_m1", "_region_p1"))
        else:
            # no matched instance anywhere (loss_utils.py:84,152: only the two warped candidates enter the min): an all-zero
            # region map over the untouched buffers -- no synthesised candidate is evaluated anywhere (no pixel of them is
            # read), the running min passes through and d/d syn is zero
            self.syn, self.syn_data = None, self.pre
            self.region = torch.zeros((B, H, W), dtype=torch.uint8, device=self.dev)
            self._put("syn" + self.tag + "_region", self.region.data_ptr())
            self._mark_sparse()
        # the cotangents of syn: this node's own buffers, which the producer's backward may turn into its result in place
        self.g_syn = [torch.empty(self.shape, dtype=torch.float32, device=self.dev) for _ in range(2)]
        self._set("syn", self.syn_data)
        self._set("g_syn", self.g_syn)

    def expose(self, out, dense):
        """what the reference's generate_images_pred leaves in the pass's outputs dict (trainer.py:1122-1125,1161-1165)"""
        sc = self.scale or 0
        out[("color", -1, sc)], out[("color", 1, sc)] = self.warp
        if self.has_ins:
            if self.sparse:
                # dense images for the caller (logging) only when maps are wanted: outside the regions syn IS the warped image
                if dense:
                    inside = (self.region & 1).bool().unsqueeze(1)
                    out[("syn", -1, sc)], out[("syn", 1, sc)] = (torch.where(inside, t, w_) for t, w_ in zip(self.syn_data, self.warp))
            else:
                out[("syn", -1, sc)], out[("syn", 1, sc)] = self.syn_data
        out["multi_has_ins" if self.student else "has_ins"] = self.has_ins

    def backward(self, stream=None):
        """the producer's own backward (linear): g_syn -> g_warp, handed to mal_loss_step_bwd.  ``stream`` (option
        "tail_overlap"): a raw stream handle that already waits for the cotangents' producer; a producer whose backward is
        ONE launch on buffers it was handed (mal_amd.dyn_utils, in place, region-only) enqueues it there.  Returns whether
        everything this call enqueued went to that stream (the caller cancels the overlap otherwise)."""
        on_stream = False
        if self.syn is None:
            g_warp = self.g_syn  # the identity producer
            on_stream = stream is not None  # (nothing was enqueued at all)
        else:
            from . import dyn_utils
            # mal_amd.dyn_utils.image_synthesis turns the cotangent buffers into its result in place
            reg = {g.data_ptr(): (self.snap[i] if self.snap is not None else None) for i, g in enumerate(self.g_syn)}
            dyn_utils.INPLACE_COTANGENTS.update(reg)
            dyn_utils.BACKWARD_STREAM["handle"], dyn_utils.BACKWARD_STREAM["used"] = stream, False
            try:
                g_warp = torch.autograd.grad(self.syn, self.leaf, self.g_syn, allow_unused=True)
            finally:
                for k in reg:
                    dyn_utils.INPLACE_COTANGENTS.pop(k, None)
                used = dyn_utils.BACKWARD_STREAM["used"]
                dyn_utils.BACKWARD_STREAM["handle"], dyn_utils.BACKWARD_STREAM["used"] = None, False
            # ... and only if the result IS those buffers (no further operation of the autograd engine on the caller's stream)
            on_stream = bool(used) and all(g is not None and g.data_ptr() == q.data_ptr() and g.is_contiguous()
                                           for g, q in zip(g_warp, self.g_syn))
            g_warp = [torch.zeros_like(w) if g is None else g.contiguous() for g, w in zip(g_warp, self.warp)]
        self.g_warp = g_warp
        self._set("g_warp", g_warp)
        return on_stream


class TemporalLossStepFn(Function):
    """The step with the temporal hint (``--temporal``, loss_utils.py:84-88; ``--main_temporal``, :152-155): three library
    calls around the producer ``synth(inputs, outputs, scale) -> has_ins`` (upstream: dyn_utils.image_synthesis), which
    reads a pass's warped images ``outputs[("color", f, 0)]`` and writes ``outputs[("syn", f, 0)]`` with ordinary autograd
    ops (dyn_utils.py:127-128,145-146,163-168).  Forward: mal_loss_step_warp -> producer, once per hinted pass, the
    teacher's first as upstream calls them (recorded by autograd on a private copy of the warped images) ->
    mal_loss_step_fwd (hands back d loss / d syn); backward: the producer's own backward (torch.autograd.grad on that
    private graph), then mal_loss_step_bwd, whose gradient sweeps add what arrives through syn to d loss / d warped colour
    before the chain rule through the warp.  ``which`` = (teacher hinted, student hinted); ``expose`` = (mono_outputs, outputs)."""

    @staticmethod
    def forward(ctx, disp_t, disp_s, aa_m1, tr_m1, aa_p1, tr_p1, consts, cfg, synth, inputs, expose, which, ens_disp=None):
        a, keep, maps = _build_args(disp_t, disp_s, aa_m1, tr_m1, aa_p1, tr_p1, consts, cfg, temporal=which[0],
                                    main_temporal=which[1], ens_disp=ens_disp)
        ctx.ens_index = 12
        B, _, H, W = keep[0][0].shape
        dev = keep[0][0].device
        hints = [_Hint(a, student, B, H, W, dev) for student, on in ((False, which[0]), (True, which[1])) if on]
        a.warp_sample_stride = 6 * H * W
        lib = L.load()
        L.check(lib.mal_loss_step_warp(C.byref(a)), "mal_loss_step_warp")
        try:  # the producer raised, or left something unusable: join what mal_loss_step_warp forked before the buffers are reused
            for h in hints:
                if h.student:  # its warped images may have been written on the library's side stream
                    L.check(lib.mal_loss_step_student_ready(C.byref(a)), "mal_loss_step_student_ready")
                h.produce(synth, inputs)
        except BaseException:
            lib.mal_loss_step_abort(C.byref(a))
            raise
        L.check(lib.mal_loss_step_fwd(C.byref(a)), "mal_loss_step_fwd")
        ctx.args, ctx.keep = a, keep
        ctx.ws_token = ops.claim_workspace(keep[2])
        ctx.hints = hints  # the C struct holds their buffers' pointers
        ctx.set_materialize_grads(False)
        for h in hints:
            h.expose(expose[1] if h.student else expose[0], cfg[5])
        outs = [keep[4], keep[3]] + [maps[k] for k in MAP_NAMES if k in maps]
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, *_):
        if g_total is None:
            return (None,) * 13
        # option "tail_overlap": the backward chain (producer's backward -> teacher's gradient sweep) on the library's side
        # stream, behind the fused sweep only -- beside the epilogue and the reduction the forward left on this stream
        lib, a = L.load(), ctx.args
        side = C.c_void_p(0)
        L.check(lib.mal_loss_step_tail_begin(C.byref(a), C.byref(side)), "mal_loss_step_tail_begin")
        overlap = side.value is not None and side.value != a.stream
        ok = True
        try:
            for h in ctx.hints:
                ok = h.backward(stream=side.value if overlap else None) and ok
        finally:
            if side.value != a.stream and not (overlap and ok):  # nothing (or not everything) went there: join it back now
                L.check(lib.mal_loss_step_tail_cancel(C.byref(a)), "mal_loss_step_tail_cancel")
        grads = _run_bwd(ctx, g_total)
        return (*grads, None, None, None, None, None, None, ctx.g_ens)


def loss_step(opt, inputs, mono_outputs, outputs, w_list=None, batch_size_scale=None, noise=None, want_maps=True,
              want_decisions=False, want_noise=False, image_synthesis=None):
    """process_batch's loss half in one call.  Reads the same dict entries as the reference:
    ``inputs[("color", f, 0)]``, ``("K", 0)``, ``("inv_K", 0)``; ``mono_outputs[("disp", 0)]``,
    ``("axisangle", 0, f)`` / ``("translation", 0, f)`` (networks/repdepth.py:155-156);
    ``outputs[("disp", 0)]``, ``"consistency_mask"``, ``"augmentation_mask"``, ``"lowest_cost"``.
    With ``opt.temporal`` the producer ``image_synthesis(inputs, outputs, scale) -> has_ins`` is called between the
    library calls (``TemporalLossStepFn``); ``mono_outputs`` then receives ``("color", f, 0)``, ``("syn", f, 0)`` and
    ``"has_ins"`` as the reference's generate_images_pred leaves them (trainer.py:1122-1125,1161-1165).  With
    ``opt.main_temporal`` (trainer.py:1164, loss_utils.py:152-155) it is called for the student's pass as well -- after the
    teacher's, as upstream -- and ``outputs`` receives the same keys and ``"multi_has_ins"``.
    Writes ``outputs["consistency_mask"]`` (x matching mask, trainer.py:592-593) when ``want_maps``.
    ``want_decisions`` (tests) adds ``maps["dec_teacher"]`` / ``["dec_student"]``: the per-pixel decisions of the two
    gradient passes (int32 (MAL_DEC_PLANES,B,H,W), include/mal_hip.h).
    Returns (losses dict, loss_list or None, maps dict)."""
    from . import config, loss_utils
    if getattr(opt, "no_ssim", False) or not getattr(opt, "distil", True) or getattr(opt, "sclm", 0) != 0:
        # (--no_ssim is read by Trainer.compute_reprojection_loss, trainer.py:1217, i.e. on the non-distil route only: the
        # distillation losses call loss_utils.compute_reprojection_loss, :46-55, which has no such branch -- and Trainer.ssim
        # does not exist then, trainer.py:318)
        raise L.MalError("loss_step covers the --distil [--temporal] [--main_temporal] [--learn_ens] [--no_ens [--dual_distil]] "
                         "single-scale configuration; use MALLossPath.compute_batch_losses for no_ssim / non-distil runs")
    ens_disp = None
    if getattr(opt, "learn_ens", False) and not getattr(opt, "no_ens", False):
        # the learnt ensemble head's disparity (loss_utils.py:240-241, trainer.py:596-597): warped by the ensemble pass,
        # distillation target where the ensemble wins, and a leaf that receives gradient there
        if "ens_disp" not in outputs:
            raise KeyError("opt.learn_ens reads outputs['ens_disp'] (the shipped RepDepth has no such head: the caller's network provides it)")
        ens_disp = outputs["ens_disp"]
    temporal, main_temporal = bool(getattr(opt, "temporal", False)), bool(getattr(opt, "main_temporal", False))
    if (temporal or main_temporal) and image_synthesis is None:
        raise L.MalError("loss_step with opt.temporal / opt.main_temporal needs image_synthesis(inputs, outputs, scale) -> has_ins "
                         "(upstream: dyn_utils.image_synthesis bound to the segmenter and the matcher, trainer.py:1161-1165)")
    color0 = inputs[("color", 0, 0)]
    B, _, H, W = color0.shape
    dev = color0.device
    aa = {f: mono_outputs[("axisangle", 0, f)] for f in (-1, 1)}
    tr = {f: mono_outputs[("translation", 0, f)] for f in (-1, 1)}
    fix = lambda t: t[:, 0] if t.dim() == 4 else t  # the pose decoder emits (B,2,1,3); frame 0 of it is used
    philox = None
    if noise is None:  # (loss_utils.compute_mono_losses adds the noise whatever --disable_automasking says, loss_utils.py:105-106)
        if config.noise_source == "philox":  # drawn inside the step's first kernel: no RNG launch, no host work
            philox = (config.noise_seed, bool(want_noise))
        else:
            noise = loss_utils.draw_noise((B, 1, H, W), dev)
            if config.noise_source == "cpu":
                torch.randn((B, 1, H, W))  # compute_main_losses' dead draw (loss_utils.py:178)
    aug = outputs["augmentation_mask"][:opt.batch_size]
    aug_is_mask = aug.dtype == torch.float32 and aug.is_contiguous()  # then 1 - mask is formed on the device
    keep = aug.reshape(B) if aug_is_mask else (1 - aug).to(torch.float32).reshape(B)
    blc = bool(getattr(opt, "loss_blc", False))
    w_main, w_distil = 1.0, 1.0
    if blc:
        scale = float(batch_size_scale if batch_size_scale is not None else opt.batch_size)
        w_main, w_distil = scale * float(w_list[0]), scale * float(w_list[1])
    consts = (color0, inputs[("color", -1, 0)], inputs[("color", 1, 0)], inputs[("K", 0)], inputs[("inv_K", 0)],
              outputs["consistency_mask"].to(torch.float32), keep, outputs["lowest_cost"], noise)
    cfg = (opt.min_depth, opt.max_depth, bool(getattr(opt, "no_ens", False)), w_main, w_distil, bool(want_maps),
           aug_is_mask, bool(want_decisions), philox, bool(getattr(opt, "dual_distil", False)))
    if temporal or main_temporal:
        res = TemporalLossStepFn.apply(mono_outputs[("disp", 0)], outputs[("disp", 0)], fix(aa[-1]), fix(tr[-1]), fix(aa[1]),
                                       fix(tr[1]), consts, cfg, image_synthesis, inputs, (mono_outputs, outputs),
                                       (temporal, main_temporal), ens_disp)
    else:
        res = LossStepFn.apply(mono_outputs[("disp", 0)], outputs[("disp", 0)], fix(aa[-1]), fix(tr[-1]), fix(aa[1]),
                               fix(tr[1]), consts, cfg, ens_disp)
    total, v = res[0].reshape(()), res[1]
    maps = {}
    names = (["mono_reproj", "multi_reproj", "consistency_mask"] + ([] if cfg[2] else ["ens_reproj"])) if want_maps else []
    names += ["noise"] if philox is not None and philox[1] else []
    names += ["dec_teacher", "dec_student"] if want_decisions else []
    maps = dict(zip(names, res[2:]))
    if want_maps:
        outputs["consistency_mask"] = maps["consistency_mask"]
    losses = {"reproj_loss/0": v[9], "consistency_loss/0": v[4], "distil_loss": v[6], "loss/0": v[11] if blc else v[10],
              "loss": total, "mono/reproj_loss/0": v[0], "mono/loss": v[2], "smooth_loss/mono": v[1],
              "smooth_loss/multi": v[5], "main/reproj_loss/0": v[3]}
    loss_list = [v[11], v[6]] if blc else None
    return losses, loss_list, maps


_CAPTURE = None  # teacher_pass_replay: receives (argument block, kept tensors) of the next LossStepFn.forward


def teacher_pass_replay(opt, inputs, mono_outputs, outputs, **kw):
    """Measurement hook (bench.py's ``roofline`` block).  Runs ONE forward of the step without the temporal hint on the
    given dicts (so the workspace holds the texels, the identity map and the camera block) and returns
    ``enqueue(launches)``: each call enqueues that many back-to-back launches of the teacher's pass -- the fused
    warp + SSIM + L1 + min + automask forward+backward sweep -- on the current stream, with the argument block of that
    forward (``mal_loss_step_teacher_replay``).  Capture ``enqueue(64)`` into a graph, replay it, time it with two events
    outside the graph: the kernel alone, as a replayed step runs it."""
    global _CAPTURE
    import copy
    o = copy.copy(opt)
    o.temporal = False
    _CAPTURE = []
    try:
        with torch.no_grad():
            loss_step(o, inputs, mono_outputs, outputs, want_maps=False, **kw)
        got = list(_CAPTURE)
    finally:
        _CAPTURE = None
    a, keep = got[0]

    def enqueue(launches):
        a.stream = ops._stream()
        L.check(L.load().mal_loss_step_teacher_replay(C.byref(a), int(launches)), "mal_loss_step_teacher_replay")

    enqueue.keep = keep  # the block holds raw pointers
    return enqueue


# ---------------------------------------------------------------------------------- sclm > 0, no --distil
_WS_MS = {}


def _workspace_ms(dev, B, H, W, sclm):
    need = L.load().mal_ms_workspace_bytes(B, H, W, sclm)
    key = (dev.index, ops._stream(), B, H, W, sclm, _WS_SLOT)
    ws = _WS_MS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WS_MS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return ws


class MultiScaleLossFn(Function):
    """leaves: disp_teacher[0..sclm], disp_student[0..sclm], axisangle_m1, translation_m1, axisangle_p1, translation_p1"""

    @staticmethod
    def forward(ctx, consts, cfg, *leaves):
        colors, colors_s, K, inv_K, cmask, keep, lowest, noises = consts
        min_depth, max_depth, sclm, aug_is_mask, philox, want_maps = cfg[:6]
        hint = cfg[6] if len(cfg) > 6 else None  # --temporal: (image_synthesis, inputs, mono_outputs)
        want_dec = len(cfg) > 9 and cfg[9]         # parity instrumentation (tests): per-scale decision planes
        S = sclm + 1
        req, p = ops._req, ops._p
        tens = [req(t, "leaf") for t in leaves]
        cons = [req(t, "input") for t in (*colors, *colors_s, K, inv_K, cmask, keep)]
        cons += [None if t is None else req(t, "input") for t in (lowest, *(noises or ()))]
        B, _, H, W = tens[0].shape
        dev = tens[0].device
        for s in range(S):
            for t in (tens[s], tens[S + s]):
                if tuple(t.shape) != (B, 1, H >> s, W >> s):
                    raise L.MalError("loss_step_multiscale: the disparity of scale %d must be (B,1,%d,%d)" % (s, H >> s, W >> s))
        a = L.MsArgs()
        a.B, a.H, a.W, a.sclm = B, H, W, sclm
        a.min_depth, a.max_depth = float(min_depth), float(max_depth)
        a.flags = (L.STEP_AUG_MASK if aug_is_mask else 0) | (L.STEP_NO_SSIM if (len(cfg) > 7 and cfg[7]) else 0) | \
                  (int(cfg[8]) if len(cfg) > 8 else 0)
        a.color0, a.color_m1, a.color_p1 = (p(t) for t in cons[:3])
        for s in range(1, S):
            a.color0_s[s] = p(cons[3 + s - 1])
        o = 3 + sclm
        a.K, a.inv_K, a.consistency_mask, a.augmentation_keep = (p(t) for t in cons[o:o + 4])
        a.lowest_cost = p(cons[o + 4])
        for s in range(S):
            a.disp_teacher[s], a.disp_student[s] = p(tens[s]), p(tens[S + s])
            if noises is not None:
                a.noise[s] = p(cons[o + 5 + s])
        a.axisangle_m1, a.translation_m1, a.axisangle_p1, a.translation_p1 = (p(t) for t in tens[2 * S:])
        if philox is not None:
            a.flags |= L.STEP_NOISE_PHILOX
            a.noise_seed = int(philox) & 0xFFFFFFFFFFFFFFFF
            a.noise_counter = noise_counter(dev).data_ptr()
        losses = torch.empty(48, dtype=torch.float32, device=dev)
        total = torch.empty(1, dtype=torch.float32, device=dev)
        a.losses, a.loss_total = p(losses), p(total)
        outs = [total, losses]
        if want_maps and lowest is not None and not (a.flags & L.STEP_NO_MOTION_MASK):
            cm = torch.empty((B, H, W), dtype=torch.float32, device=dev)
            a.consistency_mask_out = p(cm)
            outs.append(cm)
        decs = []
        if want_dec:  # one (MAL_DEC_PLANES,B,H,W) int32 block per network and scale (include/mal_hip.h MAL_DEC_*)
            decs = [torch.zeros((L.DEC_PLANES, B, H, W), dtype=torch.int32, device=dev) for _ in range(2 * S)]
            for s in range(S):
                a.dec_teacher[s], a.dec_student[s] = p(decs[s]), p(decs[S + s])
            outs += decs
        ctx.n_dec = len(decs)
        ws = _workspace_ms(dev, B, H, W, sclm)
        a.ws, a.ws_bytes, a.stream = p(ws), ws.numel(), ops._stream()
        hints = []
        if hint is not None:
            # the temporal hint on this path (trainer.py:1161-1162,1279-1283): the producer once per scale, on that scale's
            # full-resolution warp of the teacher, between mal_loss_multiscale_warp and _fwd; has_ins is the LAST call's
            synth, inputs, expose = hint
            a.flags |= L.STEP_TEMPORAL
            a.warp_sample_stride = 6 * H * W
            hints = [_Hint(a, False, B, H, W, dev, scale=s_) for s_ in range(S)]
            L.check(L.load().mal_loss_multiscale_warp(C.byref(a)), "mal_loss_multiscale_warp")
            try:  # a producer raised, or left something unusable: join what _warp forked before the buffers are reused
                for h in hints:
                    h.produce(synth, inputs, defer=True)
                has_ins = hints[-1].has_ins
                for h in hints:
                    h.finish(has_ins)
            except BaseException:
                L.load().mal_loss_multiscale_abort(C.byref(a))
                raise
        L.check(L.load().mal_loss_multiscale_fwd(C.byref(a)), "mal_loss_multiscale_fwd")
        for h in hints:
            h.expose(hint[2], want_maps)
        ctx.hints = hints  # the argument block holds their buffers' pointers
        ctx.args, ctx.keep, ctx.S = a, (tens, cons, ws, losses, total), S
        ctx.ws_token = ops.claim_workspace(ws)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, *_):
        tens, S = ctx.keep[0], ctx.S
        if g_total is not None:
            ops.check_workspace(ctx.keep[2], ctx.ws_token, "loss_step_multiscale backward")
        if g_total is None:
            return (None,) * (2 + len(tens))
        g_total = g_total.reshape(1).contiguous()
        a = ctx.args
        grads = [torch.empty_like(t) if ctx.needs_input_grad[2 + i] else None for i, t in enumerate(tens)]
        a.g_total = ops._p(g_total)
        for s in range(S):
            a.g_disp_teacher[s], a.g_disp_student[s] = ops._p(grads[s]), ops._p(grads[S + s])
        a.g_axisangle_m1, a.g_translation_m1, a.g_axisangle_p1, a.g_translation_p1 = (ops._p(g) for g in grads[2 * S:])
        for h in ctx.hints:
            h.backward()  # the producer's own backward: g_syn -> g_warp of that scale
        L.check(L.load().mal_loss_multiscale_bwd(C.byref(a)), "mal_loss_multiscale_bwd")
        return (None, None, *grads)


def loss_step_multiscale(opt, inputs, mono_outputs, outputs, noises=None, want_maps=True, image_synthesis=None,
                         want_decisions=False):
    """process_batch's loss half WITHOUT ``--distil`` (manydepth/trainer.py:573-612 with ``compute_losses``, :1248-1475,
    for both networks) over scales 0..``opt.sclm`` in one call per direction.  Reads ``inputs[("color", f, 0)]``,
    ``("color", 0, s)``, ``("K", 0)``, ``("inv_K", 0)``; ``mono_outputs[("disp", s)]``, ``("axisangle", 0, f)``,
    ``("translation", 0, f)``; ``outputs[("disp", s)]``, ``"consistency_mask"``, ``"augmentation_mask"`` and, when present,
    ``"lowest_cost"`` (then the matching mask of trainer.py:592-593 is applied and ``outputs["consistency_mask"]``
    rewritten).  ``noises``: one (B,1,H,W) N(0,1) map per scale (default: drawn as ``config.noise_source`` says).
    With ``opt.temporal`` the producer ``image_synthesis(inputs, outputs, scale) -> has_ins`` is called once per scale on the
    teacher's full-resolution warp of that scale (trainer.py:1161-1162) and, when the LAST call reported instances, every
    scale's min takes the two synthesised candidates in (:1279-1283; a scale whose own call reported none then raises the
    ``KeyError`` upstream raises); ``mono_outputs`` receives ``("color", f, s)``, ``("syn", f, s)`` and ``"has_ins"``.
    Returns (losses, mono_losses): ``losses`` as process_batch leaves it (the teacher's entries added to the
    student's, :614-616; ``losses["loss"]`` carries the gradient), ``mono_losses`` the teacher's own.
    ``want_decisions`` (tests): a third value ``{"dec_teacher": [per scale], "dec_student": [per scale]}`` -- the per-pixel
    decisions of every scale's gradient passes (int32 (MAL_DEC_PLANES,B,H,W) each, include/mal_hip.h)."""
    from . import config, loss_utils
    sclm = int(getattr(opt, "sclm", 0))
    temporal = bool(getattr(opt, "temporal", False))
    if temporal and image_synthesis is None:
        raise L.MalError("loss_step_multiscale with opt.temporal needs image_synthesis(inputs, outputs, scale) -> has_ins")
    unsupported = [k for k in ("distil", "v1_multiscale") if getattr(opt, k, False)]
    if getattr(opt, "no_ssim", False) and getattr(opt, "temporal", False):
        unsupported.append("no_ssim with temporal")
    if unsupported or sclm >= L.MS_MAX_SCALES or list(opt.frame_ids) != [0, -1, 1]:
        # (--v1_multiscale with sclm > 0 cannot run upstream for the student either: its (B,1,H>>s,W>>s) mask is multiplied
        # by the full-resolution consistency mask, manydepth/trainer.py:1322)
        raise L.MalError("loss_step_multiscale covers the non-distil sclm <= 3 configuration with frames [0,-1,1]; %s: use "
                         "MALLossPath.compute_batch_losses" % (", ".join(unsupported) or "this configuration"))
    color0 = inputs[("color", 0, 0)]
    B, _, H, W = color0.shape
    dev = color0.device
    fix = lambda t: t[:, 0] if t.dim() == 4 else t
    aa = {f: fix(mono_outputs[("axisangle", 0, f)]) for f in (-1, 1)}
    tr = {f: fix(mono_outputs[("translation", 0, f)]) for f in (-1, 1)}
    philox = None
    if getattr(opt, "disable_automasking", False):
        noises = None  # upstream still compares against the identity term (trainer.py:1296-1311): only the tie-break noise goes
    elif noises is None:
        if config.noise_source == "philox":
            philox = config.noise_seed
        else:
            noises = [loss_utils.draw_noise((B, 1, H, W), dev) for _ in range(sclm + 1)]
            if config.noise_source == "cpu":
                for _ in range(sclm + 1):
                    torch.randn((B, 1, H, W))  # the student's dead draws (trainer.py:1305-1308,1325)
    aug = outputs["augmentation_mask"][:opt.batch_size]
    aug_is_mask = aug.dtype == torch.float32 and aug.is_contiguous()
    keep = aug.reshape(B) if aug_is_mask else (1 - aug).to(torch.float32).reshape(B)
    consts = ((color0, inputs[("color", -1, 0)], inputs[("color", 1, 0)]), [inputs[("color", 0, s)] for s in range(1, sclm + 1)],
              inputs[("K", 0)], inputs[("inv_K", 0)], outputs["consistency_mask"].to(torch.float32), keep,
              outputs.get("lowest_cost"), noises)
    cfg = (opt.min_depth, opt.max_depth, sclm, aug_is_mask, philox, bool(want_maps),
           (image_synthesis, inputs, mono_outputs) if temporal else None, bool(getattr(opt, "no_ssim", False)),
           (L.STEP_NO_MOTION_MASK if getattr(opt, "disable_motion_masking", False) else 0) |
           (L.STEP_NO_AUG if getattr(opt, "no_matching_augmentation", False) else 0) |
           (L.STEP_ENSEMBLE if getattr(opt, "ensemble", False) else 0), bool(want_decisions))
    leaves = [mono_outputs[("disp", s)] for s in range(sclm + 1)] + [outputs[("disp", s)] for s in range(sclm + 1)] + \
             [aa[-1], tr[-1], aa[1], tr[1]]
    res = MultiScaleLossFn.apply(consts, cfg, *leaves)
    total, v = res[0].reshape(()), res[1]
    n_dec = 2 * (sclm + 1) if want_decisions else 0
    decs = res[len(res) - n_dec:] if n_dec else ()
    if len(res) - n_dec > 2:
        outputs["consistency_mask"] = res[2]
    elif want_maps and outputs.get("lowest_cost") is not None and getattr(opt, "disable_motion_masking", False):
        # process_batch multiplies the matching mask in whatever the loss does with it (trainer.py:592-593); the passes did not
        # form it (their weight leaves the mask out): one operator launch
        from . import layers
        with torch.no_grad():
            _, mono_depth = layers.disp_to_depth(mono_outputs[("disp", 0)].detach(), opt.min_depth, opt.max_depth)
            outputs["consistency_mask"] = ops.matching_mask(outputs["lowest_cost"], mono_depth[:, 0].contiguous(),
                                                            outputs["consistency_mask"].to(torch.float32))
    losses, mono_losses = {"loss": total}, {"loss": v[32]}
    for s in range(sclm + 1):
        losses["reproj_loss/%d" % s], losses["loss/%d" % s] = v[36 + s], v[40 + s]
        losses["consistency_loss/%d" % s] = v[(4 + s) * 4 + 1]
        losses["main/reproj_loss/%d" % s], losses["main/loss/%d" % s] = v[(4 + s) * 4], v[(4 + s) * 4 + 3]
        losses["smooth_loss/multi/%d" % s], mono_losses["smooth_loss/%d" % s] = v[(4 + s) * 4 + 2], v[s * 4 + 2]
        if getattr(opt, "ensemble", False):
            losses["ensemble_loss/%d" % s] = v[44 + s]
        mono_losses["reproj_loss/%d" % s], mono_losses["loss/%d" % s] = v[s * 4], v[s * 4 + 3]
    losses["main/loss"] = v[33]
    if want_decisions:
        return losses, mono_losses, {"dec_teacher": list(decs[:sclm + 1]), "dec_student": list(decs[sclm + 1:])}
    return losses, mono_losses
