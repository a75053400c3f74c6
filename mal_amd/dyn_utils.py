"""Temporal-hint producer: ``manydepth/dyn_utils.py`` with the per-sample arithmetic
(``generate_dynamic_instance`` / ``fill_dynamic_obj``, :6-119) on the device as three HIP launches instead of a
TorchScript loop over instances with (num,3,H,W) temporaries.  Same names and signatures as the reference, so
``from manydepth.dyn_utils import image_synthesis`` can point here; the instance segmenter (Mask2Former through
``generate_instances``) and the Hungarian matcher stay external and are passed in exactly as upstream.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import ops


class DynamicInstanceFn(Function):
    """(mask_last, mask_next, img_last, img_next, replace) -> (ori_last, ori_next); differentiable w.r.t. the images."""

    @staticmethod
    def forward(ctx, mask_last, mask_next, img_last, img_next, replace):
        if mask_last.shape != mask_next.shape or mask_last.dim() != 3:
            raise L.MalError("generate_dynamic_instance: masks must be two (num,H,W) tensors of the same shape")
        num, H, W = mask_last.shape
        il, inx = ops._req(img_last, "img_last"), ops._req(img_next, "img_next")
        C = il.shape[0]
        if il.shape != (C, H, W) or inx.shape != (C, H, W):
            raise L.MalError("generate_dynamic_instance: images must be (C,H,W) matching the masks")
        dev = il.device
        ml, mn = _as_u8(mask_last, dev), _as_u8(mask_next, dev)
        ol, on = torch.empty_like(il), torch.empty_like(inx)
        delta = torch.empty(num, 2, dtype=torch.int32, device=dev)
        flags = torch.empty(H, W, dtype=torch.uint8, device=dev)
        lib = L.load()
        ws = torch.empty(lib.mal_dyn_workspace_bytes(num), dtype=torch.uint8, device=dev)
        p = ops._p
        L.check(lib.mal_dyn_instance_fwd(p(ml), p(mn), num, p(il), p(inx), C, H, W, 1 if replace else 0, p(ol), p(on),
                                         p(delta), p(flags), p(ws), ws.numel(), ops._stream()), "mal_dyn_instance_fwd")
        ctx.save_for_backward(ml, mn, delta, flags)
        ctx.shape = (num, C, H, W)
        return ol, on

    @staticmethod
    @once_differentiable
    def backward(ctx, g_last, g_next):
        ml, mn, delta, flags = ctx.saved_tensors
        num, C, H, W = ctx.shape
        dev = ml.device
        zeros = lambda: torch.zeros(C, H, W, dtype=torch.float32, device=dev)
        g_last = zeros() if g_last is None else g_last.contiguous()
        g_next = zeros() if g_next is None else g_next.contiguous()
        gl = torch.empty_like(g_last) if ctx.needs_input_grad[2] else None
        gn = torch.empty_like(g_next) if ctx.needs_input_grad[3] else None
        if gl is None and gn is None:
            return None, None, None, None, None
        p = ops._p
        L.check(L.load().mal_dyn_instance_bwd(p(ml), p(mn), num, p(delta), p(flags), p(g_last), p(g_next), C, H, W, p(gl),
                                              p(gn), ops._stream()), "mal_dyn_instance_bwd")
        return None, None, gl, gn, None


def generate_dynamic_instance(grid_h, grid_w, mask_last, mask_next, img_last, img_next, replace: bool):
    """Signature of manydepth/dyn_utils.py:38 (``grid_h`` / ``grid_w`` are the reference's index grids; the kernels
    derive row / column indices from the thread id and do not read them)."""
    return DynamicInstanceFn.apply(mask_last, mask_next, img_last, img_next, bool(replace))


def generate_instances(images, ins_model):
    """manydepth/dyn_utils.py:172-190 runs the Mask2Former predictor on BGR uint8 copies of the images; the
    segmenter is not part of this package: pass a callable ``ins_model(images) -> list of {"instances": ...}``."""
    if not callable(ins_model):
        raise NotImplementedError("bind an instance segmenter: generate_instances(images, ins_model) calls "
                                  "ins_model(images) (upstream: Mask2Former through detectron2, dyn_utils.py:172-190)")
    return ins_model(images)


def _as_u8(mask, dev):
    """(n,H,W) instance masks as bytes on the device; bool is one byte per element already: reinterpret, do not convert"""
    m = mask.to(dev).contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)


# cotangent buffers their owner allows ``BatchSynthesisFn.backward`` to overwrite with its result, by data pointer
# (mal_amd.step registers the buffers it allocates for d loss / d syn around its call of the producer's backward).
# Value: None, or a SNAPSHOT of the same cotangent that is valid at the instances' region pixels (the step's fused sweep
# writes one when it has the region map): the backward then gathers from the snapshot and writes the region pixels of the
# registered buffer in one launch, without scratch.
INPLACE_COTANGENTS = {}
# The whole-step API (mal_amd.step, option "tail_overlap") may ask for this node's backward launch on another stream: a raw
# stream handle that already waits for the producer of the cotangents.  Honoured only by the in-place, region-only backward
# (ONE launch, no torch operation -- nothing the caching allocator or the autograd engine would have to know about), which then
# says so in BACKWARD_STREAM["used"]; the caller makes its own stream wait for that stream before it reads the result.
BACKWARD_STREAM = {"handle": None, "used": False}


class BatchSynthesisFn(Function):
    """``image_synthesis``'s tensor work for the whole batch as ONE autograd node: (color_last, color_next) (B,C,H,W) and
    ``items`` = [(b, mask_last, mask_next)] for the samples with matched instances -> (syn_last, syn_next).  Samples
    not listed keep their warped images (and pass their gradient through), exactly as the reference's
    ``clone`` + per-sample assignment (dyn_utils.py:127-128,163-168) -- without one select / copy / zero-fill / add
    node per sample in the autograd graph."""

    @staticmethod
    def forward(ctx, color_last, color_next, items, replace, prefilled=None):
        # per-sample pointers are handed to the kernels: a batch-strided tensor (samples contiguous) needs no copy
        ok = lambda t: t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t[0].is_contiguous()
        cl = color_last if ok(color_last) else ops._req(color_last, "color_last")
        cn = color_next if ok(color_next) else ops._req(color_next, "color_next")
        B, C, H, W = cl.shape
        dev = cl.device
        lib, p = L.load(), ops._p
        # the kernels write every pixel of a listed sample: only the others need the copy of the warped images ... unless
        # the caller hands buffers that hold the warped images already (the whole-step API's warp pass writes them twice):
        # then only the pixels of the instances' regions are written
        every = len({it[0] for it in items}) == B
        new = lambda: torch.empty(cl.shape, dtype=torch.float32, device=dev)
        if prefilled is not None:
            syn_last, syn_next = prefilled
            if not (ok(syn_last) and ok(syn_next) and syn_last.is_contiguous() and syn_next.is_contiguous()
                    and syn_last.shape == cl.shape and syn_next.shape == cl.shape):
                raise L.MalError("image_synthesis: the prefilled buffers must be two contiguous (B,C,H,W) float32 device tensors")
        else:
            syn_last, syn_next = (new(), new()) if every else (cl.clone(memory_format=torch.contiguous_format),
                                                               cn.clone(memory_format=torch.contiguous_format))
        if not items:
            ctx.saved, ctx.dims, ctx.every = [], (C, H, W), False
            return syn_last, syn_next, None
        # the per-pixel predicates of all samples in ONE (B,H,W) byte map (bit 0: the pixel lies in the instances' region):
        # what the backward needs, and what tells a consumer where syn can differ from the warped images at all
        flags_all = (torch.empty if every else torch.zeros)((B, H, W), dtype=torch.uint8, device=dev)
        arr = (L.DynItem * len(items))()
        saved = []
        # Host cost matters here: with the real producer the step cannot be graph-captured, and round 3 spent ~30 tensor
        # operations per listed sample on slices, views and small allocations.  Per-sample addresses are pointer arithmetic on
        # the batch tensors; the displacement tables and the extents' scratch of all items are ONE allocation each.
        nums = []
        for item in items:
            mask_last, mask_next = item[1], item[2]
            if mask_last.dim() != 3 or mask_next.dim() != 3 or tuple(mask_last.shape[1:]) != (H, W) or \
                    tuple(mask_next.shape[1:]) != (H, W):
                raise L.MalError("image_synthesis: masks must be two (n,H,W) tensors matching the images")
            if len(item) == 5:
                if item[3].numel() != item[4].numel():
                    raise L.MalError("image_synthesis: the two frames must hold the same number of matched instances")
                nums.append(int(item[3].numel()))
            else:
                if mask_last.shape != mask_next.shape:
                    raise L.MalError("image_synthesis: the two frames must hold the same number of matched instances")
                nums.append(int(mask_last.shape[0]))
        max_num = max(nums)
        ws_each = (int(lib.mal_dyn_workspace_bytes(max_num)) + 255) & ~255
        delta_all = torch.empty((len(items), max_num, 2), dtype=torch.int32, device=dev)
        ws_all = torch.empty(len(items) * ws_each, dtype=torch.uint8, device=dev)
        p_cl, p_cn, p_sl, p_sn = p(cl), p(cn), p(syn_last), p(syn_next)
        s_cl, s_cn, s_sl, s_sn = (4 * t.stride(0) for t in (cl, cn, syn_last, syn_next))
        p_delta, p_ws, p_flags = p(delta_all), p(ws_all), p(flags_all)
        keep = [delta_all, ws_all]
        for k, item in enumerate(items):
            b, mask_last, mask_next = item[:3]
            idx_last, idx_next = (item[3], item[4]) if len(item) == 5 else (None, None)
            num = nums[k]
            if idx_last is not None:  # the matcher's row selections travel to the kernels as they are (device int64)
                if not (idx_last.is_cuda and idx_last.dtype == torch.int64 and idx_last.is_contiguous()):
                    idx_last = idx_last.to(device=dev, dtype=torch.int64).contiguous()
                if not (idx_next.is_cuda and idx_next.dtype == torch.int64 and idx_next.is_contiguous()):
                    idx_next = idx_next.to(device=dev, dtype=torch.int64).contiguous()
            # bool masks are one byte per element already: their storage is handed over as it is
            ok_m = lambda m: m.is_cuda and m.is_contiguous() and m.dtype in (torch.bool, torch.uint8)
            ml = mask_last if ok_m(mask_last) else _as_u8(mask_last, dev)
            mn = mask_next if ok_m(mask_next) else _as_u8(mask_next, dev)
            a = arr[k]
            a.mask_last, a.mask_next, a.num = p(ml), p(mn), num
            a.img_last, a.img_next, a.ori_last, a.ori_next = p_cl + b * s_cl, p_cn + b * s_cn, p_sl + b * s_sl, p_sn + b * s_sn
            d_ptr, f_ptr = p_delta + k * max_num * 8, p_flags + b * H * W
            a.delta, a.flags, a.ws, a.ws_bytes = d_ptr, f_ptr, p_ws + k * ws_each, ws_each
            a.idx_last, a.idx_next = p(idx_last), p(idx_next)
            a.n_last, a.n_next = int(ml.shape[0]), int(mn.shape[0])  # a selection outside the tensor is clamped, never read
            a.prefilled = 1 if prefilled is not None else 0
            # masks and selections are handed to the kernels BY REFERENCE and read again in the backward: remember their
            # version counters -- a segmenter / matcher that re-uses its output buffers (a second producer call before this
            # backward, --temporal with --main_temporal) would otherwise scatter gradients through the wrong regions silently
            vers = tuple(None if t is None else t._version for t in (ml, mn, idx_last, idx_next))
            saved.append((b, ml, mn, num, d_ptr, f_ptr, vers, idx_last, idx_next))
        ctx.keep = keep + [flags_all]
        # all samples in one call: three launches (extents, displacements, synthesis) for up to 16 samples
        L.check(lib.mal_dyn_batch_fwd(arr, len(items), C, H, W, 1 if replace else 0, ops._stream()), "mal_dyn_batch_fwd")
        ctx.saved, ctx.dims, ctx.every = saved, (C, H, W), every
        ctx.mark_non_differentiable(flags_all)
        # no zero-filled stand-in for the cotangent of the byte map (a (B,H,W) fill launch in every backward otherwise)
        ctx.set_materialize_grads(False)
        return syn_last, syn_next, flags_all

    @staticmethod
    @once_differentiable
    def backward(ctx, g_last, g_next, _g_flags=None):
        C, H, W = ctx.dims
        if g_last is None and g_next is None:
            return None, None, None, None, None
        if g_last is None:
            g_last = torch.zeros_like(g_next)
        if g_next is None:
            g_next = torch.zeros_like(g_last)
        # cotangent buffers whose owner says they may be overwritten (the whole-step API allocates them for exactly this):
        # outside the instances' regions the gradient IS the cotangent, so only the region pixels are touched, in place
        inplace = g_last.data_ptr() in INPLACE_COTANGENTS and g_next.data_ptr() in INPLACE_COTANGENTS \
            and g_last.is_contiguous() and g_next.is_contiguous()
        g_last, g_next = g_last.contiguous(), g_next.contiguous()
        if not ctx.saved:
            return (g_last, g_next, None, None, None) if inplace else (g_last.clone(), g_next.clone(), None, None, None)
        lib, p = L.load(), ops._p
        snap_l = snap_n = None
        if inplace:
            gl, gn = g_last, g_next
            snap_l, snap_n = INPLACE_COTANGENTS[g_last.data_ptr()], INPLACE_COTANGENTS[g_next.data_ptr()]
            if snap_l is None or snap_n is None:
                snap_l = snap_n = None
                tmp_l, tmp_n = torch.empty_like(g_last), torch.empty_like(g_next)
        else:
            # samples without instances: the identity (a copy); the kernel writes every pixel of the listed ones
            gl, gn = (torch.empty_like(g_last), torch.empty_like(g_next)) if ctx.every else (g_last.clone(), g_next.clone())
        arr = (L.DynItem * len(ctx.saved))()
        at = lambda t: (p(t), 4 * t.stride(0))  # per-sample addresses by pointer arithmetic (samples are contiguous)
        (q_gl, s_gl), (q_gn, s_gn), (q_ol, s_ol), (q_on, s_on) = at(g_last), at(g_next), at(gl), at(gn)
        if snap_l is not None:
            (q_sl, s_sl), (q_sn, s_sn) = at(snap_l), at(snap_n)
        elif inplace:
            (q_tl, s_tl), (q_tn, s_tn) = at(tmp_l), at(tmp_n)
        for k, (b, ml, mn, num, delta, flags, vers, idx_last, idx_next) in enumerate(ctx.saved):
            if vers is not None and vers != tuple(None if t is None else t._version for t in (ml, mn, idx_last, idx_next)):
                raise L.MalError("image_synthesis backward: the instance masks / matched indices of sample %d were modified in place "
                                 "after the forward (a segmenter or matcher that re-uses its output buffers?): hand the producer "
                                 "private copies (mask.clone()) when it is called again before this backward" % b)
            a = arr[k]
            a.mask_last, a.mask_next, a.num, a.delta, a.flags = p(ml), p(mn), num, delta, flags  # (delta, flags: addresses)
            a.idx_last, a.idx_next = p(idx_last), p(idx_next)
            a.n_last, a.n_next = int(ml.shape[0]), int(mn.shape[0])
            a.g_ori_last, a.g_ori_next, a.g_img_last, a.g_img_next = q_gl + b * s_gl, q_gn + b * s_gn, q_ol + b * s_ol, q_on + b * s_on
            if snap_l is not None:
                a.g_ori_last, a.g_ori_next, a.region_only = q_sl + b * s_sl, q_sn + b * s_sn, 1
            elif inplace:
                a.g_tmp_last, a.g_tmp_next = q_tl + b * s_tl, q_tn + b * s_tn
        stream = ops._stream()
        if BACKWARD_STREAM["handle"] and snap_l is not None:
            stream, BACKWARD_STREAM["used"] = BACKWARD_STREAM["handle"], True
        L.check(lib.mal_dyn_batch_bwd(arr, len(ctx.saved), C, H, W, stream), "mal_dyn_batch_bwd")
        return gl, gn, None, None, None


def image_synthesis(inputs, outputs, scale, thres, ins_model, matcher):
    """manydepth/dyn_utils.py:121-170: same control flow and calls to the two external models, sample by sample; the
    tensor work of all samples with matched instances then runs as one autograd node (``BatchSynthesisFn``).
    Writes ``outputs[("syn", -1, scale)]`` / ``("syn", 1, scale)`` when any sample has matched instances."""
    bs = inputs[("color", 0, 0)].shape[0]
    instances = generate_instances(inputs[("color", 0, 0)], ins_model)
    color_last, color_next = outputs[("color", -1, scale)], outputs[("color", 1, scale)]
    items = []
    confident = []
    for b in range(bs):
        cur = instances[b]["instances"]
        instances_cur = cur[cur.scores > thres]
        if len(instances_cur) > 0:
            confident.append((b, instances_cur))
    # upstream stacks (warped last, warped next) of a sample for the segmenter, one small copy per sample (:139-140);
    # here the pairs of the whole batch are laid out by ONE copy and the segmenter sees sample b's pair as a view
    # ... or by none: the whole-step API lays the two warped images of a sample side by side and says so
    pairs = outputs.get(("color_pair", scale))
    if pairs is None and confident:
        pairs = torch.stack([color_last.detach(), color_next.detach()], dim=1)
    for b, instances_cur in confident:
        both = generate_instances(pairs[b], ins_model)
        ins_last, ins_next = both[0]["instances"], both[1]["instances"]
        slice_last, slice_next = matcher(ins_last, ins_next, instances_cur)
        if len(slice_last) + len(slice_next) == 0:
            continue
        if torch.is_tensor(slice_last) and torch.is_tensor(slice_next) and slice_last.dtype == torch.int64 \
                and slice_next.dtype == torch.int64 and ins_last.pred_masks.dtype in (torch.bool, torch.uint8):
            # the matcher's selections (index tensors) are applied inside the kernels: no gather launch per sample/frame
            items.append((b, ins_last.pred_masks, ins_next.pred_masks, slice_last, slice_next))
        else:
            items.append((b, ins_last.pred_masks[slice_last].bool(), ins_next.pred_masks[slice_next].bool()))
    if not items:
        return False
    # ("syn_prefilled", scale): two (B,3,H,W) buffers that hold the warped images already (the whole-step API's warp pass
    # writes them twice): the synthesised images are made in them, touching only the instances' regions
    # ("syn_sparse_buffers", scale): two untouched (B,3,H,W) buffers of a consumer that reads syn at region pixels only (the
    # whole-step API, MAL_STEP_SYN_SPARSE): same kernels, nothing else is written, and ("syn_sparse", scale) says so
    sparse = outputs.get(("syn_sparse_buffers", scale))
    syn_last, syn_next, region = BatchSynthesisFn.apply(color_last, color_next, items, False,
                                                        sparse if sparse is not None else outputs.get(("syn_prefilled", scale)))
    if sparse is not None:
        outputs[("syn_sparse", scale)] = True
    outputs[("syn", -1, scale)], outputs[("syn", 1, scale)] = syn_last, syn_next
    # (B,H,W) bytes, bit 0: where syn can differ from the warped images (a consumer may skip the synthesised candidates
    # elsewhere: an exact tie goes to the warped one anyway, loss_utils.py:103)
    outputs[("syn_region", scale)] = region
    return True
