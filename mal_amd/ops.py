"""Thin tensor-level wrappers over the C ABI (include/mal_hip.h).

PyTorch is used here for device memory and the current HIP stream only; every arithmetic
step happens in libmal_hip.so.  All inputs must be CUDA(HIP) float32 tensors; there is no
CPU path -- a CPU tensor raises.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib as L

_WS = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.MalError("%s: expected a CUDA/HIP tensor, got %s (mal_amd has no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise L.MalError("%s: expected float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return t.data_ptr() if t is not None else None


# Which forward owns a cached step workspace: the whole-step calls keep their intermediate maps, per-task sums and
# coefficients in a workspace that is cached per (device, stream, shape) and reused by the next forward of the same shape.
# A backward that runs AFTER a later forward has overwritten it would return another step's gradients without a word.
_WS_OWNER = {}


def claim_workspace(ws):
    """called by a whole-step forward: returns the token its backward must still find in place"""
    token = object()
    _WS_OWNER[ws.data_ptr()] = token
    return token


def check_workspace(ws, token, what):
    if _WS_OWNER.get(ws.data_ptr()) is not token:
        raise L.MalError("%s: the step's workspace was reused by a later forward of the same shape on this stream; run the "
                         "backward of a step before the next forward (or put the steps on different streams)" % what)


def workspace(dev, B, H, W):
    need = L.load().mal_workspace_bytes(B, H, W)
    key = (dev.index, _stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = ws
    return ws


def _mat(t, name, B):
    t = _req(t, name)
    if t.shape != (B, 4, 4):
        raise L.MalError("%s: expected (%d,4,4), got %s" % (name, B, tuple(t.shape)))
    return t


# ------------------------------------------------------------------ a1
def disp_to_depth(disp, min_depth, max_depth):
    disp = _req(disp, "disp")
    scaled, depth = torch.empty_like(disp), torch.empty_like(disp)
    L.check(L.load().mal_disp_to_depth(_p(disp), disp.numel(), min_depth, max_depth, _p(scaled), _p(depth), _stream()),
            "mal_disp_to_depth")
    return scaled, depth


def disp_to_depth_bwd(disp, g_scaled, g_depth, min_depth, max_depth):
    disp, g_scaled, g_depth = _req(disp, "disp"), _req(g_scaled, "g_scaled"), _req(g_depth, "g_depth")
    out = torch.empty_like(disp)
    L.check(L.load().mal_disp_to_depth_bwd(_p(disp), _p(g_scaled), _p(g_depth), disp.numel(), min_depth, max_depth,
                                           _p(out), _stream()), "mal_disp_to_depth_bwd")
    return out


# ------------------------------------------------------------------ a16
def pose_fwd(axisangles, translations, inverts):
    """[(B,1,3)]*F, [(B,1,3)]*F, [bool]*F -> [(B,4,4)]*F  (transformation_from_parameters)."""
    import ctypes as C
    aa = [_req(a, "axisangle") for a in axisangles]
    tr = [_req(t, "translation") for t in translations]
    B, F = aa[0].shape[0], len(aa)
    Ts = [torch.empty(B, 4, 4, dtype=torch.float32, device=aa[0].device) for _ in range(F)]
    inv = (C.c_int * F)(*[int(bool(i)) for i in inverts])
    L.check(L.load().mal_pose_fwd(L.ptr_array([_p(a) for a in aa]), L.ptr_array([_p(t) for t in tr]), inv, B, F,
                                  L.ptr_array([_p(t) for t in Ts]), _stream()), "mal_pose_fwd")
    return Ts


def pose_bwd(axisangles, translations, inverts, g_Ts, need_aa, need_tr):
    import ctypes as C
    aa = [_req(a, "axisangle") for a in axisangles]
    tr = [_req(t, "translation") for t in translations]
    g_Ts = [_req(g, "g_T") for g in g_Ts]
    B, F = aa[0].shape[0], len(aa)
    g_aa = [torch.empty_like(a) if n else None for a, n in zip(aa, need_aa)]
    g_tr = [torch.empty_like(t) if n else None for t, n in zip(tr, need_tr)]
    inv = (C.c_int * F)(*[int(bool(i)) for i in inverts])
    L.check(L.load().mal_pose_bwd(L.ptr_array([_p(a) for a in aa]), L.ptr_array([_p(t) for t in tr]), inv,
                                  L.ptr_array([_p(g) for g in g_Ts]), B, F, L.ptr_array([_p(g) for g in g_aa]),
                                  L.ptr_array([_p(g) for g in g_tr]), _stream()), "mal_pose_bwd")
    return g_aa, g_tr


# ------------------------------------------------------------------ a2 / a3 / a4
def backproject(depth, inv_K):
    depth = _req(depth, "depth")
    B, _, H, W = depth.shape
    inv_K = _mat(inv_K, "inv_K", B)
    pts = torch.empty(B, 4, H * W, dtype=torch.float32, device=depth.device)
    L.check(L.load().mal_backproject(_p(depth), _p(inv_K), B, H, W, _p(pts), _stream()), "mal_backproject")
    return pts


def backproject_bwd(g_points, inv_K, B, H, W):
    g_points = _req(g_points, "g_points")
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=g_points.device)
    L.check(L.load().mal_backproject_bwd(_p(g_points), _p(_mat(inv_K, "inv_K", B)), B, H, W, _p(out), _stream()),
            "mal_backproject_bwd")
    return out


def project3d(points, K, T, H, W, eps, convention, want_z=False):
    points = _req(points, "points")
    B = points.shape[0]
    K, T = _mat(K, "K", B), _mat(T, "T", B)
    grid = torch.empty(B, H, W, 2, dtype=torch.float32, device=points.device)
    z = torch.empty(B, 1, H, W, dtype=torch.float32, device=points.device) if want_z else None
    L.check(L.load().mal_project3d(_p(points), _p(K), _p(T), B, H, W, eps, convention, _p(grid), _p(z), _stream()),
            "mal_project3d")
    return grid, z


def project3d_bwd(points, K, T, g_grid, g_z, H, W, eps, convention, need_points=True, need_T=True):
    points, g_grid, g_z = _req(points, "points"), _req(g_grid, "g_grid"), _req(g_z, "g_z")
    B = points.shape[0]
    K, T = _mat(K, "K", B), _mat(T, "T", B)
    g_pts = torch.empty_like(points) if need_points else None
    g_T = torch.empty(B, 4, 4, dtype=torch.float32, device=points.device) if need_T else None
    ws = workspace(points.device, B, H, W)
    L.check(L.load().mal_project3d_bwd(_p(points), _p(K), _p(T), _p(g_grid), _p(g_z), B, H, W, eps, convention,
                                       _p(g_pts), _p(g_T), _p(ws), ws.numel(), _stream()), "mal_project3d_bwd")
    return g_pts, g_T


def grid_sample(src, grid, align_corners):
    src, grid = _req(src, "src"), _req(grid, "grid")
    B, C, H, W = src.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=src.device)
    L.check(L.load().mal_grid_sample(_p(src), _p(grid), B, C, H, W, Ho, Wo, int(bool(align_corners)), _p(out),
                                     _stream()), "mal_grid_sample")
    return out


def grid_sample_bwd(src, grid, g_out, align_corners):
    src, grid, g_out = _req(src, "src"), _req(grid, "grid"), _req(g_out, "g_out")
    B, C, H, W = src.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    g_grid = torch.empty_like(grid)
    L.check(L.load().mal_grid_sample_bwd(_p(src), _p(grid), _p(g_out), B, C, H, W, Ho, Wo, int(bool(align_corners)),
                                         _p(g_grid), _stream()), "mal_grid_sample_bwd")
    return g_grid


# ------------------------------------------------------------------ a7 / a8 / a12
def ssim(x, y):
    x, y = _req(x, "x"), _req(y, "y")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    L.check(L.load().mal_ssim(_p(x), _p(y), B, C, H, W, _p(out), _stream()), "mal_ssim")
    return out


def ssim_bwd(x, y, g_out, need_x=True, need_y=True):
    x, y, g_out = _req(x, "x"), _req(y, "y"), _req(g_out, "g_out")
    B, C, H, W = x.shape
    gx = torch.empty_like(x) if need_x else None
    gy = torch.empty_like(x) if need_y else None
    L.check(L.load().mal_ssim_bwd(_p(x), _p(y), _p(g_out), B, C, H, W, _p(gx), _p(gy), _stream()), "mal_ssim_bwd")
    return gx, gy


def reprojection_loss(pred, target, no_ssim=False):
    pred, target = _req(pred, "pred"), _req(target, "target")
    B, C, H, W = pred.shape
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=pred.device)
    L.check(L.load().mal_reprojection_loss(_p(pred), _p(target), B, C, H, W, int(no_ssim), _p(out), _stream()),
            "mal_reprojection_loss")
    return out


def reprojection_loss_bwd(pred, target, g_out, no_ssim=False, need_pred=True, need_target=False):
    pred, target, g_out = _req(pred, "pred"), _req(target, "target"), _req(g_out, "g_out")
    B, C, H, W = pred.shape
    gp = torch.empty_like(pred) if need_pred else None
    gt = torch.empty_like(pred) if need_target else None
    L.check(L.load().mal_reprojection_loss_bwd(_p(pred), _p(target), _p(g_out), B, C, H, W, int(no_ssim), _p(gp),
                                               _p(gt), _stream()), "mal_reprojection_loss_bwd")
    return gp, gt


def smooth_loss(disp, img, normalise, need_grad):
    """-> (loss: 0-dim f64 device tensor, g_disp or None)."""
    disp, img = _req(disp, "disp"), _req(img, "img")
    B, _, H, W = disp.shape
    C = img.shape[1]
    loss = torch.empty((), dtype=torch.float64, device=disp.device)
    g = torch.empty_like(disp) if need_grad else None
    ws = workspace(disp.device, B, H, W)
    L.check(L.load().mal_smooth_loss(_p(disp), _p(img), B, C, H, W, int(bool(normalise)), _p(loss), _p(g), _p(ws),
                                     ws.numel(), _stream()), "mal_smooth_loss")
    return loss, g


# ------------------------------------------------------------------ a5 materialising warp
WARP_TEXELS = 256  # MAL_WARP_TEXELS of include/mal_hip.h


def warp_fwd(disp, K, inv_K, Ts, srcs, min_depth, max_depth, eps, convention, want_depth=True, want_grid=True,
             want_warped=True):
    disp = _req(disp, "disp")
    B, _, H, W = disp.shape
    F = len(Ts)
    K, inv_K = _mat(K, "K", B), _mat(inv_K, "inv_K", B)
    Ts = [_mat(t, "T", B) for t in Ts]
    srcs = [_req(s, "src") for s in srcs]
    if pack_sources and want_warped:  # texel copies (cached per source tensor, shared with the fused passes)
        srcs, convention = [packed_source(s) for s in srcs], convention | WARP_TEXELS
    dev = disp.device
    depth = torch.empty_like(disp) if want_depth else None
    grids = [torch.empty(B, H, W, 2, dtype=torch.float32, device=dev) if want_grid else None for _ in range(F)]
    warped = [torch.empty(B, 3, H, W, dtype=torch.float32, device=dev) if want_warped else None for _ in range(F)]
    L.check(L.load().mal_warp_fwd(_p(disp), _p(K), _p(inv_K), L.ptr_array([_p(t) for t in Ts]),
                                  L.ptr_array([_p(s) for s in srcs]), B, H, W, F, min_depth, max_depth, eps, convention,
                                  _p(depth), L.ptr_array([_p(g) for g in grids]), L.ptr_array([_p(w) for w in warped]),
                                  _stream()), "mal_warp_fwd")
    return depth, grids, warped


def warp_bwd(disp, K, inv_K, Ts, srcs, g_warped, g_grid, g_depth, min_depth, max_depth, eps, convention, need_T):
    disp = _req(disp, "disp")
    B, _, H, W = disp.shape
    F = len(Ts)
    K, inv_K = _mat(K, "K", B), _mat(inv_K, "inv_K", B)
    Ts = [_mat(t, "T", B) for t in Ts]
    srcs = [_req(s, "src") for s in srcs]
    if pack_sources and any(g is not None for g in g_warped):
        srcs, convention = [packed_source(s) for s in srcs], convention | WARP_TEXELS
    g_warped = [_req(g, "g_warped") for g in g_warped]
    g_grid = [_req(g, "g_grid") for g in g_grid]
    g_depth = _req(g_depth, "g_depth")
    dev = disp.device
    g_disp = torch.empty_like(disp)
    g_T = [torch.empty(B, 4, 4, dtype=torch.float32, device=dev) if need_T[f] else None for f in range(F)]
    ws = workspace(dev, B, H, W)
    L.check(L.load().mal_warp_bwd(_p(disp), _p(K), _p(inv_K), L.ptr_array([_p(t) for t in Ts]),
                                  L.ptr_array([_p(s) for s in srcs]), L.ptr_array([_p(g) for g in g_warped]),
                                  L.ptr_array([_p(g) for g in g_grid]), _p(g_depth), B, H, W, F, min_depth, max_depth,
                                  eps, convention, _p(g_disp), L.ptr_array([_p(g) for g in g_T]), _p(ws), ws.numel(),
                                  _stream()), "mal_warp_bwd")
    return g_disp, g_T


# ------------------------------------------------------------------ a8-a10 on materialised candidates
def photo_fwd(target, cands, ident=None, noise=None, ext_mask=None, flags=0, want_min=True, want_argmin=True,
              want_weight=True):
    target = _req(target, "target")
    B, _, H, W = target.shape
    cands = [_req(c, "cand") for c in cands]
    ident, noise, ext_mask = _req(ident, "ident"), _req(noise, "noise"), _req(ext_mask, "ext_mask")
    dev = target.device
    mn = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) if want_min else None
    am = torch.empty(B, 1, H, W, dtype=torch.uint8, device=dev) if want_argmin else None
    wt = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) if want_weight else None
    sums = torch.zeros(8, dtype=torch.float64, device=dev)
    ws = workspace(dev, B, H, W)
    L.check(L.load().mal_photo_fwd(_p(target), L.ptr_array([_p(c) for c in cands]), len(cands), _p(ident), _p(noise),
                                   _p(ext_mask), B, H, W, flags, _p(mn), _p(am), _p(wt), _p(sums), _p(ws), ws.numel(),
                                   _stream()), "mal_photo_fwd")
    return mn, am, wt, sums


def photo_bwd(target, cands, argmin, weight, scale, sums, flags, need):
    target = _req(target, "target")
    B, _, H, W = target.shape
    cands = [_req(c, "cand") for c in cands]
    g = [torch.empty_like(c) if n else None for c, n in zip(cands, need)]
    L.check(L.load().mal_photo_bwd(_p(target), L.ptr_array([_p(c) for c in cands]), len(cands), _p(argmin), _p(weight),
                                   _p(scale), _p(sums), B, H, W, flags, L.ptr_array([_p(x) for x in g]), _stream()),
            "mal_photo_bwd")
    return g


# ------------------------------------------------------------------ texel packing of the sources
_PACKED = {}
pack_sources = True  # fused pass gathers whole texels from (B,H,W,n) copies of the sources (n = mal_texel_floats())


def packed_source(src):
    """(B,3,H,W) -> (B,H,W,n) texel copy, cached per live tensor object and version: the sources of a
    batch are shared by the teacher, ensemble and student passes (manydepth/trainer.py:573-612)."""
    key = (id(src), _stream())
    hit = _PACKED.get(key)
    if hit is not None and hit[0]() is src and hit[1] == src._version:
        return hit[2]
    if len(_PACKED) > 16:
        _PACKED.clear()
    src_c = _req(src, "image")
    B, _, H, W = src_c.shape
    dst = torch.empty(B, H, W, L.load().mal_texel_floats(), dtype=torch.float32, device=src_c.device)
    L.check(L.load().mal_pack_texels(_p(src_c), B, H, W, _p(dst), _stream()), "mal_pack_texels")
    _PACKED[key] = (weakref.ref(src), src._version, dst)
    return dst


def clear_packed_sources():
    _PACKED.clear()


# ------------------------------------------------------------------ the fused pass
# tests set this to a list: every instrumentable fused gradient pass then appends its decision planes (MAL_DEC_*)
DECISION_SINK = None


def pass_fused(disp, K, inv_K, Ts, srcs, target, ident=None, noise=None, ext_mask=None, mono_depth=None,
               mono_reproj=None, ens_reproj=None, min_depth=0.1, max_depth=100.0, eps=1e-7, convention=0, flags=0,
               want_min=True, want_cons_target=False, want_depth=False, disp2=None, sample_scale=None):
    disp, disp2, sample_scale = _req(disp, "disp"), _req(disp2, "disp2"), _req(sample_scale, "sample_scale")
    B, _, H, W = disp.shape
    K, inv_K = _mat(K, "K", B), _mat(inv_K, "inv_K", B)
    Ts = [_mat(t, "T", B) for t in Ts]
    srcs = [_req(s, "src") for s in srcs]
    target = _req(target, "target")
    ident, noise, ext_mask = _req(ident, "ident"), _req(noise, "noise"), _req(ext_mask, "ext_mask")
    mono_depth, mono_reproj, ens_reproj = (_req(mono_depth, "mono_depth"), _req(mono_reproj, "mono_reproj"),
                                           _req(ens_reproj, "ens_reproj"))
    dev = disp.device
    new = lambda: torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
    grad, pose, epi = bool(flags & L.F_GRAD), bool(flags & L.F_POSE_GRAD), bool(flags & L.F_EPILOGUE)
    out = dict(min_reproj=new() if want_min else None, sums=torch.empty(8, dtype=torch.float64, device=dev),
               g_reproj=new() if grad else None, g_cons=new() if (grad and epi) else None,
               g_distil=new() if (grad and epi) else None,
               g_T=[torch.empty(B, 4, 4, dtype=torch.float32, device=dev) if pose else None for _ in range(2)],
               cons_target=new() if (epi and want_cons_target) else None, depth=new() if want_depth else None)
    ws = workspace(dev, B, H, W)
    if pack_sources:
        srcs = [packed_source(s) for s in srcs]
        target = packed_source(target)
        flags |= L.F_SRC_PACKED | L.F_TGT_PACKED
    if DECISION_SINK is not None and grad and ((pose and (flags & L.F_AUTOMASK) and not epi) or
                                               (epi and not pose and not (flags & L.F_AUTOMASK))):
        dec = torch.zeros((L.DEC_PLANES, B, H, W), dtype=torch.int32, device=dev)  # parity instrumentation (tests)
        L.check(L.load().mal_decisions_next_pass(_p(dec)), "mal_decisions_next_pass")
        DECISION_SINK.append(dec)
    L.check(L.load().mal_pass_fused(
        _p(disp), _p(disp2), _p(K), _p(inv_K), L.ptr_array([_p(t) for t in Ts]), L.ptr_array([_p(s) for s in srcs]),
        _p(target), _p(ident), _p(noise), _p(ext_mask), _p(sample_scale), _p(mono_depth), _p(mono_reproj), _p(ens_reproj), B, H, W, 2, min_depth,
        max_depth, eps, convention, flags, _p(out["min_reproj"]), _p(out["sums"]), _p(out["g_reproj"]),
        _p(out["g_cons"]), _p(out["g_distil"]), L.ptr_array([_p(g) for g in out["g_T"]]), _p(out["cons_target"]),
        _p(out["depth"]), _p(ws), ws.numel(), _stream()), "mal_pass_fused")
    return out


def distil_epilogue(multi_depth, mono_depth, multi_reproj, mono_reproj, ens_reproj, ext_mask, flags=0, need_grad=True,
                    want_cons_target=True, ens_depth=None):
    """``ens_depth`` (--learn_ens): the learnt ensemble's depth map; then the fourth gradient returned is d/d ens_depth
    (instead of d/d mono_depth of --dual_distil)."""
    multi_depth, mono_depth = _req(multi_depth, "multi_depth"), _req(mono_depth, "mono_depth")
    multi_reproj, mono_reproj = _req(multi_reproj, "multi_reproj"), _req(mono_reproj, "mono_reproj")
    ens_reproj, ext_mask = _req(ens_reproj, "ens_reproj"), _req(ext_mask, "ext_mask")
    ens_depth = _req(ens_depth, "ens_depth")
    B, _, H, W = multi_depth.shape
    dev = multi_depth.device
    new = lambda: torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
    sums = torch.zeros(8, dtype=torch.float64, device=dev)
    g_cons = new() if need_grad else None
    g_dist = new() if need_grad else None
    ct = new() if want_cons_target else None
    ws = workspace(dev, B, H, W)
    if ens_depth is not None:
        if ens_reproj is None or flags:
            raise L.MalError("distil_epilogue: ens_depth (--learn_ens) needs ens_reproj and takes no flags")
        g_ens = new() if need_grad else None
        L.check(L.load().mal_distil_epilogue_learned(_p(multi_depth), _p(mono_depth), _p(ens_depth), _p(multi_reproj),
                                                     _p(mono_reproj), _p(ens_reproj), _p(ext_mask), B, H, W, _p(sums), _p(g_cons),
                                                     _p(g_dist), _p(g_ens), _p(ct), _p(ws), ws.numel(), _stream()),
                "mal_distil_epilogue_learned")
        return sums, g_cons, g_dist, g_ens, ct
    g_mono = new() if (need_grad and (flags & L.F_DUAL_DISTIL)) else None
    L.check(L.load().mal_distil_epilogue(_p(multi_depth), _p(mono_depth), _p(multi_reproj), _p(mono_reproj),
                                         _p(ens_reproj), _p(ext_mask), B, H, W, flags, _p(sums), _p(g_cons), _p(g_dist),
                                         _p(g_mono), _p(ct), _p(ws), ws.numel(), _stream()), "mal_distil_epilogue")
    return sums, g_cons, g_dist, g_mono, ct


def matching_mask(lowest_cost, mono_depth, consistency_mask=None):
    lowest_cost, mono_depth = _req(lowest_cost, "lowest_cost"), _req(mono_depth, "mono_depth")
    consistency_mask = _req(consistency_mask, "consistency_mask")
    out = torch.empty_like(lowest_cost)
    L.check(L.load().mal_matching_mask(_p(lowest_cost), _p(mono_depth), _p(consistency_mask), lowest_cost.numel(),
                                       _p(out), _stream()), "mal_matching_mask")
    return out


# ------------------------------------------------------------------ assembly
def axpy_maps(maps, scales=None, denoms=None, mults=None, eps=None, out=None, accumulate=False):
    """out = [out +] sum_k mult_k * (*scale_k) / (*denom_k + eps_k) * maps[k]; scalars stay on the device."""
    n = len(maps)
    maps = [_req(m, "map") for m in maps]
    if out is None:
        out = torch.empty_like(maps[0])
        accumulate = False
    scales = scales or [None] * n
    denoms = denoms or [None] * n
    L.check(L.load().mal_axpy_maps(n, L.ptr_array([_p(m) for m in maps]), L.ptr_array([_p(s) for s in scales]),
                                   L.ptr_array([_p(d) for d in denoms]), L.f32_array(mults or [1.0] * n),
                                   L.f32_array(eps or [0.0] * n), maps[0].numel(), _p(out), int(accumulate), _stream()),
            "mal_axpy_maps")
    return out


def finish_scalars(num, den=None, eps=0.0, mult=1.0):
    """f64 device scalars -> f32: out[k] = mult*num[k] / (den[k] + eps)."""
    n = num.numel()
    out = torch.empty(num.shape, dtype=torch.float32, device=num.device)
    L.check(L.load().mal_finish_scalars(_p(num), _p(den), eps, mult, n, _p(out), _stream()), "mal_finish_scalars")
    return out


def sum_f64(x):
    x = _req(x, "x")
    out = torch.empty((), dtype=torch.float64, device=x.device)
    ws = workspace(x.device, 1, 2, 2)
    L.check(L.load().mal_sum_f64(_p(x), x.numel(), _p(out), _p(ws), ws.numel(), _stream()), "mal_sum_f64")
    return out


# ------------------------------------------------------------------ per-scale disparity -> full resolution
def upsample_bilinear(x, H, W):
    """F.interpolate(x, [H, W], mode="bilinear", align_corners=False) of a (B,1,h,w) map (trainer.py:1094-1096), with
    ATen's association of the four products"""
    x = _req(x, "x")
    B, _, h, w = x.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    L.check(L.load().mal_upsample_bilinear(_p(x), B, h, w, H, W, _p(out), _stream()), "mal_upsample_bilinear")
    return out


def upsample_bilinear_adjoint(g_out, h, w):
    """its adjoint for power-of-two factors <= 8, gathered in a fixed order (ATen scatters with atomics)"""
    g_out = _req(g_out, "g_out")
    B, _, H, W = g_out.shape
    g = torch.empty((B, 1, h, w), dtype=torch.float32, device=g_out.device)
    L.check(L.load().mal_upsample_bilinear_adjoint(_p(g_out), B, h, w, H, W, _p(g), _stream()), "mal_upsample_bilinear_adjoint")
    return g
