"""ctypes binding of libmal_hip.so (include/mal_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.  The
library is built in-tree by ``python -m mal_amd.build`` (``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

# torch first: the process must hold ONE HIP runtime, the one PyTorch-ROCm loads.  If libmal_hip.so were loaded
# before torch, the system libamdhip64 would come in first and kernels registered with it fail to launch on
# torch's streams ("HIP kernel launch failed").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAL_HIP_LIB points at another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("MAL_HIP_LIB") or os.path.join(_HERE, "lib", "libmal_hip.so")

c_fp = C.c_void_p      # device float*
c_pp = C.c_void_p      # host array of device pointers (we pass a ctypes array)
i32, f32, sz, vp = C.c_int, C.c_float, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); mirrors include/mal_hip.h one to one
SIGNATURES = {
    "mal_version": (i32, []),
    "mal_strerror": (C.c_char_p, [i32]),
    "mal_device_check": (i32, []),
    "mal_workspace_bytes": (sz, [i32, i32, i32]),
    "mal_disp_to_depth": (i32, [c_fp, sz, f32, f32, c_fp, c_fp, vp]),
    "mal_disp_to_depth_bwd": (i32, [c_fp, c_fp, c_fp, sz, f32, f32, c_fp, vp]),
    "mal_backproject": (i32, [c_fp, c_fp, i32, i32, i32, c_fp, vp]),
    "mal_backproject_bwd": (i32, [c_fp, c_fp, i32, i32, i32, c_fp, vp]),
    "mal_project3d": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, f32, i32, c_fp, c_fp, vp]),
    "mal_project3d_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, f32, i32, c_fp, c_fp, vp, sz, vp]),
    "mal_grid_sample": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, vp]),
    "mal_grid_sample_bwd": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, vp]),
    "mal_ssim": (i32, [c_fp, c_fp, i32, i32, i32, i32, c_fp, vp]),
    "mal_ssim_bwd": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, vp]),
    "mal_reprojection_loss": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, vp]),
    "mal_reprojection_loss_bwd": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp, vp]),
    "mal_smooth_loss": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp, vp, sz, vp]),
    "mal_warp_fwd": (i32, [c_fp, c_fp, c_fp, c_pp, c_pp, i32, i32, i32, i32, f32, f32, f32, i32, c_fp, c_pp, c_pp, vp]),
    "mal_warp_bwd": (i32, [c_fp, c_fp, c_fp, c_pp, c_pp, c_pp, c_pp, c_fp, i32, i32, i32, i32, f32, f32, f32, i32,
                           c_fp, c_pp, vp, sz, vp]),
    "mal_photo_fwd": (i32, [c_fp, c_pp, i32, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, vp, sz, vp]),
    "mal_photo_bwd": (i32, [c_fp, c_pp, i32, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_pp, vp]),
    "mal_pass_fused": (i32, [c_fp, c_fp, c_fp, c_fp, c_pp, c_pp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32,
                             i32, f32, f32, f32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp, c_pp, c_fp, c_fp, vp, sz, vp]),
    "mal_distil_epilogue": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp,
                                  c_fp, vp, sz, vp]),
    "mal_distil_epilogue_learned": (i32, [c_fp] * 7 + [i32, i32, i32] + [c_fp] * 5 + [vp, sz, vp]),
    "mal_matching_mask": (i32, [c_fp, c_fp, c_fp, sz, c_fp, vp]),
    "mal_texel_floats": (i32, []),
    "mal_costvol_channel_last": (i32, []),
    "mal_pack_texels": (i32, [c_fp, i32, i32, i32, c_fp, vp]),
    "mal_axpy_maps": (i32, [i32, c_pp, c_pp, c_pp, C.POINTER(f32), C.POINTER(f32), sz, c_fp, i32, vp]),
    "mal_finish_scalars": (i32, [c_fp, c_fp, f32, f32, i32, c_fp, vp]),
    "mal_sum_f64": (i32, [c_fp, sz, c_fp, vp, sz, vp]),
    "mal_pose_fwd": (i32, [c_pp, c_pp, C.POINTER(i32), i32, i32, c_pp, vp]),
    "mal_pose_bwd": (i32, [c_pp, c_pp, C.POINTER(i32), c_pp, i32, i32, c_pp, c_pp, vp]),
    "mal_cost_volume": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, f32, i32, c_fp, c_fp, c_fp,
                              c_fp, c_fp, vp]),
    "mal_dyn_workspace_bytes": (sz, [i32]),
    "mal_dyn_instance_fwd": (i32, [vp, vp, i32, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, vp, vp, vp, sz, vp]),
    "mal_dyn_instance_bwd": (i32, [vp, vp, i32, vp, vp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, vp]),
    "mal_dyn_batch_fwd": (i32, [vp, i32, i32, i32, i32, i32, vp]),
    "mal_dyn_batch_bwd": (i32, [vp, i32, i32, i32, i32, vp]),
    "mal_epipolar_coords": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, f32, f32, c_fp, c_fp, c_fp, vp]),
    "mal_coord_sample_l1": (i32, [c_fp, vp, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, vp]),
    "mal_epipolar_coords_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "mal_epipolar_coords_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, f32, f32, c_fp, c_fp, c_fp,
                                      vp, sz, vp]),
    "mal_coord_sample_l1_bwd": (i32, [c_fp, vp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, vp, c_fp, vp, sz, vp]),
    "mal_coord_sample_l1_bwd_workspace_bytes": (sz, [i32]),
    "mal_epipolar_coords_of_depths": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, vp]),
    "mal_epipolar_gradcoords": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, vp]),
    "mal_direct_align_workspace_bytes": (sz, [i32, i32, i32]),
    "mal_direct_align_update": (i32, [c_fp, c_fp, c_fp, i32, c_fp, c_fp, vp]),
    "mal_direct_align_normal_eq": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp, vp, sz, vp]),
    "mal_epipolar_gradcoords_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "mal_epipolar_gradcoords_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, vp, sz, vp]),
    "mal_direct_align_normal_eq_bwd": (i32, [c_fp] * 10 + [i32] * 5 + [c_fp] * 7 + [vp, sz, vp]),
    "mal_direct_align_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "mal_direct_align_update_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp, c_fp, vp]),
    "mal_step_workspace_bytes": (sz, [i32, i32, i32]),
    "mal_loss_step_fwd": (i32, [vp]),
    "mal_loss_step_bwd": (i32, [vp]),
    "mal_loss_step_warp": (i32, [vp]),
    "mal_loss_step_student_ready": (i32, [vp]),
    "mal_loss_step_abort": (i32, [vp]),
    "mal_loss_step_tail_begin": (i32, [vp, vp]),
    "mal_loss_step_tail_cancel": (i32, [vp]),
    "mal_loss_step_teacher_replay": (i32, [vp, i32]),
    "mal_tiebreak_noise": (i32, [C.c_uint64, C.c_uint64, i32, i32, i32, c_fp, vp]),
    "mal_ms_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "mal_loss_multiscale_warp": (i32, [vp]),
    "mal_loss_multiscale_abort": (i32, [vp]),
    "mal_loss_multiscale_fwd": (i32, [vp]),
    "mal_loss_multiscale_bwd": (i32, [vp]),
    "mal_dr_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "mal_dr_loss_fwd": (i32, [vp]),
    "mal_dr_loss_bwd": (i32, [vp]),
    "mal_upsample_bilinear": (i32, [c_fp, i32, i32, i32, i32, i32, c_fp, vp]),
    "mal_upsample_bilinear_adjoint": (i32, [c_fp, i32, i32, i32, i32, i32, c_fp, vp]),
    "mal_set_option": (i32, [C.c_char_p, i32]),
    "mal_build_has_experiments": (i32, []),
    "mal_struct_bytes": (sz, [i32]),
    "mal_event_create": (vp, []),
    "mal_event_destroy": (i32, [vp]),
    "mal_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f32)]),
    "mal_profile_next_pass": (i32, [vp, vp]),
    "mal_march_geometry": (i32, [i32, i32, i32, i32] + [C.POINTER(i32)] * 4),
    "mal_clock_probe": (i32, [vp, i32, vp]),
    "mal_decisions_next_pass": (i32, [vp]),
}

class DynItem(C.Structure):
    """mal_dyn_item (include/mal_hip.h)."""
    _fields_ = [("mask_last", vp), ("mask_next", vp), ("num", i32), ("img_last", vp), ("img_next", vp),
                ("ori_last", vp), ("ori_next", vp), ("delta", vp), ("flags", vp), ("ws", vp), ("ws_bytes", sz),
                ("g_ori_last", vp), ("g_ori_next", vp), ("g_img_last", vp), ("g_img_next", vp), ("idx_last", vp), ("idx_next", vp),
                ("prefilled", i32), ("g_tmp_last", vp), ("g_tmp_next", vp), ("region_only", i32), ("n_last", i32), ("n_next", i32)]


class StepArgs(C.Structure):
    """mal_step_args (include/mal_hip.h)."""
    _fields_ = ([("B", i32), ("H", i32), ("W", i32), ("min_depth", f32), ("max_depth", f32), ("flags", i32),
                 ("w_main", f32), ("w_distil", f32)] +
                [(n, vp) for n in ("color0", "color_m1", "color_p1", "K", "inv_K", "disp_teacher", "disp_student",
                                   "axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1",
                                   "consistency_mask", "augmentation_keep", "lowest_cost", "noise", "losses", "loss_total",
                                   "mono_reproj", "ens_reproj", "multi_reproj", "consistency_mask_out", "g_total",
                                   "g_disp_teacher", "g_disp_student", "g_axisangle_m1", "g_translation_m1",
                                   "g_axisangle_p1", "g_translation_p1", "ws")] +
                [("ws_bytes", sz), ("stream", vp), ("dec_teacher", vp), ("dec_student", vp),
                 ("noise_seed", C.c_uint64), ("noise_step", C.c_uint64), ("noise_counter", vp), ("noise_out", vp)] +
                [(n, vp) for n in ("warp_m1", "warp_p1", "syn_m1", "syn_p1", "g_syn_m1", "g_syn_p1", "g_warp_m1", "g_warp_p1")] +
                [("warp_sample_stride", i32), ("warp2_m1", vp), ("warp2_p1", vp), ("syn_region", vp), ("g_syn_region_m1", vp),
                 ("g_syn_region_p1", vp), ("ens_disp", vp), ("g_ens_disp", vp)] +
                [(n, vp) for n in ("warp_s_m1", "warp_s_p1", "syn_s_m1", "syn_s_p1", "g_syn_s_m1", "g_syn_s_p1", "g_warp_s_m1",
                                   "g_warp_s_p1", "syn_s_region", "g_syn_s_region_m1", "g_syn_s_region_p1")])


class MsArgs(C.Structure):
    """mal_ms_args (include/mal_hip.h)."""
    _fields_ = ([("B", i32), ("H", i32), ("W", i32), ("sclm", i32), ("min_depth", f32), ("max_depth", f32), ("flags", i32)] +
                [(n, vp) for n in ("color0", "color_m1", "color_p1")] + [("color0_s", vp * 4), ("K", vp), ("inv_K", vp),
                 ("disp_teacher", vp * 4), ("disp_student", vp * 4)] +
                [(n, vp) for n in ("axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1", "consistency_mask",
                                   "augmentation_keep", "lowest_cost")] +
                [("noise", vp * 4), ("noise_seed", C.c_uint64), ("noise_step", C.c_uint64), ("noise_counter", vp),
                 ("losses", vp), ("loss_total", vp), ("consistency_mask_out", vp), ("g_total", vp),
                 ("g_disp_teacher", vp * 4), ("g_disp_student", vp * 4)] +
                [(n, vp) for n in ("g_axisangle_m1", "g_translation_m1", "g_axisangle_p1", "g_translation_p1", "ws")] +
                [("ws_bytes", sz), ("stream", vp)] +
                [(n, vp * 4) for n in ("warp_m1", "warp_p1", "syn_m1", "syn_p1", "g_syn_m1", "g_syn_p1", "g_warp_m1", "g_warp_p1",
                                       "syn_region", "g_syn_region_m1", "g_syn_region_p1")] +
                [("warp_sample_stride", i32), ("syn_sparse", i32), ("dec_teacher", vp * 4), ("dec_student", vp * 4)])


class DrArgs(C.Structure):
    """mal_dr_args (include/mal_hip.h)."""
    _fields_ = ([("B", i32), ("H", i32), ("W", i32), ("n_iters", i32), ("min_depth", f32), ("max_depth", f32),
                 ("smooth_weight", f32), ("flags", i32)] +
                [(n, vp) for n in ("color0", "color_m1", "color_p1", "K", "inv_K")] +
                [("disp", vp * 4), ("T_m1", vp * 4), ("T_p1", vp * 4), ("consistency_mask", vp), ("noise", vp * 4),
                 ("noise_seed", C.c_uint64), ("noise_step", C.c_uint64), ("noise_counter", vp), ("losses", vp), ("loss_total", vp),
                 ("g_total", vp), ("g_disp", vp * 4), ("g_T_m1", vp * 4), ("g_T_p1", vp * 4),
                 ("ws", vp), ("ws_bytes", sz), ("stream", vp), ("scale", i32), ("color0_s", vp), ("disp_lo", vp * 4),
                 ("g_disp_lo", vp * 4), ("texels_from", vp), ("dec", vp * 4)] +
                [(n, vp) for n in ("pu_disp_m1", "pu_disp_p1", "pu_T_m1", "pu_T_p1", "pu_noise", "g_pu_total", "g_pu_disp_m1",
                                   "g_pu_disp_p1", "g_pu_T_m1", "g_pu_T_p1", "pu_dec", "pu_loss_total")] +
                [(n, i32) for n in ("pu_disp_m1_into", "pu_disp_p1_into", "pu_T_m1_into", "pu_T_p1_into")])


DR_MAX_ITERS = 4
DR_NO_AUTOMASK, DR_NO_MOTION_MASK, DR_NOISE_PHILOX, DR_AVG, DR_NO_SSIM, DR_POSE_UPDATE = 1, 2, 4, 8, 16, 32
DR_POSE_NOISE_KEY = 0x706f73655f757064
MS_MAX_SCALES = 4
STEP_NO_ENS, STEP_AUG_MASK, STEP_NOISE_PHILOX, STEP_TEMPORAL, STEP_SYN_SPARSE, STEP_DUAL_DISTIL, STEP_TEXEL_INPUTS = 1, 2, 4, 8, 16, 32, 64
STEP_MAIN_TEMPORAL, STEP_SYN_S_SPARSE, STEP_NO_SSIM, STEP_NO_MOTION_MASK, STEP_NO_AUG, STEP_ENSEMBLE = 128, 256, 512, 1024, 2048, 4096
# decision planes of mal_step_args.dec_teacher / dec_student (MAL_DEC_*)
DEC_WIN, DEC_DISTIL, DEC_SMOOTH_X, DEC_SMOOTH_Y, DEC_TAP0, DEC_TAP1, DEC_L1, DEC_PLANES = 0, 1, 2, 3, 4, 5, 6, 7

# flags (include/mal_hip.h)
F_AUTOMASK, F_GRAD, F_POSE_GRAD, F_NO_SSIM, F_AVG, F_EPILOGUE, F_DUAL_DISTIL, F_SRC_PACKED, F_TGT_PACKED = 1, 2, 4, 8, 16, 32, 64, 128, 256

_lib = None


class MalError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MalError("%s not found: build it with `python -m mal_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback for the MAL loss path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # the argument blocks are laid out twice (include/mal_hip.h, the ctypes Structures above): a stale library or a field added on
    # one side only must not get as far as a kernel launch
    for which, cls in enumerate((StepArgs, MsArgs, DrArgs, DynItem)):
        if lib.mal_struct_bytes(which) != C.sizeof(cls):
            raise MalError("%s: sizeof(%s) is %d in the library, %d in mal_amd/_lib.py -- rebuild (python -m mal_amd.build)"
                           % (LIB_PATH, cls.__name__, lib.mal_struct_bytes(which), C.sizeof(cls)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise MalError("%s failed: %s (%d)" % (what, load().mal_strerror(rc).decode(), rc))


def ptr_array(ptrs):
    """Host array of device pointers (None -> NULL)."""
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p if p else None
    return arr


def f32_array(vals):
    arr = (C.c_float * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = float(v)
    return arr
