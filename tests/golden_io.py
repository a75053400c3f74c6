"""Load the committed golden vectors (tests/golden/*.npz, written by oracle/gen_golden.py
from the reference's own functions) back into the batch layout of mal_amd.synthetic."""
import ast
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STEP_CASES = [
    "step_b2_32x64_distil", "step_b2_32x64_noens", "step_b2_32x64_lossblc", "step_b2_32x64_dual",
    "step_b2_32x64_temporal", "step_b2_32x64_temporal_main", "step_b3_37x50_distil", "step_b2_32x64_learnens",
]
BIG_CASE = "step_b2_192x640_distil"
LAYER_CASES = ["layers_b2_24x40", "layers_b1_19x33"]


def load(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def batch_from_golden(z):
    b = {}
    for k in ("color0", "color_m1", "color_p1"):
        b[k] = torch.from_numpy(z["in/" + k].astype(np.float32)) / 255
    for k in ("disp_teacher", "disp_student", "lowest_cost"):
        b[k] = torch.from_numpy(z["in/" + k].astype(np.float32))
    for k in ("axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1", "K", "inv_K"):
        b[k] = torch.from_numpy(z["in/" + k].copy())
    b["consistency_mask"] = torch.from_numpy(z["in/consistency_mask"].astype(np.float32))
    b["augmentation_mask"] = torch.from_numpy(z["in/augmentation_mask"].astype(np.float32))
    if "in/disp_ens" in z:
        b["disp_ens"] = torch.from_numpy(z["in/disp_ens"].astype(np.float32))
    if "in/syn_rects" in z:
        b["syn_rects"] = [tuple(int(v) for v in r) for r in z["in/syn_rects"]]
    return b


MULTISCALE_CASE = "multiscale_b2_48x96_sclm3"
MULTISCALE_TEMPORAL_CASE = "multiscale_b2_48x96_sclm3_temporal"  # --temporal on the non-distil path (trainer.py:1161-1162,1279-1283)


def multiscale_dicts(z, pose_fn, device="cpu"):
    """the sclm>0 fixture (oracle/gen_golden.py run_reference_multiscale) as the reference's dicts: per-scale disparities
    for both networks (leaves), ("color", 0, s) = the target pooled by 2**s"""
    from mal_amd.synthetic import to_dicts
    b = batch_from_golden(z)
    sclm = int(z["sclm"])
    inputs, mono_outputs, outputs, leaves = to_dicts(b, pose_fn, device=device)
    for s in range(1, sclm + 1):
        inputs[("color", 0, s)] = torch.nn.functional.avg_pool2d(b["color0"], 2 ** s).to(device)
        for name, outs in (("disp_teacher", mono_outputs), ("disp_student", outputs)):
            leaf = torch.from_numpy(z["in/%s_s%d" % (name, s)].astype(np.float32)).to(device).requires_grad_(True)
            leaves["%s_s%d" % (name, s)] = leaf
            outs[("disp", s)] = leaf
    return b, sclm, inputs, mono_outputs, outputs, leaves


DUALREFINE_CASES = ["dualrefine_b2_40x72", "dualrefine_b2_40x72_scales0123"]


def dualrefine_dicts(z, pose_fn, device="cpu", dtype=torch.float32):
    """a DualRefine fixture (oracle/gen_golden_dr.py: the reference's own Trainer methods called unbound) as that trainer's
    dicts: 4-tuple keys ("disp", scale, deq_iter); the second case carries upstream's default scales [0, 1, 2, 3]
    (scale 1 is never read, scale 3 has iteration 0 only: dualrefine/trainer.py:403-407).  Returns
    (batch, scales, units, inputs, outputs, leaves); units = the (scale, iteration) pairs in the order the loops visit them."""
    b = batch_from_golden(z)
    mv = lambda t: (t.to(dtype) if t.is_floating_point() else t).to(device)
    scales = [int(v) for v in z["scales"]] if "scales" in z else [0]
    inputs = {("color", f, 0): mv(b[k]) for f, k in ((0, "color0"), (-1, "color_m1"), (1, "color_p1"))}
    inputs[("K", 0)], inputs[("inv_K", 0)] = mv(b["K"]), mv(b["inv_K"])
    leaves = {k: mv(b[k]).clone().requires_grad_(True) for k in ("disp_teacher", "disp_student", "axisangle_m1", "translation_m1",
                                                                 "axisangle_p1", "translation_p1")}
    T_m1 = pose_fn(leaves["axisangle_m1"], leaves["translation_m1"], True)
    T_p1 = pose_fn(leaves["axisangle_p1"], leaves["translation_p1"], False)
    outputs = {("disp", 0, 0): leaves["disp_teacher"], ("disp", 0, 1): leaves["disp_student"],
               ("cam_T_cam", 0, -1): T_m1, ("cam_T_cam", 0, 1): T_p1, ("cam_T_cam", 0, -1, 1): T_m1 * 1.0,
               "consistency_mask": mv(b["consistency_mask"]).unsqueeze(1)}
    units = [(0, 0), (0, 1)]
    for s in scales:
        if s == 0:
            continue
        inputs[("color", 0, s)] = mv(torch.nn.functional.avg_pool2d(b["color0"], 2 ** s))
        for it in (0, 1):
            key = "in/disp_s%d_it%d" % (s, it)
            if key in z:
                leaf = mv(torch.from_numpy(z[key].astype(np.float32))).requires_grad_(True)
                leaves["disp_s%d_it%d" % (s, it)] = leaf
                outputs[("disp", s, it)] = leaf
                units.append((s, it))
    return b, scales, units, inputs, outputs, leaves


def multiscale_noises(z, shape, sclm):
    torch.manual_seed(int(z["in/noise_seed"]))
    return [torch.randn(shape) for _ in range(sclm + 1)], [torch.randn(shape) for _ in range(sclm + 1)]


def opt_kwargs(z):
    return dict(ast.literal_eval(str(z["opt"])))


def noises(z, shape):
    if "in/noise_mono" in z:
        return torch.from_numpy(z["in/noise_mono"].copy()), torch.from_numpy(z["in/noise_main"].copy())
    torch.manual_seed(int(z["in/noise_seed"]))
    n0 = torch.randn(shape)
    n1 = torch.randn(shape)
    return n0, n1


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-6)


def assert_close(a, b, rtol, name="", floor=1e-6, max_bad_frac=0.0):
    """|a-b| <= rtol*max(|b|, floor) elementwise; ``max_bad_frac`` of elements may miss
    (used only for per-pixel maps at argmin ties / clamp boundaries, stated at call sites)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    bad = np.abs(a - b) > rtol * np.maximum(np.abs(b), floor)
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_bad_frac, "%s: %.3g of elements off (allowed %.3g), worst rel %.3g" % (
        name, frac, max_bad_frac, (np.abs(a - b) / np.maximum(np.abs(b), floor)).max())
