"""Golden vectors for DualRefine's loss loops from the REFERENCE's own Trainer methods.

TEST INFRASTRUCTURE ONLY.  Run in the authoring container only (needs /root/reference):

    python -m oracle.gen_golden_dr

``dualrefine/trainer.py`` cannot be imported as shipped: it imports cv2 and tensorboardX (absent here), and
``dualrefine/networks/__init__.py`` pulls the never-committed ``networks/lib``.  None of that is touched by the four
methods on the hot path -- ``Trainer.generate_images_pred`` (:395-451), ``compute_losses`` (:530-697),
``pose_update_generate_images_pred`` (:457-480) and ``compute_pose_update_losses`` (:699-767) -- so the module is
imported with inert stand-ins for those imports (empty modules; ``dualrefine.networks`` as a bare namespace whose
``utils.utils`` is the real file, as oracle/gen_golden_epi.py does) and the methods are called UNBOUND on a plain
namespace object that carries what they read from ``self`` (``opt``, the reference's own ``BackprojectDepth`` /
``Project3D`` / ``SSIM`` objects, ``device``, ``f_thres``, ``num_scales``).  The debug ``print``s of the shipped methods
run; the ``exit(0)`` at :484 is not reached through ``pose_update_generate_images_pred``'s arithmetic and is trapped.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_trainer():
    sys.path.insert(0, REF)
    for name in ("cv2", "tensorboardX"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.SummaryWriter = object
            sys.modules[name] = m
    import dualrefine  # noqa: F401  (the package itself: an empty __init__)
    for name, path in (("dualrefine.networks", os.path.join(REF, "dualrefine", "networks")),
                       ("dualrefine.networks.utils", os.path.join(REF, "dualrefine", "networks", "utils")),
                       ("dualrefine.datasets", None)):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path] if path else []
            sys.modules[name] = m
    import dualrefine.layers as DL
    import dualrefine.trainer as DT
    return DL, DT.Trainer


def case(tag, scales):
    """``scales`` [0]: the refinement iterations of the full resolution; [0, 1, 2, 3]: upstream's default list
    (dualrefine/options.py:65-69) -- the loops visit scale 0 and 2 with n_losses+1 iterations, skip scale 1 and take iteration 0
    of scale 3 (trainer.py:403-407,536-547), every disparity upsampled to full resolution, total / 4."""
    sys.path.insert(0, ROOT)
    from mal_amd.synthetic import make_batch
    from oracle.gen_golden import quantize_batch, pack_inputs
    from oracle.mal_oracle import dr_default_opt
    DL, Trainer = import_trainer()
    torch.set_num_threads(8)
    B, H, W = 2, 40, 72
    q = quantize_batch(make_batch(B, H, W, seed=321))
    opt = dr_default_opt(height=H, width=W, batch_size=B, n_losses=1, scales=list(scales))
    me = types.SimpleNamespace(opt=opt, device="cpu", f_thres=1, num_scales=len(opt.scales), ssim=DL.SSIM(),
                               backproject_depth={0: DL.BackprojectDepth(B, H, W)}, project_3d={0: DL.Project3D(B, H, W)})
    me.compute_reprojection_loss = types.MethodType(Trainer.compute_reprojection_loss, me)
    me.compute_loss_masks = Trainer.compute_loss_masks  # a staticmethod upstream
    inputs = {("color", f, 0): q[k] for f, k in ((0, "color0"), (-1, "color_m1"), (1, "color_p1"))}
    inputs[("K", 0)], inputs[("inv_K", 0)] = q["K"], q["inv_K"]
    leaves = {k: q[k].clone().requires_grad_(True) for k in ("disp_teacher", "disp_student", "axisangle_m1", "translation_m1",
                                                            "axisangle_p1", "translation_p1")}
    T_m1 = DL.transformation_from_parameters(leaves["axisangle_m1"], leaves["translation_m1"], True)
    T_p1 = DL.transformation_from_parameters(leaves["axisangle_p1"], leaves["translation_p1"], False)
    outputs = {("disp", 0, 0): leaves["disp_teacher"], ("disp", 0, 1): leaves["disp_student"],
               ("cam_T_cam", 0, -1): T_m1, ("cam_T_cam", 0, 1): T_p1, ("cam_T_cam", 0, -1, 1): T_m1 * 1.0,
               "consistency_mask": q["consistency_mask"].unsqueeze(1)}
    extra = {}
    for s in scales:
        if s == 0:
            continue
        inputs[("color", 0, s)] = F.avg_pool2d(q["color0"], 2 ** s)
        for it, name in ((0, "disp_teacher"), (1, "disp_student")):
            if s == 1 or (s == 3 and it > 0):
                continue  # never read (trainer.py:404-407)
            leaf = F.avg_pool2d(q[name], 2 ** s).half().float().clone().requires_grad_(True)
            extra["disp_s%d_it%d" % (s, it)] = leaf
            outputs[("disp", s, it)] = leaf
    leaves.update(extra)
    noise_seed = 4321
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):  # the shipped methods print debug values
        Trainer.generate_images_pred(me, inputs, outputs)
        torch.manual_seed(noise_seed)  # compute_losses draws randn per (scale, deq_iter) from the global CPU generator
        losses = Trainer.compute_losses(me, inputs, outputs)
        losses["loss"].backward()
        grads = {k: t.grad.clone() for k, t in leaves.items()}
        try:
            Trainer.pose_update_generate_images_pred(me, inputs, outputs)
        except SystemExit:
            pass  # :484 `exit(0)` after the warp has been written
        torch.manual_seed(noise_seed + 1)
        pl = Trainer.compute_pose_update_losses(me, inputs, outputs)
    d = {"in/noise_seed": np.int64(noise_seed)}
    for k, v in losses.items():
        d["losses/" + k] = np.float64(v.item())
    for k, v in pl.items():
        d["pose_losses/" + k] = np.float64(v.item())
    for k, g in grads.items():
        d["grad/" + k] = g.numpy()
    d["color_m1_pose"] = outputs[("color", -1, 0, 0, 1)].detach().numpy()
    d["depth_0_1"] = outputs[("depth", 0, 0, 1)].detach().numpy()
    for k, t in extra.items():
        d["in/" + k] = t.detach().half().numpy()
    d["scales"] = np.array(list(scales), dtype=np.int64)
    d.update(pack_inputs(q))
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: round(v.item(), 6) for k, v in losses.items()},
          {k: round(v.item(), 6) for k, v in pl.items()})



def main():
    if sys.argv[1:] != ["scales"]:  # `python -m oracle.gen_golden_dr scales` writes the four-scale case only
        case("dualrefine_b2_40x72", [0])
    case("dualrefine_b2_40x72_scales0123", [0, 1, 2, 3])


if __name__ == "__main__":
    main()
