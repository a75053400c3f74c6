"""TEST INFRASTRUCTURE (authoring container only): golden vectors of DualRefine's epipolar correlation lookup from the
REFERENCE's own classes -> tests/golden/epi_*.npz.

``dualrefine/networks/__init__.py`` imports the never-committed ``networks/lib``, so the package is
registered as a bare namespace and the two module files that hold this path -- ``networks/utils/utils.py``
(``Reprojections``) and ``networks/corr.py`` (``CoordSampler``) -- are imported on their own; nothing of them is
modified or stood in for.  Run: python -m oracle.gen_golden_epi
"""
import importlib
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_modules():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import dualrefine  # noqa: F401
    if "dualrefine.networks" not in sys.modules:
        pkg = types.ModuleType("dualrefine.networks")
        pkg.__path__ = [os.path.join(REF, "dualrefine", "networks")]
        sys.modules["dualrefine.networks"] = pkg
    return (importlib.import_module("dualrefine.networks.utils.utils"), importlib.import_module("dualrefine.networks.corr"))


def make_case(B, C, h, w, seed, trans=0.15):
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4).repeat(B, 1, 1)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 0.58 * w, 1.92 * h, 0.5 * w, 0.5 * h
    low = torch.nn.functional.interpolate(torch.rand(B, 1, max(h // 4, 2), max(w // 4, 2), generator=g), size=(h, w),
                                          mode="bilinear", align_corners=False)
    depth = (1.0 + 8.0 * low).contiguous()
    poses = torch.eye(4).repeat(B, 1, 1)
    poses[:, :3, 3] = trans * torch.randn(B, 3, generator=g)
    ang = 0.02 * torch.randn(B, generator=g)
    poses[:, 0, 0], poses[:, 0, 2], poses[:, 2, 0], poses[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
    # features quantised to fp16-representable values so the fixture stores them exactly in half the bytes
    f1 = torch.randn(B, C, h, w, generator=g).half().float()
    f2 = torch.randn(B, C, h, w, generator=g).half().float()
    return K, depth, poses, f1, f2


CASES = {  # tag: (B, C, h, w, r, levels, heads, gap_factor, seed)
    "epi_b2_c16_12x20_r4_l3": (2, 16, 12, 20, 4, 3, 1, "depth", 11),
    "epi_b1_c8_9x13_r2_l2_h2": (1, 8, 9, 13, 2, 2, 2, "depth", 12),
}


def main():
    U, Cn = reference_modules()
    os.makedirs(OUT, exist_ok=True)
    for tag, (B, C, h, w, r, L, heads, gf, seed) in CASES.items():
        K, depth, poses, f1, f2 = make_case(B, C, h, w, seed)
        args = SimpleNamespace(corr_radius=r, disable_pose_updates=True, gap_factor=gf, gap_factor_depth_ratio=8, num_levels=L,
                               min_depth=0.1, max_depth=100.0, use_depth_bins_for_masking=False)
        R = U.Reprojections(args)
        with torch.no_grad():
            R.delta.fill_(0.7)
        R._reg_intrinsics(K)
        R.update_depth_bins(9.0, 1.0, 4.0, 4.0)
        S = Cn.CoordSampler(args)
        with torch.no_grad():
            c, max_dx, ds = R.depth2epipolarcoords(poses, depth)
            S.register(f1, f2, num_levels=L)
            corr = S(c, L, heads)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **{
            "in/K": K.numpy(), "in/depth": depth.numpy(), "in/poses": poses.numpy(), "in/f1": f1.half().numpy(),
            "in/f2": f2.half().numpy(), "in/meta": np.array([r, L, heads, 1 if gf == "minmax" else 0], dtype=np.int64),
            "in/delta": np.float32(0.7), "in/minmax": np.array([9.0, 1.0], dtype=np.float32),
            "out/coords": c.numpy(), "out/max_dx": max_dx.numpy(), "out/depths": ds.numpy(), "out/corr": corr.numpy()})
        print(tag, tuple(c.shape), tuple(corr.shape), float(corr.mean()))
        # ---- VJPs: autograd through the reference's own Reprojections / CoordSampler
        gg = torch.Generator().manual_seed(seed + 500)
        w_corr, w_ds, w_mx = torch.randn(corr.shape, generator=gg), 0.1 * torch.randn(ds.shape, generator=gg), torch.randn(max_dx.shape, generator=gg)
        dg, pg = depth.clone().requires_grad_(True), poses.clone().requires_grad_(True)
        f1g, f2g = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        R.delta.grad = None
        cg, mg, sg = R.depth2epipolarcoords(pg, dg)
        S2 = Cn.CoordSampler(args)
        S2.register(f1g, f2g, num_levels=L)
        ((S2(cg, L, heads) * w_corr).sum() + (sg * w_ds).sum() + (mg * w_mx).sum()).backward()
        np.savez_compressed(os.path.join(OUT, tag.replace("epi_", "epi_grad_") + ".npz"), **{
            "in/w_corr": w_corr.numpy(), "in/w_ds": w_ds.numpy(), "in/w_mx": w_mx.numpy(),
            "grad/depth": dg.grad.numpy(), "grad/poses": pg.grad.numpy(), "grad/delta": R.delta.grad.numpy().copy(),
            "grad/f1": f1g.grad.numpy(), "grad/f2": f2g.grad.numpy()})
        print("  grads", float(dg.grad.abs().sum()), float(pg.grad.abs().sum()), float(R.delta.grad))
        # pose refinement step (utils.py:219-236, 303-368): PoseUpdate.direct_align without --robust_pose_loss
        g2 = torch.Generator().manual_seed(seed + 100)
        f2s = (0.8 * f1 + 0.2 * f2).half().float()  # a target that resembles the source, so the step is well conditioned
        src_w, tgt_w = 0.5 + torch.rand(B, 1, h, w, generator=g2), 0.5 + torch.rand(B, 1, h, w, generator=g2)
        weight = 0.5 + torch.rand(B, 1, h, w, generator=g2)
        args.disable_fixed_pose_weight, args.robust_pose_loss, args.mixed_precision = True, False, False
        P = U.PoseUpdate(args, C, norm_fn="none")
        with torch.no_grad():
            c_p, P2 = R.depth2gradcoords(poses, depth, K)
            P.compute_feat(f1, f2s)
            P.src_w, P.tgt_w = src_w.clone(), tgt_w.clone()
            new_poses, update = P.direct_align(poses, K, c_p, P2, weight.clone())
        np.savez_compressed(os.path.join(OUT, tag.replace("epi_", "epi_align_") + ".npz"), **{
            "in/K": K.numpy(), "in/depth": depth.numpy(), "in/poses": poses.numpy(), "in/f1": f1.half().numpy(),
            "in/f2": f2s.half().numpy(), "in/src_w": src_w.numpy(), "in/tgt_w": tgt_w.numpy(), "in/weight": weight.numpy(),
            "out/c_p": c_p.numpy(), "out/P2": P2.numpy(), "out/new_poses": new_poses.numpy(), "out/update": update.numpy()})
        print("  align", tuple(c_p.shape), update.flatten()[:3].tolist())
        # ---- VJP of the refinement step, with and without --robust_pose_loss (utils.py:344-355): autograd through the
        # reference's own depth2gradcoords + PoseUpdate.direct_align
        g3 = torch.Generator().manual_seed(seed + 700)
        Wn, Wu = torch.randn(B, 4, 4, generator=g3), torch.randn(B, 6, 1, generator=g3)
        for robust in (False, True):
            args.robust_pose_loss = robust
            Pg = U.PoseUpdate(args, C, norm_fn="none")
            leaves = {k: v.clone().requires_grad_(True) for k, v in dict(
                poses=poses, depth=depth, f1=f1, f2=f2s, src_w=src_w, tgt_w=tgt_w, weight=weight).items()}
            cg, Pg2 = R.depth2gradcoords(leaves["poses"], leaves["depth"], K)
            Pg.compute_feat(leaves["f1"], leaves["f2"])
            Pg.src_w, Pg.tgt_w = leaves["src_w"], leaves["tgt_w"]
            npg, upg = Pg.direct_align(leaves["poses"], K, cg, Pg2, leaves["weight"])
            ((npg * Wn).sum() + (upg * Wu).sum()).backward()
            name = tag.replace("epi_", "epi_aligngrad_") + ("_robust" if robust else "")
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **{
                "in/Wn": Wn.numpy(), "in/Wu": Wu.numpy(), "out/new_poses": npg.detach().numpy(), "out/update": upg.detach().numpy(),
                **{"grad/" + k: v.grad.numpy() for k, v in leaves.items()}})
            print("  aligngrad robust=%d" % robust, upg.detach().flatten()[:3].tolist(),
                  {k: float(v.grad.abs().sum()) for k, v in leaves.items()})
        args.robust_pose_loss = False
        # the masking lookup (depth_pose.py:561-580): depthbins2coords (both branches) + CoordSampler.__corr__
        if B * h * w > 200:  # 96 hypotheses per pixel: keep only the small case as a fixture
            continue
        outs = {}
        with torch.no_grad():
            for name, flag in (("lin", False), ("bins", True)):
                args.use_depth_bins_for_masking = flag
                c0, ds0 = R.depthbins2coords(poses, depth)
                outs["out/c0_" + name], outs["out/ds0_" + name] = c0.numpy(), ds0.numpy()
                outs["out/corr0_" + name] = S.__corr__(c0).numpy()
        np.savez_compressed(os.path.join(OUT, tag.replace("epi_", "epi_bins_") + ".npz"), **{
            "in/K": K.numpy(), "in/depth": depth.numpy(), "in/poses": poses.numpy(), "in/f1": f1.half().numpy(),
            "in/f2": f2.half().numpy(), "in/range": np.array([0.1, 100.0, 1.0, 9.0], dtype=np.float32), **outs})
        print("  bins", tuple(outs["out/c0_lin"].shape), float(outs["out/corr0_bins"].mean()))


if __name__ == "__main__":
    main()
