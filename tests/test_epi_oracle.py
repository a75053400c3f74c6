"""The N4 oracle (oracle/epi_oracle.py: DualRefine's epipolar correlation lookup, forward) reproduces the golden vectors
made by the reference's own Reprojections / CoordSampler (oracle/gen_golden_epi.py) bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import epi_oracle as E

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["epi_b2_c16_12x20_r4_l3", "epi_b1_c8_9x13_r2_l2_h2"]


def load(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    t = lambda k: torch.from_numpy(z[k].astype(np.float32))
    r, L, heads, _ = (int(v) for v in z["in/meta"])
    return z, t("in/K"), t("in/depth"), t("in/poses"), t("in/f1"), t("in/f2"), r, L, heads, torch.tensor([float(z["in/delta"])])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_reproduces_reference(tag):
    z, K, depth, poses, f1, f2, r, L, heads, delta = load(tag)
    c, max_dx, ds = E.depth2epipolarcoords(poses, depth, K, delta, r=r, num_levels=L)
    corr = E.coord_sample(f1, E.pyramid(f2, L), c, L, heads)
    for got, key in ((c, "out/coords"), (max_dx, "out/max_dx"), (ds, "out/depths"), (corr, "out/corr")):
        assert np.array_equal(got.numpy(), z[key]), key


ALIGN_CASES = ["epi_align_b2_c16_12x20_r4_l3", "epi_align_b1_c8_9x13_r2_l2_h2"]


def load_align(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    t = lambda k: torch.from_numpy(z[k].astype(np.float32))
    return z, {k[3:]: t(k) for k in z.files if k.startswith("in/")}


@pytest.mark.parametrize("tag", ALIGN_CASES)
def test_direct_align_oracle_reproduces_reference(tag):
    """PoseUpdate.direct_align (utils.py:303-368) through depth2gradcoords (:219-236): bit for bit"""
    z, i = load_align(tag)
    c_p, P2 = E.depth2gradcoords(i["poses"], i["depth"], i["K"])
    new_poses, update = E.direct_align(i["poses"], i["f1"], i["f2"], i["src_w"], i["tgt_w"], i["K"], c_p, P2, i["weight"])
    for got, key in ((c_p, "out/c_p"), (P2, "out/P2"), (update, "out/update"), (new_poses, "out/new_poses")):
        assert np.array_equal(got.numpy(), z[key]), key


BINS_CASES = ["epi_bins_b1_c8_9x13_r2_l2_h2"]


def load_bins(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    t = lambda k: torch.from_numpy(z[k].astype(np.float32))
    return z, t("in/K"), t("in/depth"), t("in/poses"), t("in/f1"), t("in/f2"), [float(v) for v in z["in/range"]]


@pytest.mark.parametrize("tag", BINS_CASES)
def test_depthbins_lookup_oracle_reproduces_reference(tag):
    """depthbins2coords (utils.py:231-255, both branches) + CoordSampler.__corr__ (corr.py:52-75): bit for bit"""
    z, K, depth, poses, f1, f2, (dmin, dmax, bmin, bmax) = load_bins(tag)
    for name, rng in (("lin", None), ("bins", (bmin, bmax))):
        c0, ds0 = E.depthbins2coords(poses, depth, K, dmin, dmax, 96, rng)
        corr0 = E.corr_all_channels(f1, E.pyramid(f2, 2), c0, 1)
        for got, key in ((c0, "out/c0_"), (ds0, "out/ds0_"), (corr0, "out/corr0_")):
            assert np.array_equal(got.numpy(), z[key + name]), key + name


GRAD_CASES = ["epi_grad_b2_c16_12x20_r4_l3", "epi_grad_b1_c8_9x13_r2_l2_h2"]


def oracle_lookup_grads(K, depth, poses, f1, f2, r, L, heads, delta, w_corr, w_ds, w_mx):
    """autograd through the restated lookup with the fixture's cotangents -> grads w.r.t. depth, poses, delta, f1, f2"""
    dg, pg = depth.clone().requires_grad_(True), poses.clone().requires_grad_(True)
    f1g, f2g = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    dl = delta.clone().requires_grad_(True)
    c, max_dx, ds = E.depth2epipolarcoords(pg, dg, K, dl, r=r, num_levels=L)
    corr = E.coord_sample(f1g, E.pyramid(f2g, L), c, L, heads)
    ((corr * w_corr).sum() + (ds * w_ds).sum() + (max_dx * w_mx).sum()).backward()
    return {"depth": dg.grad, "poses": pg.grad, "delta": dl.grad, "f1": f1g.grad, "f2": f2g.grad}


@pytest.mark.parametrize("tag", GRAD_CASES)
def test_lookup_vjp_oracle_reproduces_reference(tag):
    """gradients of the lookup taken by autograd through the reference's own Reprojections / CoordSampler
    (oracle/gen_golden_epi.py) = autograd through the restatement, bit for bit"""
    z, K, depth, poses, f1, f2, r, L, heads, delta = load(tag.replace("epi_grad_", "epi_"))
    zg = np.load(os.path.join(GOLDEN, tag + ".npz"))
    t = lambda k: torch.from_numpy(zg[k])
    g = oracle_lookup_grads(K, depth, poses, f1, f2, r, L, heads, delta, t("in/w_corr"), t("in/w_ds"), t("in/w_mx"))
    for k, v in g.items():
        assert np.array_equal(v.numpy().reshape(zg["grad/" + k].shape), zg["grad/" + k]), k


ALIGNGRAD_CASES = [t + r for t in ("epi_aligngrad_b2_c16_12x20_r4_l3", "epi_aligngrad_b1_c8_9x13_r2_l2_h2") for r in ("", "_robust")]
ALIGN_LEAVES = ("poses", "depth", "f1", "f2", "src_w", "tgt_w", "weight")


def load_aligngrad(tag):
    """inputs of the align fixture of the same case + the cotangents / outputs / gradients of the VJP fixture"""
    robust = tag.endswith("_robust")
    base = tag[:-len("_robust")] if robust else tag
    z, i = load_align(base.replace("epi_aligngrad_", "epi_align_"))
    g = np.load(os.path.join(GOLDEN, tag + ".npz"))
    return i, {k: torch.from_numpy(g[k]) for k in g.files}, robust


def oracle_align_grads(i, Wn, Wu, robust):
    lv = {k: i[k].clone().requires_grad_(True) for k in ALIGN_LEAVES}
    c_p, P2 = E.depth2gradcoords(lv["poses"], lv["depth"], i["K"])
    new_poses, update = E.direct_align(lv["poses"], lv["f1"], lv["f2"], lv["src_w"], lv["tgt_w"], i["K"], c_p, P2, lv["weight"],
                                       robust=robust)
    ((new_poses * Wn).sum() + (update * Wu).sum()).backward()
    return new_poses.detach(), update.detach(), {k: v.grad for k, v in lv.items()}


@pytest.mark.parametrize("tag", ALIGNGRAD_CASES)
def test_direct_align_vjp_oracle_reproduces_reference(tag):
    """the refinement step's outputs (incl. --robust_pose_loss, utils.py:344-355) and its gradients w.r.t. every input, as
    autograd takes them through the reference's own depth2gradcoords + PoseUpdate.direct_align"""
    i, g, robust = load_aligngrad(tag)
    new_poses, update, grads = oracle_align_grads(i, g["in/Wn"], g["in/Wu"], robust)
    assert np.array_equal(new_poses.numpy(), g["out/new_poses"].numpy()) and np.array_equal(update.numpy(), g["out/update"].numpy())
    for k in ALIGN_LEAVES:
        r = g["grad/" + k]
        assert float((grads[k] - r).abs().max()) <= 1e-5 * float(r.abs().max()) + 1e-9, k
