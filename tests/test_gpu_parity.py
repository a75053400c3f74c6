"""GPU parity: the HIP loss path (through the C ABI) against the CPU oracle and the golden
vectors of the reference, on the same seeded inputs.

Tolerances (north_star: 1e-4 rel, fp32).  The loss has genuine discontinuities -- the
per-pixel argmin over candidates, the automask comparison, the 3-way distillation argmin,
the bilinear tap switch when a sampling position crosses an integer, and the border clip --
so two correct fp32 evaluations (e.g. ATen on two CPUs: the oracle re-run on the GPU box's
host already differs from the golden file in the 7th digit) can take different branches at
pixels that sit within rounding distance of a tie.  The tests therefore
  * compare scalars at 1e-4 rel (the distillation mean gets the explicit allowance of its
    near-tie pixels);
  * compare per-pixel maps/gradients at 1e-4 of the map's scale OUTSIDE the pixels the
    oracle itself marks as near-tie (listed by kind, their fraction bounded);
  * bound summed gradients (poses) by 1e-4 or by the distance between the fp32 oracle and
    the same oracle evaluated in fp64, whichever is larger: not further from the reference
    than the reference is from exact arithmetic.
"""
import numpy as np
import pytest
import torch

from tests import golden_io as G
from tests import hip_harness as HH

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)
    assert torch.cuda.is_available(), "GPU tests need a HIP device"


def _l2rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def _check_case(tag, fuse, full):
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    kw = G.opt_kwargs(z)
    o = HH.run_oracle(b, kw, n0, n1)
    h = HH.run_hip(b, kw, n0, n1, fuse=fuse)
    N = B * H * W

    # ---- exact / elementwise things
    G.assert_close(h["mono_depth"], o["mono_depth"], 1e-6, "mono depth")
    G.assert_close(h["multi_depth"], o["multi_depth"], 1e-6, "multi depth")
    assert (h["consistency_mask"] != o["consistency_mask"]).mean() <= 1e-5, "matching mask"
    if "cons_target" in h:
        G.assert_close(h["cons_target"], o["cons_target"], 1e-6, "consistency_target")

    # ---- per-pixel minimum reprojection maps (SSIM's sigma = E[x^2]-mu^2 cancellation against
    # C2 = 9e-4 turns a 1-ulp difference of a warped pixel into ~3e-5 relative)
    for name in ("mono_reproj", "ens"):
        if o[name] is None:
            continue
        d = np.abs(h[name] - o[name])
        assert d.max() <= 1e-4, (name, d.max())
        assert (d > 5e-5 + 1e-4 * np.abs(o[name])).mean() <= 1e-4, name

    # ---- scalars
    temporal = "syn_rects" in b
    amb_distil = HH.near_tie(np.concatenate([m for m in (o["mono_reproj"], o["ens"], o["multi_cands"].min(1, keepdims=True))
                                             if m is not None], 1), 2e-4)
    allow_distil = float((np.abs(o["mono_depth"] - o["multi_depth"]) * amb_distil).sum() / N)
    allow_auto, renorm, any_auto = HH.automask_tie_allowance(o, n0)  # automask pixels at rounding distance of the threshold
    for k, v in o["losses"].items():
        tol = 1e-4 * abs(v) + (allow_distil if ("distil" in k or k.startswith("loss")) else 0.0)
        tol += 0.0 if ("distil" in k or "consistency" in k) else allow_auto
        assert abs(h["losses"][k] - v) <= tol, (k, h["losses"][k], v, tol)
        if not temporal:  # golden = the reference's own run in the authoring container
            gv = float(z["losses/" + k])
            assert abs(h["losses"][k] - gv) <= 2e-4 * abs(gv) + allow_distil + allow_auto, ("golden", k, h["losses"][k], gv)
    assert abs(h["final"] - o["final"]) <= 1e-4 * abs(o["final"]) + B * (allow_distil + allow_auto)

    # ---- per-pixel disparity gradients outside the near-tie pixels
    idn = o["ident"] + n0.numpy() * np.float32(1e-5)
    amb_t = HH.dilate3(HH.near_tie(o["mono_cands"], 2e-4, distinct=temporal) | (np.abs(o["mono_reproj"] - idn) <= 1e-4))
    amb_t |= HH.sample_ambiguous(o["mono_sample"], H, W)
    amb_s = HH.dilate3(HH.near_tie(o["multi_cands"], 2e-4, distinct=temporal)) | HH.sample_ambiguous(o["multi_sample"], H, W) | amb_distil
    amb_s |= np.abs(o["mono_depth"] - o["multi_depth"]) <= 1e-6 * np.abs(o["mono_depth"])
    if kw.get("dual_distil"):  # the teacher's depth then also receives the distillation gradient
        amb_t |= amb_distil
    if temporal:  # the synthesised candidates add their own ties (exactly equal candidates count as one)
        amb_t |= HH.dilate3(HH.near_tie(o["mono_cands"], 2e-4, distinct=True))
    for key, amb in (("disp_teacher", amb_t), ("disp_student", amb_s)):
        g, r = h["grads"][key], o["grads"][key]
        sc = np.abs(r).max()
        good = ~amb
        if good.any():
            err = np.abs(g - r)[good]
            tol = 2e-4 + (renorm if key == "disp_teacher" else 0.0)
            assert (err > tol * sc).mean() <= 2e-5, (key, err.max() / sc, (err > tol * sc).mean())
        assert amb.mean() <= 0.05, (key, "near-tie fraction", amb.mean())

    # ---- summed gradients: never further from the fp32 reference than it is from fp64
    # (per-pixel leaves: over the pixels outside the near-tie set, where one flipped branch cannot
    # dominate the norm of a small map)
    o64 = HH.oracle_fp64_grads(b, kw, n0, n1)
    for key in HH.LEAVES:
        g, r, r64 = h["grads"][key], o["grads"][key], o64[key]
        if key in ("disp_teacher", "disp_student"):
            keep = ~(amb_t if key == "disp_teacher" else amb_s)
            if not keep.any():
                continue
            g, r, r64 = g[keep], r[keep], r64[keep]
        floor = _l2rel(r, r64)
        if any_auto and g.ndim != 4 and key != "disp_student":
            # an automask pixel sits within rounding of its threshold: a pose gradient -- a sum over all pixels of cancelling
            # terms -- moves by per cent with the side that ONE pixel takes, in the oracle as much as in the kernels, so the
            # free-running comparison says nothing here (rounds 1-4 allowed 2e-2).  The same route on the same fixture is held
            # at max(1e-4, 1.25 x floor) with the decisions forced: tests/test_gpu_decisions.py (the one-call step) and
            # tests/test_gpu_step.py::test_loss_step_equals_operator_route (this route == the step at 2e-5).
            continue
        extra = 0.0 if key == "disp_student" else renorm
        assert _l2rel(g, r) <= max(1e-4, 1.5 * floor) + extra, (key, _l2rel(g, r), floor)


@pytest.mark.parametrize("fuse", [True, False], ids=["fused", "explicit"])
@pytest.mark.parametrize("tag", G.STEP_CASES)
def test_step_parity(tag, fuse):
    _check_case(tag, fuse, True)


def test_learn_ens_gradient_reaches_the_ensemble_head():
    """--learn_ens (loss_utils.py:240-241, trainer.py:596-600): the ensemble pass warps with outputs["ens_disp"], its depth is
    the distillation target where the ensemble wins the three-way min, and THAT is where ens_disp receives gradient; held
    against the reference's own numbers (the golden file) away from the near-tie pixels of the three-way argmin"""
    z = G.load("step_b2_32x64_learnens")
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    kw = G.opt_kwargs(z)
    assert kw.get("learn_ens") is True and "disp_ens" in b
    o = HH.run_oracle(b, kw, n0, n1)
    assert np.array_equal(o["grads"]["disp_ens"], z["grad/disp_ens"])  # the oracle IS the reference here (bit for bit)
    for fuse in (True, False):
        h = HH.run_hip(b, kw, n0, n1, fuse=fuse)
        amb = HH.near_tie(np.concatenate([o["mono_reproj"], o["ens"], o["multi_cands"].min(1, keepdims=True)], 1), 2e-4)
        g, r = h["grads"]["disp_ens"], z["grad/disp_ens"]
        assert np.abs(r).max() > 0 and (r != 0).mean() > 0.02          # the ensemble does win somewhere
        assert ((g != 0) != (r != 0))[~amb].mean() == 0.0                # ... and the same pixels carry gradient
        assert np.abs(g - r)[~amb].max() <= 1e-4 * np.abs(r).max(), np.abs(g - r)[~amb].max() / np.abs(r).max()
        assert amb.mean() <= 0.05
        gv = float(z["losses/distil_loss"])
        allow = float((np.abs(o["mono_depth"] - o["multi_depth"]) * amb).sum() / (B * H * W)) * 2
        assert abs(h["losses"]["distil_loss"] - gv) <= 1e-4 * abs(gv) + allow


def test_step_parity_full_size():
    """B=2 192x640 (the size BASELINE.json's metric is quoted on, per sample)."""
    _check_case(G.BIG_CASE, True, False)


def test_fused_and_explicit_routes_agree():
    """The fused kernel and the materialising kernels share their device arithmetic up to the
    association of the 3x3 window sums: the two routes agree far inside the 1e-4 budget."""
    z = G.load("step_b3_37x50_distil")
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    a = HH.run_hip(b, {}, n0, n1, fuse=True)
    c = HH.run_hip(b, {}, n0, n1, fuse=False)
    for k in a["losses"]:
        assert abs(a["losses"][k] - c["losses"][k]) <= 1e-4 * abs(c["losses"][k]), k
    for k in HH.LEAVES:
        ga, gc = a["grads"][k], c["grads"][k]
        if ga.ndim == 4:  # per-pixel: a near-tie pixel may take the other branch in one of the routes
            assert (np.abs(ga - gc) > 1e-4 * np.abs(gc).max()).mean() <= 1e-3, k
        else:
            assert _l2rel(ga, gc) <= 1e-3, k


def test_determinism():
    """Two runs on the same inputs are bitwise identical (no atomics anywhere)."""
    z = G.load("step_b2_32x64_distil")
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    a = HH.run_hip(b, {}, n0, n1, fuse=True)
    c = HH.run_hip(b, {}, n0, n1, fuse=True)
    assert a["final"] == c["final"]
    for k in HH.LEAVES:
        assert np.array_equal(a["grads"][k], c["grads"][k]), k


def test_cpu_tensor_raises():
    from mal_amd import layers, _lib
    with pytest.raises(_lib.MalError):
        layers.disp_to_depth(torch.rand(1, 1, 4, 4), 0.1, 100.0)
