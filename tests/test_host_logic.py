"""CPU: host-side logic of the package (no kernels): pose composition against the golden
vectors, loss balancing against the oracle, mask rule, synthetic generator, option defaults,
and that the product package never imports the oracle."""
import os
import re
import pytest

import numpy as np
import torch

from tests import golden_io as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pose_composition_matches_reference():
    from mal_amd import layers
    z = G.load("layers_b2_24x40")
    b = G.batch_from_golden(z)
    for inv in (False, True):
        T = layers.transformation_from_parameters(b["axisangle_m1"], b["translation_m1"], invert=inv)
        G.assert_close(T, z["T_inv%d" % inv], 1e-6, "T", floor=1e-3)
    G.assert_close(layers.rot_from_axisangle(b["axisangle_p1"]), z["rot"], 1e-6, floor=1e-3)
    G.assert_close(layers.get_translation_matrix(b["translation_p1"]), z["trans"], 0)


def test_pose_composition_gradient_matches_oracle():
    from mal_amd import layers
    from oracle import mal_oracle as O
    torch.manual_seed(0)
    aa, tr = 0.01 * torch.randn(3, 1, 3), 0.05 * torch.randn(3, 1, 3)
    w = torch.randn(3, 4, 4)
    gs = []
    for fn in (layers.transformation_from_parameters, O.transformation_from_parameters):
        a, t = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        (fn(a, t, True) * w).sum().backward()
        gs.append((a.grad, t.grad))
    assert torch.allclose(gs[0][0], gs[1][0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(gs[0][1], gs[1][1], rtol=1e-5, atol=1e-6)


def test_loss_masks_rule():
    from mal_amd import loss_utils
    from oracle import mal_oracle as O
    torch.manual_seed(1)
    r, i = torch.rand(2, 1, 5, 7), torch.rand(2, 1, 5, 7)
    i[0, 0, 0, 0] = r[0, 0, 0, 0]  # tie -> index 0 wins -> mask 1
    assert torch.equal(loss_utils.compute_loss_masks(r, i), O.compute_loss_masks(r, i))
    assert torch.equal(loss_utils.compute_loss_masks(r, None), torch.ones_like(r))


def test_loss_balancing_matches_oracle():
    from mal_amd import loss_utils
    from oracle import mal_oracle as O
    a, b = loss_utils.LossBalancing(2, 10, 4), O.LossBalancing(2, 10, 4)
    rng = np.random.RandomState(0)
    for it in range(4):  # the last iteration runs off the end of the dataset (10 records, bs 4)
        ll = [torch.tensor(float(rng.rand() + 0.5)), torch.tensor(float(rng.rand() + 0.1))]
        la, lb = a.compute_loss(ll, it), b.compute_loss(ll, it)
        assert abs(float(la) - float(lb)) < 1e-6
        if it < 3:
            wa, wb = a.update_weight(it, 3.0), b.update_weight(it, 3.0)
            assert np.allclose(wa, wb)
    assert np.allclose(a.train_scores, b.train_scores)


def test_loss_balancing_against_the_reference_fixture():
    """the PRODUCT's LossBalancing (mal_amd/loss_utils.py) against the reference's own (tests/golden/loss_balancing.npz, written by
    oracle/gen_golden.py blc from manydepth/loss_utils.py:283-345): weights and running state bit for bit over eight steps"""
    import os
    from mal_amd import loss_utils
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_balancing.npz"))
    lb = loss_utils.LossBalancing(int(g["num_loss"]), int(g["num_data"]), int(g["bs"]))
    for it, (sc, lam) in enumerate(zip(g["scores"], g["lambdas"])):
        lb.compute_loss([torch.tensor(float(sc[0]), dtype=torch.float64), torch.tensor(float(sc[1]), dtype=torch.float64)], it)
        w = lb.update_weight(it, float(lam))
        assert np.array_equal(np.array(w, dtype=np.float64), g["weights"][it]), (it, w, g["weights"][it])
        assert float(lb.previous_total_loss) == float(g["previous_total_loss"][it])
        assert np.array_equal(np.asarray(lb.previous_loss, dtype=np.float64), g["previous_loss"][it])
    assert np.array_equal(lb.train_scores, g["train_scores"])


def test_synthetic_batch_contract_and_determinism():
    from mal_amd.synthetic import make_batch, kitti_intrinsics
    a, b = make_batch(2, 16, 24, seed=5), make_batch(2, 16, 24, seed=5)
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
    assert a["color0"].shape == (2, 3, 16, 24) and a["color0"].min() >= 0 and a["color0"].max() <= 1
    assert a["disp_teacher"].shape == (2, 1, 16, 24) and 0 < a["disp_teacher"].min() and a["disp_teacher"].max() < 1
    K, iK = kitti_intrinsics(1, 192, 640)
    assert abs(K[0, 0, 0] - 0.58 * 640) < 1e-3 and abs(K[0, 1, 1] - 1.92 * 192) < 1e-3
    assert torch.allclose(K[0] @ iK[0], torch.eye(4), atol=1e-5)


def test_option_defaults_follow_reference():
    from mal_amd import trainer
    o = trainer.default_options()
    assert (o.height, o.width, o.batch_size, o.min_depth, o.max_depth) == (192, 640, 12, 0.1, 100.0)
    assert o.frame_ids == [0, -1, 1] and o.sclm == 0 and o.disparity_smoothness == 1e-3


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under mal_amd/ may import or execute it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mal_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), os.path.join(dirpath, f)


def test_workspace_ownership_tokens():
    """ops.claim_workspace / check_workspace: the latest forward owns a cached step workspace; an older token is refused"""
    import pytest
    from mal_amd import _lib, ops

    class Ws:  # stands in for the uint8 device tensor: only data_ptr() is read
        def data_ptr(self):
            return 0x1234

    ws = Ws()
    first = ops.claim_workspace(ws)
    ops.check_workspace(ws, first, "first")
    second = ops.claim_workspace(ws)
    ops.check_workspace(ws, second, "second")
    with pytest.raises(_lib.MalError):
        ops.check_workspace(ws, first, "first, after the second forward")


def test_dualrefine_one_call_step_refuses_what_it_does_not_cover():
    """DualRefineLossPath.loss_step covers scales out of [0,1,2,3] and n_losses < MAL_DR_MAX_ITERS; everything else is
    refused by name before any device work (the operator-level methods remain the route for it)"""
    import pytest
    from mal_amd import _lib, dualrefine
    for kw in (dict(scales=[0, 4]), dict(scales=[0, 0]), dict(scales=[]), dict(n_losses=_lib.DR_MAX_ITERS),
               dict(frame_ids=[0, -1]), dict(v1_multiscale=True)):
        lp = dualrefine.DualRefineLossPath(dualrefine.default_options(**kw))
        with pytest.raises(_lib.MalError):
            lp.loss_step({("color", 0, 0): torch.zeros(1, 3, 8, 8)}, {})


def _can_price():
    import shutil
    from mal_amd import build
    return shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc") or os.path.exists(build.VALU_JSON)


@pytest.mark.skipif(not _can_price(), reason="needs hipcc (or a built mal_amd/lib/valu_cost.json) to price the row loops")
def test_valu_price_list_finds_the_row_loops():
    """build() treats the vector-ALU price list as best effort (it parses a compiler listing by mangled names); the hard
    assertion lives here: the shipped sources yield a row loop and a gradient-only loop for the north-star kernel, and the
    hand-written DPP blocks kept for A/B (-DMAL_HSUM_DPP) open with the five wait states their hazards need (mal_pairs.h)."""
    from mal_amd import build
    rep = build.valu_report()
    for name in ("teacher", "teacher_temporal", "student", "ensemble"):
        assert name in rep["kernels"], (name, sorted(rep["kernels"]))
        assert rep["kernels"][name]["pipe_cycles"] > 0 and rep["kernels"][name]["valu_instructions"] > 100
    assert rep["kernels"]["teacher"]["drain"]["valu_instructions"] > 0
    # round 4: the 18 partial planes' horizontal sums go through LDS (measured faster); the 24 statistic planes keep the 48
    # written-out DPP adds (through LDS they measured slower: profiles/r04_hsum_variants_ab.txt) -- not the compiler's peephole
    # (a range, not an equality: a compiler update may fold or split a few)
    for name in ("teacher", "student"):
        assert 40 <= rep["kernels"][name]["classes"]["dpp"]["instr"] <= 64, rep["kernels"][name]["classes"]
    src = open(os.path.join(os.path.dirname(build.CSRC), "csrc", "mal_pairs.h")).read()
    assert src.count('asm("s_nop 4') == 2 and 'asm("s_nop 1' not in src
