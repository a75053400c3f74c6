"""The CPU restatement of the temporal-hint producer against the reference's own outputs (tests/golden/dyn_*.npz,
made by oracle/gen_golden_dyn.py from manydepth/dyn_utils.py): bit-exact images and gradients."""
import numpy as np
import pytest
import torch

from oracle import dyn_oracle as D
from tests import golden_io as G

CASES = ["dyn_n4_24x40", "dyn_n1_19x33", "dyn_n7_48x80"]


def load(tag):
    z = G.load(tag)
    ml, mn = torch.from_numpy(z["in/mask_last"]), torch.from_numpy(z["in/mask_next"])
    il = torch.from_numpy(z["in/img_last"].astype(np.float32)) / 255
    inx = torch.from_numpy(z["in/img_next"].astype(np.float32)) / 255
    return z, ml, mn, il, inx


@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("tag", CASES)
def test_restatement_matches_the_reference(tag, replace):
    z, ml, mn, il, inx = load(tag)
    il.requires_grad_(True), inx.requires_grad_(True)
    sfx = "_replace" if replace else ""
    ol, on = D.generate_dynamic_instance(ml, mn, il, inx, replace)
    assert np.array_equal(ol.detach().numpy(), z["out/ori_last" + sfx])
    assert np.array_equal(on.detach().numpy(), z["out/ori_next" + sfx])
    ct_l, ct_n = torch.from_numpy(z["in/ct_last" + sfx]), torch.from_numpy(z["in/ct_next" + sfx])
    gl, gn = torch.autograd.grad((ol * ct_l).sum() + (on * ct_n).sum(), [il, inx])
    assert np.array_equal(gl.numpy(), z["out/g_img_last" + sfx])
    assert np.array_equal(gn.numpy(), z["out/g_img_next" + sfx])


def test_row_and_column_zero_are_invisible_to_the_extents():
    m = torch.zeros(1, 6, 7, dtype=torch.bool)
    m[0, 0, :] = True
    m[0, :, 0] = True
    assert D.extents(m).tolist() == [[5, 1, 6, 1]]  # rows/cols 1.. are present through column/row 0's pixels
    m[:] = False
    m[0, 0, 0] = True
    assert D.extents(m).tolist() == [[0, 0, 0, 0]]


def test_image_synthesis_matches_the_reference():
    """oracle/dyn_oracle.image_synthesis -- the producer the CPU baseline of bench.py and the headline parity test drive --
    against the reference's OWN image_synthesis (dyn_utils.py:121-170) on a batch that walks every branch: a sample without
    a confident instance, one whose match is empty, two that are synthesised; images and gradients bit for bit."""
    from oracle.gen_golden_dyn import synthesis_stubs
    z = G.load("dyn_synthesis_b4_24x40")
    f = lambda k: torch.from_numpy(z[k].astype(np.float32)) / 255
    tgt, cl, cn = f("in/target"), f("in/color_last").requires_grad_(True), f("in/color_next").requires_grad_(True)
    B, _, H, W = tgt.shape
    ins_model, matcher, _ = synthesis_stubs(B, H, W, int(z["in/stub_seed"]))
    outputs = {("color", -1, 0): cl, ("color", 1, 0): cn}
    assert D.image_synthesis({("color", 0, 0): tgt}, outputs, 0, 0.5, ins_model, matcher) is True
    sl, sn = outputs[("syn", -1, 0)], outputs[("syn", 1, 0)]
    assert np.array_equal(sl.detach().numpy(), z["out/syn_last"]) and np.array_equal(sn.detach().numpy(), z["out/syn_next"])
    ct_l, ct_n = torch.from_numpy(z["in/ct_last"]), torch.from_numpy(z["in/ct_next"])
    gl, gn = torch.autograd.grad((sl * ct_l).sum() + (sn * ct_n).sum(), [cl, cn])
    assert np.array_equal(gl.numpy(), z["out/g_last"]) and np.array_equal(gn.numpy(), z["out/g_next"])
    # no sample matched: nothing is written and has_ins is False (loss_utils.py:84 then ignores the hint)
    empty = lambda a, b, c: (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64))
    out2 = {("color", -1, 0): cl, ("color", 1, 0): cn}
    ins_model2, _, _ = synthesis_stubs(B, H, W, int(z["in/stub_seed"]))
    assert D.image_synthesis({("color", 0, 0): tgt}, out2, 0, 0.5, ins_model2, empty) is False and ("syn", -1, 0) not in out2
