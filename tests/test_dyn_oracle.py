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
