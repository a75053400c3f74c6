"""GPU parity of the temporal-hint producer kernels (mal_dyn_instance_fwd/_bwd through mal_amd.dyn_utils)
against the reference's own outputs (tests/golden/dyn_*.npz) and, at full size, against the CPU checker."""
import numpy as np
import pytest
import torch

from tests import golden_io as G
from tests.test_dyn_oracle import CASES, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("tag", CASES)
def test_golden_bit_exact(tag, replace):
    from mal_amd import dyn_utils
    z, ml, mn, il, inx = load(tag)
    il, inx = il.to(DEV).requires_grad_(True), inx.to(DEV).requires_grad_(True)
    sfx = "_replace" if replace else ""
    ol, on = dyn_utils.generate_dynamic_instance(None, None, ml.to(DEV), mn.to(DEV), il, inx, replace)
    assert np.array_equal(ol.detach().cpu().numpy(), z["out/ori_last" + sfx])
    assert np.array_equal(on.detach().cpu().numpy(), z["out/ori_next" + sfx])
    ct_l, ct_n = torch.from_numpy(z["in/ct_last" + sfx]).to(DEV), torch.from_numpy(z["in/ct_next" + sfx]).to(DEV)
    gl, gn = torch.autograd.grad((ol * ct_l).sum() + (on * ct_n).sum(), [il, inx])
    # the cotangents are multiples of 1/64: every sum is exact, whatever the order
    assert np.array_equal(gl.cpu().numpy(), z["out/g_img_last" + sfx])
    assert np.array_equal(gn.cpu().numpy(), z["out/g_img_next" + sfx])


def test_full_size_against_the_cpu_checker():
    from mal_amd import dyn_utils
    from oracle import dyn_oracle as D
    from oracle.gen_golden_dyn import make_masks
    num, H, W = 12, 192, 640
    ml, mn = make_masks(num, H, W, seed=9)
    g = torch.Generator().manual_seed(4)
    il = (torch.round(torch.rand(3, H, W, generator=g) * 255) / 255).requires_grad_(True)
    inx = (torch.round(torch.rand(3, H, W, generator=g) * 255) / 255).requires_grad_(True)
    ct_l = torch.round(torch.randn(3, H, W, generator=g) * 64) / 64
    ct_n = torch.round(torch.randn(3, H, W, generator=g) * 64) / 64
    ol, on = D.generate_dynamic_instance(ml, mn, il, inx, False)
    gl, gn = torch.autograd.grad((ol * ct_l).sum() + (on * ct_n).sum(), [il, inx])
    dl, dn = il.detach().to(DEV).requires_grad_(True), inx.detach().to(DEV).requires_grad_(True)
    hl, hn = dyn_utils.generate_dynamic_instance(None, None, ml.to(DEV), mn.to(DEV), dl, dn, False)
    hgl, hgn = torch.autograd.grad((hl * ct_l.to(DEV)).sum() + (hn * ct_n.to(DEV)).sum(), [dl, dn])
    assert torch.equal(hl.cpu(), ol.detach()) and torch.equal(hn.cpu(), on.detach())
    assert torch.equal(hgl.cpu(), gl) and torch.equal(hgn.cpu(), gn)


class _Instances:
    """the slice of detectron2's Instances that image_synthesis touches"""

    def __init__(self, scores, masks):
        self.scores, self.pred_masks = scores, masks

    def __len__(self):
        return len(self.scores)

    def __getitem__(self, sel):
        return _Instances(self.scores[sel], self.pred_masks[sel])


@pytest.mark.parametrize("as_tensor", [False, True], ids=["list_selection", "index_tensor_selection"])
def test_image_synthesis_control_flow(as_tensor):
    """dyn_utils.py:121-170: samples without confident or matched instances keep their warped images.  The matcher's
    selection arrives as Python lists (applied by torch indexing, as upstream) or as int64 index tensors (applied inside
    the kernels: no gather launch per sample and frame)."""
    from mal_amd import dyn_utils
    from oracle import dyn_oracle as D
    from oracle.gen_golden_dyn import make_masks
    B, H, W = 3, 24, 40
    g = torch.Generator().manual_seed(8)
    color = {f: torch.rand(B, 3, H, W, generator=g).to(DEV) for f in (-1, 0, 1)}
    masks = {b: make_masks(3, H, W, seed=20 + b, edge_cases=False) for b in range(B)}
    scores = {0: torch.tensor([0.9, 0.8, 0.7]), 1: torch.tensor([0.1, 0.2, 0.1]), 2: torch.tensor([0.9, 0.9, 0.9])}
    calls = {"n": 0}

    def ins_model(images):
        calls["n"] += 1
        if images.shape[0] == B:  # the target frames: only the scores matter
            return [{"instances": _Instances(scores[b], torch.zeros(3, H, W, dtype=torch.bool))} for b in range(B)]
        b = calls["b"]            # (warped last, warped next) of sample b
        return [{"instances": _Instances(scores[b], masks[b][0].to(DEV))},
                {"instances": _Instances(scores[b], masks[b][1].to(DEV))}]

    def matcher(ins_last, ins_next, cur):
        b = calls["b"]
        if b != 0:
            return [], []  # sample 2 has confident instances but no match
        return (torch.tensor([2, 0], device=DEV), torch.tensor([2, 0], device=DEV)) if as_tensor else ([2, 0], [2, 0])

    # image_synthesis calls generate_instances per sample in order; track which sample is being processed
    order = iter([0, 2])
    orig = dyn_utils.generate_instances

    def tracking(images, model):
        if images.shape[0] == 2:
            calls["b"] = next(order)
        return orig(images, model)

    dyn_utils.generate_instances = tracking
    leaves = {f: color[f].clone().requires_grad_(True) for f in (-1, 1)}
    try:
        inputs = {("color", 0, 0): color[0]}
        outputs = {("color", -1, 0): leaves[-1], ("color", 1, 0): leaves[1]}
        has = dyn_utils.image_synthesis(inputs, outputs, 0, 0.5, ins_model, matcher)
    finally:
        dyn_utils.generate_instances = orig
    assert has is True
    sl, sn = outputs[("syn", -1, 0)].detach().cpu(), outputs[("syn", 1, 0)].detach().cpu()
    for b in (1, 2):
        assert torch.equal(sl[b], color[-1][b].cpu()) and torch.equal(sn[b], color[1][b].cpu())
    ml, mn = masks[0][0][[2, 0]], masks[0][1][[2, 0]]
    rl, rn = color[-1][0].cpu().clone().requires_grad_(True), color[1][0].cpu().clone().requires_grad_(True)
    ol, on = D.generate_dynamic_instance(ml, mn, rl, rn, False)
    assert torch.equal(sl[0], ol.detach()) and torch.equal(sn[0], on.detach())
    # the whole batch is one autograd node: samples without instances pass their cotangent through, sample 0 gets
    # the adjoint of its synthesis (bit-exact against the oracle's)
    wl, wn = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    ((outputs[("syn", -1, 0)] * wl.to(DEV)).sum() + (outputs[("syn", 1, 0)] * wn.to(DEV)).sum()).backward()
    ((ol * wl[0]).sum() + (on * wn[0]).sum()).backward()
    gl, gn = leaves[-1].grad.cpu(), leaves[1].grad.cpu()
    assert torch.equal(gl[0], rl.grad) and torch.equal(gn[0], rn.grad)
    for b in (1, 2):
        assert torch.equal(gl[b], wl[b]) and torch.equal(gn[b], wn[b])
