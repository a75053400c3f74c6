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


@pytest.mark.parametrize("small_blocks", [1, 0], ids=["256-thread extents (default)", "1024-thread extents"])
def test_full_size_against_the_cpu_checker(small_blocks):
    from mal_amd import _lib, dyn_utils
    from oracle import dyn_oracle as D
    from oracle.gen_golden_dyn import make_masks
    assert _lib.load().mal_set_option(b"dyn_small_blocks", small_blocks) == 0  # (same bits either way: the extents' band reduction)
    try:
        _full_size_against_the_cpu_checker(dyn_utils, D, make_masks)
    finally:
        _lib.load().mal_set_option(b"dyn_small_blocks", 1)


def _full_size_against_the_cpu_checker(dyn_utils, D, make_masks):
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


def _stub_items(B, H, W, listed, seed):
    from oracle.gen_golden_dyn import make_masks
    items = []
    for b in listed:
        ml, mn = make_masks(3, H, W, seed=seed + b, edge_cases=False)
        items.append((b, ml.to(DEV), mn.to(DEV)))
    return items


@pytest.mark.parametrize("shape,listed", [((3, 24, 40), (0, 2)), ((2, 32, 64), (0, 1)), ((2, 21, 37), (1,))],
                         ids=["w40-one-sample-skipped", "w64-every-sample", "w37-scalar-kernels"])
def test_prefilled_and_in_place_equal_the_plain_node(shape, listed):
    """the whole-step API's protocol -- syn made in buffers that hold the warped images already, the cotangent buffers
    turned into the gradients in place, the region map handed out -- against the plain autograd node: bit for bit"""
    from mal_amd import dyn_utils
    B, H, W = shape
    g = torch.Generator().manual_seed(31)
    cl, cn = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    wl, wn = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    items = _stub_items(B, H, W, listed, 40)
    a_l, a_n = cl.clone().requires_grad_(True), cn.clone().requires_grad_(True)
    sl, sn, region = dyn_utils.BatchSynthesisFn.apply(a_l, a_n, items, False, None)
    gl, gn = torch.autograd.grad([sl, sn], [a_l, a_n], [wl.clone(), wn.clone()])
    # prefilled + in place
    b_l, b_n = cl.clone().requires_grad_(True), cn.clone().requires_grad_(True)
    pre = (cl.clone(), cn.clone())
    pl, pn, region2 = dyn_utils.BatchSynthesisFn.apply(b_l, b_n, items, False, pre)
    assert pl.data_ptr() == pre[0].data_ptr() and pn.data_ptr() == pre[1].data_ptr()
    assert torch.equal(pl, sl) and torch.equal(pn, sn) and torch.equal(region2, region)
    for snapshots in (False, True):
        ct = [wl.clone(), wn.clone()]
        reg = {t.data_ptr(): None for t in ct}
        if snapshots:  # a second copy of the cotangent that is valid at region pixels only (NaN elsewhere: never read)
            rg = (region & 1).bool()[:, None].expand(B, 3, H, W)
            snap = [torch.where(rg, t, torch.full_like(t, float("nan"))) for t in ct]
            reg = {t.data_ptr(): s_ for t, s_ in zip(ct, snap)}
        dyn_utils.INPLACE_COTANGENTS.update(reg)
        try:
            hl, hn = torch.autograd.grad([pl, pn], [b_l, b_n], ct, retain_graph=True)
        finally:
            for k in reg:
                dyn_utils.INPLACE_COTANGENTS.pop(k, None)
        assert hl.data_ptr() == ct[0].data_ptr() and hn.data_ptr() == ct[1].data_ptr()  # really in place
        assert torch.equal(hl, gl) and torch.equal(hn, gn), snapshots
    # the region map: bit 0 = some instance's mask (either frame) holds the pixel; zero for samples not listed
    for b in range(B):
        want = torch.zeros(H, W, dtype=torch.bool, device=DEV)
        for (bb, ml, mn) in items:
            if bb == b:
                want = (ml.bool() | mn.bool()).any(0)
        assert torch.equal((region[b] & 1).bool(), want), b
        assert torch.equal(sl[b][:, ~want], cl[b][:, ~want])  # syn differs from the warped image only there


def test_masks_overwritten_before_the_backward_are_refused():
    """the node keeps the matcher's masks by reference (no copy per step); a producer that writes into the same mask storage
    again before the backward (a segmenter with static output buffers, called for the student's pass) must not lead to
    gradients through the wrong regions: the backward sees the changed version counter and raises"""
    from mal_amd import dyn_utils
    from mal_amd._lib import MalError
    B, H, W = 2, 24, 40
    g = torch.Generator().manual_seed(5)
    cl, cn = (torch.rand(B, 3, H, W, generator=g).to(DEV).requires_grad_(True) for _ in range(2))
    items = _stub_items(B, H, W, (0, 1), 70)
    sl, sn, _ = dyn_utils.BatchSynthesisFn.apply(cl, cn, items, False, None)
    torch.autograd.grad([sl, sn], [cl, cn], [torch.ones_like(sl), torch.ones_like(sn)], retain_graph=True)  # untouched: fine
    items[1][1].zero_()  # the "next call" of the segmenter lands in the same buffer
    with pytest.raises(MalError, match="modified in place"):
        torch.autograd.grad([sl, sn], [cl, cn], [torch.ones_like(sl), torch.ones_like(sn)])


def test_step_with_the_region_map_equals_the_dense_path():
    """loss_step with mal_amd.dyn_utils.image_synthesis (sparse syn buffers -- only the region pixels are written and read,
    MAL_STEP_SYN_SPARSE --, in-place producer backward, region map:
    synthesised candidates skipped where their window cannot differ) against the same step driven by a producer that
    offers none of that (every candidate evaluated everywhere): same losses, same gradients; and with the map no
    synthesised candidate ever wins outside the dilated region (an exact tie goes to the warped one, loss_utils.py:103)"""
    from mal_amd import _lib, dyn_utils, step, trainer
    from mal_amd.synthetic import instance_stub, make_batch, to_dicts
    B, H, W = 3, 64, 128
    batch = make_batch(B, H, W, seed=11)
    opt = trainer.default_options(height=H, width=W, batch_size=B, temporal=True)
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(B, 1, H, W, generator=g).to(DEV)

    def run(dense):
        ins_model, matcher = instance_stub(B, H, W, n_inst=2, seed=5, device=DEV)

        def synth(inputs, outputs, scale):
            if not dense:
                return dyn_utils.image_synthesis(inputs, outputs, scale, 0.5, ins_model, matcher)
            plain = {k: v for k, v in outputs.items() if k[0] not in ("syn_prefilled", "syn_sparse_buffers")}
            has = dyn_utils.image_synthesis(inputs, plain, scale, 0.5, ins_model, matcher)
            for k in (("syn", -1, scale), ("syn", 1, scale)):
                if k in plain:
                    outputs[k] = plain[k]
            seen["region"] = plain.get(("syn_region", scale))
            return has

        inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=DEV)
        for f, s in ((-1, "m1"), (1, "p1")):
            mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
            mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
        losses, _, maps = step.loss_step(opt, inputs, mono_outputs, outputs, noise=noise.clone(), want_decisions=True,
                                         image_synthesis=synth)
        losses["loss"].backward()
        torch.cuda.synchronize()
        return ({k: float(v.detach()) for k, v in losses.items()}, {k: t.grad.cpu() for k, t in leaves.items()},
                maps["dec_teacher"][_lib.DEC_WIN].cpu(), maps["dec_student"].cpu())

    seen = {}
    l_dense, g_dense, win_dense, dec_s_dense = run(True)
    l_sparse, g_sparse, win_sparse, dec_s_sparse = run(False)
    for k, v in l_dense.items():
        assert abs(l_sparse[k] - v) <= 2e-6 * abs(v) + 1e-9, (k, l_sparse[k], v)
    region = (seen["region"] & 1).bool().cpu()
    near = torch.nn.functional.max_pool2d(region[:, None].float(), 3, 1, 1)[:, 0] > 0
    # Decision-exact (round 3 allowed 1e-3 of the elements to be off): away from the dilated region syn_f IS warp_f, so the
    # dense path's winner 2 / 3 there is the same image as 0 / 1 -- compared modulo 2 --; what then still differs is a
    # handful of genuine near-ties (two kernels round the synthesised candidates' SSIM differently), and EVERY element of
    # the teacher's disparity gradient that differs lies within reach of one of them: the 3x3 window of the decision, the
    # 3x3 window of the partials, and the producer's patch shift (<= 8 px horizontally in this stub) on the way back
    # through syn.
    wd, ws_ = win_dense.long(), win_sparse.long()
    same_image = ~near
    d = ((wd & 3) != (ws_ & 3)) & ~(same_image & (((wd & 3) % 2) == ((ws_ & 3) % 2)))
    d |= ((wd >> 2) & 1) != ((ws_ >> 2) & 1)  # the automask bit
    assert int(d.sum()) <= 3e-4 * d.numel() + 8, int(d.sum())
    reach = torch.nn.functional.max_pool2d(d[:, None].float(), 2 * 12 + 1, 1, 12) > 0
    r, g = g_dense["disp_teacher"], g_sparse["disp_teacher"]
    off = (g - r).abs() > 1e-4 * float(r.abs().max())
    assert not bool((off & ~reach).any()), ("teacher gradient differs away from every differing decision",
                                            torch.nonzero(off & ~reach)[:5].tolist())
    # the student sees the hint only through the teacher's min map (the distillation argmin, pointwise): its decisions
    # differ where that map moved by a rounding across a three-way near-tie, and its gradient only there
    ds = (dec_s_dense != dec_s_sparse).any(0)
    assert int(ds.sum()) <= 3e-4 * ds.numel() + 8, int(ds.sum())
    rs, gs = g_dense["disp_student"], g_sparse["disp_student"]
    off_s = (gs - rs).abs() > 1e-4 * float(rs.abs().max())
    assert not bool((off_s[:, 0] & ~ds).any()), torch.nonzero(off_s[:, 0] & ~ds)[:5].tolist()
    for k in ("axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1"):  # global sums: a few near-ties move them by
        sc = float(g_dense[k].abs().max())                                        # their share of the pixels
        assert float((g_sparse[k] - g_dense[k]).abs().max()) <= (1e-4 + 4.0 * float(d.float().mean())) * sc, k
    syn_won = (win_sparse & 3) >= 2
    assert not bool((syn_won & ~near).any())          # with the map: never outside the dilated region
    assert bool((syn_won & near).any())               # ... and the synthesised candidates do win somewhere inside


def test_selection_outside_the_mask_tensor_is_clamped():
    """the matcher's int64 selections are applied inside the kernels; torch indexing raised IndexError on a bad one, a
    kernel cannot -- the row is clamped into the tensor (mal_dyn_item.n_last / n_next), nothing outside it is addressed"""
    from mal_amd import dyn_utils
    from oracle.gen_golden_dyn import make_masks
    B, H, W = 1, 24, 40
    g = torch.Generator().manual_seed(3)
    cl, cn = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    ml, mn = make_masks(3, H, W, seed=5, edge_cases=False)
    ml, mn = ml.to(DEV), mn.to(DEV)
    idx = lambda v: torch.tensor(v, dtype=torch.int64, device=DEV)
    bad = dyn_utils.BatchSynthesisFn.apply(cl, cn, [(0, ml, mn, idx([7, -4, 1]), idx([99, 0, 1]))], False)
    ok = dyn_utils.BatchSynthesisFn.apply(cl, cn, [(0, ml, mn, idx([2, 0, 1]), idx([2, 0, 1]))], False)
    torch.cuda.synchronize()
    assert torch.equal(bad[0], ok[0]) and torch.equal(bad[1], ok[1])
