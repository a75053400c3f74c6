"""Golden vectors for the temporal-hint producer (manydepth/dyn_utils.py:6-119) from the REFERENCE's own
TorchScript functions.  TEST INFRASTRUCTURE ONLY; run in the authoring container only:

    python -m oracle.gen_golden_dyn

Imports ``manydepth.dyn_utils`` from /root/reference (it needs nothing but torch, numpy and PIL) and calls
``generate_dynamic_instance`` on synthetic instance masks (blobs that move between the two frames, some
touching row/column 0 and the image border, some empty, overlapping shifted copies) with random images;
stores inputs, outputs and the gradients of a random cotangent w.r.t. both images.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"


def make_masks(num, H, W, seed, edge_cases=True):
    g = torch.Generator().manual_seed(seed)
    last = torch.zeros(num, H, W, dtype=torch.bool)
    nxt = torch.zeros(num, H, W, dtype=torch.bool)
    for i in range(num):
        h = int(torch.randint(3, max(4, H // 2), (1,), generator=g))
        w = int(torch.randint(3, max(4, W // 2), (1,), generator=g))
        y0 = int(torch.randint(0, H - h + 1, (1,), generator=g))
        x0 = int(torch.randint(0, W - w + 1, (1,), generator=g))
        dy = int(torch.randint(-7, 8, (1,), generator=g))
        dx = int(torch.randint(-9, 10, (1,), generator=g))
        blob = torch.rand(h, w, generator=g) > 0.25
        last[i, y0:y0 + h, x0:x0 + w] = blob
        y1, x1 = min(max(y0 + dy, 0), H - h), min(max(x0 + dx, 0), W - w)
        nxt[i, y1:y1 + h, x1:x1 + w] = torch.rand(h, w, generator=g) > 0.25
    if edge_cases and num >= 3:
        nxt[num - 1] = False                  # an instance that vanished
        last[num - 2, 0, :] = True            # row 0 is invisible to the extent computation (mask * index)
        last[num - 2, :, 0] = True
    return last, nxt


def main():
    sys.path.insert(0, REF)
    import manydepth.dyn_utils as DU
    os.makedirs(OUT, exist_ok=True)
    for tag, (num, H, W, seed) in {"dyn_n4_24x40": (4, 24, 40, 1), "dyn_n1_19x33": (1, 19, 33, 2),
                                   "dyn_n7_48x80": (7, 48, 80, 3)}.items():
        last, nxt = make_masks(num, H, W, seed)
        g = torch.Generator().manual_seed(100 + seed)
        img_last = (torch.round(torch.rand(3, H, W, generator=g) * 255) / 255).requires_grad_(True)
        img_next = (torch.round(torch.rand(3, H, W, generator=g) * 255) / 255).requires_grad_(True)
        x, y = torch.arange(H), torch.arange(W)
        grid_h, grid_w = torch.meshgrid(x, y, indexing="ij")
        d = {}
        for replace in (False, True):
            ol, on = DU.generate_dynamic_instance(grid_h, grid_w, last, nxt, img_last, img_next, replace)
            ct_l = torch.round(torch.randn(3, H, W, generator=g) * 64) / 64
            ct_n = torch.round(torch.randn(3, H, W, generator=g) * 64) / 64
            gl, gn = torch.autograd.grad((ol * ct_l).sum() + (on * ct_n).sum(), [img_last, img_next])
            sfx = "_replace" if replace else ""
            d.update({"out/ori_last" + sfx: ol.detach().numpy(), "out/ori_next" + sfx: on.detach().numpy(),
                      "in/ct_last" + sfx: ct_l.numpy(), "in/ct_next" + sfx: ct_n.numpy(),
                      "out/g_img_last" + sfx: gl.numpy(), "out/g_img_next" + sfx: gn.numpy()})
        d.update({"in/mask_last": last.numpy(), "in/mask_next": nxt.numpy(),
                  "in/img_last": (img_last.detach() * 255).round().to(torch.uint8).numpy(),
                  "in/img_next": (img_next.detach() * 255).round().to(torch.uint8).numpy()})
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **d)
        print(tag, {k: v.shape for k, v in d.items() if k.startswith("out/ori")})


class _Inst:
    """the slice of detectron2's Instances that image_synthesis touches"""

    def __init__(self, scores, masks):
        self.scores, self.pred_masks = scores, masks

    def __len__(self):
        return len(self.scores)

    def __getitem__(self, sel):
        return _Inst(self.scores[sel], self.pred_masks[sel])


def synthesis_stubs(B, H, W, seed):
    """stand-ins for the segmenter and the matcher that drive ``image_synthesis`` (dyn_utils.py:121-170) through every
    branch: sample 0 has no confident instance in the target frame (:135-136), sample 1 has matched instances, sample 2
    (if any) a confident target but an empty match (:146-147), further samples are matched."""
    masks = {b: make_masks(3, H, W, seed + b, edge_cases=False) for b in range(B)}
    state = {"pairs": [b for b in range(B) if b != 0]}

    def ins_model(images):
        n = len(images)
        if n != 2:  # the target frames: only the scores are read
            state["next"] = list(state["pairs"])
            return [{"instances": _Inst(torch.tensor([0.2, 0.3, 0.1] if b == 0 else [0.9, 0.95, 0.4]),
                                        torch.zeros(3, H, W, dtype=torch.bool))} for b in range(n)]
        b = state["next"].pop(0)
        state["cur"] = b
        return [{"instances": _Inst(torch.full((3,), 0.9), masks[b][0])}, {"instances": _Inst(torch.full((3,), 0.9), masks[b][1])}]

    def matcher(ins_last, ins_next, cur):
        if state["cur"] == 2:
            return torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)
        return torch.tensor([2, 0]), torch.tensor([1, 0])

    return ins_model, matcher, masks


def synthesis_case():
    """the reference's own ``image_synthesis`` (dyn_utils.py:121-170) on a (B,3,H,W) pair of warped images"""
    import manydepth.dyn_utils as DU
    B, H, W = 4, 24, 40
    g = torch.Generator().manual_seed(77)
    mk = lambda: (torch.round(torch.rand(B, 3, H, W, generator=g) * 255) / 255)
    tgt, cl, cn = mk(), mk().requires_grad_(True), mk().requires_grad_(True)
    ins_model, matcher, _ = synthesis_stubs(B, H, W, 40)
    outputs = {("color", -1, 0): cl, ("color", 1, 0): cn}
    has = DU.image_synthesis({("color", 0, 0): tgt}, outputs, 0, 0.5, ins_model, matcher)
    assert has
    ct_l = torch.round(torch.randn(B, 3, H, W, generator=g) * 64) / 64
    ct_n = torch.round(torch.randn(B, 3, H, W, generator=g) * 64) / 64
    sl, sn = outputs[("syn", -1, 0)], outputs[("syn", 1, 0)]
    gl, gn = torch.autograd.grad((sl * ct_l).sum() + (sn * ct_n).sum(), [cl, cn])
    u8 = lambda t: (t.detach() * 255).round().to(torch.uint8).numpy()
    d = {"in/target": u8(tgt), "in/color_last": u8(cl), "in/color_next": u8(cn), "in/ct_last": ct_l.numpy(), "in/ct_next": ct_n.numpy(),
         "out/syn_last": sl.detach().numpy(), "out/syn_next": sn.detach().numpy(), "out/g_last": gl.numpy(), "out/g_next": gn.numpy(),
         "in/stub_seed": np.int64(40)}
    np.savez_compressed(os.path.join(OUT, "dyn_synthesis_b4_24x40.npz"), **d)
    changed = [bool((sl[b] != cl[b]).any()) for b in range(B)]
    print("dyn_synthesis_b4_24x40 samples changed:", changed)
    assert changed == [False, True, False, True]


if __name__ == "__main__":
    main()
    synthesis_case()
