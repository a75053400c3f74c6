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


if __name__ == "__main__":
    main()
