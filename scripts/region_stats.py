"""how much of the image the bench's stand-in instances cover, and how many tiles of the syn sweep they touch"""
import sys, torch
sys.path.insert(0, ".")
from mal_amd.synthetic import instance_stub
B, H, W = 12, 192, 640
ins_model, matcher = instance_stub(B, H, W, n_inst=3, seed=1234, device="cpu")
tot = 0.0; tiles = 0; ntiles = 0
for b in range(B):
    out = ins_model(torch.zeros(2, 3, H, W))
    m = (out[0]["instances"].pred_masks | out[1]["instances"].pred_masks).any(0)
    tot += float(m.float().mean())
    d = torch.nn.functional.max_pool2d(m[None, None].float(), 5, 1, 2)[0, 0] > 0
    for y0 in range(0, H, 16):
        for x0 in range(0, W, 60):
            ntiles += 1
            tiles += int(d[y0:y0 + 16, x0:x0 + 60].any())
print("region fraction %.3f, tiles touched %.3f" % (tot / B, tiles / ntiles))
