"""Weights for the network parity fixtures (N1) that do not have to be stored: every entry of a state dict is filled from a
generator seeded by the entry's NAME, so the reference's ``RepDepth`` (oracle/gen_golden_net.py, authoring container) and
``mal_amd.networks.RepDepth`` (the tests) hold identical parameters exactly when their state-dict keys and shapes
agree -- which is itself part of what is pinned (checkpoints interchange, trainer.py:1605-1636).  165 MB of parameters
become a seed."""
import zlib

import torch


def named_fill_(module, seed=0):
    """in place; returns the number of tensors written.  Convolution / linear weights ~ N(0, 2/fan_in) (activations
    keep their scale through 18 layers), biases and BatchNorm shifts small, BatchNorm scales around 1, running
    statistics as after a few training steps (so eval mode differs from train mode)."""
    sd = module.state_dict()
    n = 0
    for name, t in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if not t.dtype.is_floating_point:  # num_batches_tracked
            t.fill_(3)
            n += 1
            continue
        r = torch.randn(t.shape, generator=g, dtype=torch.float32)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = 0.5 + r.abs()
        elif leaf == "running_mean":
            v = 0.1 * r
        elif leaf == "weight" and t.dim() == 1:  # BatchNorm scale
            v = 1.0 + 0.1 * r
        elif leaf == "bias":
            v = 0.05 * r
        elif t.dim() >= 2:
            fan_in = t[0].numel()
            v = r * (2.0 / fan_in) ** 0.5
        else:
            v = 0.1 * r
        t.copy_(v.to(t.dtype))
        n += 1
    return n


def seeded_images(B, H, W, seed, frames=(0, -1, 1)):
    """colour frames with values k/255 (exact in the uint8 the fixture stores): a smooth random texture, the neighbouring
    frames shifted by a few pixels"""
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=g), size=(H + 16, W + 16),
                                           mode="bilinear", align_corners=False)
    base = (base + 0.08 * torch.rand(B, 3, H + 16, W + 16, generator=g)).clamp(0, 1)
    out = {}
    for f in frames:
        dx, dy = 8 + 3 * f, 8 + f
        img = base[:, :, dy:dy + H, dx:dx + W]
        out[f] = torch.round(img * 255).to(torch.uint8)
    return out
