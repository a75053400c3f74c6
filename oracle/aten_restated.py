"""Explicit restatements of the PyTorch ATen operators the reference's hot path calls.

TEST INFRASTRUCTURE ONLY (see oracle/mal_oracle.py).  The reference delegates this
arithmetic to a third-party dependency that is not vendored in /root/reference: PyTorch
(ATen), unpinned upstream (no requirements file), 2.10.0 in this image.  The published
algorithms are restated here as plain index arithmetic so that (i) the oracle does not
depend on ATen being right to check the HIP kernels, and (ii) the border / reflect /
rounding rules the kernels must reproduce are written down once:

* ``F.grid_sample(mode="bilinear", padding_mode="border")`` -- call sites
  manydepth/trainer.py:1122-1125,1196-1199 (align_corners=True) and
  dualrefine/trainer.py:444-447,477-480 (align_corners=False).
  unnormalise: ac=True  ix = (g+1)/2*(W-1);  ac=False  ix = ((g+1)*W-1)/2
  border: ix clipped to [0, W-1]; d(ix)/d(g) is zero where the clip is active,
  *inclusive* of the two end points (ATen treats the border itself as out of bounds
  for the gradient), else (W-1)/2 resp. W/2.
  bilinear: x0=floor(ix); the four taps weighted (1-tx)(1-ty)...; taps whose index falls
  outside the image (only x0+1==W / y0+1==H, with zero weight) contribute nothing.
* ``nn.ReflectionPad2d(1)`` then ``nn.AvgPool2d(3, 1)`` -- manydepth/layers.py:232-238:
  index -1 -> 1 and index n -> n-2; the pooled value is the sum of the nine taps
  (row-major) divided by 9.
"""
import torch


def unnormalize(g, size, align_corners):
    if align_corners:
        return ((g + 1) / 2) * (size - 1)
    return ((g + 1) * size - 1) / 2


def _clip_border(ix, size):
    """Clip to [0, size-1]; gradient is cut where ix <= 0 or ix >= size-1."""
    hi = float(size - 1)
    inside = (ix > 0) & (ix < hi)
    return torch.where(inside, ix, ix.detach().clamp(0.0, hi))


def grid_sample_bilinear_border(src, grid, align_corners=True):
    """src (B,C,H,W), grid (B,Ho,Wo,2) -> (B,C,Ho,Wo).  Differentiable wrt grid (and src)."""
    B, C, H, W = src.shape
    ix = _clip_border(unnormalize(grid[..., 0], W, align_corners), W)
    iy = _clip_border(unnormalize(grid[..., 1], H, align_corners), H)
    x0f, y0f = torch.floor(ix.detach()), torch.floor(iy.detach())
    tx, ty = ix - x0f, iy - y0f
    x0, y0 = x0f.long(), y0f.long()
    x1, y1 = x0 + 1, y0 + 1
    flat = src.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H))
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *xx.shape[1:])
        return v * ok.unsqueeze(1).to(src.dtype)

    w_nw = ((1 - tx) * (1 - ty)).unsqueeze(1)
    w_ne = (tx * (1 - ty)).unsqueeze(1)
    w_sw = ((1 - tx) * ty).unsqueeze(1)
    w_se = (tx * ty).unsqueeze(1)
    return tap(y0, x0) * w_nw + tap(y0, x1) * w_ne + tap(y1, x0) * w_sw + tap(y1, x1) * w_se


def grid_sample_forced_taps(src, grid, x0, y0, clipx, clipy, align_corners=True):
    """The same bilinear border sampling with its two discontinuous choices TAKEN from the caller instead of
    re-decided: ``x0``/``y0`` (B,Ho,Wo) long = floor of the clipped position, ``clipx``/``clipy`` bool = the clip is
    active (no gradient to the grid).  Used by the decision-forced parity tests: a position within rounding distance
    of an integer or of the border may fall on either side in two correct fp32 evaluations; the value is continuous
    there, only d/d(grid) differs.  With the decisions this function itself would take it equals
    ``grid_sample_bilinear_border`` exactly."""
    B, C, H, W = src.shape
    ix = unnormalize(grid[..., 0], W, align_corners)
    iy = unnormalize(grid[..., 1], H, align_corners)
    ix = torch.where(clipx, ix.detach().clamp(0.0, float(W - 1)), ix)
    iy = torch.where(clipy, iy.detach().clamp(0.0, float(H - 1)), iy)
    tx, ty = ix - x0.to(ix.dtype), iy - y0.to(iy.dtype)
    x1, y1 = x0 + 1, y0 + 1
    flat = src.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H))
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).reshape(B, C, *xx.shape[1:])
        return v * ok.unsqueeze(1).to(src.dtype)

    w_nw = ((1 - tx) * (1 - ty)).unsqueeze(1)
    w_ne = (tx * (1 - ty)).unsqueeze(1)
    w_sw = ((1 - tx) * ty).unsqueeze(1)
    w_se = (tx * ty).unsqueeze(1)
    return tap(y0, x0) * w_nw + tap(y0, x1) * w_ne + tap(y1, x0) * w_sw + tap(y1, x1) * w_se


def taps_of(grid, H, W, align_corners=True):
    """The decisions ``grid_sample_bilinear_border`` takes for ``grid``: (x0, y0, clipx, clipy)."""
    ix = unnormalize(grid[..., 0].detach(), W, align_corners)
    iy = unnormalize(grid[..., 1].detach(), H, align_corners)
    clipx, clipy = ~((ix > 0) & (ix < W - 1)), ~((iy > 0) & (iy < H - 1))
    ix, iy = ix.clamp(0.0, float(W - 1)), iy.clamp(0.0, float(H - 1))
    return torch.floor(ix).long(), torch.floor(iy).long(), clipx, clipy


def reflection_pad1(x):
    """(B,C,H,W) -> (B,C,H+2,W+2), pad 1 with reflection (edge not repeated)."""
    x = torch.cat([x[:, :, 1:2, :], x, x[:, :, -2:-1, :]], 2)
    return torch.cat([x[:, :, :, 1:2], x, x[:, :, :, -2:-1]], 3)


def avg_pool3(xp):
    """3x3 stride-1 valid average: row-major sum of the nine taps, then / 9."""
    Hp, Wp = xp.shape[-2:]
    H, W = Hp - 2, Wp - 2
    acc = None
    for dy in range(3):
        for dx in range(3):
            t = xp[:, :, dy:dy + H, dx:dx + W]
            acc = t if acc is None else acc + t
    return acc / 9
