"""TEST INFRASTRUCTURE -- not part of the product (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
import anything under oracle/).

CPU restatement of DualRefine's epipolar correlation lookup (forward):

  depth2epipolarcoords   dualrefine/networks/utils/utils.py:180-217 (with _iproj :132-153 and _proj :155-168)
  CoordSampler           dualrefine/networks/corr.py:11-50 (register + __call__)

as DEQDepthPose's fixed-point function calls them (dualrefine/networks/depth_pose.py:433-435).  Plain torch on the CPU;
``F.grid_sample`` / ``F.avg_pool2d`` are ATen's, as upstream.  Pinned bit-exactly to the reference's own classes by
oracle/gen_golden_epi.py -> tests/golden/epi_*.npz (tests/test_epi_oracle.py).
"""
import torch
import torch.nn.functional as F


def intrinsics_of(K):
    """utils.py:170-173"""
    return K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]


def iproj(Z, K):
    """utils.py:132-153: Z (B,1,D,h,w) -> homogeneous points (B,4,D,h,w)"""
    fx, fy, cx, cy = intrinsics_of(K)
    ht, wd = Z.shape[-2:]
    y, x = torch.meshgrid(torch.arange(ht).float(), torch.arange(wd).float(), indexing="ij")
    X = (x[None] - cx[:, None, None]) / fx[:, None, None]
    Y = (y[None] - cy[:, None, None]) / fy[:, None, None]
    X, Y = X[:, None, None], Y[:, None, None]
    return torch.cat([Z * X, Z * Y, Z, torch.ones_like(Z)], dim=1)


def proj(Xs, K):
    """utils.py:155-168: (B,4,N) -> pixel coordinates (B,2,N); inverse depth clamped at 100"""
    fx, fy, cx, cy = intrinsics_of(K)
    X, Y, Z, _ = Xs.unbind(dim=1)
    d = torch.clamp(1.0 / Z, max=100)
    x = fx[:, None] * (X * d) + cx[:, None]
    y = fy[:, None] * (Y * d) + cy[:, None]
    return torch.stack([x, y], dim=1)


def depth2epipolarcoords(poses, depths, K, delta, r=8, num_levels=3, ratio=8):
    """utils.py:180-217 with ``--gap_factor depth`` (the default; the other choice, "minmax", evaluates
    ``self.minmax(r)``, which does not exist upstream -- the method is called ``minmax_gap`` -- and raises)
    -> (coords (B,2,L,2r+1,h,w), max_dx (B,1,h,w), depths (B,1,L*(2r+1),h,w)).
    ``delta``: the module's learnable scalar (softplus'd here as upstream)."""
    bsz, _, ht, wd = depths.shape
    dx = torch.linspace(-r, r, 2 * r + 1)[None, None, :, None, None]
    depths = depths[:, None]
    dd = F.softplus(delta)
    gap = dd * depths / ratio / r
    deltas, max_dx = [], None
    for level in range(num_levels):
        d = (2 ** level) * depths.new_ones(depths.shape) * gap
        if level == 0:
            max_dx = (dx * d).max(dim=2, keepdim=True)[0][:, 0]
        deltas.append(dx * d)
    deltas = torch.cat(deltas, 2)
    depths = depths + deltas
    X0 = iproj(depths, K)
    X1 = poses.type(X0.dtype) @ X0.reshape(bsz, 4, -1)
    c1 = proj(X1, K).reshape(bsz, 2, num_levels, 2 * r + 1, ht, wd)
    return c1, max_dx, depths


def pyramid(fmap2, num_levels):
    """corr.py:11-23"""
    out, f2 = [fmap2], fmap2
    for _ in range(num_levels - 1):
        f2 = F.avg_pool2d(f2, 2, stride=2)
        out.append(f2)
    return out


def coord_sample(fmap1, f2_pyramid, coords, num_levels=1, num_head=1):
    """corr.py:25-50 -> (B, num_levels * num_head * d1, h, w): mean over channel groups of |fmap1 - f2(coords)|"""
    batch, _, n1, d1, h1, w1 = coords.shape
    coords = coords.permute(2, 0, 4, 5, 3, 1).reshape(num_levels, batch, h1 * w1, d1, 2)
    f1 = fmap1[..., None]
    outs = []
    for i in range(num_levels):
        xg, yg = coords[i].split([1, 1], dim=-1)
        grid = torch.cat([2 * (xg + 0.5) / w1 - 1, 2 * (yg + 0.5) / h1 - 1], dim=-1)
        f2 = F.grid_sample(f2_pyramid[i], grid, align_corners=False).view(batch, -1, h1, w1, d1)
        corr = torch.abs(f1 - f2).view(batch, num_head, -1, h1, w1, d1).mean(2)
        outs.append(corr.permute(0, 2, 3, 1, 4).reshape(batch, h1, w1, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()
