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


def depthbins2coords(poses, depths, K, min_depth, max_depth, num_depth_bins=96, bins_range=None):
    """utils.py:231-255 -> (coords (B,2,1,num_depth_bins,h,w), depth hypotheses (B,1,num_depth_bins,h,w)).
    ``bins_range`` = (min_depth_bin, max_depth_bin) selects the ``--use_depth_bins_for_masking`` branch; otherwise the
    hypotheses run from min_depth to 8x the pixel's depth (clamped to max_depth), linearly."""
    bsz, _, ht, wd = depths.shape
    if bins_range is not None:
        d = torch.linspace(bins_range[0], bins_range[1], num_depth_bins)
        d = d[None, None, :, None, None].repeat(bsz, 1, 1, ht, wd)
    else:
        lin = torch.linspace(0, 1, num_depth_bins)
        depths_ = 8 * (depths - min_depth) + min_depth
        depths_ = torch.clamp(depths_, max=max_depth)
        lin_ = (depths_ - min_depth) / (depths - min_depth)
        lin = lin[None, None, :, None, None] * lin_[:, None]
        d = lin * (depths[:, None] - min_depth) + min_depth
    X0 = iproj(d, K)
    X1 = poses.type(X0.dtype) @ X0.reshape(bsz, 4, -1)
    return proj(X1, K).reshape(bsz, 2, 1, num_depth_bins, ht, wd), d


def corr_all_channels(fmap1, f2_pyramid, coords, num_levels=1):
    """CoordSampler.__corr__ (corr.py:52-75): the mean over ALL channels, i.e. ``coord_sample`` with one head"""
    return coord_sample(fmap1, f2_pyramid, coords, num_levels, 1)


# ---------------------------------------------------------------- pose refinement step (forward)
def depth2gradcoords(poses, depths, K):
    """utils.py:219-236 -> (c1 (B,2,1,5,h,w): the projection and its +-1 px neighbours in x and y, X1 (B,4,h*w))"""
    bsz, _, ht, wd = depths.shape
    X0 = iproj(depths[:, None], K)
    X1 = poses.type(X0.dtype) @ X0.reshape(bsz, 4, -1)
    c1 = proj(X1, K).reshape(bsz, 2, 1, 1, ht, wd)
    p_dx = torch.tensor([1., 0.]).reshape(1, 2, 1, 1, 1, 1)
    p_dy = torch.tensor([0., 1.]).reshape(1, 2, 1, 1, 1, 1)
    return torch.cat([c1, c1 + torch.cat([p_dx, -p_dx, p_dy, -p_dy], 3)], 3), X1


def se3_exp(vec):
    """dualrefine/layers.py:29-55"""
    rho, phi = vec[:, :3], vec[:, 3:]
    theta = torch.norm(phi, 2, 1, keepdim=True)
    a = phi / theta
    a_skew = torch.zeros((vec.shape[0], 3, 3))
    a_skew[:, 0, 1] = -a[:, 2, 0]
    a_skew[:, 0, 2] = a[:, 1, 0]
    a_skew[:, 1, 0] = a[:, 2, 0]
    a_skew[:, 1, 2] = -a[:, 0, 0]
    a_skew[:, 2, 0] = -a[:, 1, 0]
    a_skew[:, 2, 1] = a[:, 0, 0]
    eye = torch.eye(3).unsqueeze(0)
    aat = torch.bmm(a, a.permute(0, 2, 1))
    R = torch.cos(theta) * eye + (1 - torch.cos(theta)) * aat + torch.sin(theta) * a_skew
    J = (torch.sin(theta) / theta) * eye + (1 - (torch.sin(theta) / theta)) * aat + (1 - torch.cos(theta)) / theta * a_skew
    T = torch.eye(4).unsqueeze(0).repeat(vec.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, -1:] = torch.bmm(J, rho.type(J.dtype))
    return T


def sample_tgt(tgt_feat, tgt_w, p2):
    """utils.py:370-392 -> (warped features (B,C,h,w), their central-difference gradients (B,C,h,w,2), warped weight)"""
    batch, _, n1, d1, h1, w1 = p2.shape
    p2 = p2.permute(2, 0, 4, 5, 3, 1).reshape(batch, h1 * w1, d1, 2)
    xg, yg = p2.split([1, 1], dim=-1)
    grid = torch.cat([2 * (xg + 0.5) / w1 - 1, 2 * (yg + 0.5) / h1 - 1], dim=-1)
    f = F.grid_sample(tgt_feat, grid, align_corners=False).view(batch, -1, h1, w1, d1)
    grads = torch.stack([(f[..., 1] - f[..., 2]) / 2, (f[..., 3] - f[..., 4]) / 2], dim=-1)
    w = F.grid_sample(tgt_w, grid[:, :, :1], align_corners=False).reshape(batch, 1, h1, w1)
    return f[..., 0], grads, w


def robust_weights(res, p2, height, width):
    """--robust_pose_loss, utils.py:344-348: scaled_barron(0, 0.1) of the squared residual (losses.py:8-19,41-90 with
    alpha = 0: first derivative 2 / (x + 2) at x = cost / 0.1**2) x the in-image test of the projected centre with a
    2 px margin (utils.py:409-412)"""
    batch_size = p2.shape[0]
    cost = (res[..., 0] ** 2).sum(-1).reshape(batch_size, 1, height, width)
    w_loss = 2 / (cost / (0.1 ** 2) + 2)
    pts, pad = p2[:, :, 0, 0], 2
    hi = torch.tensor([width - pad - 1, height - pad - 1]).to(pts).reshape(1, 2, 1, 1)
    valid = torch.all((pts >= pad) & (pts <= hi), 1, keepdim=True)
    return w_loss * valid.float()


def normal_equations(src_feat, tgt_feat, src_w, tgt_w, K, p2, P2, weight=None, robust=False):
    """utils.py:303-355 up to the solve: H (B,6,6), b (B,6)"""
    batch_size, channels, height, width = src_feat.shape
    warped, grads, warped_w = sample_tgt(tgt_feat, tgt_w, p2)
    X, Y, Z = P2[:, 0], P2[:, 1], P2[:, 2]
    fx, fy = K[:, 0, 0].reshape(-1, 1), K[:, 1, 1].reshape(-1, 1)
    fxz, fyz = fx / Z, fy / Z
    fxxz2, fyyz2 = fxz * X / Z, fyz * Y / Z
    zeros = torch.zeros(fxxz2.shape)
    J_pixel_xi = torch.stack((torch.stack((fxz, zeros, -fxxz2, -fxxz2 * Y, fx + fxxz2 * X, -fxz * Y), 1),
                              torch.stack((zeros, fyz, -fyyz2, -fy - fyyz2 * Y, fyyz2 * X, fyz * X), 1)), 1)
    J_pixel_xi = J_pixel_xi.permute(0, 3, 1, 2)
    J_img_pixel = grads.reshape(batch_size, channels, height * width, 2).permute(0, 2, 1, 3)
    J = -J_img_pixel @ J_pixel_xi
    res = (src_feat - warped).permute(0, 2, 3, 1).reshape(-1, height * width, channels, 1)
    w = src_w * warped_w
    if weight is not None:
        w = w * weight
    if robust:
        w = w * robust_weights(res, p2, height, width)
    JW = J * w.reshape(batch_size, height * width, 1, 1)
    H = (JW.transpose(2, 3) @ J).sum(1)
    b = (-res * JW).sum(2).sum(1)
    return H, b


def direct_align(poses, src_feat, tgt_feat, src_w, tgt_w, K, p2, P2, weight=None, robust=False):
    """utils.py:303-368: one Gauss-Newton step on the feature-metric error -> (new poses (B,4,4), update (B,6,1))"""
    H, b = normal_equations(src_feat, tgt_feat, src_w, tgt_w, K, p2, P2, weight, robust)
    L = torch.linalg.cholesky(H)
    update = torch.cholesky_solve(b[..., None], L)
    return torch.bmm(se3_exp(update).type(poses.dtype), poses), update
