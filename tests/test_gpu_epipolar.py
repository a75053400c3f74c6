"""GPU parity of the epipolar correlation lookup (mal_amd.epipolar -> mal_epipolar_coords / mal_coord_sample_l1) against
the golden run of the reference's own Reprojections / CoordSampler (tests/golden/epi_*.npz) and, at DualRefine's size
(B=8, 128 channels, 48x160, radius 8, 3 levels), against the CPU oracle.  fp32, tolerance 1e-4 (north_star): the pose
product and the bilinear blend are summed in a different order than ATen's; a sample whose position lies within 1e-4 px
of a tap boundary may take the neighbouring taps (the blend is continuous there, so the value still agrees)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.test_epi_oracle import CASES, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _args(r, L):
    return SimpleNamespace(corr_radius=r, disable_pose_updates=True, gap_factor="depth", gap_factor_depth_ratio=8, num_levels=L)


def run(K, depth, poses, f1, f2, r, L, heads, delta):
    from mal_amd import epipolar
    d = lambda t: t.to(DEV)
    R = epipolar.Reprojections(_args(r, L)).to(DEV)
    with torch.no_grad():
        R.delta.fill_(float(delta))
        R._reg_intrinsics(d(K))
        c, max_dx, ds = R.depth2epipolarcoords(d(poses), d(depth))
        S = epipolar.CoordSampler(_args(r, L))
        S.register(d(f1), d(f2), num_levels=L)
        corr = S(c, L, heads)
    return c.cpu(), max_dx.cpu(), ds.cpu(), corr.cpu()


def check(got, ref):
    c, max_dx, ds, corr = got
    rc, rmax, rds, rcorr = ref
    assert torch.allclose(ds, rds, rtol=1e-6, atol=1e-6)
    assert torch.allclose(max_dx, rmax, rtol=1e-6, atol=1e-7)
    # coordinates in pixels: 1e-4 relative to the image size
    assert (c - rc).abs().max() <= 1e-4 * max(1.0, float(rc.abs().max()))
    assert (corr - rcorr).abs().max() <= 1e-4 * max(1.0, float(rcorr.abs().max()))


@pytest.mark.parametrize("tag", CASES)
def test_golden(tag):
    z, K, depth, poses, f1, f2, r, L, heads, delta = load(tag)
    t = lambda k: torch.from_numpy(z[k])
    check(run(K, depth, poses, f1, f2, r, L, heads, float(delta)),
          (t("out/coords"), t("out/max_dx"), t("out/depths"), t("out/corr")))


def test_dualrefine_size_against_the_cpu_checker():
    from oracle import epi_oracle as E
    from oracle.gen_golden_epi import make_case
    B, C, h, w, r, L = 8, 128, 48, 160, 8, 3
    K, depth, poses, f1, f2 = make_case(B, C, h, w, seed=3)
    delta = torch.tensor([0.7])
    with torch.no_grad():
        rc, rmax, rds = E.depth2epipolarcoords(poses, depth, K, delta, r=r, num_levels=L)
        rcorr = E.coord_sample(f1, E.pyramid(f2, L), rc, L, 1)
    check(run(K, depth, poses, f1, f2, r, L, 1, 0.7), (rc, rmax, rds, rcorr))


def test_forward_only_is_enforced():
    from mal_amd import epipolar, _lib
    S = epipolar.CoordSampler(_args(2, 1))
    with pytest.raises(_lib.MalError):
        S.register(torch.randn(1, 4, 8, 8, device=DEV, requires_grad=True), torch.randn(1, 4, 8, 8, device=DEV))
