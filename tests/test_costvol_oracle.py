"""The restated cost-volume construction (oracle/costvol_oracle.py, its own BackprojectDepth/Project3D) against the
golden run that used the reference's layer objects (tests/golden/costvol_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import costvol_oracle as CO
from tests import golden_io as G

# costvol_ref_*: the same cases produced by the reference's OWN ResnetEncoderMatching.compute_depth_bins /
# match_features / compute_confidence_mask / indices_to_disparity, called unbound (oracle/gen_golden_costvol.py)
CASES = ["costvol_b2_f2_16x28", "costvol_b1_f1_11x17", "costvol_ref_b2_f2_16x28", "costvol_ref_b1_f1_11x17"]


def load(tag):
    z = G.load(tag)
    t = lambda k: torch.from_numpy(z[k].astype(np.float32))
    return z, t("in/current"), t("in/lookup"), t("in/poses"), t("in/K"), t("in/invK"), t("in/bins")


@pytest.mark.parametrize("tag", CASES)
def test_restatement_matches_the_golden_run(tag):
    z, cur, look, poses, K, invK, bins = load(tag)
    with torch.no_grad():
        cv, miss = CO.match_features(cur, look, poses, K, invK, bins, True)
        cvm, low, conf = CO.encoder_outputs(cv, miss, bins)
    assert np.array_equal(miss.numpy(), z["out/missing"])
    assert np.array_equal(conf.numpy(), z["out/confidence"])
    np.testing.assert_allclose(cv.numpy(), z["out/cost_volume"], rtol=0, atol=0)
    np.testing.assert_allclose(cvm.numpy(), z["out/masked_cost_volume"], rtol=0, atol=0)
    assert np.array_equal(low.numpy(), z["out/lowest_cost"])


def test_depth_bins():
    assert torch.allclose(CO.depth_bins(0.5, 8.0, 4, "linear"), torch.tensor([0.5, 3.0, 5.5, 8.0]))
    b = CO.depth_bins(0.5, 8.0, 5, "inverse")
    assert abs(float(b[0]) - 0.5) < 1e-6 and abs(float(b[-1]) - 8.0) < 1e-6 and bool((b[1:] > b[:-1]).all())
