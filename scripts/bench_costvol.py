"""Cost volume (N3) at MAL's size: B=12, one lookup frame, 64 channels, 96 bins, 48x160 (192x640 / 4).
HIP kernels (mal_cost_volume; argv[1] = costvol_impl, default 1) vs the reference's formulation run with torch on the
same GPU (the oracle's code moved to the device) and on the host CPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mal_amd import costvol
from oracle import costvol_oracle as CO
from oracle.gen_golden_costvol import make_case
B, F_, C, h, w, D = 12, 1, 64, 48, 160, 96
cur, look, poses, K, invK = make_case(B, F_, C, h, w, D, seed=5)
poses[:, :, :3, 3] *= 0.25
bins = CO.depth_bins(0.5, 20.0, D, "linear")
dev = torch.device("cuda:0")
g = [t.to(dev) for t in (cur, look, poses, K, invK)]
from mal_amd import _lib
if len(sys.argv) > 1:
    _lib.check(_lib.load().mal_set_option(b"costvol_impl", int(sys.argv[1])), "costvol_impl")
def hip():
    return costvol.cost_volume_outputs(*g, bins, True)
for _ in range(3): hip()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): hip()
torch.cuda.synchronize(); th = (time.perf_counter() - t) / 20
t = time.perf_counter()
with torch.no_grad():
    cv, miss = CO.match_features(cur, look, poses, K, invK, bins, True)
    CO.encoder_outputs(cv, miss, bins)
tc = time.perf_counter() - t
# algorithmic traffic: read both feature maps once, write the volume + masks
alg = B * h * w * (2 * C * 4 + D * 4 + 8)
print("HIP %.0f us (%.1f GB/s algorithmic; %d (pixel,bin) pairs -> %.1f G pair-channels/s)   CPU checker %.2f s (%d threads)" %
      (th * 1e6, alg / th / 1e9, B * h * w * D, B * h * w * D * C / th / 1e9, tc, torch.get_num_threads()))
