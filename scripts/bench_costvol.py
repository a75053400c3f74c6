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

# What bounds these gather kernels is the vector-memory address path, not HBM: scripts/vmem_probe.hip measures, per CU,
# 5.4 cycles per dword wave-load whose 64 lanes fall in one 128-byte line (chip: 102 G wave-loads/s), 16 cycles over 2-4
# lines (37.5 G/s), 32 over 8 (18.8 G/s), 64 over 16 or more (9.5 G/s) -- all L1 hits (profiles/r02_vmem_probe.txt).
WAVE_LOAD_CEILING = {1: 102.0e9, 4: 37.5e9, 8: 18.8e9, 16: 9.5e9}
wave_loads = B * h * w * D * C * 4 / 64.0  # four taps per (pixel, bin, channel), 64 pixels per wave
# algorithmic traffic: read both feature maps once, write the volume + masks
alg = B * h * w * (2 * C * 4 + D * 4 + 8)
print("HIP %.0f us (%.1f GB/s algorithmic; %d (pixel,bin) pairs -> %.1f G pair-channels/s)   CPU checker %.2f s (%d threads)" %
      (th * 1e6, alg / th / 1e9, B * h * w * D, B * h * w * D * C / th / 1e9, tc, torch.get_num_threads()))
print("bound: vector-memory address path: %.1f M dword wave-loads -> %.1f G wave-loads/s = %.2f of the coalesced (one line "
      "per load) ceiling %.0f G/s measured by scripts/vmem_probe.hip [whole call incl. the finishing kernel]" %
      (wave_loads / 1e6, wave_loads / th / 1e9, wave_loads / th / WAVE_LOAD_CEILING[1], WAVE_LOAD_CEILING[1] / 1e9))
