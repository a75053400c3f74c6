"""which association of the four bilinear products does ATen's device kernel use?  (run on the GPU box)"""
import torch
from mal_amd import ops
x = torch.rand(2, 1, 6, 12, generator=torch.Generator().manual_seed(5)).cuda()
H, W = 48, 96
ref = torch.nn.functional.interpolate(x, [H, W], mode="bilinear", align_corners=False)
up = ops.upsample_bilinear(x, H, W)
print("lib vs aten max", float((up - ref).abs().max()), "frac differing", float((up != ref).float().mean()))
cpu = torch.nn.functional.interpolate(x.cpu(), [H, W], mode="bilinear", align_corners=False)
print("aten cuda vs cpu", float((ref.cpu() - cpu).abs().max()), float((ref.cpu() != cpu).float().mean()))
print("lib vs cpu", float((up.cpu() - cpu).abs().max()), float((up.cpu() != cpu).float().mean()))
# candidates in float64-emulated fma
import numpy as np
xs = x.cpu().numpy().astype(np.float32)
def taps(n_out, n_in):
    s = np.float32(n_in) / np.float32(n_out)
    d = np.arange(n_out, dtype=np.float32)
    src = np.maximum(s * (d + np.float32(0.5)) - np.float32(0.5), np.float32(0))
    i0 = src.astype(np.int32); i1 = i0 + (i0 < n_in - 1)
    l1 = (src - i0).astype(np.float32); l0 = (np.float32(1) - l1).astype(np.float32)
    return i0, i1, l0, l1
y0, y1, h0, h1 = taps(H, 6); x0, x1, w0, w1 = taps(W, 12)
a = xs[:, :, y0][:, :, :, x0]; b = xs[:, :, y0][:, :, :, x1]; c = xs[:, :, y1][:, :, :, x0]; d = xs[:, :, y1][:, :, :, x1]
h0_, h1_ = h0[None, None, :, None], h1[None, None, :, None]; w0_, w1_ = w0[None, None, None, :], w1[None, None, None, :]
f32 = np.float32
def fma(p, q, r): return (p.astype(np.float64) * q.astype(np.float64) + r.astype(np.float64)).astype(np.float32)
def mul(p, q): return (p * q).astype(np.float32)
cands = {
  "plain": mul(h0_, (mul(w0_, a) + mul(w1_, b)).astype(f32)) + mul(h1_, (mul(w0_, c) + mul(w1_, d)).astype(f32)),
  "fma(w0,a,w1*b); fma(h0,i1,h1*i2)": fma(h0_ + 0 * a, fma(w0_ + 0 * a, a, mul(w1_, b)), mul(h1_, fma(w0_ + 0 * a, c, mul(w1_, d)))),
  "fma(w1,b,w0*a); fma(h1,i2,h0*i1)": fma(h1_ + 0 * a, fma(w1_ + 0 * a, d, mul(w0_, c)), mul(h0_, fma(w1_ + 0 * a, b, mul(w0_, a)))),
  "fma(w0,a,w1*b); fma(h1,i2,h0*i1)": fma(h1_ + 0 * a, fma(w0_ + 0 * a, c, mul(w1_, d)), mul(h0_, fma(w0_ + 0 * a, a, mul(w1_, b)))),
  "fma(w1,b,w0*a); fma(h0,i1,h1*i2)": fma(h0_ + 0 * a, fma(w1_ + 0 * a, b, mul(w0_, a)), mul(h1_, fma(w1_ + 0 * a, d, mul(w0_, c)))),
}
for k, v in cands.items():
    v = v.astype(np.float32)
    print("%-40s vs aten-cuda: %d differing; vs cpu: %d; vs lib: %d" % (k, int((v != ref.cpu().numpy()).sum()), int((v != cpu.numpy()).sum()), int((v != up.cpu().numpy()).sum())))
