"""Diagnostic (GPU box): where do the full-size gradient outliers sit?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import golden_io as G
from tests import hip_harness as HH
from mal_amd import loss_utils, build
build.build(verbose=False)
tag = sys.argv[1] if len(sys.argv) > 1 else G.BIG_CASE
z = G.load(tag); b = G.batch_from_golden(z)
B, _, H, W = b["color0"].shape
n0, n1 = G.noises(z, (B, 1, H, W)); kw = G.opt_kwargs(z)
stash = {}
orig = loss_utils.compute_mono_losses
def wrap(*a, **k):
    l, mr = orig(*a, **k); stash["mono_reproj"] = mr.detach().cpu().numpy(); return l, mr
loss_utils.compute_mono_losses = wrap
o = HH.run_oracle(b, kw, n0, n1)
h = HH.run_hip(b, kw, n0, n1, fuse=True)
mr_h, mr_o = stash["mono_reproj"], o["mono_reproj"]
print("mono_reproj: max abs diff %.3e, frac rel>1e-4: %.3e, frac rel>1e-5 %.3e" % (np.abs(mr_h-mr_o).max(), (np.abs(mr_h-mr_o) > 1e-4*np.abs(mr_o)).mean(), (np.abs(mr_h-mr_o) > 1e-5*np.abs(mr_o)).mean()))
idn = o["ident"] + n0.numpy() * np.float32(1e-5)
margin_mask = mr_o - idn                      # automask decision margin (oracle)
cands = o["mono_cands"]
margin_min = np.abs(cands[:, 0:1] - cands[:, 1:2])
mask_o = (mr_o <= idn); mask_h = (mr_h <= idn)
print("automask flips: %d of %d; their |margin|:" % ((mask_o != mask_h).sum(), mask_o.size), np.sort(np.abs(margin_mask[mask_o != mask_h]))[:20])
g_h, g_o = h["grads"]["disp_teacher"], o["grads"]["disp_teacher"]
err = np.abs(g_h - g_o); sc = np.abs(g_o).max()
bad = err > 1e-4 * sc
print("teacher grad: bad px %d" % bad.sum())
flip = HH.dilate3(mask_o != mask_h)
tie_min = HH.dilate3(margin_min < 1e-5)
tie_mask = HH.dilate3(np.abs(margin_mask) < 1e-6)
print(" bad explained by automask flips: %d ; by near-tie argmin(<1e-5): %d ; by near-tie mask(<1e-6): %d; unexplained: %d" % ((bad & flip).sum(), (bad & tie_min).sum(), (bad & tie_mask).sum(), (bad & ~flip & ~tie_min & ~tie_mask).sum()))
idx = np.argwhere(bad & ~flip & ~tie_min)
for (bb, _, y, x) in idx[:12]:
    print("  px b%d y%d x%d  ref %.4e hip %.4e | rp %.6f idn %.6f | c0 %.6f c1 %.6f | mask_o %d mask_h %d" % (bb, y, x, g_o[bb,0,y,x], g_h[bb,0,y,x], mr_o[bb,0,y,x], idn[bb,0,y,x], cands[bb,0,y,x], cands[bb,1,y,x], mask_o[bb,0,y,x], mask_h[bb,0,y,x]))
good = ~(flip | tie_min | tie_mask)
print("teacher grad outside ambiguous: max err/max %.3e, L2 rel %.3e" % (err[good].max()/sc, np.linalg.norm((g_h-g_o)[good])/np.linalg.norm(g_o[good])))
# pose grad sensitivity
for k in ("axisangle_m1","translation_m1","axisangle_p1","translation_p1"):
    print(k, "hip", h["grads"][k].ravel()[:3], "oracle", o["grads"][k].ravel()[:3])
# fp64 oracle for the pose gradients
b64 = {k: (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in b.items()}
try:
    o64 = HH.run_oracle(b64, kw, n0.double(), n1.double())
    for k in HH.LEAVES:
        r64, r32, g = o64["grads"][k], o["grads"][k], h["grads"][k]
        print("%-16s |o32-o64|/|o64| %.3e   |hip-o64|/|o64| %.3e   |hip-o32|/|o32| %.3e" % (k, np.linalg.norm(r32-r64)/np.linalg.norm(r64), np.linalg.norm(g-r64)/np.linalg.norm(r64), np.linalg.norm(g-r32)/np.linalg.norm(r32)))
    print("final: o32 %.9f o64 %.9f hip %.9f" % (o["final"], o64["final"], h["final"]))
except Exception as e:
    import traceback; traceback.print_exc()
