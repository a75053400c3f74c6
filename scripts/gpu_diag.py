"""Diagnostic (GPU box): HIP loss path vs CPU oracle on the golden cases; prints error metrics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import golden_io as G
from tests import hip_harness as HH

def rel(a, b, floor=1e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)

def report(tag, fuse):
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    kw = G.opt_kwargs(z)
    o = HH.run_oracle(b, kw, n0, n1)
    h = HH.run_hip(b, kw, n0, n1, fuse=fuse)
    print("== %s fuse=%s  final hip %.8f oracle %.8f golden %.8f rel %.2e" % (tag, fuse, h["final"], o["final"], float(z["final_loss"]), rel(h["final"], o["final"])))
    for k in o["losses"]:
        print("   %-22s hip %.8f  oracle %.8f  rel %.2e" % (k, h["losses"][k], o["losses"][k], rel(h["losses"][k], o["losses"][k])))
    for k in HH.LEAVES:
        g, r = h["grads"][k], o["grads"][k]
        sc = np.abs(r).max() + 1e-30
        e = np.abs(g - r)
        print("   grad %-16s max|ref| %.3e  max err/max %.2e  L2 rel %.2e  frac>1e-4*max %.2e" % (
            k, sc, e.max() / sc, np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-30), (e > 1e-4 * sc).mean()))
    print("   depth multi max rel %.2e" % rel(h["multi_depth"], o["multi_depth"]).max())
    if "cons_target" in h:
        print("   cons_target max rel %.2e" % rel(h["cons_target"], o["cons_target"]).max())
    print("   consistency_mask mismatches %d" % int((h["consistency_mask"] != o["consistency_mask"]).sum()))

if __name__ == "__main__":
    from mal_amd import build
    build.build(verbose=False)
    print(torch.cuda.get_device_name(0))
    cases = G.STEP_CASES if len(sys.argv) < 2 else sys.argv[1:]
    for tag in cases:
        for fuse in (True, False):
            try:
                report(tag, fuse)
            except Exception as e:
                import traceback; traceback.print_exc()
                print("!! %s fuse=%s failed: %r" % (tag, fuse, e))
    report(G.BIG_CASE, True)
