"""Exploration for tests/test_gpu_decisions.py: how many decisions differ between the HIP kernels and the free-running
oracle, and what gradient agreement is left once the oracle is told the kernels' decisions."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mal_amd.synthetic import make_batch
from tests import hip_harness as HH
from tests import golden_io as G
from tests.test_gpu_step import run_step


def run_step_dec(batch, kw, n0):
    from mal_amd import step, trainer
    from mal_amd.synthetic import to_dicts
    B, _, H, W = batch["color0"].shape
    dev = torch.device("cuda:0")
    opt = trainer.default_options(height=H, width=W, batch_size=B, **kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=dev)
    for f, s in ((-1, "m1"), (1, "p1")):
        mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
        mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
    losses, loss_list, maps = step.loss_step(opt, inputs, mono_outputs, outputs, w_list=[0.7, 0.3], noise=n0.to(dev), want_decisions=True)
    losses["loss"].backward()
    torch.cuda.synchronize()
    return dict(losses={k: float(v.detach()) for k, v in losses.items()}, maps={k: v.cpu() for k, v in maps.items()},
                grads={k: t.grad.cpu().numpy() for k, t in leaves.items()})


def explore(name, b, kw, n0, n1):
    B, _, H, W = b["color0"].shape
    t0 = time.time()
    h = run_step_dec(b, kw, n0)
    o = HH.run_oracle(b, kw, n0, n1)
    kd = HH.kernel_decisions(h["maps"])
    od = HH.oracle_decisions(o, b, n0, no_ens=bool(kw.get("no_ens")))
    diff = HH.decision_differences(kd, od)
    print("==", name, "N =", B * H * W, "decisions that differ:", {k: int(v.sum()) for k, v in diff.items()})
    f = HH.run_oracle(b, kw, n0, n1, forced=kd)
    for k in ("reproj_loss/0", "consistency_loss/0", "distil_loss"):
        print("  loss %-20s hip %.8f forced %.8f free %.8f  rel(forced) %.2e rel(free) %.2e" % (
            k, h["losses"][k], f["losses"][k], o["losses"][k], abs(h["losses"][k] - f["losses"][k]) / abs(f["losses"][k]),
            abs(h["losses"][k] - o["losses"][k]) / abs(o["losses"][k])))
    print("  final hip %.8f forced %.8f free %.8f" % (h["losses"]["loss"], f["final"], o["final"]))
    for k in HH.LEAVES:
        g, rf, r0 = h["grads"][k], f["grads"][k], o["grads"][k]
        sc = np.abs(rf).max()
        e = np.abs(g - rf)
        l2 = np.linalg.norm((g - rf).ravel()) / np.linalg.norm(rf.ravel())
        l2free = np.linalg.norm((g - r0).ravel()) / np.linalg.norm(r0.ravel())
        msg = "  grad %-16s forced: max|d|/max|r| %.2e  L2rel %.2e   free: L2rel %.2e" % (k, e.max() / sc, l2, l2free)
        if g.ndim == 4:
            msg += "  pixels > 1e-4*max: %d, > 1e-5: %d" % (int((e > 1e-4 * sc).sum()), int((e > 1e-5 * sc).sum()))
            if (e > 1e-4 * sc).any():
                idx = np.argwhere(e > 1e-4 * sc)[:6]
                msg += "  e.g. " + str([tuple(int(v) for v in i) for i in idx])
        print(msg)
    print("  (%.1f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    for tag in ("step_b2_32x64_distil", "step_b3_37x50_distil", "step_b2_32x64_noens", G.BIG_CASE):
        z = G.load(tag)
        b = G.batch_from_golden(z)
        B, _, H, W = b["color0"].shape
        n0, n1 = G.noises(z, (B, 1, H, W))
        explore(tag, b, G.opt_kwargs(z), n0, n1)
    for (B, H, W) in ((12, 192, 640), (12, 192, 512)):
        b = make_batch(B, H, W, seed=77)
        g = torch.Generator().manual_seed(5)
        n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
        explore("synthetic B=%d %dx%d" % (B, H, W), b, {}, n0, n1)
