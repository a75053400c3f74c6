"""Decision-exact gradient parity of the whole-step kernels (mal_loss_step_fwd/_bwd) against the CPU oracle.

The loss is piecewise smooth: per pixel it takes an argmin over candidates (loss_utils.py:103), the automask comparison
(:27-44,105-110), the distillation argmin (:237-254), the matching-mask thresholds (trainer.py:1066-1076), the bilinear
tap cell and the border clip of grid_sample, and the signs inside the L1 and smoothness terms.  Two correct fp32
evaluations may decide a near-tie differently, and a flipped decision changes gradients by O(1) at that pixel -- which
round 1's tests absorbed with percent-level allowances.  Here instead
  (i)  the kernels EXPORT their decisions (mal_step_args.dec_teacher / dec_student, MAL_DEC_* planes); they must equal
       the free-running oracle's except on a handful of pixels (<= 3e-4 N + 8 per kind), every one of which is shown to
       be a near-tie in the oracle's own numbers;
  (ii) the oracle is re-run taking the kernels' decisions (``forced=``): now both sides evaluate the same smooth function
       and EVERYTHING is held at the north star's 1e-4 -- loss scalars at 1e-5 rel, every per-pixel disparity gradient
       and the four pose gradients against the same forced oracle evaluated in fp64 (the exact value of that function),
       within 1e-4 or 1.25x the distance of the fp32 forced oracle from it, whichever is larger: at B=12 192x640 the
       reference's own fp32 arithmetic is 2e-4...6e-4 (L2 rel) from exact on the pose gradients -- sums of 1.5 M
       cancelling terms -- and the kernels are as close to exact as it is.  No pixel is exempted.
"""
import numpy as np
import pytest
import torch

from tests import golden_io as G
from tests import hip_harness as HH

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _l2rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def run_step_with_decisions(batch, opt_kw, n0, w_list=(0.7, 0.3), device="cuda:0"):
    from mal_amd import step, trainer
    from mal_amd.synthetic import to_dicts
    B, _, H, W = batch["color0"].shape
    dev = torch.device(device)
    opt = trainer.default_options(height=H, width=W, batch_size=B, **opt_kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=dev)
    for f, s in ((-1, "m1"), (1, "p1")):
        mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
        mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
    from mal_amd.synthetic import fake_image_synthesis
    synth = fake_image_synthesis(batch["syn_rects"]) if "syn_rects" in batch else None
    losses, loss_list, maps = step.loss_step(opt, inputs, mono_outputs, outputs, w_list=list(w_list), noise=n0.to(dev),
                                             want_decisions=True, image_synthesis=synth)
    losses["loss"].backward()
    torch.cuda.synchronize()
    return dict(losses={k: float(v.detach()) for k, v in losses.items()}, maps={k: v.cpu() for k, v in maps.items()},
                loss_list=None if loss_list is None else [float(l.detach()) for l in loss_list],
                grads={k: (t.grad if t.grad is not None else torch.zeros_like(t)).cpu().numpy() for k, t in leaves.items()})


def _frac_dist(sample, H, W):
    """distance (pixels) of the sampling positions to the nearest integer / border, min over frames and axes"""
    d = None
    for f in (-1, 1):
        g = sample[f].astype(np.float64)
        for v in ((g[..., 0] + 1) / 2 * (W - 1), (g[..., 1] + 1) / 2 * (H - 1)):
            dd = np.abs(v - np.round(v))
            d = dd if d is None else np.minimum(d, dd)
    return d[:, None]


def _gap2(stack):
    s = np.sort(stack, axis=1)
    return s[:, 1:2] - s[:, 0:1]


def check_decisions_are_near_ties(diff, o, b, n0, N):
    """every pixel where the kernels decided differently from the free-running oracle is a near-tie in the oracle"""
    B, _, H, W = b["color0"].shape
    idn = o["ident"] + n0.numpy() * np.float32(1e-5)
    rp_s = o["multi_cands"].min(1, keepdims=True)
    trio = np.concatenate([m for m in (o["mono_reproj"], o["ens"], rp_s) if m is not None], 1)
    tgt = b["color0"].numpy()

    def l1_gap(preds, cands):
        win = cands.argmin(1)[:, None]
        pred = preds[0]
        for i in range(1, len(preds)):
            pred = np.where(win == i, preds[i], pred)
        return np.abs(pred - tgt).min(1, keepdims=True)

    mono = o["mono_depth"]
    matching = 1.0 / b["lowest_cost"].numpy()[:, None]
    ratio = np.minimum(np.abs((matching - mono) / mono - 1.0), np.abs((mono - matching) / matching - 1.0))
    near = {
        "win_t": _gap2(o["mono_cands"]) <= 1e-4, "win_s": _gap2(o["multi_cands"]) <= 1e-4,
        "automask": np.abs(o["mono_reproj"] - idn) <= 1e-4, "distil": _gap2(trio) <= 1e-4,
        "tap_t": _frac_dist(o["mono_sample"], H, W) <= 1e-3, "tap_s": _frac_dist(o["multi_sample"], H, W) <= 1e-3,
        "l1_t": l1_gap(o["mono_preds"], o["mono_cands"]) <= 1e-5,
        "l1_s": l1_gap([o["multi_color"][-1], o["multi_color"][1]], o["multi_cands"]) <= 1e-5,
        "cmask": ratio <= 1e-5,
        "smooth_t": HH.smooth_sign_ambiguous(b["disp_teacher"].numpy()), "smooth_s": HH.smooth_sign_ambiguous(b["disp_student"].numpy()),
    }
    # a flipped winner also changes which candidate's L1 signs are reported
    near["l1_t"] = near["l1_t"] | diff["win_t"]
    near["l1_s"] = near["l1_s"] | diff["win_s"]
    counts = {}
    for k in HH.DEC_KINDS:
        d = diff[k]
        counts[k] = int(d.sum())
        assert counts[k] <= 3e-4 * N + 8, ("too many differing decisions", k, counts[k], N)
        unexplained = d & ~near[k]
        assert not unexplained.any(), ("decision differs away from any tie", k, np.argwhere(unexplained)[:5].tolist())
    return counts


def _to64(d):
    if isinstance(d, dict):
        return {k: _to64(v) for k, v in d.items()}
    if isinstance(d, (tuple, list)):
        return type(d)(_to64(v) for v in d)
    return d.double() if torch.is_tensor(d) and d.dtype == torch.float32 else d


def check_step_decision_exact(b, kw, n0, n1, w_list=(0.7, 0.3), return_runs=False):
    B, _, H, W = b["color0"].shape
    N = B * H * W
    h = run_step_with_decisions(b, kw, n0, w_list)
    o = HH.run_oracle(b, kw, n0, n1, w_list)
    kd = HH.kernel_decisions(h["maps"])
    od = HH.oracle_decisions(o, b, n0, no_ens=bool(kw.get("no_ens")))
    # the kernels report the L1 signs of the winning WARPED candidate; where a synthesised image won (temporal hint) its
    # signs are taken from the oracle's own
    syn_won = kd["teacher"]["win"] >= 2
    if syn_won.any():
        kd["teacher"]["l1"] = torch.where(syn_won, od["teacher"]["l1"], kd["teacher"]["l1"])
    counts = check_decisions_are_near_ties(HH.decision_differences(kd, od), o, b, n0, N)
    # ---- same decisions on both sides: hold everything at 1e-4
    f = HH.run_oracle(b, kw, n0, n1, w_list, forced=kd)
    pairs = [("reproj_loss/0", f["losses"]["reproj_loss/0"]), ("consistency_loss/0", f["losses"]["consistency_loss/0"]),
             ("distil_loss", f["losses"]["distil_loss"]), ("mono/loss", f["mono_losses"]["loss"]),
             ("mono/reproj_loss/0", f["mono_losses"]["reproj_loss/0"]), ("loss", f["final"])]
    for k, v in pairs:
        assert abs(h["losses"][k] - v) <= 1e-5 * abs(v), (k, h["losses"][k], v)
    b64 = {k: (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in b.items()}
    f64 = HH.run_oracle(b64, kw, n0.double(), n1.double(), w_list, forced=_to64(kd))
    report = {}
    for key in HH.LEAVES:
        g, r32, r64 = h["grads"][key], f["grads"][key], f64["grads"][key]
        tol = max(1e-4, 1.25 * _l2rel(r32, r64))
        report[key] = (_l2rel(g, r64), _l2rel(r32, r64))
        assert _l2rel(g, r64) <= tol, (key, "L2 rel to the exact (fp64) forced oracle", _l2rel(g, r64), "fp32 oracle:", _l2rel(r32, r64))
        if key.startswith("disp"):  # EVERY pixel, no exemptions
            sc = np.abs(r64).max()
            tol_px = max(1e-4, 1.25 * np.abs(r32 - r64).max() / sc)
            worst = np.abs(g - r64).max() / sc
            assert worst <= tol_px, (key, "worst pixel / map scale", worst, tol_px, np.unravel_index(np.abs(g - r64).argmax(), g.shape))
    if return_runs:
        return (h, o), counts, report
    return counts, report


def test_small_cases_need_no_floor_clause():
    """at the golden sizes plain 1e-4 (L2 rel to the exact forced oracle) holds for all six leaves"""
    for tag in ("step_b2_32x64_distil", "step_b3_37x50_distil"):
        z = G.load(tag)
        b = G.batch_from_golden(z)
        B, _, H, W = b["color0"].shape
        n0, n1 = G.noises(z, (B, 1, H, W))
        counts, report = check_step_decision_exact(b, G.opt_kwargs(z), n0, n1)
        assert all(v[0] <= 1e-4 for v in report.values()), report


@pytest.mark.parametrize("tag", ["step_b2_32x64_temporal"])
def test_temporal_hint_decision_exact(tag):
    """--temporal through the one-call step: the two synthesised candidates join the teacher's min (loss_utils.py:84-88);
    the golden vector holds the reference's own losses and gradients, incl. those that reach the teacher's disparity
    and the poses through syn -> warped image (dyn_utils.py:127-128,163-168)"""
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    kw = G.opt_kwargs(z)
    (h, o), counts, report = check_step_decision_exact(b, kw, n0, n1, return_runs=True)
    assert all(v[0] <= 1e-4 for v in report.values()), report
    # against the reference's own numbers (the golden file): scalars; gradients where no decision differs
    for k in ("reproj_loss/0", "consistency_loss/0", "distil_loss"):
        gv = float(z["losses/" + k])
        assert abs(h["losses"][k] - gv) <= 2e-3 * abs(gv), (k, h["losses"][k], gv)
    if sum(counts.values()) == 0:
        for key in HH.LEAVES:
            g, r = h["grads"][key], z["grad/" + key]
            assert _l2rel(g, r.reshape(g.shape)) <= 1e-4, (key, _l2rel(g, r.reshape(g.shape)))


def test_temporal_at_baseline_size():
    """--temporal --distil at B=12 192x640 (BASELINE.json configs[1] as written), rectangles standing in for the
    instance patches as in the golden vectors"""
    from mal_amd.synthetic import make_batch
    B, H, W = 12, 192, 640
    b = make_batch(B, H, W, seed=78, with_syn=True)
    g = torch.Generator().manual_seed(6)
    n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    counts, report = check_step_decision_exact(b, {"temporal": True}, n0, n1)
    print("temporal: differing decisions", counts, "L2 rel to fp64 (hip, fp32 oracle)", report)


def test_baseline_size_report():
    """BASELINE.json configs[1] shape (per GPU), the synthetic batch bench.py times: prints the decision counts and
    the distances that DESIGN.md quotes (the same case is asserted in tests/test_gpu_step.py)"""
    from mal_amd.synthetic import make_batch
    B, H, W = 12, 192, 640
    b = make_batch(B, H, W, seed=77)
    g = torch.Generator().manual_seed(5)
    n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    counts, report = check_step_decision_exact(b, {}, n0, n1)
    print("differing decisions", counts, "L2 rel to fp64 (hip, fp32 oracle)", report)


def test_decisions_do_not_change_results():
    """the instrumented instantiations run the same arithmetic: bitwise equal losses and gradients"""
    from tests.test_gpu_step import run_step
    from mal_amd.synthetic import make_batch
    b = make_batch(2, 40, 130, seed=41)
    g = torch.Generator().manual_seed(9)
    n0 = torch.randn(2, 1, 40, 130, generator=g)
    a = run_step(b, {}, n0)
    c = run_step_with_decisions(b, {}, n0)
    for k, v in a["losses"].items():
        assert c["losses"][k] == v, k
    for k, v in a["grads"].items():
        assert np.array_equal(c["grads"][k], v), k
