"""GPU parity of the whole-step entry point (mal_loss_step_fwd/_bwd through mal_amd.step.loss_step)
against the CPU oracle and against the operator-level route of the same library."""
import numpy as np
import pytest
import torch

from mal_amd.synthetic import to_dicts
from tests import golden_io as G
from tests import hip_harness as HH

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["step_b2_32x64_distil", "step_b2_32x64_noens", "step_b2_32x64_lossblc", "step_b3_37x50_distil", "step_b2_32x64_learnens",
         "step_b2_32x64_dual"]


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _l2rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def run_step(batch, opt_kw, n0, w_list=(0.7, 0.3), device="cuda:0"):
    from mal_amd import step, trainer
    B, _, H, W = batch["color0"].shape
    dev = torch.device(device)
    opt = trainer.default_options(height=H, width=W, batch_size=B, **opt_kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=dev)
    for f, s in ((-1, "m1"), (1, "p1")):
        mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
        mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
    losses, loss_list, maps = step.loss_step(opt, inputs, mono_outputs, outputs, w_list=list(w_list), noise=n0.to(dev))
    losses["loss"].backward()
    torch.cuda.synchronize()
    return dict(losses={k: float(v.detach()) for k, v in losses.items()}, maps={k: v.cpu().numpy() for k, v in maps.items()},
                loss_list=None if loss_list is None else [float(l.detach()) for l in loss_list],
                grads={k: (t.grad if t.grad is not None else torch.zeros_like(t)).cpu().numpy() for k, t in leaves.items()})


@pytest.mark.parametrize("tag", CASES + [G.BIG_CASE])
def test_loss_step_against_oracle(tag):
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    _check_step(b, G.opt_kwargs(z), n0, n1)


@pytest.mark.parametrize("B,H,W", [(12, 192, 640), (12, 192, 512)], ids=["kitti_b12_192x640", "cityscapes_b12_192x512"])
def test_loss_step_at_baseline_sizes(B, H, W):
    """BASELINE.json configs[1] / configs[3] shapes on the synthetic batch bench.py uses: the oracle still
    finishes in seconds there, so the full-size step is held against it directly."""
    from mal_amd.synthetic import make_batch
    b = make_batch(B, H, W, seed=77)
    g = torch.Generator().manual_seed(5)
    n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    _check_step(b, {}, n0, n1)


@pytest.mark.parametrize("B,H,W", [(1, 8, 72), (1, 16, 61), (2, 10, 121), (1, 24, 8)],
                         ids=["two_strips", "strip_edge_61", "three_strips_ragged", "narrow_8"])
def test_loss_step_small_and_ragged_shapes(B, H, W):
    """strip / segment boundaries of the marching kernels: widths just past a 60-column strip, a few rows only,
    images narrower than one wavefront"""
    from mal_amd.synthetic import make_batch
    b = make_batch(B, H, W, seed=31)
    g = torch.Generator().manual_seed(6)
    n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    _check_step(b, {}, n0, n1)


def test_march_direction_option():
    """mal_set_option("march_flip"): odd row segments walking bottom-up (default) or top-down is a scheduling choice --
    same losses and gradients up to the fp32 order of the vertical window sums"""
    from mal_amd import _lib
    from mal_amd.synthetic import make_batch
    b = make_batch(2, 40, 130, seed=41)
    g = torch.Generator().manual_seed(9)
    n0 = torch.randn(2, 1, 40, 130, generator=g)
    lib = _lib.load()
    res = {}
    try:
        for flip in (0, 1):
            _lib.check(lib.mal_set_option(b"march_flip", flip), "march_flip")
            res[flip] = run_step(b, {}, n0)
    finally:
        lib.mal_set_option(b"march_flip", 1)
    for k, v in res[1]["losses"].items():
        assert abs(res[0]["losses"][k] - v) <= 2e-6 * max(abs(v), 1e-3), k
    for k, gq in res[1]["grads"].items():
        assert _l2rel(res[0]["grads"][k], gq) <= 2e-4, k  # a handful of near-tie pixels may take the other branch


def _check_step(b, kw, n0, n1):
    """free-running oracle: scalars and maps (with the explicit allowance of the near-tie pixels, which move a MEAN by
    at most their own terms); gradients: decision-exact (tests/test_gpu_decisions.py) -- the kernels' per-pixel
    decisions are shown to differ from the oracle's only at near-ties, then every leaf is held at 1e-4 / the fp32
    reference's own distance from exact arithmetic with the decisions forced.  No percent-level allowance anywhere."""
    from tests.test_gpu_decisions import check_step_decision_exact
    B, _, H, W = b["color0"].shape
    (h, o), _, _ = check_step_decision_exact(b, kw, n0, n1, return_runs=True)
    N = B * H * W
    amb_distil = HH.near_tie(np.concatenate([m for m in (o["mono_reproj"], o["ens"], o["multi_cands"].min(1, keepdims=True))
                                             if m is not None], 1), 2e-4)
    allow = float((np.abs(o["mono_depth"] - o["multi_depth"]) * amb_distil).sum() / N)
    pairs = [("reproj_loss/0", o["losses"]["reproj_loss/0"]), ("consistency_loss/0", o["losses"]["consistency_loss/0"]),
             ("distil_loss", o["losses"]["distil_loss"]), ("mono/loss", o["mono_losses"]["loss"]),
             ("mono/reproj_loss/0", o["mono_losses"]["reproj_loss/0"])]
    allow_auto, renorm, any_auto = HH.automask_tie_allowance(o, n0)  # automask pixels at rounding distance of the threshold
    for k, v in pairs:
        tol = 1e-4 * abs(v) + (allow if "distil" in k else 0.0) + (allow_auto if "distil" not in k and "consistency" not in k else 0.0)
        assert abs(h["losses"][k] - v) <= tol, (k, h["losses"][k], v, allow_auto)
    scale = B if kw.get("loss_blc") else 1
    assert abs(h["losses"]["loss"] - o["final"]) <= 1e-4 * abs(o["final"]) + scale * (allow + allow_auto), (h["losses"]["loss"], o["final"])
    if kw.get("loss_blc"):
        assert abs(h["loss_list"][0] - o["loss_list"][0]) <= 1e-4 * abs(o["loss_list"][0]) + allow_auto
    assert np.abs(h["maps"]["mono_reproj"].numpy() - o["mono_reproj"]).max() <= 1e-4
    if o["ens"] is not None:
        assert np.abs(h["maps"]["ens_reproj"].numpy() - o["ens"]).max() <= 1e-4
    assert (h["maps"]["consistency_mask"].numpy() != o["consistency_mask"]).mean() <= 1e-5


@pytest.mark.parametrize("tag", CASES + ["step_b2_32x64_temporal", "step_b2_32x64_temporal_main", G.BIG_CASE])
def test_loss_step_equals_operator_route(tag):
    """one C call vs ~60 operator launches: same kernels underneath, same numbers -- so the decision-exact parity shown
    for the one-call step (tests/test_gpu_decisions.py) carries over to the operator-level drop-ins (mal_amd.loss_utils /
    MALLossPath) for every configuration both routes take."""
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, n1 = G.noises(z, (B, 1, H, W))
    kw = G.opt_kwargs(z)
    if "syn_rects" in b:
        from tests.test_gpu_decisions import run_step_with_decisions
        a = run_step_with_decisions(b, kw, n0)
    else:
        a = run_step(b, kw, n0)
    c = HH.run_hip(b, kw, n0, n1, fuse=True)
    want = a["losses"]["loss"] if not kw.get("loss_blc") else None
    if want is not None:
        assert abs(want - c["final"]) <= 2e-6 * abs(c["final"]), (want, c["final"])
    else:  # LossBalancing: bs * sum_i w_i L_i is formed on the device by the step, on the host side by the operator route
        assert abs(a["losses"]["loss"] - c["final"]) <= 2e-6 * abs(c["final"]), (a["losses"]["loss"], c["final"])
    for k in list(HH.LEAVES) + (["disp_ens"] if "disp_ens" in b else []):
        ga, gc = a["grads"][k], c["grads"][k]
        assert np.abs(ga - gc).max() <= 2e-5 * np.abs(gc).max(), (k, np.abs(ga - gc).max() / np.abs(gc).max())
    if "disp_ens" in b:  # --learn_ens: ... and against the reference's own gradient w.r.t. the ensemble head's disparity
        from tests import hip_harness as H2
        o = H2.run_oracle(b, kw, n0, n1)
        amb = H2.near_tie(np.concatenate([o["mono_reproj"], o["ens"], o["multi_cands"].min(1, keepdims=True)], 1), 2e-4)
        g, r = a["grads"]["disp_ens"], z["grad/disp_ens"]
        assert np.abs(r).max() > 0 and np.abs(g - r)[~amb].max() <= 1e-4 * np.abs(r).max()


@pytest.mark.parametrize("kw", [{"temporal": True}, {"temporal": True, "main_temporal": True}, {"main_temporal": True}],
                         ids=["temporal", "both", "student_only"])
def test_temporal_hint_steps_replay_from_a_graph(kw):
    """the three library calls around the producer(s), forward and backward, captured into one HIP graph (as bench.py replays the
    headline; the side stream's fork / join are event-based and capturable): a replay leaves the same loss and gradients as
    the eager step, bit for bit"""
    from mal_amd import step, trainer
    from mal_amd.synthetic import make_batch
    B, H, W = 2, 40, 130
    b = make_batch(B, H, W, seed=43, with_syn=True)
    g_ = torch.Generator().manual_seed(9)
    n0 = torch.randn(B, 1, H, W, generator=g_).to(DEV)
    dev = torch.device(DEV)
    opt = trainer.default_options(height=H, width=W, batch_size=B, **kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(b, lambda a, t, inv: None, device=dev)
    for f, sfx in ((-1, "m1"), (1, "p1")):
        mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + sfx]
        mono_outputs[("translation", 0, f)] = leaves["translation_" + sfx]
    synth = HH.producer_of(b, dev)
    one_ = torch.ones((), device=dev)
    hold = {}

    def one():
        for t in leaves.values():
            t.grad = None
        losses, _, _ = step.loss_step(opt, inputs, dict(mono_outputs), dict(outputs), w_list=[0.7, 0.3], noise=n0, want_maps=False,
                                      image_synthesis=synth)
        losses["loss"].backward(gradient=one_)
        hold["loss"] = losses["loss"].detach()

    # every eager step on a side stream, as PyTorch's capture rules ask (and bench.py does): a leaf's gradient accumulator
    # remembers the stream of the leaf's first use and the backward synchronises with it -- were that the legacy default stream,
    # the captured backward would try to pull it into the capture, which invalidates the capture (and this runtime then dies
    # in capture_end instead of raising)
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        one()
        one()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    ref_loss = float(hold["loss"])
    ref = {k: t.grad.clone() for k, t in leaves.items()}
    graph = torch.cuda.CUDAGraph()
    # thread_local, as bench.py captures: the backward's nodes run on autograd's device thread, whose allocations a capture in
    # "global" mode forbids (the runtime then dies in capture_end instead of raising)
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        one()
    grads = {k: t.grad for k, t in leaves.items()}  # the buffers the captured backward writes
    for _ in range(2):
        for t in grads.values():
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert float(hold["loss"]) == ref_loss
        for k, t in grads.items():
            assert torch.equal(t, ref[k]), k


def test_loss_step_rejects_unsupported_options():
    from mal_amd import step, trainer, _lib
    opt = trainer.default_options(temporal=True)
    with pytest.raises(_lib.MalError):
        step.loss_step(opt, {("color", 0, 0): torch.zeros(1, 3, 4, 4, device="cuda")}, {}, {})


@pytest.mark.parametrize("tag", ["step_b3_37x50_distil", "step_b2_32x64_distil"])
def test_temporal_step_without_instances_equals_the_distil_step(tag):
    """opt.temporal with a producer that finds no matched instance (has_ins False): the reference then takes the min
    over the two warped candidates only (loss_utils.py:84), i.e. the --distil step.  B >= 2, so a sweep that indexed
    the batch-strided halves of the warped pair as if they were contiguous would read other samples' images."""
    from mal_amd import step, trainer
    z = G.load(tag)
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    assert B >= 2
    n0, _ = G.noises(z, (B, 1, H, W))
    ref = run_step(b, {}, n0)
    dev = torch.device("cuda:0")
    opt = trainer.default_options(height=H, width=W, batch_size=B, temporal=True)
    inputs, mono_outputs, outputs, leaves = to_dicts(b, lambda a, t, inv: None, device=dev)
    for f, s in ((-1, "m1"), (1, "p1")):
        mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
        mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
    calls = []

    def no_instances(inputs_, outputs_, scale):
        calls.append(tuple(outputs_[("color", -1, scale)].shape))
        return False

    losses, _, maps = step.loss_step(opt, inputs, mono_outputs, outputs, w_list=[0.7, 0.3], noise=n0.to(dev),
                                     image_synthesis=no_instances)
    losses["loss"].backward()
    torch.cuda.synchronize()
    assert calls == [(B, 3, H, W)] and mono_outputs["has_ins"] is False and ("syn", -1, 0) not in mono_outputs
    for k, v in ref["losses"].items():
        assert abs(float(losses[k]) - v) <= 2e-6 * max(abs(v), 1e-3), (k, float(losses[k]), v)
    assert np.abs(maps["mono_reproj"].cpu().numpy() - ref["maps"]["mono_reproj"]).max() <= 1e-6
    for k, t in leaves.items():
        g = (t.grad if t.grad is not None else torch.zeros_like(t)).cpu().numpy()
        assert np.abs(g - ref["grads"][k]).max() <= 2e-5 * np.abs(ref["grads"][k]).max(), k


def test_a_raising_producer_leaves_the_library_usable():
    """opt.temporal: mal_loss_step_warp forks the ensemble pass onto a side stream; if the producer then raises, the step
    joins it (mal_loss_step_abort) before the exception travels on, and the next step runs as if nothing had happened"""
    from mal_amd import step, trainer
    z = G.load("step_b2_32x64_distil")
    b = G.batch_from_golden(z)
    B, _, H, W = b["color0"].shape
    n0, _ = G.noises(z, (B, 1, H, W))
    dev = torch.device("cuda:0")
    opt = trainer.default_options(height=H, width=W, batch_size=B, temporal=True)

    def run(producer):
        inputs, mono_outputs, outputs, leaves = to_dicts(b, lambda a, t, inv: None, device=dev)
        for f, s in ((-1, "m1"), (1, "p1")):
            mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
            mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
        losses, _, _ = step.loss_step(opt, inputs, mono_outputs, outputs, w_list=[0.7, 0.3], noise=n0.to(dev), image_synthesis=producer)
        losses["loss"].backward()
        torch.cuda.synchronize()
        return float(losses["loss"]), leaves["disp_teacher"].grad.cpu().numpy()

    class Boom(Exception):
        pass

    def raising(inputs_, outputs_, scale):
        raise Boom("segmenter failed")

    ref = run(lambda i, o, s: False)
    for _ in range(2):
        with pytest.raises(Boom):
            run(raising)
    again = run(lambda i, o, s: False)
    assert again[0] == ref[0] and np.array_equal(again[1], ref[1])


@pytest.mark.parametrize("option,temporal", [("march3", False), ("temporal_spec", True)], ids=["march3", "temporal_spec"])
def test_alternative_teacher_schedules_hold_against_the_oracle(option, temporal):
    """Two schedules of the teacher's gradient pass that were measured SLOWER and are off by default (LABBOOK.md 6), kept
    for same-box A/B -- mal_set_option("march3", 1): three cooperating waves per strip (warp | statistics | gradient row);
    mal_set_option("temporal_spec", 1): the --temporal step's pass in front of the producer already takes the gradient and
    the sweep behind it only redoes the tasks near the region map.  Each is held against the oracle by the same
    decision-exact check as the default schedule."""
    from mal_amd import _lib
    from mal_amd.synthetic import make_batch
    lib = _lib.load()
    if not lib.mal_build_has_experiments():
        # the default library does not contain the losing schedules and says so instead of silently running the default
        assert lib.mal_set_option(option.encode(), 1) != 0 and lib.mal_set_option(option.encode(), 0) == 0
        pytest.skip("built without -DMAL_EXPERIMENTS (MAL_EXPERIMENTS=1 python -m mal_amd.build)")
    _lib.check(lib.mal_set_option(option.encode(), 1), option)
    try:
        if temporal:
            from tests.test_gpu_decisions import test_temporal_hint_decision_exact, test_temporal_at_baseline_size
            test_temporal_hint_decision_exact("step_b2_32x64_temporal")
            test_temporal_at_baseline_size(12, 192, 640)
        else:
            for (B, H, W), seed in (((2, 40, 130), 41), ((12, 192, 640), 77), ((1, 16, 61), 31)):
                b = make_batch(B, H, W, seed=seed)
                g = torch.Generator().manual_seed(5)
                n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
                _check_step(b, {}, n0, n1)
    finally:
        lib.mal_set_option(option.encode(), 0)


@pytest.mark.parametrize("what", ["temporal_step", "four_scales", "dualrefine"])
def test_specialised_passes_equal_the_generic_ones_bit_for_bit(what):
    """mal_set_option("march_lean"): the teacher / student / refinement entry points of the whole-step lists only compile
    out code for operands the lists never pass (march_body SPEC), so every loss and every gradient is the SAME number as
    with the generic instantiations -- temporal teacher sweep + scale-0 student (headline step), the students of the lower
    scales (four-scale list), DualRefine's refinement pass; the plain teacher is covered by
    test_decisions_do_not_change_results (instrumented = generic against production = specialised)."""
    from mal_amd import _lib, dualrefine, dyn_utils, layers, step, trainer
    from mal_amd.synthetic import instance_stub, make_batch, to_dicts
    lib = _lib.load()
    B, H, W = 3, 48, 136
    batch = make_batch(B, H, W, seed=53)
    g = torch.Generator().manual_seed(3)
    noises = [torch.randn(B, 1, H, W, generator=g).to(DEV) for _ in range(4)]

    def run():
        if what == "dualrefine":
            from tests.test_gpu_decisions import _dr_build
            inputs, outputs, gl = _dr_build(batch, DEV, layers.transformation_from_parameters)
            lp = dualrefine.DualRefineLossPath(dualrefine.default_options(height=H, width=W, batch_size=B, n_losses=1), fuse=True)
            losses = lp.loss_step(inputs, outputs, noises=noises[:2])
            leaves = gl
        else:
            inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=DEV)
            for f, s in ((-1, "m1"), (1, "p1")):
                mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
                mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
            if what == "temporal_step":
                ins_model, matcher = instance_stub(B, H, W, n_inst=2, seed=5, device=DEV)
                synth = lambda i, o, sc: dyn_utils.image_synthesis(i, o, sc, 0.5, ins_model, matcher)
                opt = trainer.default_options(height=H, width=W, batch_size=B, temporal=True)
                losses, _, _ = step.loss_step(opt, inputs, mono_outputs, outputs, noise=noises[0].clone(), image_synthesis=synth)
            else:
                for s in range(1, 4):
                    inputs[("color", 0, s)] = torch.nn.functional.avg_pool2d(batch["color0"], 2 ** s).to(DEV)
                    for name, outs in (("disp_teacher", mono_outputs), ("disp_student", outputs)):
                        leaf = torch.nn.functional.avg_pool2d(batch[name], 2 ** s).to(DEV).clone().requires_grad_(True)
                        leaves["%s_s%d" % (name, s)] = leaf
                        outs[("disp", s)] = leaf
                opt = trainer.default_options(height=H, width=W, batch_size=B, sclm=3, distil=False)
                losses, _ = step.loss_step_multiscale(opt, inputs, mono_outputs, outputs, noises=[n.clone() for n in noises])
        losses["loss"].backward()
        torch.cuda.synchronize()
        return ({k: float(v.detach()) for k, v in losses.items()},
                {k: t.grad.clone() for k, t in leaves.items() if t.grad is not None})

    res = {}
    try:
        for lean in (0, 1):
            _lib.check(lib.mal_set_option(b"march_lean", lean), "march_lean")
            res[lean] = run()
    finally:
        lib.mal_set_option(b"march_lean", 1)
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) >= 6
    for k, v in res[0][1].items():
        assert torch.equal(v, res[1][1][k]), k


def test_a_backward_after_a_later_forward_of_the_same_shape_is_refused():
    """the whole-step calls keep their maps, sums and coefficients in a workspace cached per (device, stream, shape): a
    backward that runs after a later forward reused it must fail loudly, not return the other step's gradients"""
    from mal_amd import _lib, step, trainer
    from mal_amd.synthetic import make_batch
    B, H, W = 2, 32, 64
    opt = trainer.default_options(height=H, width=W, batch_size=B)
    held = []
    for seed in (3, 4):
        batch = make_batch(B, H, W, seed=seed)
        inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=DEV)
        for f, s in ((-1, "m1"), (1, "p1")):
            mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
            mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
        losses, _, _ = step.loss_step(opt, inputs, mono_outputs, outputs, want_maps=False)
        held.append((losses["loss"], leaves))
    with pytest.raises(_lib.MalError):
        held[0][0].backward()
    held[1][0].backward()  # the latest forward still owns the workspace
    torch.cuda.synchronize()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in held[1][1].values())


@pytest.mark.parametrize("temporal", [False, True], ids=["distil", "temporal"])
def test_channels_last_inputs_are_the_texels(temporal):
    """Zero-copy texels: the three images handed over in torch.channels_last memory format are used as the (B,H,W,3) texel
    images directly (MAL_STEP_TEXEL_INPUTS: the first sweep writes no texel copy) -- losses and every gradient are
    bit-identical to the NCHW step (the same loads, in another place)."""
    from mal_amd import step, trainer
    from mal_amd.synthetic import make_batch
    B, H, W = 3, 40, 130
    batch = make_batch(B, H, W, seed=17, with_syn=temporal)
    g = torch.Generator().manual_seed(4)
    noise = torch.randn(B, 1, H, W, generator=g).to(DEV)
    opt = trainer.default_options(height=H, width=W, batch_size=B, temporal=temporal)
    synth = HH.producer_of(batch, torch.device(DEV)) if temporal else None

    def run(channels_last):
        inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=torch.device(DEV))
        for f, s in ((-1, "m1"), (1, "p1")):
            mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
            mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
        if channels_last:
            for f in (0, -1, 1):
                inputs[("color", f, 0)] = inputs[("color", f, 0)].contiguous(memory_format=torch.channels_last)
                assert not inputs[("color", f, 0)].is_contiguous()
        losses, _, maps = step.loss_step(opt, inputs, mono_outputs, outputs, noise=noise.clone(), image_synthesis=synth)
        losses["loss"].backward()
        torch.cuda.synchronize()
        return ({k: float(v.detach()) for k, v in losses.items()}, {k: t.grad.cpu() for k, t in leaves.items()},
                {k: v.cpu() for k, v in maps.items()})

    la, ga, ma = run(False)
    lb, gb, mb = run(True)
    assert la == lb, (la, lb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    for k in ma:
        assert torch.equal(ma[k], mb[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("main_temporal", [False, True], ids=["temporal", "temporal+main_temporal"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_tail_overlap_equals_the_serial_backward(graph, main_temporal):
    """option "tail_overlap" (default 1, round 5): a --temporal step's backward chain -- the producer's backward, then the teacher's
    gradient sweep -- runs on the library's side stream behind the fused sweep only, beside the epilogue and the reduction that
    the forward left on the caller's stream; the sweep then leaves the unnormalised map and the assembly finishes it (as in the
    step without the hint).  Same losses to the bit, the teacher's disparity gradient to rounding (one fused multiply-add
    association), every other gradient to the bit -- eagerly and replayed from a captured graph, three steps in a row."""
    import bench
    from mal_amd import _lib, config
    from mal_amd import step as step_mod
    lib = _lib.load()
    dev = torch.device("cuda:0")
    res = {}
    old_noise = config.noise_source, config.noise_seed
    try:
        for opt in (0, 1):
            assert lib.mal_set_option(b"tail_overlap", opt) == 0
            step = bench.Step(dev, 4321, "step", main_temporal=main_temporal)
            step_mod.noise_counter(dev).zero_()  # the in-kernel tie-break noise: the same draws for both runs
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                runs = []
                if graph:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
                        loss = step()
                for _ in range(3):
                    if graph:
                        g.replay()
                    else:
                        loss = step()
                    torch.cuda.synchronize()
                    runs.append((float(loss.detach()), {k: t.grad.detach().clone() for k, t in step.leaves.items() if t.grad is not None}))
                res[opt] = runs
    finally:
        lib.mal_set_option(b"tail_overlap", 1)
        config.noise_source, config.noise_seed = old_noise  # (bench.Step switches the noise source)
    for (l0, g0), (l1, g1) in zip(res[0], res[1]):
        assert l0 == l1
        assert set(g0) == set(g1)
        for k in g0:
            if k == "disp_teacher":  # (with --main_temporal the student's sweep left the unnormalised map before, too)
                sc = float(g0[k].abs().max())
                assert float((g0[k] - g1[k]).abs().max()) <= 2e-6 * sc, (k, float((g0[k] - g1[k]).abs().max()) / sc)
            else:
                assert torch.equal(g0[k], g1[k]), k
