"""The in-kernel tie-break noise of the whole-step API (MAL_STEP_NOISE_PHILOX): same values as the CPU restatement of
Philox4x32-10 + Box-Muller, a fresh draw every step -- also when the step is replayed from a HIP graph --, and a step
whose result equals the step run with that very noise handed in (so the decision-exact parity tests, which pass the
noise explicitly, cover it)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import noise_oracle as NO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def _noise(seed, step, B, H, W):
    from mal_amd import _lib, ops
    out = torch.empty(B, 1, H, W, device="cuda")
    _lib.check(_lib.load().mal_tiebreak_noise(C.c_uint64(seed), C.c_uint64(step), B, H, W, out.data_ptr(), ops._stream()),
               "mal_tiebreak_noise")
    torch.cuda.synchronize()
    return out


def test_matches_cpu_restatement():
    for seed, step, shape in ((1234, 0, (2, 37, 50)), (0x4d414c5eed, (1 << 33) + 5, (1, 192, 640))):
        g = _noise(seed, step, *shape).cpu().numpy()
        r = NO.tiebreak_noise(seed, step, *shape)
        # the device uses the fast log / sin / cos: a few 1e-6 relative on values of order 1 (the noise is scaled by 1e-5)
        assert np.abs(g - r).max() <= 2e-5 * max(1.0, np.abs(r).max()), np.abs(g - r).max()
        assert abs(g.mean()) < 0.02 and abs(g.std() - 1) < 0.02 if g.size > 1e5 else True


def _step(batch, want_noise=True, noise=None):
    from mal_amd import step, trainer
    from mal_amd.synthetic import to_dicts
    B, _, H, W = batch["color0"].shape
    dev = torch.device("cuda:0")
    opt = trainer.default_options(height=H, width=W, batch_size=B)
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=dev)
    for f, s in ((-1, "m1"), (1, "p1")):
        mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
        mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
    losses, _, maps = step.loss_step(opt, inputs, mono_outputs, outputs, noise=noise, want_noise=want_noise)
    losses["loss"].backward()
    torch.cuda.synchronize()
    return float(losses["loss"]), maps, {k: t.grad.cpu().numpy() for k, t in leaves.items()}


def test_step_draws_fresh_noise_and_equals_the_explicit_run():
    from mal_amd import config, step
    from mal_amd.synthetic import make_batch
    b = make_batch(2, 40, 130, seed=3)
    old = config.noise_source, config.noise_seed
    config.noise_source, config.noise_seed = "philox", 777
    try:
        ctr = step.noise_counter(torch.device("cuda:0"))
        c0 = int(ctr.item())
        l1, m1, g1 = _step(b)
        assert int(ctr.item()) == c0 + 1
        assert torch.equal(m1["noise"], _noise(777, c0, 2, 40, 130))
        l2, m2, g2 = _step(b)
        assert int(ctr.item()) == c0 + 2 and not torch.equal(m1["noise"], m2["noise"])
        config.noise_source = "cuda"
        l3, m3, g3 = _step(b, want_noise=False, noise=m1["noise"].clone())
        assert l3 == l1
        for k in g1:
            assert np.array_equal(g1[k], g3[k]), k
    finally:
        config.noise_source, config.noise_seed = old


def test_graph_replay_advances_the_stream():
    from mal_amd import config, step
    from mal_amd.synthetic import make_batch
    b = make_batch(1, 24, 70, seed=4)
    old = config.noise_source, config.noise_seed
    config.noise_source, config.noise_seed = "philox", 99
    try:
        from mal_amd import trainer
        from mal_amd.synthetic import to_dicts
        dev = torch.device("cuda:0")
        opt = trainer.default_options(height=24, width=70, batch_size=1)
        inputs, mono_outputs, outputs, leaves = to_dicts(b, lambda a, t, inv: None, device=dev)
        for f, s in ((-1, "m1"), (1, "p1")):
            mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
            mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
        holder = {}

        def one():
            losses, _, maps = step.loss_step(opt, inputs, mono_outputs, dict(outputs), want_maps=False, want_noise=True)
            holder["noise"] = maps["noise"]

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            one()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one()
        ctr = step.noise_counter(dev)
        torch.cuda.synchronize()
        c0 = int(ctr.item())
        g.replay()
        torch.cuda.synchronize()
        n1 = holder["noise"].clone()
        g.replay()
        torch.cuda.synchronize()
        n2 = holder["noise"].clone()
        assert int(ctr.item()) == c0 + 2
        assert torch.equal(n1, _noise(99, c0, 1, 24, 70)) and torch.equal(n2, _noise(99, c0 + 1, 1, 24, 70))
    finally:
        config.noise_source, config.noise_seed = old
